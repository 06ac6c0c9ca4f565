"""Static instruction mix of one kernel's gfx950 code, basic block by basic block (hipcc -S --cuda-device-only): what a
round without a GPU can still say about a change whose purpose is fewer instructions -- LDS reads and writes, FP64
vector and matrix instructions, branches per block, and the kernel's registers / LDS / scratch from its metadata.  Loop
trip counts are not in the code: the caller multiplies (a block that branches to itself is a loop body).
usage: python scripts/isa_block_mix.py <source.hip> <kernel name substring> [extra compiler flags ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bito_amd", "csrc")
FLAGS = {"gs_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "_gs_round5.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
         "walk_lds.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def main():
    src, name, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *FLAGS.get(os.path.basename(src), []),
               *extra, "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
        text = open(out).read()
    symbols = [m.group(1) for m in re.finditer(r"^(_Z\w+):", text, re.M) if name in m.group(1)]
    for sym in symbols:
        body = text[text.index(sym + ":"):]
        body = body[:body.index(".Lfunc_end")]
        meta = text[text.index(".name:           " + sym):]
        meta = meta[:meta.index("- .agpr_count", 10) if "- .agpr_count" in meta[10:] else 1500]
        pick = {k: re.search(r"\." + k + r":\s+(\d+)", meta) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size")}
        agpr = re.search(r"\.agpr_count:\s+(\d+)", text[:text.index(".name:           " + sym)][-400:])
        print(sym)
        print("  " + ", ".join(f"{k} {v.group(1)}" for k, v in pick.items() if v) + (f", agpr_count {agpr.group(1)}" if agpr else ""))
        blocks, cur = [], ("entry", [])
        for line in body.splitlines()[1:]:
            line = line.strip()
            if re.match(r"^\.LBB\d+_\d+:", line):
                blocks.append(cur)
                cur = (line.split(":")[0], [])
            elif line and not line.startswith((";", ".")):
                cur[1].append(line)
        blocks.append(cur)
        total = {}
        for label, ins in blocks:
            kinds = {}
            for i in ins:
                op = i.split()[0]
                key = ("lds " + op if op.startswith("ds_") else "mfma" if "mfma" in op else
                       "fp64 valu" if op.startswith("v_") and "f64" in op else "global/buffer" if op.startswith(("global_", "buffer_", "flat_")) else
                       "s_barrier" if op == "s_barrier" else None)
                if key:
                    kinds[key] = kinds.get(key, 0) + 1
                    total[key] = total.get(key, 0) + 1
            loops = [i.split()[-1] for i in ins if i.startswith("s_cbranch") and i.split()[-1] == label]
            if len(ins) >= 16 or loops:
                print(f"  {label:10s} {len(ins):5d} instructions{' (loops to itself)' if loops else ''}: " +
                      ", ".join(f"{k} {v}" for k, v in sorted(kinds.items())))
        print("  whole kernel (static): " + ", ".join(f"{k} {v}" for k, v in sorted(total.items())))


if __name__ == "__main__":
    main()
