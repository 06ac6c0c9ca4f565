#!/usr/bin/env python3
"""Build-time check of walk_pipe.hip's compiled device code (hipcc -S --cuda-device-only output).

walk_pipe_kernel keeps the matrix images of a whole tree in a[0 .. WALK_PIPE_IMAGE_REGS) across several asm
statements.  Those AGPRs are on every statement's clobber list, which keeps the compiler from holding values in
them ACROSS a statement; this script verifies the stronger property the kernel relies on: outside the asm
statements the compiler never touches them at all (it has v0..v31 and the AGPRs above for its own values).
Fails (exit 1) on the first stray use.  usage: check_walk_pipe_asm.py walk_pipe.gfx950.s"""
import re
import sys

import os

# WALK_PIPE_IMAGE_REGS of the generated loops this build includes
_INC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bito_amd", "csrc", "walk_pipe_gen.inc")
LIMIT = int(re.search(r"#define WALK_PIPE_IMAGE_REGS (\d+)", open(_INC).read()).group(1))
# ... and of the other layouts' kernels (fourth template argument: ...ELb?ELi1EE wide, ...ELb?ELi2EE two waves per SIMD)
WIDE_LIMIT = int(re.search(r"#define WALK_PIPE_W_IMAGE_REGS (\d+)", open(_INC).read()).group(1))
TWO_LIMIT = int(re.search(r"#define WALK_PIPE_T_IMAGE_REGS (\d+)", open(_INC).read()).group(1))


def main(path):
    inside = False
    kernel = None
    stray = 0
    kernels = 0
    meta = ""
    spilled = []
    for line in open(path):
        m = re.match(r"^(_ZN8bito_amd16walk_pipe_kernel\w+):", line)
        if m:
            kernel = m.group(1)
            kernels += 1
        if "#ASMSTART" in line:
            inside = True
            continue
        if "#ASMEND" in line:
            inside = False
            continue
        text = line.strip()
        # a kernel's code ends at its .Lfunc_end label / .size directive, not at the first s_endpgm: LLVM may emit
        # several exits and place blocks behind the first one
        if kernel and (text.startswith(".Lfunc_end") or text.startswith(".size")):
            kernel = None
            continue
        # the kernels' metadata records (end of the file).  Spilled VGPRs are parked in AGPRs (above the images:
        # the scan below covers every instruction of the function, spill code included) or in scratch; reported only.
        m = re.match(r"^\.name:\s+(\S+)", text)
        if m:
            meta = m.group(1)
        m = re.match(r"^\.vgpr_spill_count:\s+(\d+)", text)
        if m and "walk_pipe_kernel" in meta and int(m.group(1)) > 0:
            spilled.append((meta, int(m.group(1))))
        if inside or not kernel or not text or text[0] in ";.":
            continue
        layout = re.search(r"walk_pipe_kernelILi\dELi\dELb[01]ELi(\d)EE", kernel)
        if not layout:
            print(f"cannot read the layout of {kernel}", file=sys.stderr)
            return 1
        limit = {0: LIMIT, 1: WIDE_LIMIT, 2: TWO_LIMIT, 3: LIMIT}[int(layout.group(1))]
        for mm in re.finditer(r"\ba\[?(\d+)", text.split(";")[0]):
            if int(mm.group(1)) < limit:
                stray += 1
                if stray <= 5:
                    print(f"stray AGPR use in {kernel}: {text}", file=sys.stderr)
    # the two-wave kernels must have been built with the register split their loops were generated for: every AGPR the
    # images need, the rest of the wave's 256 registers as VGPRs (Makefile: amdgpu-agpr-alloc forced by name)
    text_all = open(path).read()
    for m in re.finditer(r"\.amdhsa_kernel (_ZN8bito_amd16walk_pipe_kernelILi\dELi\dELb[01]ELi2EE\w+)(.*?)\.end_amdhsa_kernel", text_all, re.S):
        acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", m.group(2)).group(1))
        nxt = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2)).group(1))
        if acc != 256 - TWO_LIMIT or nxt > 256:
            print(f"{m.group(1)}: accum_offset {acc}, next_free_vgpr {nxt}; the two-wave loops need {256 - TWO_LIMIT} VGPRs + {TWO_LIMIT} AGPRs = 256", file=sys.stderr)
            return 1
    if kernels == 0:
        print("no walk_pipe_kernel instantiation found in " + path, file=sys.stderr)
        return 1
    if stray:
        print(f"{stray} uses of a0..a{LIMIT - 1} outside the asm statements", file=sys.stderr)
        return 1
    print(f"{kernels} walk_pipe_kernel instantiations: a0..a{LIMIT - 1} (wide layout: a0..a{WIDE_LIMIT - 1}, two waves per SIMD: a0..a{TWO_LIMIT - 1}) untouched outside the asm statements"
          + (f" ({len(spilled)} instantiations spill VGPRs, at most {max(c for _, c in spilled)}: none into the image registers)" if spilled else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
