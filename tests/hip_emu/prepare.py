"""Copies of the product's sources for the emulated (CPU) build, with the two constructs the stand-in runtime cannot take
as they are rewritten (tests/hip_emu/hip/hip_runtime.h; test infrastructure only):
  extern __shared__ T name[];           ->  T* name = reinterpret_cast<T*>(hip_emu::DynamicShared());
  asm volatile("" : "+v"(x));           ->  (nothing: an optimisation barrier on a vector register)
usage: prepare.py <out dir> <source> ..."""
import os
import re
import sys


def interpreted_asm(text):
    """walk_pipe.hip: every `asm volatile(text : outputs : inputs : clobbers)` becomes hip_emu::RunAsm(text, {outputs},
    {inputs}) with each operand `[name] "constraint"(expr)` as hip_emu::Op("name", "constraint", expr) -- the statement is
    then interpreted (gfx950_asm.hpp); LDS addresses become offsets into the launch's dynamic LDS."""
    text = text.replace("__device__ __forceinline__ unsigned LdsAddress(const void* p) { return (unsigned)(size_t)p; }",
                        "__device__ __forceinline__ unsigned LdsAddress(const void* p) { return (unsigned)(static_cast<const char*>(p) - "
                        "static_cast<const char*>(hip_emu::DynamicShared())); }")
    assert "hip_emu::DynamicShared()));" in text
    # the image part hands exponentials round a wave through LDS without a barrier (a wave's LDS instructions execute in
    # order on the hardware); the emulation's lanes are fibers that run one after the other: they meet here
    lockstep = "exps[lane] = exp(lam * time);"
    assert text.count(lockstep) == 1
    text = text.replace(lockstep, lockstep + " hip_emu::WaveBarrier();")
    text = re.sub(r'\[(\w+)\]\s*"([=&+]*[sv])"\s*\(([^()]+)\)', r'hip_emu::Op("\1", "\2", \3)', text)
    out, at = [], 0
    while True:
        start = text.find("asm volatile(", at)
        if start < 0:
            out.append(text[at:])
            break
        out.append(text[at:start])
        depth, k = 0, start + len("asm volatile")
        while True:
            if text[k] == "(":
                depth += 1
            elif text[k] == ")":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        inside = text[start + len("asm volatile("):k]
        # sections are separated by ':' at the top level (none of the operands holds one; "::" does not occur)
        parts, cur, level = [], "", 0
        for i, ch in enumerate(inside):
            if ch in "([":
                level += 1
            elif ch in ")]":
                level -= 1
            scope = ch == ":" and (inside[i + 1:i + 2] == ":" or inside[i - 1:i] == ":")  # (the :: of hip_emu::Op)
            if ch == ":" and level == 0 and not scope:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        parts.append(cur)
        assert len(parts) == 4, parts
        clean = lambda s: s.replace("\\\n", " ").strip()  # noqa: E731  (line continuations of the macro)
        body = "hip_emu::RunAsm(%s, {%s}, {%s})" % (parts[0].rstrip(" \\\n"), clean(parts[1]), clean(parts[2]))
        # keep the macro's line structure: one logical line
        out.append(body)
        at = k + 1
    return "".join(out)


def lds_walk_asm(text):
    """walk_lds.hip: its asm statements are scalar loads of a step descriptor into an SGPR tuple and the waits for them
    (hidden from the compiler's own wait counting on purpose): a copy and nothing."""
    pairs = (('asm volatile("s_load_dwordx16 %0, %1, 0x0" : "+s"(w) : "s"(p));', "std::memcpy(&w, p, 64);"),
             ('asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w)::"memory");', ";"),
             ('asm volatile("s_load_dwordx8 %0, %1, 0x0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(p) : "memory");', "std::memcpy(&w, p, 32);"),
             ('asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(w) : "s"(p));', "std::memcpy(&w, p, 32);"),
             ('asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");', ";"))
    for old, new in pairs:
        assert old in text, old
        text = text.replace(old, new)
    assert "asm volatile" not in text
    return text


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    for path in sys.argv[2:]:
        text = open(path).read()
        text = re.sub(r"extern\s+__shared__\s+([A-Za-z_0-9:]+)\s+([A-Za-z_0-9]+)\[\];",
                      r"\1* \2 = reinterpret_cast<\1*>(hip_emu::DynamicShared());", text)
        text = re.sub(r'asm volatile\(""\s*:\s*"\+v"\([A-Za-z_0-9]+\)\);', ";", text)
        text = re.sub(r"__shared__\s+(alignas\(\d+\))", r"\1 __shared__", text)  # (`static alignas(16) T x` is not C++)
        # (gs_kernels.hip keeps its assembly behind GS_ASM_FETCH, which the emulated build sets to 0: the builtin form)
        if os.path.basename(path) == "walk_pipe.hip":
            text = interpreted_asm(text)
        if os.path.basename(path) == "walk_lds.hip":
            text = lds_walk_asm(text)
        if ("asm volatile" in text or "__asm__" in text) and "GS_ASM_FETCH" not in text:
            raise SystemExit(f"{path}: holds assembly the emulation cannot run")
        target = os.path.join(out, os.path.basename(path))
        if not (os.path.exists(target) and open(target).read() == text):
            open(target, "w").write(text)


if __name__ == "__main__":
    main()
