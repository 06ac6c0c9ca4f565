"""Copies of the product's sources for the emulated (CPU) build, with the two constructs the stand-in runtime cannot take
as they are rewritten (tests/hip_emu/hip/hip_runtime.h; test infrastructure only):
  extern __shared__ T name[];           ->  T* name = reinterpret_cast<T*>(hip_emu::DynamicShared());
  asm volatile("" : "+v"(x));           ->  (nothing: an optimisation barrier on a vector register)
usage: prepare.py <out dir> <source> ..."""
import os
import re
import sys


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    for path in sys.argv[2:]:
        text = open(path).read()
        text = re.sub(r"extern\s+__shared__\s+([A-Za-z_0-9:]+)\s+([A-Za-z_0-9]+)\[\];",
                      r"\1* \2 = reinterpret_cast<\1*>(hip_emu::DynamicShared());", text)
        text = re.sub(r'asm volatile\(""\s*:\s*"\+v"\([A-Za-z_0-9]+\)\);', ";", text)
        # (gs_kernels.hip keeps its assembly behind GS_ASM_FETCH, which the emulated build sets to 0: the builtin form)
        if ("asm volatile" in text or "__asm__" in text) and "GS_ASM_FETCH" not in text:
            raise SystemExit(f"{path}: holds assembly the emulation cannot run")
        target = os.path.join(out, os.path.basename(path))
        if not (os.path.exists(target) and open(target).read() == text):
            open(target, "w").write(text)


if __name__ == "__main__":
    main()
