#!/bin/bash
# Debug build: bito_amd/libbito_amd_dbg.so = the library with cycle stamps at the phase boundaries of
# walk_lds_kernel (read back by scripts/gpu_stamps.py).  The tracked sources are left untouched.
set -e
cd "$(dirname "$0")/../bito_amd/csrc"
cp walk_lds.hip /tmp/walk_lds_keep.hip
trap 'cp /tmp/walk_lds_keep.hip walk_lds.hip; make >/dev/null' EXIT
python3 - <<'PY'
s=open('/tmp/walk_lds_keep.hip').read()
def rep(a,b):
    global s
    assert a in s, a[:60]
    s=s.replace(a,b,1)
rep("template <int C, int G, bool GRAD>\n__global__ void __launch_bounds__(kLdsWaves * 64, 1)\nwalk_lds_kernel(","__device__ unsigned long long g_walk_stamps[8 * 64];\n#define STAMP(i) do { if (blockIdx.x >= 16000 && blockIdx.x < 16064 && threadIdx.x == 0) g_walk_stamps[(blockIdx.x - 16000) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)\n\ntemplate <int C, int G, bool GRAD>\n__global__ void __launch_bounds__(kLdsWaves * 64, 1)\nwalk_lds_kernel(")
rep("  extern __shared__ double lds[];\n","  extern __shared__ double lds[];\n  STAMP(0);\n")
rep("  __syncthreads();\n\n#define TIP_AT","  __syncthreads();\n  STAMP(1);\n\n#define TIP_AT")
rep("  // ---------------- root: site likelihoods","  STAMP(2);\n  // ---------------- root: site likelihoods")
rep("  // ---------------- pre-order + edge derivatives","  STAMP(3);\n  // ---------------- pre-order + edge derivatives")
rep("  // ---------------- workgroup sums, fixed order","  STAMP(4);\n  // ---------------- workgroup sums, fixed order")
rep("      out[e] = s;\n    }\n  }\n}\n","      out[e] = s;\n    }\n  }\n  STAMP(5);\n}\n\nextern \"C\" int bito_amd_debug_walk_stamps(unsigned long long* out) {\n  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_stamps), sizeof(unsigned long long) * 8 * 64);\n}\n")
open('walk_lds.hip','w').write(s)
PY
make >/dev/null
cp ../libbito_amd.so ../libbito_amd_dbg.so
