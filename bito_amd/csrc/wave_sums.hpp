// wave_sums.hpp -- sums across the lanes of a wave on the vector ALU alone (DPP, v_permlane32_swap, v_permlane16_swap:
// no LDS round trips, no address registers).  Device code only (included by the .hip files).  Fixed order.
#pragma once
#include <hip/hip_runtime.h>

namespace bito_amd {

// Two sums over the 64 lanes at once, without LDS round trips or address registers: v_permlane32_swap puts
// a's upper half beside its lower half in lanes 0-31 and b's likewise in lanes 32-63 (one addition halves
// both), then five DPP steps reduce each half: lane 31 ends up with the sum of a, lane 63 with the sum of b.
// Fixed order.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double DppAdd(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, kRowMask, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, kRowMask, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double PairSum(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  double v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
  v = DppAdd<0x111, 0xf>(v);  // row_shr:1
  v = DppAdd<0x112, 0xf>(v);  // row_shr:2
  v = DppAdd<0x114, 0xf>(v);  // row_shr:4
  v = DppAdd<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row's sum
  v = DppAdd<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3 -> lanes 31 and 63 hold the halves' sums
  return v;
}

// v + (v of the lane 32 away), in every lane: v_permlane32_swap of two copies puts the lower half beside the upper half
__device__ __forceinline__ double SwapSum32(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// v + (v of the lane 16 away: rows 0 and 1, rows 2 and 3), in every lane: v_permlane16_swap
__device__ __forceinline__ double SwapSum16(double v) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

}  // namespace bito_amd
