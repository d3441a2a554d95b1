// Wavefront-level helpers shared by the oxDNA and MARTINI kernels.
#ifndef MYTHOS_WAVE_OPS_H
#define MYTHOS_WAVE_OPS_H

#include <hip/hip_runtime.h>

namespace mythos {

// Cross-lane moves inside a 16-lane row as DPP modifiers (full-rate VALU, no trip through the LDS
// crossbar that __shfl's ds_bpermute takes): quad_perm [1,0,3,2] / [2,3,0,1] exchange with lane^1 /
// lane^2, row_half_mirror and row_mirror reflect inside 8 / 16 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}

// Sum over the G lanes of a group (G <= 16: butterfly in a fixed order, every lane gets the total).
template <int G, typename R>
__device__ __forceinline__ R group_sum(R v) {
  if constexpr (G <= 16) {
    if constexpr (G >= 2) v += dpp_move<0xB1>(v);   // lane ^ 1
    if constexpr (G >= 4) v += dpp_move<0x4E>(v);   // lane ^ 2: quads now hold their sum
    if constexpr (G >= 8) v += dpp_move<0x141>(v);  // row_half_mirror: the other quad of the 8
    if constexpr (G >= 16) v += dpp_move<0x140>(v); // row_mirror: the other half of the 16
  } else {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  }
  return v;
}

}  // namespace mythos

#endif  // MYTHOS_WAVE_OPS_H
