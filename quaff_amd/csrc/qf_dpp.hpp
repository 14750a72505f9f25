// qf_dpp.hpp — neighbour-lane exchange by DPP row / wave shifts (gfx950), shared by the fill kernels.
// One v_mov_b32_dpp per dword instead of a ds_bpermute through LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace qf {

// lane index inside the wavefront (loop-invariant: the compiler keeps the compare below as a scalar mask)
__device__ __forceinline__ int dpp_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ZERO: the group's edge lane receives 0.0 (both dwords zero-filled by bound_ctrl) instead of -inf
// G = 32 (two groups per wavefront): a wave shift, after which the second group's edge lane is patched (two selects per value)
template <int G, bool ZERO>
__device__ __forceinline__ double dpp_from_below(double v) {  // lane l-1's value; -inf (or 0) in the group's lane 0
  constexpr int ctrl = G == 16 ? 0x111 : 0x138;              // row_shr:1 / wave_shr:1
  const long long bits = __double_as_longlong(v);
  int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xFFFFFFFFll), ctrl, 0xF, 0xF, true);  // bound_ctrl: 0 at the edge
  int hi = ZERO ? __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), ctrl, 0xF, 0xF, true)
                : __builtin_amdgcn_update_dpp((int)0xFFF00000, (int)(bits >> 32), ctrl, 0xF, 0xF, false);
  if (G == 32) {
    const bool edge = dpp_lane() == 32;
    lo = edge ? 0 : lo;
    hi = edge ? (ZERO ? 0 : (int)0xFFF00000) : hi;
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int G, bool ZERO>
__device__ __forceinline__ double dpp_from_above(double v) {  // lane l+1's value; -inf (or 0) in the group's last lane
  constexpr int ctrl = G == 16 ? 0x101 : 0x130;              // row_shl:1 / wave_shl:1
  const long long bits = __double_as_longlong(v);
  int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xFFFFFFFFll), ctrl, 0xF, 0xF, true);  // bound_ctrl: 0 at the edge
  int hi = ZERO ? __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), ctrl, 0xF, 0xF, true)
                : __builtin_amdgcn_update_dpp((int)0xFFF00000, (int)(bits >> 32), ctrl, 0xF, 0xF, false);
  if (G == 32) {
    const bool edge = dpp_lane() == 31;
    lo = edge ? 0 : lo;
    hi = edge ? (ZERO ? 0 : (int)0xFFF00000) : hi;
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int G>
__device__ __forceinline__ float dpp_f32_from_above(float v) {  // lane l+1's value; 0 in the group's last lane
  constexpr int ctrl = G == 16 ? 0x101 : 0x130;              // row_shl:1 / wave_shl:1
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true);
  if (G == 32 && dpp_lane() == 31) r = 0;                    // (the first group's last lane would see the second group's first)
  return __int_as_float(r);
}

// acc = 2 * acc + (x > y): the compare's lane mask goes straight into an add-with-carry (no select / shift / or)
__device__ __forceinline__ uint32_t shift_in_gt(uint32_t acc, double x, double y) {
  const unsigned long long mask = __builtin_amdgcn_fcmp(x, y, 2 /* ordered > */);
  unsigned long long carry_out;
  asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(acc), "=s"(carry_out) : "v"(acc), "s"(mask));
  return acc;
}

// Traceback words are written once by a fill and read, sparsely, by the traceback kernel after it: a non-temporal store keeps them from
// churning the L2 on their way out (QF_TB_NT: A/B)
#ifndef QF_TB_NT
#define QF_TB_NT 0
#endif
typedef uint32_t qf_u4v __attribute__((ext_vector_type(4), aligned(4)));   // (the words of a unit start on a 4-byte boundary)
__device__ __forceinline__ void tb_store(uint32_t* p, uint32_t v) {
  if (QF_TB_NT) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void tb_store4(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {   // one 16-byte store
  if (QF_TB_NT) __builtin_nontemporal_store(qf_u4v{a, b, c, d}, (qf_u4v*)p);
  else *(qf_u4v*)p = qf_u4v{a, b, c, d};
}

}  // namespace qf
