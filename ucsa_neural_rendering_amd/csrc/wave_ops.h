// Wave64 scan / reduce helpers (deterministic: fixed tree, no atomics).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float wave_incl_scan_mul(float v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v = o * v;
  }
  return v;
}

__device__ __forceinline__ float wave_incl_scan_add(float v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v = o + v;
  }
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ float wave_bcast(float v, int src) {
  return __shfl(v, src, 64);
}

__device__ __forceinline__ uint32_t wave_incl_scan_add_u32(uint32_t v,
                                                           uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= (uint32_t)d) v = o + v;
  }
  return v;
}

// ---------------------------------------------------------------------------
// The same scans on the DPP data path (no LDS round trip): row_shr 1 / 2 / 4 /
// 8 inside the four 16-lane rows, then row_bcast15 into rows 1 and 3 and
// row_bcast31 into rows 2 and 3.  A __shfl_up is a ds_bpermute_b32 (~24 issue
// cycles per wave and an LDS latency in a dependent chain, six per scan); a DPP
// step is one VALU instruction (tools/ubench/wave_scan.hip: 48 vs 13.6 G
// scans/s chip-wide).  Lanes without a source keep the identity.  The partial
// results associate differently from the shuffle ladder (row totals instead of
// distance-16 / 32 partials): ordinary fp32 round-off apart.  Used by the
// run() path's weight / cdf scans (composite*.hip, sampling.hip); the marcher
// keeps the shuffle form its oracle was matched to bit for bit.
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_src(float v, float identity) {
  return __int_as_float(__builtin_amdgcn_update_dpp(
      __float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_incl_scan_mul_dpp(float v) {
  v = v * dpp_src<0x111, 0xf>(v, 1.0f);  // row_shr:1
  v = v * dpp_src<0x112, 0xf>(v, 1.0f);  // row_shr:2
  v = v * dpp_src<0x114, 0xf>(v, 1.0f);  // row_shr:4
  v = v * dpp_src<0x118, 0xf>(v, 1.0f);  // row_shr:8
  v = v * dpp_src<0x142, 0xa>(v, 1.0f);  // row_bcast:15 -> rows 1, 3
  v = v * dpp_src<0x143, 0xc>(v, 1.0f);  // row_bcast:31 -> rows 2, 3
  return v;
}

__device__ __forceinline__ float wave_incl_scan_add_dpp(float v) {
  v = v + dpp_src<0x111, 0xf>(v, 0.0f);
  v = v + dpp_src<0x112, 0xf>(v, 0.0f);
  v = v + dpp_src<0x114, 0xf>(v, 0.0f);
  v = v + dpp_src<0x118, 0xf>(v, 0.0f);
  v = v + dpp_src<0x142, 0xa>(v, 0.0f);
  v = v + dpp_src<0x143, 0xc>(v, 0.0f);
  return v;
}

// exclusive from inclusive: lane i takes lane i - 1 (wave_shr:1), lane 0 the identity
__device__ __forceinline__ float wave_shift_up1(float incl, float identity) {
  return dpp_src<0x138, 0xf>(incl, identity);
}

// lane 63's value in every lane (one v_readlane_b32)
__device__ __forceinline__ float wave_last(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Reduction over the four 16-lane rows (lanes l, l^16, l^32, l^48), result in
// all of them: v_permlane32_swap / v_permlane16_swap (gfx950) exchange half
// waves / neighbouring rows between two copies of the value -- mov + swap + op
// per step (~13 issue cycles, no LDS) instead of a ds_bpermute_b32 (~24 plus
// its latency in a dependent chain).
typedef unsigned int wave_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rows4_sum(float x) {
  const wave_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const wave_u2 q = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float rows4_max(float x) {
  const wave_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  // (+ 0.0f: marks the operands as canonical, so fmaxf stays one v_max_f32)
  const float s = fmaxf(__uint_as_float(r[0]) + 0.0f, __uint_as_float(r[1]) + 0.0f);
  const wave_u2 q = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return fmaxf(__uint_as_float(q[0]) + 0.0f, __uint_as_float(q[1]) + 0.0f);
}
