// Pieces shared by the fused composite kernel (composite.hip) and the split
// inference pair (composite_split.hip): identical arithmetic, so both produce
// the same bits.
#pragma once
#include "mfma_mlp_f16.h"
#include "wave_ops.h"

#define ROW_FINE 0x80000000u

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Sigmoid and softmax of the shaded samples use the hardware exponential
// (v_exp_f32 on x*log2e, ~2 ulp): 27 of them per sample were 6 % of the
// kernel as libm expf (measured), and the outputs are probabilities / colours
// compared at 1e-4.  The weights (phase A) keep expf: they decide the mask.
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
// ... and the hardware reciprocal (v_rcp_f32, 1 ulp) for the sigmoid and the
// softmax normalisation: an IEEE division is an 11-instruction sequence, four
// of them per sample were ~9 % of the shading kernel's VALU work.
// max of two finite-or-infinite floats as ONE instruction (v_med3_f32 with +inf
// as the third operand); fmaxf costs two (it first quiets a signalling NaN)
__device__ __forceinline__ float fast_max(float a, float b) {
  return __builtin_amdgcn_fmed3f(a, b, __builtin_inff());
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) {
  return fast_rcp(1.0f + fast_exp(-x));
}

__device__ __forceinline__ void sh4_select(float dx, float dy, float dz,
                                           uint32_t g, float (&o)[4]) {
  // oracle: d01 = (d + 1) / 2; x = d01 * 2 - 1   (keep the same rounding)
  const float x = ((dx + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float y = ((dy + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float z = ((dz + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float xy = x * y, xz = x * z, yz = y * z;
  const float x2 = x * x, y2 = y * y, z2 = z * z;
  if (g == 0) {
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
  } else if (g == 1) {
    o[0] = 1.0925484305920792f * xy;
    o[1] = -1.0925484305920792f * yz;
    o[2] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[3] = -1.0925484305920792f * xz;
  } else if (g == 2) {
    o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[2] = 2.8906114426405538f * xy * z;
    o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  } else {
    o[0] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[1] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[2] = 1.4453057213202769f * z * (x2 - y2);
    o[3] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  }
}

