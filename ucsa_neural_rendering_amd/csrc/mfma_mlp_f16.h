// fp16-input MFMA variant of the three MLPs (inference option, tiny-cuda-nn's
// own numerics: fp16 weights and layer inputs, fp32 accumulation).
//
// v_mfma_f32_16x16x32_f16: D[16x16] = A[16x32] * B[32x16] + C.  Lane l = 16g+j
// holds A[i=j'][k = 8g + e] and B[k = 8g + e][j], e = 0..7 (8 halves = 4
// VGPRs); C/D as in the fp32 form (row 4g + r, column j).  Chaining works the
// same way as in mfma_mlp.h: k-slot (s, g, e) of the next layer is DEFINED to
// be neuron 16*(2s + (e>>2)) + 4g + (e&3), i.e. what lane (g, j) already holds
// in acc[2s][0..3], acc[2s+1][0..3]; ucsa_mlp_pack_f16 permutes the weights.
#pragma once
#include "mfma_mlp.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_h(half8 a, half8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

typedef short s16x8 __attribute__((ext_vector_type(8)));

// two accumulator blocks (ReLU) -> one 32-wide k-step operand.  Rounding is
// monotonic and keeps the sign, so the ReLU runs AFTER the conversion, on the
// packed halves (v_pk_max_i16 on the bit patterns: one instruction per pair).
__device__ __forceinline__ half8 chain_relu_h(f32x4 lo, f32x4 hi) {
  half8 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] = (_Float16)lo[r];
    v[4 + r] = (_Float16)hi[r];
  }
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(
      half8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
}

// A fragment f of a packed fp16 weight buffer (16 B per lane)
__device__ __forceinline__ half8 frag_h(const void* packed, int f,
                                        uint32_t lane) {
  return reinterpret_cast<const half8*>(packed)[f * 64 + lane];
}

// fragment counts: sigma 4 + 2 ; colour 4 + 8 + 2 ; sem 4 + 2*nrb
#define SIGMA_H_FRAGS 6
#define COLOR_H_FRAGS 14
#define SEM_H_FRAGS(nrb) (4 + 2 * (nrb))
