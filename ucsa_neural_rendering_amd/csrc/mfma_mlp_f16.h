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

// two floats -> two halves (round to nearest even) in one dword: ONE
// v_cvt_pk_f16_f32 (gfx950) instead of two v_cvt_f16_f32 and a v_perm_b32 --
// conversions issue at half the rate of plain fp32 arithmetic
// (tools/ubench/valu_rates.hip)
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2_t cvt_pk_h(float a, float b) {
  return __builtin_convertvector(f32x2_t{a, b}, half2_t);
}

// two accumulator blocks (ReLU) -> one 32-wide k-step operand.  Rounding is
// monotonic and keeps the sign, so the ReLU runs AFTER the conversion, on the
// packed halves (v_pk_max_i16 on the bit patterns: one instruction per pair).
__device__ __forceinline__ half8 chain_relu_h(f32x4 lo, f32x4 hi) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t w = {__builtin_bit_cast(uint32_t, cvt_pk_h(lo[0], lo[1])),
                     __builtin_bit_cast(uint32_t, cvt_pk_h(lo[2], lo[3])),
                     __builtin_bit_cast(uint32_t, cvt_pk_h(hi[0], hi[1])),
                     __builtin_bit_cast(uint32_t, cvt_pk_h(hi[2], hi[3]))};
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(
      half8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, w), z));
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

// dW[ob][ib] += dY_tile^T X_tile over the 16 samples of the tile on the f16
// pipe: ONE v_mfma_f32_16x16x16_f16 per 16x16 weight tile instead of four
// f32-input k-steps (which run on the vector ALU at 32 cycles each).  Lane
// (k = lane >> 4, i = lane & 15) holds samples 4k .. 4k+3 of neuron 16*ob + i
// (A) resp. 16*ib + i (B) -- the same LDS words as the f32 form reads, four
// per operand, rounded to half (dY carries the loss scale; X is what the f16
// forward fed into the layer).  fp32 accumulation across the whole kernel.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

template <int OB, int IB>
__device__ __forceinline__ void dw_accumulate_h(const float* dy_tile,
                                                const float* x_tile,
                                                uint32_t lane,
                                                f32x4 (&dw)[OB][IB]) {
  const uint32_t i = lane & 15u, k = lane >> 4;
  half4 a[OB], b[IB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      a[ob][e] = (_Float16)dy_tile[(4 * k + e) * TILE_LD + 16 * ob + i];
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      b[ib][e] = (_Float16)x_tile[(4 * k + e) * TILE_LD + 16 * ib + i];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
      dw[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[ob], b[ib], dw[ob][ib], 0, 0, 0);
}
