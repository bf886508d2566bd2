// fp32 MFMA building blocks for the three bias-free width-64 MLPs.
//
// Mapping (v_mfma_f32_16x16x4_f32: D[16x16] = A[16x4] * B[4x16] + C):
//   rows    i = output neurons of the layer (16 per row block "rb"),
//   columns j = samples (16 per MFMA),
//   k         = input neurons (4 per k-step "ks").
// Lane l = 16*g + j holds  A[i=j'][k=g] (j' = l&15)  and  B[k=g][j],
// and after the MFMA   D[row = 4*g + r][col = j]   in accumulator reg r.
//
// Layer chaining without any cross-lane traffic: the order of the contraction
// index is free as long as A and B agree, so k-step ks = 4*rb + r of the NEXT
// layer is defined to be neuron 16*rb + 4*g + r -- exactly what lane (g, j)
// already holds in acc[rb][r].  The A fragments (weights) are pre-permuted
// accordingly by ucsa_mlp_pack, once per parameter update.
//
// f32 MFMA is bit-for-bit a k-ordered fmaf chain (MI355X guide), so the fp32
// mode of this library has ordinary fp32 round-off vs the CPU oracle.
#pragma once
#include "ucsa_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ReLU as ONE instruction (v_max_i32 on the bit pattern: a float is negative
// iff its bit pattern is a negative integer; -0 -> +0).  fmaxf(x, 0) costs two
// (the compiler first quiets a possible signalling NaN with v_max_f32 x, x).
// Not inline asm: the operand usually comes straight out of an MFMA and goes
// into the next one, and the compiler's hazard recogniser (which inserts the
// s_nop an MFMA result needs before a VALU may read it) does not look inside
// asm statements -- measured: wrong results.  v_max_f32 and v_max_i32 issue at
// the same (half) rate on gfx950 anyway (tools/ubench/valu_rates.hip).
__device__ __forceinline__ float relu1(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b > 0 ? b : 0);
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  f32x4 r;
  r[0] = relu1(v[0]);
  r[1] = relu1(v[1]);
  r[2] = relu1(v[2]);
  r[3] = relu1(v[3]);
  return r;
}

// Fragment counts / offsets inside a packed weight vector (in 64-float frags).
//   sigma: L1 32->64 (4 rb x 8 ks = 32), L2 64->16 (1 x 16)            = 48
//   color: L1 32->64 (32), L2 64->64 (4 x 16 = 64), L3 64->16 (16)     = 112
//   sem:   L1 16->64 (4 x 4 = 16), L2 64->out_pad (nrb x 16)           = 16+16*nrb
#define SIGMA_L1_FRAGS 32
#define SIGMA_L2_FRAGS 16
#define COLOR_L1_FRAGS 32
#define COLOR_L2_FRAGS 64
#define COLOR_L3_FRAGS 16
#define SEM_L1_FRAGS 16

// One dense layer: KS k-steps, NRB row blocks, A fragments read through `wf`
// (a callable (rb, ks) -> float so weights may live in registers or LDS).
template <int KS, int NRB, typename WF>
__device__ __forceinline__ void mfma_layer(const float (&xin)[KS], WF wf,
                                           f32x4 (&acc)[NRB]) {
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) acc[rb] = mfma16(wf(rb, ks), xin[ks], acc[rb]);
  }
}

// acc[4] (64 neurons) -> next layer's 16 k-step operands, with ReLU.
__device__ __forceinline__ void chain_relu(const f32x4 (&acc)[4],
                                           float (&xin)[16]) {
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) xin[rb * 4 + r] = relu1(acc[rb][r]);
}

// ---------------------------------------------------------------------------
// Backward helpers.
//
// dX = W^T dY uses "transposed" A fragments (ucsa_mlp_pack_t): rows = input
// neurons of the layer, k-steps follow dY's accumulator layout, so dY's
// registers are the B operands directly and dX comes out in the layout of the
// forward activations it belongs to.
//
// dW = dY X^T contracts over SAMPLES, which sit on the column index of every
// activation register, so both operands are transposed through a small LDS
// tile [16 samples][neurons] (row stride TILE_LD floats):
//     A[i = o][k = s]  <- dY_tile[s][16*ob + i]
//     B[k = s][j = c]  <- X_tile [s][16*ib + j]
// and accumulated in registers for the whole kernel (one 16x16 tile = 4 regs).
// ---------------------------------------------------------------------------
#define TILE_LD 68

// store lane (g,j)'s accumulator block rb (neurons 16*rb+4g..+3 of sample j)
__device__ __forceinline__ void tile_store(float* tile, uint32_t g, uint32_t j,
                                           int rb, f32x4 v) {
  *reinterpret_cast<f32x4*>(tile + j * TILE_LD + 16 * rb + 4 * g) = v;
}

// dW[ob][ib] += dY_tile^T X_tile over the 16 samples of the tile
template <int OB, int IB>
__device__ __forceinline__ void dw_accumulate(const float* dy_tile,
                                              const float* x_tile,
                                              uint32_t lane,
                                              f32x4 (&dw)[OB][IB]) {
  const uint32_t i = lane & 15u, k = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    float a[OB], b[IB];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) a[ob] = dy_tile[(4 * ks + k) * TILE_LD + 16 * ob + i];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) b[ib] = x_tile[(4 * ks + k) * TILE_LD + 16 * ib + i];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) dw[ob][ib] = mfma16(a[ob], b[ib], dw[ob][ib]);
  }
}

// write one wave's dW tiles into its slot of the partial buffer, tcnn layout
// (row-major [out, in_cols]); lane (g', j') holds rows 4g'+r, column j'.
template <int OB, int IB>
__device__ __forceinline__ void dw_store(float* dst, uint32_t in_cols,
                                         uint32_t lane,
                                         const f32x4 (&dw)[OB][IB]) {
  const uint32_t gq = lane >> 4, jq = lane & 15u;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dst[(16 * ob + 4 * gq + r) * in_cols + 16 * ib + jq] = dw[ob][ib][r];
}

template <int OB, int IB>
__device__ __forceinline__ void dw_zero(f32x4 (&dw)[OB][IB]) {
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) dw[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
}
