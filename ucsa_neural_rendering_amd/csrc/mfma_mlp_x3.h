// "bf16x3": fp32-grade contraction of the three MLPs on the dense bf16 MFMA
// pipe (v_mfma_f32_16x16x32_bf16, 16x the rate of the f32-input form).
//
// Every fp32 operand is split EXACTLY into three bf16 terms,
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1
// (round to nearest even, v_cvt_pk_bf16_f32; both differences are exact in
// fp32 and the last one has at most 8 significant bits), and a product x*w is
// accumulated in fp32 from the six partial products of order <= 2:
//     x2*w0 + x1*w1 + x0*w2 + x1*w0 + x0*w1 + x0*w0      (smallest first).
// Each partial product of two bf16 values is exact in fp32; what is dropped
// (x1*w2 + x2*w1 + x2*w2) is at most 2^-23 |x*w|, the size of the rounding of
// the fp32 product itself, and unbiased.  The same scheme is what "fp32 matmul precision =
// highest" means on bf16 matrix units elsewhere (six bf16 passes).  It is NOT
// bit-identical to the k-ordered fmaf chain of the f32-input MFMA; both sit
// at ordinary fp32 round-off from an fp64 evaluation
// (tests/test_gpu_parity.py::test_bf16x3_*).
//
// Fragment layout: as mfma_mlp_f16.h (lane l = 16g+j holds k-slots 8g+e,
// e = 0..7, of row / column j), one 16-byte fragment per term:
// packed[(f * 3 + term) * 64 + lane].
#pragma once
#include "mfma_mlp_f16.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct X3 {  // 8 values, three bf16 terms each (2 values per dword)
  u32x4 t[3];
};

__device__ __forceinline__ f32x4 mfma_b(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(
      __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two floats -> two bf16 (round to nearest even) in one dword, a in the low half
__device__ __forceinline__ uint32_t bf16_pair(float a, float b) {
  return __builtin_bit_cast(uint32_t,
                            __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float pair_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float pair_hi(uint32_t u) {
  return __uint_as_float(u & 0xFFFF0000u);
}

// The residuals a - lo(p), b - hi(p) of a bf16 pair p as ONE instruction each:
// v_dot2_f32_bf16  D = A.lo * B.lo + A.hi * B.hi + C  with B = (-1, 0) resp.
// (0, -1) and C = the fp32 value.  Both products are exact (x * -1, x * 0) and
// the sum is the exactly representable difference, so the result equals the
// shift / mask + subtract form bit for bit (tools/ubench/dot2_check.hip: 0
// mismatches over 2^21 pairs) at 2 instead of 4 instructions per pair
// (8.5 vs 11.3 issue cycles, profiles/r03_valu_rates.txt).  The two selector
// constants live in registers (made opaque once per kernel) so that they cannot
// be folded into an inline operand of unknown half placement.
struct X3Sel {
  uint32_t lo, hi;   // (-1, 0) and (0, -1) as bf16 pairs
};
__device__ __forceinline__ X3Sel x3_selectors() {
  X3Sel s{0x0000BF80u, 0xBF800000u};
  asm volatile("" : "+v"(s.lo), "+v"(s.hi));
  return s;
}
__device__ __forceinline__ float resid_lo(uint32_t p, float a, const X3Sel& s) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p),
                                         __builtin_bit_cast(bf16x2, s.lo), a, false);
}
__device__ __forceinline__ float resid_hi(uint32_t p, float b, const X3Sel& s) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p),
                                         __builtin_bit_cast(bf16x2, s.hi), b, false);
}

// exact three-term split of two values into dword `d` of each term
__device__ __forceinline__ void split_pair(float a, float b, X3& x, int d,
                                           const X3Sel& sel) {
  const uint32_t p0 = bf16_pair(a, b);
  const float ra = resid_lo(p0, a, sel), rb = resid_hi(p0, b, sel);
  const uint32_t p1 = bf16_pair(ra, rb);
  const float sa = resid_lo(p1, ra, sel), sb = resid_hi(p1, rb, sel);
  x.t[0][d] = p0;
  x.t[1][d] = p1;
  x.t[2][d] = bf16_pair(sa, sb);
}
__device__ __forceinline__ void split_pair(float a, float b, X3& x, int d) {
  split_pair(a, b, x, d, x3_selectors());
}

// two accumulator blocks (ReLU) -> one 32-wide k-step operand
__device__ __forceinline__ X3 chain_relu_x3(f32x4 lo, f32x4 hi, const X3Sel& sel) {
  X3 x;
  split_pair(relu1(lo[0]), relu1(lo[1]), x, 0, sel);
  split_pair(relu1(lo[2]), relu1(lo[3]), x, 1, sel);
  split_pair(relu1(hi[0]), relu1(hi[1]), x, 2, sel);
  split_pair(relu1(hi[2]), relu1(hi[3]), x, 3, sel);
  return x;
}
__device__ __forceinline__ X3 chain_relu_x3(f32x4 lo, f32x4 hi) {
  return chain_relu_x3(lo, hi, x3_selectors());
}

struct W3 {  // one A fragment, three terms
  u32x4 t[3];
};

__device__ __forceinline__ W3 frag_x3(const void* packed, int f, uint32_t lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(packed) + (f * 3) * 64 + lane;
  W3 w;
  w.t[0] = p[0];
  w.t[1] = p[64];
  w.t[2] = p[128];
  return w;
}

// acc += W * x, six partial products, smallest magnitude first
__device__ __forceinline__ f32x4 mfma_x3(const W3& w, const X3& x, f32x4 acc) {
  acc = mfma_b(w.t[2], x.t[0], acc);
  acc = mfma_b(w.t[1], x.t[1], acc);
  acc = mfma_b(w.t[0], x.t[2], acc);
  acc = mfma_b(w.t[1], x.t[0], acc);
  acc = mfma_b(w.t[0], x.t[1], acc);
  acc = mfma_b(w.t[0], x.t[0], acc);
  return acc;
}

// ---------------------------------------------------------------------------
// "bf16x2": the two-term form, for GRADIENT contractions (round 4,
// k_shade_bwd<.., B2>).  x = x0 + x1 with x0 = bf16(x), x1 = bf16(x - x0)
// carries 16 significant bits; a product is accumulated from the three partial
// products of order <= 1 (x1*w0 + x0*w1 + x0*w0), what is dropped is at most
// 2^-16 |x*w|.  Gradients are compared with the oracle at 2e-3 relative L2
// (tests/test_gpu_backward.py, test_gpu_configs.py); the forward stays bf16x3.
// The backward kernels RECOMPUTE the hidden layers with these products: a
// unit whose pre-activation is within ~1e-5 of zero may get the other ReLU
// gate than in the forward -- the other subgradient at a kink, on ~1e-5 of the
// units (as in any reduced-precision backward, tiny-cuda-nn's fp16 included).
// Weight fragments in LDS keep terms 0 and 1 of the x3 pack:
// lds[(f * 2 + term) * 64 + lane].
// ---------------------------------------------------------------------------
struct X2 {
  u32x4 t[2];
};
struct W2 {
  u32x4 t[2];
};

__device__ __forceinline__ void split2_pair(float a, float b, X2& x, int d,
                                            const X3Sel& sel) {
  const uint32_t p0 = bf16_pair(a, b);
  const float ra = resid_lo(p0, a, sel), rb = resid_hi(p0, b, sel);
  x.t[0][d] = p0;
  x.t[1][d] = bf16_pair(ra, rb);
}

// two accumulator blocks -> one 32-wide k-step operand, with / without ReLU
__device__ __forceinline__ X2 chain_relu_x2(f32x4 lo, f32x4 hi, const X3Sel& sel) {
  X2 x;
  split2_pair(relu1(lo[0]), relu1(lo[1]), x, 0, sel);
  split2_pair(relu1(lo[2]), relu1(lo[3]), x, 1, sel);
  split2_pair(relu1(hi[0]), relu1(hi[1]), x, 2, sel);
  split2_pair(relu1(hi[2]), relu1(hi[3]), x, 3, sel);
  return x;
}
__device__ __forceinline__ X2 chain_x2(f32x4 lo, f32x4 hi, const X3Sel& sel) {
  X2 x;
  split2_pair(lo[0], lo[1], x, 0, sel);
  split2_pair(lo[2], lo[3], x, 1, sel);
  split2_pair(hi[0], hi[1], x, 2, sel);
  split2_pair(hi[2], hi[3], x, 3, sel);
  return x;
}

// A fragment f of a two-term weight buffer in LDS
__device__ __forceinline__ W2 frag_x2(const void* lds2, int f, uint32_t lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(lds2) + (f * 2) * 64 + lane;
  W2 w;
  w.t[0] = p[0];
  w.t[1] = p[64];
  return w;
}

// acc += W * x, three partial products, smallest first
__device__ __forceinline__ f32x4 mfma_x2(const W2& w, const X2& x, f32x4 acc) {
  acc = mfma_b(w.t[1], x.t[0], acc);
  acc = mfma_b(w.t[0], x.t[1], acc);
  acc = mfma_b(w.t[0], x.t[0], acc);
  return acc;
}

// dW[ob][ib] += dY_tile^T X_tile over the 16 samples of the tile
// (v_mfma_f32_16x16x16_bf16, three passes): lane (k = lane >> 4, i = lane & 15)
// holds samples 4k .. 4k+3 of neuron 16*ob + i (A) resp. 16*ib + i (B), the
// LDS words dw_accumulate reads, each split into two bf16 terms.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct B2x4 {  // four values, two bf16 terms each
  u32x2 t[2];
};
__device__ __forceinline__ B2x4 split2_quad(float a, float b, float c, float d,
                                            const X3Sel& sel) {
  B2x4 q;
  const uint32_t p0 = bf16_pair(a, b), p1 = bf16_pair(c, d);
  q.t[0][0] = p0;
  q.t[0][1] = p1;
  q.t[1][0] = bf16_pair(resid_lo(p0, a, sel), resid_hi(p0, b, sel));
  q.t[1][1] = bf16_pair(resid_lo(p1, c, sel), resid_hi(p1, d, sel));
  return q;
}
__device__ __forceinline__ f32x4 mfma_b16(u32x2 a, u32x2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a),
                                                   __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

template <int OB, int IB>
__device__ __forceinline__ void dw_accumulate_b2(const float* dy_tile,
                                                 const float* x_tile, uint32_t lane,
                                                 f32x4 (&dw)[OB][IB],
                                                 const X3Sel& sel) {
  const uint32_t i = lane & 15u, k = lane >> 4;
  B2x4 a[OB], b[IB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
    const float* p = dy_tile + (4 * k) * TILE_LD + 16 * ob + i;
    a[ob] = split2_quad(p[0], p[TILE_LD], p[2 * TILE_LD], p[3 * TILE_LD], sel);
  }
#pragma unroll
  for (int ib = 0; ib < IB; ++ib) {
    const float* p = x_tile + (4 * k) * TILE_LD + 16 * ib + i;
    b[ib] = split2_quad(p[0], p[TILE_LD], p[2 * TILE_LD], p[3 * TILE_LD], sel);
  }
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
      f32x4 acc = dw[ob][ib];
      acc = mfma_b16(a[ob].t[1], b[ib].t[0], acc);
      acc = mfma_b16(a[ob].t[0], b[ib].t[1], acc);
      acc = mfma_b16(a[ob].t[0], b[ib].t[0], acc);
      dw[ob][ib] = acc;
    }
}
