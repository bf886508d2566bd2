// "f16x2": fp32-grade contraction of the MLPs on the dense f16 MFMA pipe with
// HALF the matrix passes of bf16x3 (mfma_mlp_x3.h): three instead of six.
//
// Every fp32 operand is written as two f16 terms, the second one SCALED:
//     x ~ xh + xl * 2^-11,   xh = f16(x),  xl = f16((x - xh) * 2^11)
// (round to nearest even).  x - xh is exact in fp32 and at most half an ulp of
// xh, i.e. <= 2^-12 |x| for a normal xh; scaled by 2^11 it is a normal f16
// again whenever |x| >= 2^-13, and then the pair carries 22 significant bits:
// |x - (xh + xl 2^-11)| <= 2^-23 |x|, the rounding of an fp32 value itself.
// Below that (xl, and eventually xh, f16 subnormals: an f16-subnormal xh only
// shifts more of x into xl) the error is ABSOLUTE, <= 2^-36 per operand --
// measured 6e-10 on the sigma net's outputs for features of 1e-4
// (test_f16x2_sigma_mlp_is_fp32_grade).  Without the 2^11 scale the second
// term would be subnormal for every |x| < 2^-2 and hash-grid features
// (1e-4 ... 1e-1) would keep 11-14 bits only.
//
// A product x*w is accumulated in fp32 from
//     cross = xh*wl + xl*wh     (both exact products of f16 values, scale 2^11)
//     main  = xh*wh
// as  acc + cross * 2^-11 + main  (one fp32 multiply-add per accumulator element
// and layer to fold the cross sum in; what is dropped, xl*wl 2^-22, is <=
// 2^-24 |x*w|).  Same error class as bf16x3 (2^-23 per product, fp32
// accumulation) and as the f32-input MFMA chain: tests/test_gpu_parity.py
// ::test_f16x2_nets_are_fp32_grade.
//
// RANGE: the first terms are f16, so the range is that of the reference's own
// nets (tiny-cuda-nn computes them in fp16): |x| <= 65504 for every layer input
// and every weight.  The MLPs have no biases and ReLU is positively homogeneous,
// so ucsa_mlp_pack_h2 stores the first layer's weights times 2^-4 and the last
// layer's times 2^4: hidden activations 16 x smaller, same outputs (powers of
// two: exact) -- hidden layers overflow beyond 2^20 only.  Values beyond the
// range are NOT detected: the f16 conversion gives inf, the two partial sums
// give inf - inf = NaN, and a NaN pre-activation passes ReLU (v_max_f32) as 0,
// here as in every other mode of these kernels.  bf16x3 has fp32's range.
//
// Fragment layout: as mfma_mlp_x3.h with two terms,
// packed[(f * 2 + term) * 64 + lane], 16 bytes each.
#pragma once
#include "mfma_mlp_x3.h"

#define H2_LO_SCALE 2048.0f            // 2^11
#define H2_LO_UNSCALE 0.00048828125f   // 2^-11
#define H2_HIDDEN_SCALE 0.0625f        // first layer x 2^-4, last layer x 2^4

struct H2X {  // 8 values, two f16 terms each (2 values per dword)
  u32x4 t[2];
};
struct H2W {
  u32x4 t[2];
};

__device__ __forceinline__ f32x4 mfma_hh(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a),
                                                __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

__device__ __forceinline__ uint32_t f16_pair(float a, float b) {
  return __builtin_bit_cast(uint32_t, cvt_pk_h(a, b));
}

// residuals a - lo(p), b - hi(p) of an f16 pair p as one v_dot2_f32_f16 each
// (D = A.lo * B.lo + A.hi * B.hi + C with B = (-1, 0) resp. (0, -1): exact)
struct H2Sel {
  uint32_t lo, hi;   // (-1, 0) and (0, -1) as f16 pairs
};
__device__ __forceinline__ H2Sel h2_selectors() {
  H2Sel s{0x0000BC00u, 0xBC000000u};
  asm volatile("" : "+v"(s.lo), "+v"(s.hi));
  return s;
}
__device__ __forceinline__ float h2_resid_lo(uint32_t p, float a, const H2Sel& s) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, p),
                                __builtin_bit_cast(half2_t, s.lo), a, false);
}
__device__ __forceinline__ float h2_resid_hi(uint32_t p, float b, const H2Sel& s) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, p),
                                __builtin_bit_cast(half2_t, s.hi), b, false);
}

// two values into dword `d` of both terms
__device__ __forceinline__ void h2_split_pair(float a, float b, H2X& x, int d,
                                              const H2Sel& sel) {
  const uint32_t p0 = f16_pair(a, b);
  const float ra = h2_resid_lo(p0, a, sel), rb = h2_resid_hi(p0, b, sel);
  x.t[0][d] = p0;
  x.t[1][d] = f16_pair(ra * H2_LO_SCALE, rb * H2_LO_SCALE);
}

// two accumulator blocks (ReLU) -> one 32-wide k-step operand
__device__ __forceinline__ H2X h2_chain_relu(f32x4 lo, f32x4 hi, const H2Sel& sel) {
  H2X x;
  h2_split_pair(relu1(lo[0]), relu1(lo[1]), x, 0, sel);
  h2_split_pair(relu1(lo[2]), relu1(lo[3]), x, 1, sel);
  h2_split_pair(relu1(hi[0]), relu1(hi[1]), x, 2, sel);
  h2_split_pair(relu1(hi[2]), relu1(hi[3]), x, 3, sel);
  return x;
}

__device__ __forceinline__ H2W h2_frag(const void* packed, int f, uint32_t lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(packed) + (f * 2) * 64 + lane;
  H2W w;
  w.t[0] = p[0];
  w.t[1] = p[64];
  return w;
}

// the two halves of a product: `cross` collects the scaled second-order terms
// of a layer's k-steps, `main` the first-order ones on top of the folded sum
__device__ __forceinline__ f32x4 h2_cross(const H2W& w, const H2X& x, f32x4 cross) {
  cross = mfma_hh(w.t[1], x.t[0], cross);
  return mfma_hh(w.t[0], x.t[1], cross);
}
__device__ __forceinline__ f32x4 h2_fold(f32x4 cross) {
  return f32x4{cross[0] * H2_LO_UNSCALE, cross[1] * H2_LO_UNSCALE,
               cross[2] * H2_LO_UNSCALE, cross[3] * H2_LO_UNSCALE};
}
__device__ __forceinline__ f32x4 h2_main(const H2W& w, const H2X& x, f32x4 acc) {
  return mfma_hh(w.t[0], x.t[0], acc);
}
// W * x for a one-k-step layer, and Wa * xa + Wb * xb for a two-k-step one
__device__ __forceinline__ f32x4 h2_mul1(const H2W& w, const H2X& x) {
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  return h2_main(w, x, h2_fold(h2_cross(w, x, z4)));
}
__device__ __forceinline__ f32x4 h2_mul2(const H2W& wa, const H2X& xa, const H2W& wb,
                                         const H2X& xb) {
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc = h2_fold(h2_cross(wb, xb, h2_cross(wa, xa, z4)));
  acc = h2_main(wa, xa, acc);
  return h2_main(wb, xb, acc);
}
