// Micro-benchmark: issue cost of the VALU instructions the shading kernels are
// made of (gfx950), alone and beside 16-cycle bf16 MFMAs.  Per wave and
// iteration: 128 instructions of one kind over 8 independent register chains;
// 1 / 2 / 4 waves per SIMD.  Output: cycles (2.4 GHz nominal) per instruction
// per SIMD.  Question it answers: does packed fp32
// (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) halve the issue cost per
// element, or does a packed instruction cost two plain ones?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { FMA, PK_FMA, PK_ADD, PK_MUL, ADD, CVT_BF16, EXP, RCP, PERM, AND, LSHL, MAXI, FMAMIX, CVT_F16,
       MUL, MAXF, MED3, CNDMASK, MOV, OR, XOR, ADDU, LSHLADD, LSHLADD64, MAD64, MULLO, MUL24, MAD24,
       CVTF16, CVTF32, CVTPKRNE, PKMAXI16, DPPADD, READLANE, BPERMUTE, FLOOR, CVTI32, BFE, ANDOR, LSHR,
       MINF, CMPSEL, DOT2BF16, DOT2CBF16, PERMSWAP32, PERMSWAP16, BITOP3, N_KIND };
static const char* NAMES[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_sub_f32",
                              "v_cvt_pk_bf16_f32", "v_exp_f32", "v_rcp_f32", "v_perm_b32", "v_and_b32",
                              "v_lshlrev_b32", "v_max_i32", "v_fma_mix_f32", "v_cvt_pkrtz_f16_f32",
                              "v_mul_f32", "v_max_f32", "v_med3_f32", "v_cndmask_b32 (vcc)", "v_mov_b32",
                              "v_or_b32", "v_xor_b32", "v_add_u32", "v_lshl_add_u32", "v_lshl_add_u64",
                              "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24",
                              "v_cvt_f16_f32", "v_cvt_f32_f16", "v_cvt_pk_f16_f32 (RNE)", "v_pk_max_i16",
                              "v_add_f32 dpp row_ror:4", "v_readlane_b32 (+v_mov)", "ds_bpermute_b32",
                              "v_floor_f32", "v_cvt_i32_f32", "v_bfe_u32", "v_and_or_b32",
                              "v_lshrrev_b32", "v_min_f32", "v_cmp_gt_f32 + v_cndmask",
                              "v_dot2_f32_bf16", "v_dot2c_f32_bf16", "v_permlane32_swap_b32",
                              "v_permlane16_swap_b32", "v_bitop3_b32"};

template <int KIND>
__device__ __forceinline__ void op(f32x2& x, const f32x2& y, const f32x2& z) {
  if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
  if constexpr (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
  if constexpr (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if constexpr (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if constexpr (KIND == ADD) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == CVT_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[0]));
  if constexpr (KIND == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[0]));
  if constexpr (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
  if constexpr (KIND == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == LSHL) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x[0]));
  if constexpr (KIND == MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == FMAMIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
  if constexpr (KIND == CVT_F16) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == MAXF) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == MINF) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
  if constexpr (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[0]) : "v"(y[0]) : "vcc");
  if constexpr (KIND == CMPSEL) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[0]) : "v"(y[0]) : "vcc");
  if constexpr (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(x) : "v"(y));
  if constexpr (KIND == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(y[0]), "v"(z[0]) : "vcc");
  if constexpr (KIND == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
  if constexpr (KIND == CVTF16) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x[0]));
  if constexpr (KIND == CVTF32) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x[0]));
  if constexpr (KIND == CVTPKRNE) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == PKMAXI16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
  if constexpr (KIND == DPPADD) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x[0]));
  if constexpr (KIND == READLANE) asm volatile("v_readlane_b32 s20, %0, 3\n v_mov_b32 %0, s20" : "+v"(x[0]) : : "s20");
  if constexpr (KIND == BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x[0]) : "v"(y[1]));
  if constexpr (KIND == FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(x[0]));
  if constexpr (KIND == CVTI32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[0]));
  if constexpr (KIND == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(x[0]));
  if constexpr (KIND == ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
  if constexpr (KIND == LSHR) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(x[0]));
  if constexpr (KIND == DOT2BF16) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
  if constexpr (KIND == DOT2CBF16) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
  if constexpr (KIND == PERMSWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[0]), "+v"(x[1]));
  if constexpr (KIND == PERMSWAP16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[0]), "+v"(x[1]));
  if constexpr (KIND == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[0]) : "v"(y[0]), "v"(z[0]));
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mf(f32x4& acc, const f32x4& a, const f32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// MFMAS: bf16 MFMAs per iteration, interleaved evenly with the 128 VALU ops
template <int KIND, int MFMAS>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  f32x2 v[8];
  f32x4 acc[4], a, b;
  for (int i = 0; i < 8; ++i) v[i] = f32x2{1.0f + threadIdx.x * 1e-3f + i, 0.5f + i};
  for (int i = 0; i < 4; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    a[i] = 1.0f + i;
    b[i] = 0.5f + i;
  }
  const f32x2 y = {0.999f, 1.001f}, z = {1e-3f, -1e-3f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 128; ++q) {
      if constexpr (MFMAS > 0)
        if (q % (128 / MFMAS) == 0) mf(acc[(q / (128 / MFMAS)) & 3], a, b);
      op<KIND>(v[q & 7], y, z);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int MFMAS>
float run(int waves_per_simd, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const dim3 grid(256), block(64 * 4 * waves_per_simd);
  hipLaunchKernelGGL((k<KIND, MFMAS>), grid, block, 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, MFMAS>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f * 2.4e9f / iters;   // cycles per iteration of all waves of a SIMD
}

template <int KIND>
void one(float* out) {
  printf("%-28s", NAMES[KIND]);
  for (int w = 1; w <= 4; w *= 2) {
    const float c = run<KIND, 0>(w, out);
    printf("  %dw/SIMD %6.2f cyc/inst", w, c / (128.0f * w));
  }
  // beside MFMAs: 16 and 32 bf16 MFMAs per 128 VALU, 4 waves per SIMD
  const float m16 = run<KIND, 16>(4, out), m32 = run<KIND, 32>(4, out);
  printf("  | 4w + 16 MFMA: %7.1f  + 32 MFMA: %7.1f cyc/iter/wave (VALU alone %7.1f, 16 MFMA alone 256-272)\n",
         m16 / 4, m32 / 4, run<KIND, 0>(4, out) / 4);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  one<FMA>(out); one<PK_FMA>(out); one<PK_ADD>(out); one<PK_MUL>(out); one<ADD>(out);
  one<CVT_BF16>(out); one<CVT_F16>(out); one<EXP>(out); one<RCP>(out); one<PERM>(out);
  one<AND>(out); one<LSHL>(out); one<MAXI>(out); one<FMAMIX>(out);
  one<MUL>(out); one<MAXF>(out); one<MINF>(out); one<MED3>(out); one<CNDMASK>(out); one<CMPSEL>(out);
  one<MOV>(out); one<OR>(out); one<XOR>(out); one<ADDU>(out); one<LSHLADD>(out); one<LSHLADD64>(out);
  one<MAD64>(out); one<MULLO>(out); one<MUL24>(out); one<MAD24>(out); one<CVTF16>(out);
  one<CVTF32>(out); one<CVTPKRNE>(out); one<PKMAXI16>(out); one<DPPADD>(out); one<READLANE>(out);
  one<BPERMUTE>(out); one<FLOOR>(out); one<CVTI32>(out); one<BFE>(out); one<ANDOR>(out); one<LSHR>(out);
  one<DOT2BF16>(out); one<DOT2CBF16>(out); one<PERMSWAP32>(out); one<PERMSWAP16>(out); one<BITOP3>(out);
  return 0;
}
