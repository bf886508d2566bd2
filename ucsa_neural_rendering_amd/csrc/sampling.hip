// Hierarchical resampling (first half): coarse weights ->
// pdf over the interior bins -> inverse CDF at caller-supplied uniforms.
// reference nr4seg/nerf/renderer_semantics.py:182-207 and sample_pdf :10-46.
//
// One wave per ray; the ray's z / weights / cdf / bins live in LDS.  Products
// and sums are wave scans (fp32), so cdf differs from the CPU oracle's
// sequential double-accumulated cumsum by fp32 round-off; the inverse CDF is
// continuous, tests state the tolerance.
#include <type_traits>

#include "ucsa_common.h"
#include "wave_ops.h"

#define RS_WAVES 4

extern __shared__ __attribute__((aligned(16))) float rs_smem[];

// Bitonic sort of 64 R values held by a wave in registers: element r * 64 + lane in
// x[r].  Exchanges across lanes by __shfl_xor, across registers in the lane -- no LDS
// round trip and no fence per stage (the LDS network below costs 28 dependent stages of
// two reads, two conditional writes and a fence for 128 values; round 6).
template <int R>
__device__ __forceinline__ void wave_bitonic_sort(float (&x)[R], uint32_t lane) {
#pragma unroll
  for (uint32_t k = 2; k <= 64u * R; k <<= 1) {
#pragma unroll
    for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
      if (jj >= 64u) {   // partner in the same lane: registers r and r ^ (jj / 64)
        const uint32_t dr = jj >> 6;
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)R; ++r) {
          if (r & dr) continue;
          const bool up = (((r << 6) | lane) & k) == 0u;
          const float a = x[r], b = x[r | dr];
          const float mn = fminf(a, b), mx = fmaxf(a, b);
          x[r] = up ? mn : mx;
          x[r | dr] = up ? mx : mn;
        }
      } else {           // partner in lane ^ jj, same register
#pragma unroll
        for (uint32_t r = 0; r < (uint32_t)R; ++r) {
          const bool up = (((r << 6) | lane) & k) == 0u;
          const float a = x[r], b = __shfl_xor(a, (int)jj, 64);
          const bool keep_min = ((lane & jj) == 0u) == up;
          x[r] = keep_min ? fminf(a, b) : fmaxf(a, b);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(64 * RS_WAVES)
k_resample(const float* __restrict__ z, const float* __restrict__ sigma,
           const float* __restrict__ u, uint32_t N, uint32_t T, uint32_t t,
           uint32_t tpad, float density_scale, float* __restrict__ new_z) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t r = blockIdx.x * RS_WAVES + wid;
  if (r >= N) return;  // whole wave exits; no block barriers below
  // per-wave LDS: zs[T], cdf[T], bins[T], us[tpad]
  float* zs = rs_smem + (size_t)wid * (3 * T + tpad);
  float* cdf = zs + T;
  float* bins = cdf + T;
  const float* zr = z + (size_t)r * T;
  const float* sr = sigma + (size_t)r * T;

  for (uint32_t i = lane; i < T; i += 64) zs[i] = zr[i];
  __builtin_amdgcn_wave_barrier();

  // pass 1: weights w_i = alpha_i * prod_{j<i}(1 - alpha_j + 1e-15)
  // (kept in cdf[] temporarily), and the pdf normaliser.
  float carry = 1.0f;  // exclusive transmittance entering this 64-chunk
  float wsum = 0.0f;
  for (uint32_t base = 0; base < T; base += 64) {
    const uint32_t i = base + lane;
    float alpha = 0.0f, zi = 0.0f, delta = 0.0f;
    if (i < T) {
      zi = zs[i];
      delta = (i + 1 < T) ? zs[i + 1] - zi : 1e10f;
      alpha = 1.0f - expf(-delta * density_scale * sr[i]);
      if (i + 1 < T) bins[i] = zi + 0.5f * delta;
    }
    const float fac = (i < T) ? (1.0f - alpha + 1e-15f) : 1.0f;
    const float incl = wave_incl_scan_mul_dpp(fac);
    const float excl = wave_shift_up1(incl, 1.0f);
    const float w = alpha * (carry * excl);
    if (i < T) cdf[i] = w;
    if (i >= 1 && i + 1 < T) wsum += w + 1e-5f;
    carry = carry * wave_last(incl);
  }
  wsum = wave_sum(wsum);
  __builtin_amdgcn_wave_barrier();

  // pass 2: cdf[0] = 0, cdf[k] = sum_{i=1..k} pdf_i, k = 1..T-2  (T-1 entries)
  float run = 0.0f;
  for (uint32_t base = 0; base + 1 < T; base += 64) {
    const uint32_t k = base + lane;  // output index
    float p = 0.0f;
    if (k >= 1 && k + 1 < T) p = (cdf[k] + 1e-5f) / wsum;
    const float incl = wave_incl_scan_add_dpp(p);
    __builtin_amdgcn_wave_barrier();
    if (k + 1 < T) cdf[k] = run + incl;
    run = run + wave_last(incl);
  }
  __builtin_amdgcn_wave_barrier();

  // pass 3: invert.  n_cdf = T-1 entries; searchsorted(right=True).
  // The uniforms are sorted first (bitonic network in LDS): the reference
  // consumes new_z only through sort/merge (renderer_semantics.py:221-222),
  // so their order within the ray is unobservable, and ascending fine samples
  // make consecutive lanes neighbours in space (coherent gathers, and the
  // hash-grid backward can combine runs instead of contending on atomics).
  const uint32_t n_cdf = T - 1;
  const float* ur = u + (size_t)r * t;
  float* out = new_z + (size_t)r * t;
  float* us = bins + T;  // [tpad]
  // up to 256 uniforms per ray: sorted in registers (1, 2 or 4 per lane), then handed
  // to the inversion loop through the same LDS array
  if (tpad <= 256u) {
    auto sort_in_regs = [&](auto tag) {
      constexpr int R = decltype(tag)::value;
      float x[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t q = (uint32_t)r * 64u + lane;
        x[r] = q < t ? ur[q] : INFINITY;
      }
      wave_bitonic_sort<R>(x, lane);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t q = (uint32_t)r * 64u + lane;
        if (q < tpad) us[q] = x[r];
      }
    };
    if (tpad <= 64u) sort_in_regs(std::integral_constant<int, 1>{});
    else if (tpad == 128u) sort_in_regs(std::integral_constant<int, 2>{});
    else sort_in_regs(std::integral_constant<int, 4>{});
    __builtin_amdgcn_wave_barrier();
  } else {
  for (uint32_t q = lane; q < tpad; q += 64) us[q] = q < t ? ur[q] : INFINITY;
  __builtin_amdgcn_wave_barrier();
  for (uint32_t k = 2; k <= tpad; k <<= 1) {
    for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
      for (uint32_t p = lane; p < (tpad >> 1); p += 64) {
        // p-th compare-exchange of this stage
        const uint32_t lo_i = ((p & ~(jj - 1)) << 1) | (p & (jj - 1));
        const uint32_t hi_i = lo_i | jj;
        const bool up = (lo_i & k) == 0;
        const float a0 = us[lo_i], a1 = us[hi_i];
        if ((a0 > a1) == up) { us[lo_i] = a1; us[hi_i] = a0; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  }
  for (uint32_t q = lane; q < t; q += 64) {
    const float uu = us[q];
    uint32_t lo = 0, hi = n_cdf;  // first index with cdf[idx] > uu
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
    }
    const uint32_t below = lo > 0 ? lo - 1 : 0;
    const uint32_t above = lo < n_cdf - 1 ? lo : n_cdf - 1;
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = bins[below], b1 = bins[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    out[q] = b0 + (uu - c0) / denom * (b1 - b0);
  }
}

extern "C" int32_t ucsa_resample(const float* z, const float* sigma,
                                 const float* u, uint32_t N, uint32_t T,
                                 uint32_t t, float density_scale, float* new_z,
                                 void* stream) {
  UCSA_CHECK_ARG(z, 0);
  UCSA_CHECK_ARG(sigma, 1);
  UCSA_CHECK_ARG(u, 2);
  UCSA_CHECK_ARG(T >= 3 && T <= 4096, 4);
  UCSA_CHECK_ARG(new_z, 7);
  if (N == 0 || t == 0) return 0;
  UCSA_CHECK_ARG(t <= 4096, 5);
  uint32_t tpad = 2;
  while (tpad < t) tpad <<= 1;
  const size_t smem = (size_t)RS_WAVES * (3 * T + tpad) * sizeof(float);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_resample, dim3(ucsa_div_up(N, RS_WAVES)),
                     dim3(64 * RS_WAVES), smem, (hipStream_t)stream, z, sigma,
                     u, N, T, t, tpad, density_scale, new_z);
  return ucsa_launch_status();
}
