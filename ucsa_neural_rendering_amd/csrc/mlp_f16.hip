// fp16-MFMA inference option: weight packing and the sigma MLP.
// (call sites: reference nr4seg/nerf/network_tcnn_semantics.py:48-58,133-139;
// tiny-cuda-nn itself computes these nets in fp16 with fp32 accumulation.)
#include <cstdlib>
#include "mfma_mlp_h2.h"

__device__ __forceinline__ uint32_t chain_col_h(uint32_t s, uint32_t g,
                                                uint32_t e) {
  return 16u * (2u * s + (e >> 2)) + 4u * g + (e & 3u);
}

// the weight at k-slot e of lane l of A fragment f (both 16-bit layouts)
__device__ __forceinline__ float pack_h_value(int kind,
                                              const float* __restrict__ params,
                                              uint32_t f, uint32_t l,
                                              uint32_t e) {
  const uint32_t g = l >> 4, i = l & 15u;
  float v = 0.f;
  if (kind == UCSA_MLP_SIGMA) {
    if (f < 4) {  // L1: feature 2*(4q+g)+c, e = 2q+c
      v = params[(f * 16 + i) * 32 + 2u * (4u * (e >> 1) + g) + (e & 1u)];
    } else {
      v = params[64 * 32 + i * 64 + chain_col_h(f - 4, g, e)];
    }
  } else if (kind == UCSA_MLP_COLOR) {
    if (f < 4) {
      uint32_t col;
      if (e < 4) col = 4u * g + e;  // SH
      else { const uint32_t m = 4u * g + (e - 4); col = m == 0 ? 31u : 15u + m; }
      v = params[(f * 16 + i) * 32 + col];
    } else if (f < 12) {
      const uint32_t rb = (f - 4) >> 1, s = (f - 4) & 1u;
      v = params[64 * 32 + (rb * 16 + i) * 64 + chain_col_h(s, g, e)];
    } else {
      v = params[64 * 32 + 64 * 64 + i * 64 + chain_col_h(f - 12, g, e)];
    }
  } else {
    if (f < 4) {
      if (e < 4) {
        const uint32_t m = 4u * g + e;
        v = params[(f * 16 + i) * 16 + (m == 0 ? 15u : m - 1u)];
      }  // e >= 4: K padding, zero
    } else {
      const uint32_t rb = (f - 4) >> 1, s = (f - 4) & 1u;
      v = params[64 * 16 + (rb * 16 + i) * 64 + chain_col_h(s, g, e)];
    }
  }
  return v;
}

__global__ void k_mlp_pack_f16(int kind, const float* __restrict__ params,
                               _Float16* __restrict__ packed, uint32_t n_total) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_total) return;
  packed[idx] =
      (_Float16)pack_h_value(kind, params, idx >> 9, (idx >> 3) & 63u, idx & 7u);
}

// bf16x3 (mfma_mlp_x3.h): the same fragments, each weight as three bf16 terms
// w = w0 + w1 + w2 (exact), fragment (f, term) at [(f * 3 + term) * 64 + lane]
__global__ void k_mlp_pack_x3(int kind, const float* __restrict__ params,
                              uint16_t* __restrict__ packed, uint32_t n_total) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_total) return;
  const uint32_t e = idx & 7u, l = (idx >> 3) & 63u, f = idx >> 9;
  float r = pack_h_value(kind, params, f, l, e);
#pragma unroll
  for (uint32_t term = 0; term < 3; ++term) {
    const uint32_t bits = bf16_pair(r, 0.f) & 0xFFFFu;
    packed[((f * 3 + term) * 64 + l) * 8 + e] = (uint16_t)bits;
    r -= __uint_as_float(bits << 16);
  }
}

// f16x2 (mfma_mlp_h2.h): each weight as two f16 terms, the second scaled by
// 2^11; fragment (f, term) at [(f * 2 + term) * 64 + lane].  First-layer
// fragments carry the weights x 2^-4, last-layer ones x 2^4 (exact; the hidden
// activations are 16 x smaller, the outputs unchanged).
// range_host (may be NULL): the largest |value| this pack converted to f16, as
// its fp32 bit pattern (non-negative floats order like unsigned integers; a
// NaN's pattern is above every finite one).  The waves raise a DEVICE word
// (scratch[0]) with device-scope atomics; the last workgroup to finish
// (scratch[1] counts them) publishes it with ONE system-scope atomic max to
// range_host, which may live in pinned host memory: the host then simply reads
// it -- no copy, stream, event or wait on its side.  (One system-scope atomic per
// WAVE, ~110 per pack over PCIe, cost 0.1 ms per pack.)  Neither word is reset
// here: the host sees the worst pack since it last cleared range_host.
__global__ void k_mlp_pack_h2(int kind, const float* __restrict__ params,
                              _Float16* __restrict__ packed, uint32_t n_total,
                              uint32_t* __restrict__ range_host,
                              uint32_t* __restrict__ scratch) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t b = 0u;
  if (idx < n_total) {
    const uint32_t e = idx & 7u, l = (idx >> 3) & 63u, f = idx >> 9;
    const bool first = f < 4;
    const bool last = kind == UCSA_MLP_COLOR ? f >= 12 : f >= 4;
    const float v = pack_h_value(kind, params, f, l, e) *
                    (first ? H2_HIDDEN_SCALE : (last ? 1.0f / H2_HIDDEN_SCALE : 1.0f));
    const _Float16 hi = (_Float16)v;
    packed[((f * 2 + 0) * 64 + l) * 8 + e] = hi;
    packed[((f * 2 + 1) * 64 + l) * 8 + e] = (_Float16)((v - (float)hi) * H2_LO_SCALE);
    b = __float_as_uint(v) & 0x7FFFFFFFu;
  }
  if (!range_host) return;   // (kernel-uniform)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)b, d, 64);
    b = o > b ? o : b;
  }
  if ((threadIdx.x & 63u) == 0u) atomicMax(&scratch[0], b);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&scratch[1], 1u) == gridDim.x - 1u) {   // the last workgroup
      scratch[1] = 0u;
      const uint32_t m = atomicMax(&scratch[0], 0u);
      __hip_atomic_fetch_max(range_host, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

extern "C" uint32_t ucsa_mlp_pack_f16_halves(int32_t kind, uint32_t n_classes) {
  const uint32_t nrb = ((n_classes ? n_classes : 1) + 15u) / 16u;
  const uint32_t frags = kind == UCSA_MLP_SIGMA ? SIGMA_H_FRAGS
                         : kind == UCSA_MLP_COLOR ? COLOR_H_FRAGS
                                                  : SEM_H_FRAGS(nrb);
  return frags * 64 * 8;
}

extern "C" int32_t ucsa_mlp_pack_f16(int32_t kind, const float* params,
                                     void* packed_half, uint32_t n_classes,
                                     void* stream) {
  UCSA_CHECK_ARG(kind >= 0 && kind <= 2, 0);
  UCSA_CHECK_ARG(params, 1);
  UCSA_CHECK_ARG(packed_half, 2);
  UCSA_CHECK_ARG(kind != UCSA_MLP_SEM || (n_classes >= 1 && n_classes <= 61), 3);
  const uint32_t n_total = ucsa_mlp_pack_f16_halves(kind, n_classes);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_mlp_pack_f16, dim3(ucsa_div_up(n_total, 256)), dim3(256),
                     0, (hipStream_t)stream, (int)kind, params,
                     (_Float16*)packed_half, n_total);
  return ucsa_launch_status();
}

extern "C" uint32_t ucsa_mlp_pack_h2_bytes(int32_t kind, uint32_t n_classes) {
  return ucsa_mlp_pack_f16_halves(kind, n_classes) * 2u * 2u;
}

extern "C" int32_t ucsa_mlp_pack_h2(int32_t kind, const float* params,
                                    void* packed_h2, uint32_t n_classes,
                                    void* stream) {
  UCSA_CHECK_ARG(kind >= 0 && kind <= 2, 0);
  UCSA_CHECK_ARG(params, 1);
  UCSA_CHECK_ARG(packed_h2, 2);
  UCSA_CHECK_ARG(kind != UCSA_MLP_SEM || (n_classes >= 1 && n_classes <= 61), 3);
  const uint32_t n_total = ucsa_mlp_pack_f16_halves(kind, n_classes);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_mlp_pack_h2, dim3(ucsa_div_up(n_total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (int)kind, params, (_Float16*)packed_h2,
                     n_total, (uint32_t*)nullptr, (uint32_t*)nullptr);
  return ucsa_launch_status();
}

// ucsa_mlp_pack_h2 that also records the range of what it packed: *range_bits
// (uint32 in pinned, device-mapped host memory -- or device memory --, zeroed by
// the caller) = max over every value converted to f16 of its |.| as an fp32 bit
// pattern, accumulated over calls; scratch: two device uint32, zeroed by the
// caller once.  A pattern >= 0x477FE000 (65504.0f) means a weight left f16x2's
// range (or is not finite).
extern "C" int32_t ucsa_mlp_pack_h2_checked(int32_t kind, const float* params,
                                            void* packed_h2, uint32_t n_classes,
                                            uint32_t* range_bits, uint32_t* scratch,
                                            void* stream) {
  UCSA_CHECK_ARG(kind >= 0 && kind <= 2, 0);
  UCSA_CHECK_ARG(params, 1);
  UCSA_CHECK_ARG(packed_h2, 2);
  UCSA_CHECK_ARG(kind != UCSA_MLP_SEM || (n_classes >= 1 && n_classes <= 61), 3);
  UCSA_CHECK_ARG(range_bits, 4);
  UCSA_CHECK_ARG(scratch, 5);
  const uint32_t n_total = ucsa_mlp_pack_f16_halves(kind, n_classes);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_mlp_pack_h2, dim3(ucsa_div_up(n_total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (int)kind, params, (_Float16*)packed_h2,
                     n_total, range_bits, scratch);
  return ucsa_launch_status();
}

extern "C" uint32_t ucsa_mlp_pack_x3_bytes(int32_t kind, uint32_t n_classes) {
  return ucsa_mlp_pack_f16_halves(kind, n_classes) * 2u * 3u;
}

extern "C" int32_t ucsa_mlp_pack_x3(int32_t kind, const float* params,
                                    void* packed_x3, uint32_t n_classes,
                                    void* stream) {
  UCSA_CHECK_ARG(kind >= 0 && kind <= 2, 0);
  UCSA_CHECK_ARG(params, 1);
  UCSA_CHECK_ARG(packed_x3, 2);
  UCSA_CHECK_ARG(kind != UCSA_MLP_SEM || (n_classes >= 1 && n_classes <= 61), 3);
  const uint32_t n_total = ucsa_mlp_pack_f16_halves(kind, n_classes);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_mlp_pack_x3, dim3(ucsa_div_up(n_total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (int)kind, params, (uint16_t*)packed_x3,
                     n_total);
  return ucsa_launch_status();
}

// ---------------------------------------------------------------------------
// Transposed fp16 A fragments for dX = W^T dY of the colour / semantics nets
// (training with precision="fp16": composite_bwd.hip, k_shade_bwd<.., HALF>).
// Rows i of a fragment = neurons of the layer's INPUT (natural order, so the
// result lands in the accumulator layout of the forward activations it
// gates); k-slot (s, g, e) = the layer's OUTPUT neuron chain_col_h(s, g, e),
// i.e. what lane (g, j) holds of dY after packing two accumulator blocks into
// a half8 (chain without ReLU).
//   colour: [L3^T 4 frags (k-slots e<4 = output row 4g+e, rest 0)
//            | L2^T 8 (rb, s) | L1^T restricted to the 16 h-row slots: 2]  = 14
//   sem   : [L2^T 4*NS (rb, s), NS = ceil(nrb/2), classes >= 16*nrb -> 0
//            | L1^T (h-row slots) 2]
// ---------------------------------------------------------------------------
__device__ __forceinline__ float pack_t_value(int kind, const float* __restrict__ params,
                                              uint32_t f, uint32_t l, uint32_t e,
                                              uint32_t nrb) {
  const uint32_t g = l >> 4, i = l & 15u;
  float v = 0.f;
  if (kind == UCSA_MLP_SIGMA) {
    // [L2^T 4 frags (k-slots e<4 = output row 4g+e, rest 0) | L1^T (rb, s): 4]
    if (f < 4) {
      if (e < 4) v = params[64 * 32 + (4u * g + e) * 64 + 16u * f + i];
    } else {
      const uint32_t rb = (f - 4) >> 1, sidx = (f - 4) & 1u;
      v = params[chain_col_h(sidx, g, e) * 32 + 16u * rb + i];
    }
  } else if (kind == UCSA_MLP_COLOR) {
    if (f < 4) {
      if (e < 4) v = params[64 * 32 + 64 * 64 + (4u * g + e) * 64 + 16u * f + i];
    } else if (f < 12) {
      const uint32_t rb = (f - 4) >> 1, sidx = (f - 4) & 1u;
      v = params[64 * 32 + chain_col_h(sidx, g, e) * 64 + 16u * rb + i];
    } else {
      const uint32_t col = i == 0 ? 31u : 15u + i;
      v = params[chain_col_h(f - 12, g, e) * 32 + col];
    }
  } else {
    const uint32_t ns = (nrb + 1u) / 2u;
    if (f < 4 * ns) {
      const uint32_t rb = f / ns, sidx = f % ns;
      const uint32_t n = chain_col_h(sidx, g, e);
      if (n < 16u * nrb) v = params[64 * 16 + n * 64 + 16u * rb + i];
    } else {
      const uint32_t col = i == 0 ? 15u : i - 1u;
      v = params[chain_col_h(f - 4 * ns, g, e) * 16 + col];
    }
  }
  return v;
}

__global__ void k_mlp_pack_t_f16(int kind, const float* __restrict__ params,
                                 _Float16* __restrict__ packed,
                                 uint32_t n_total, uint32_t nrb) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_total) return;
  packed[idx] = (_Float16)pack_t_value(kind, params, idx >> 9, (idx >> 3) & 63u, idx & 7u, nrb);
}

// the same transposed fragments as three exact bf16 terms (layout of
// k_mlp_pack_x3): dX = W^T dY of k_shade_bwd's bf16x2 mode reads terms 0, 1
__global__ void k_mlp_pack_t_x3(int kind, const float* __restrict__ params,
                                uint16_t* __restrict__ packed, uint32_t n_total,
                                uint32_t nrb) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_total) return;
  const uint32_t e = idx & 7u, l = (idx >> 3) & 63u, f = idx >> 9;
  float r = pack_t_value(kind, params, f, l, e, nrb);
#pragma unroll
  for (uint32_t term = 0; term < 3; ++term) {
    const uint32_t bits = bf16_pair(r, 0.f) & 0xFFFFu;
    packed[((f * 3 + term) * 64 + l) * 8 + e] = (uint16_t)bits;
    r -= __uint_as_float(bits << 16);
  }
}

extern "C" uint32_t ucsa_mlp_pack_t_f16_halves(int32_t kind, uint32_t n_classes) {
  const uint32_t nrb = ((n_classes ? n_classes : 1) + 15u) / 16u;
  const uint32_t frags = kind == UCSA_MLP_COLOR ? 14u : 4u * ((nrb + 1u) / 2u) + 2u;
  return frags * 64 * 8;
}

extern "C" int32_t ucsa_mlp_pack_t_f16(int32_t kind, const float* params,
                                       void* packed_half, uint32_t n_classes,
                                       void* stream) {
  UCSA_CHECK_ARG(kind == UCSA_MLP_COLOR || kind == UCSA_MLP_SEM, 0);
  UCSA_CHECK_ARG(params, 1);
  UCSA_CHECK_ARG(packed_half, 2);
  UCSA_CHECK_ARG(kind != UCSA_MLP_SEM || (n_classes >= 1 && n_classes <= 61), 3);
  const uint32_t n_total = ucsa_mlp_pack_t_f16_halves(kind, n_classes);
  const uint32_t nrb = ((n_classes ? n_classes : 1) + 15u) / 16u;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_mlp_pack_t_f16, dim3(ucsa_div_up(n_total, 256)), dim3(256),
                     0, (hipStream_t)stream, (int)kind, params,
                     (_Float16*)packed_half, n_total, nrb);
  return ucsa_launch_status();
}

#define SIGMA_T_FRAGS 8

extern "C" uint32_t ucsa_mlp_pack_t_x3_bytes(int32_t kind, uint32_t n_classes) {
  if (kind == UCSA_MLP_SIGMA) return SIGMA_T_FRAGS * 64 * 8 * 2u * 3u;
  return ucsa_mlp_pack_t_f16_halves(kind, n_classes) * 2u * 3u;
}

extern "C" int32_t ucsa_mlp_pack_t_x3(int32_t kind, const float* params,
                                      void* packed_x3, uint32_t n_classes,
                                      void* stream) {
  UCSA_CHECK_ARG(kind >= 0 && kind <= 2, 0);
  UCSA_CHECK_ARG(params, 1);
  UCSA_CHECK_ARG(packed_x3, 2);
  UCSA_CHECK_ARG(kind != UCSA_MLP_SEM || (n_classes >= 1 && n_classes <= 61), 3);
  const uint32_t n_total = ucsa_mlp_pack_t_x3_bytes(kind, n_classes) / 6u;
  const uint32_t nrb = ((n_classes ? n_classes : 1) + 15u) / 16u;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_mlp_pack_t_x3, dim3(ucsa_div_up(n_total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (int)kind, params, (uint16_t*)packed_x3,
                     n_total, nrb);
  return ucsa_launch_status();
}

// sigma MLP, fp16 inputs/weights, fp32 accumulate, fp32 outputs.
// 6 MFMAs per 16 samples: load/store bound, 8 column blocks per iteration.
#define SIGH_UNROLL 8

typedef _Float16 sig_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void feat_to_h(const float2 v, _Float16& a, _Float16& b) {
  a = (_Float16)v.x;
  b = (_Float16)v.y;
}
__device__ __forceinline__ void feat_to_h(const sig_half2 v, _Float16& a, _Float16& b) {
  a = v[0];
  b = v[1];
}

// FT: float2 features (rounded to half here) or half2 features (already
// rounded by ucsa_hashgrid_encode_rays_h16): the same operands either way
template <typename FT>
__global__ void __launch_bounds__(256)
k_sigma_mlp_f16(const FT* __restrict__ feat, const void* __restrict__ packed,
                uint64_t M, float* __restrict__ h, float* __restrict__ sigma,
                const uint32_t* __restrict__ slot) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  half8 w1[4], w2[2];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) w1[rb] = frag_h(packed, rb, lane);
#pragma unroll
  for (int s = 0; s < 2; ++s) w2[s] = frag_h(packed, 4 + s, lane);
  const uint64_t span = 16 * SIGH_UNROLL;
  for (uint64_t base = wave * span; base < M; base += nwaves * span) {
    half8 xin[SIGH_UNROLL];
#pragma unroll
    for (int sb = 0; sb < SIGH_UNROLL; ++sb) {
      uint64_t m = base + sb * 16 + j;
      if (m >= M) m = M - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        _Float16 fa, fb;
        feat_to_h(feat[(uint64_t)(4 * q + g) * M + m], fa, fb);
        xin[sb][2 * q] = fa;
        xin[sb][2 * q + 1] = fb;
      }
    }
#pragma unroll
    for (int sb = 0; sb < SIGH_UNROLL; ++sb) {
      const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 a1[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a1[rb] = mfma_h(w1[rb], xin[sb], z4);
      f32x4 out = mfma_h(w2[0], chain_relu_h(a1[0], a1[1]), z4);
      out = mfma_h(w2[1], chain_relu_h(a1[2], a1[3]), out);
      const uint64_t m = base + sb * 16 + j;
      if (m < M) {
        const uint64_t mo = slot ? slot[m] : m;   // see k_sigma_mlp (mlp.hip)
        *reinterpret_cast<f32x4*>(h + mo * 16 + 4 * g) = out;
        if (g == 0) sigma[mo] = expf(out[0]);
      }
    }
  }
}

extern "C" int32_t ucsa_sigma_mlp_fwd_f16(const float* feat,
                                          const void* packed_sigma_half,
                                          uint32_t M, uint32_t n_levels,
                                          float* h, float* sigma,
                                          void* stream) {
  UCSA_CHECK_ARG(feat, 0);
  UCSA_CHECK_ARG(packed_sigma_half, 1);
  UCSA_CHECK_ARG(n_levels == 16, 3);
  UCSA_CHECK_ARG(h && sigma, 4);
  if (M == 0) return 0;
  const uint32_t need = ucsa_div_up(M, 16 * SIGH_UNROLL * 4);
  const uint32_t blocks = need < 2048u ? need : 2048u;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sigma_mlp_f16<float2>, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)feat, packed_sigma_half,
                     (uint64_t)M, h, sigma, (const uint32_t*)nullptr);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_sigma_mlp_fwd_f16_h(const void* feat_half,
                                            const void* packed_sigma_half,
                                            uint32_t M, uint32_t n_levels,
                                            float* h, float* sigma,
                                            void* stream) {
  UCSA_CHECK_ARG(feat_half, 0);
  UCSA_CHECK_ARG(packed_sigma_half, 1);
  UCSA_CHECK_ARG(n_levels == 16, 3);
  UCSA_CHECK_ARG(h && sigma, 4);
  if (M == 0) return 0;
  const uint32_t need = ucsa_div_up(M, 16 * SIGH_UNROLL * 4);
  const uint32_t blocks = need < 2048u ? need : 2048u;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sigma_mlp_f16<sig_half2>, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, (const sig_half2*)feat_half,
                     packed_sigma_half, (uint64_t)M, h, sigma, (const uint32_t*)nullptr);
  return ucsa_launch_status();
}

// sigma MLP on the bf16 MFMA pipe with three-term operands (mfma_mlp_x3.h):
// fp32-grade h and sigma, 36 16-cycle MFMAs per 16 samples instead of 48
// 32-cycle f32-input ones.
#define SIGX_UNROLL 4

__global__ void __launch_bounds__(256)
k_sigma_mlp_x3(const float2* __restrict__ feat, const void* __restrict__ packed,
               uint64_t M, float* __restrict__ h, float* __restrict__ sigma,
               const uint32_t* __restrict__ slot) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  W3 w1[4], w2[2];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) w1[rb] = frag_x3(packed, rb, lane);
#pragma unroll
  for (int s = 0; s < 2; ++s) w2[s] = frag_x3(packed, 4 + s, lane);
  const X3Sel sel = x3_selectors();   // the two dot2 selector constants, once
  const uint64_t span = 16 * SIGX_UNROLL;
  for (uint64_t base = wave * span; base < M; base += nwaves * span) {
    float2 raw[SIGX_UNROLL][4];
#pragma unroll
    for (int sb = 0; sb < SIGX_UNROLL; ++sb) {
      uint64_t m = base + sb * 16 + j;
      if (m >= M) m = M - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[sb][q] = feat[(uint64_t)(4 * q + g) * M + m];
    }
#pragma unroll
    for (int sb = 0; sb < SIGX_UNROLL; ++sb) {
      const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
      X3 xin;
#pragma unroll
      for (int q = 0; q < 4; ++q) split_pair(raw[sb][q].x, raw[sb][q].y, xin, q, sel);
      f32x4 a1[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a1[rb] = mfma_x3(w1[rb], xin, z4);
      f32x4 out = mfma_x3(w2[0], chain_relu_x3(a1[0], a1[1], sel), z4);
      out = mfma_x3(w2[1], chain_relu_x3(a1[2], a1[3], sel), out);
      const uint64_t m = base + sb * 16 + j;
      if (m < M) {
        const uint64_t mo = slot ? slot[m] : m;   // see k_sigma_mlp (mlp.hip)
        *reinterpret_cast<f32x4*>(h + mo * 16 + 4 * g) = out;
        if (g == 0) sigma[mo] = expf(out[0]);
      }
    }
  }
}

// the same on the f16 pipe with two-term operands (mfma_mlp_h2.h, "f16x2"):
// 18 MFMAs per 16 samples instead of 36
__global__ void __launch_bounds__(256)
k_sigma_mlp_h2(const float2* __restrict__ feat, const void* __restrict__ packed,
               uint64_t M, float* __restrict__ h, float* __restrict__ sigma,
               const uint32_t* __restrict__ slot) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  H2W w1[4], w2[2];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) w1[rb] = h2_frag(packed, rb, lane);
#pragma unroll
  for (int s = 0; s < 2; ++s) w2[s] = h2_frag(packed, 4 + s, lane);
  const H2Sel sel = h2_selectors();
  const uint64_t span = 16 * SIGX_UNROLL;
  for (uint64_t base = wave * span; base < M; base += nwaves * span) {
    float2 raw[SIGX_UNROLL][4];
#pragma unroll
    for (int sb = 0; sb < SIGX_UNROLL; ++sb) {
      uint64_t m = base + sb * 16 + j;
      if (m >= M) m = M - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[sb][q] = feat[(uint64_t)(4 * q + g) * M + m];
    }
#pragma unroll
    for (int sb = 0; sb < SIGX_UNROLL; ++sb) {
      H2X xin;
#pragma unroll
      for (int q = 0; q < 4; ++q) h2_split_pair(raw[sb][q].x, raw[sb][q].y, xin, q, sel);
      f32x4 a1[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a1[rb] = h2_mul1(w1[rb], xin);
      const f32x4 out = h2_mul2(w2[0], h2_chain_relu(a1[0], a1[1], sel), w2[1],
                                h2_chain_relu(a1[2], a1[3], sel));
      const uint64_t m = base + sb * 16 + j;
      if (m < M) {
        const uint64_t mo = slot ? slot[m] : m;   // see k_sigma_mlp (mlp.hip)
        *reinterpret_cast<f32x4*>(h + mo * 16 + 4 * g) = out;
        if (g == 0) sigma[mo] = expf(out[0]);
      }
    }
  }
}

extern "C" int32_t ucsa_sigma_mlp_fwd_h2(const float* feat,
                                         const void* packed_sigma_h2, uint32_t M,
                                         uint32_t n_levels, float* h,
                                         float* sigma, void* stream) {
  UCSA_CHECK_ARG(feat, 0);
  UCSA_CHECK_ARG(packed_sigma_h2, 1);
  UCSA_CHECK_ARG(n_levels == 16, 3);
  UCSA_CHECK_ARG(h && sigma, 4);
  if (M == 0) return 0;
  const uint32_t need = ucsa_div_up(M, 16 * SIGX_UNROLL * 4);
  const uint32_t blocks = need < 4096u ? need : 4096u;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sigma_mlp_h2, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)feat, packed_sigma_h2,
                     (uint64_t)M, h, sigma, (const uint32_t*)nullptr);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_sigma_mlp_fwd_x3(const float* feat,
                                         const void* packed_sigma_x3, uint32_t M,
                                         uint32_t n_levels, float* h,
                                         float* sigma, void* stream) {
  UCSA_CHECK_ARG(feat, 0);
  UCSA_CHECK_ARG(packed_sigma_x3, 1);
  UCSA_CHECK_ARG(n_levels == 16, 3);
  UCSA_CHECK_ARG(h && sigma, 4);
  if (M == 0) return 0;
  const uint32_t need = ucsa_div_up(M, 16 * SIGX_UNROLL * 4);
  const uint32_t blocks = need < 4096u ? need : 4096u;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sigma_mlp_x3, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)feat, packed_sigma_x3,
                     (uint64_t)M, h, sigma, (const uint32_t*)nullptr);
  return ucsa_launch_status();
}


// The sigma MLP of any arithmetic reading a DEPTH-ORDERED feature array
// (ucsa_tile_depth_order / ucsa_hashgrid_encode_sorted) and writing h / sigma
// to the ray-major slots: sample m of the feature array -> row slot[m].
int32_t ucsa_sigma_mlp_fwd_f32_slot(const float* feat, const float* packed_sigma,
                                    uint32_t M, const uint32_t* slot, float* h,
                                    float* sigma, void* stream);   // mlp.hip
extern "C" int32_t ucsa_sigma_mlp_fwd_scatter(int32_t mode, const void* feat,
                                              const void* packed_sigma,
                                              uint32_t M, uint32_t n_levels,
                                              const uint32_t* slot, float* h,
                                              float* sigma, void* stream) {
  UCSA_CHECK_ARG(mode >= 0 && mode <= 3, 0);
  UCSA_CHECK_ARG(feat, 1);
  UCSA_CHECK_ARG(packed_sigma, 2);
  UCSA_CHECK_ARG(n_levels == 16, 4);
  UCSA_CHECK_ARG(slot, 5);
  UCSA_CHECK_ARG(h && sigma, 6);
  if (M == 0) return 0;
  if (mode == 0)
    return ucsa_sigma_mlp_fwd_f32_slot((const float*)feat, (const float*)packed_sigma,
                                       M, slot, h, sigma, stream);
  UCSA_CLEAR_ERR();
  if (mode == 1) {   // f16 nets on fp16 features
    const uint32_t need = ucsa_div_up(M, 16 * SIGH_UNROLL * 4);
    hipLaunchKernelGGL(k_sigma_mlp_f16<sig_half2>, dim3(need < 2048u ? need : 2048u),
                       dim3(256), 0, (hipStream_t)stream, (const sig_half2*)feat,
                       packed_sigma, (uint64_t)M, h, sigma, slot);
  } else {
    const uint32_t need = ucsa_div_up(M, 16 * SIGX_UNROLL * 4);
    const dim3 g(need < 4096u ? need : 4096u);
    if (mode == 2)
      hipLaunchKernelGGL(k_sigma_mlp_x3, g, dim3(256), 0, (hipStream_t)stream,
                         (const float2*)feat, packed_sigma, (uint64_t)M, h, sigma, slot);
    else
      hipLaunchKernelGGL(k_sigma_mlp_h2, g, dim3(256), 0, (hipStream_t)stream,
                         (const float2*)feat, packed_sigma, (uint64_t)M, h, sigma, slot);
  }
  return ucsa_launch_status();
}

// ---------------------------------------------------------------------------
// sigma MLP backward with every contraction on the bf16 MFMA pipe as two-term
// splits (mfma_mlp_x3.h "bf16x2"; the f32-input form is k_sigma_mlp_bwd,
// mlp_bwd.hip -- same inputs, outputs and partial layout).  Per 16 samples:
// 12 (forward L1 recompute) + 12 + 12 (dX) bf16 16x16x32 MFMAs and 12 + 24
// (dW, 16x16x16) instead of 128 f32-input MFMAs.
// ---------------------------------------------------------------------------
#define SIGX_WAVES 4
extern __shared__ __attribute__((aligned(16))) float sigx_smem[];

__global__ void __launch_bounds__(64 * SIGX_WAVES)
k_sigma_mlp_bwd_x2(const float2* __restrict__ feat, const float* __restrict__ d_h,
                   const void* __restrict__ packed, const void* __restrict__ packed_t,
                   uint64_t M, float2* __restrict__ d_feat, float* __restrict__ partial) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint64_t wave = (uint64_t)blockIdx.x * SIGX_WAVES + wid;
  const uint64_t nwaves = (uint64_t)gridDim.x * SIGX_WAVES;
  float* dy_tile = sigx_smem + (size_t)wid * 2 * 16 * TILE_LD;
  float* x_tile = dy_tile + 16 * TILE_LD;
  const X3Sel sel = x3_selectors();
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // terms 0 and 1 of the three-term packs, in registers for the whole kernel
  auto frag2 = [&](const void* p, int f) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p) + (f * 3) * 64 + lane;
    W2 w;
    w.t[0] = q[0];
    w.t[1] = q[64];
    return w;
  };
  W2 w1[4], w2t[4], w1t[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    w1[rb] = frag2(packed, rb);
    w2t[rb] = frag2(packed_t, rb);
    w1t[rb] = frag2(packed_t, 4 + rb);
  }
  f32x4 dw1[4][2], dw2[1][4];
  dw_zero(dw1);
  dw_zero(dw2);

  for (uint64_t base = wave * 16; base < M; base += nwaves * 16) {
    uint64_t m = base + j;
    const bool live = m < M;
    if (!live) m = M - 1;
    float xin[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float2 v = feat[(uint64_t)(4 * q + g) * M + m];
      xin[2 * q] = v.x;
      xin[2 * q + 1] = v.y;
    }
    f32x4 dh = *reinterpret_cast<const f32x4*>(d_h + m * 16 + 4 * g);
    if (!live) dh = z4;  // padded columns add nothing
    X2 xb;
#pragma unroll
    for (int q = 0; q < 4; ++q) split2_pair(xin[2 * q], xin[2 * q + 1], xb, q, sel);
    f32x4 acc1[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc1[rb] = mfma_x2(w1[rb], xb, z4);

    // dW2 += dh (x) relu(acc1)
    tile_store(dy_tile, g, j, 0, dh);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) tile_store(x_tile, g, j, rb, relu4(acc1[rb]));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    dw_accumulate_b2<1, 4>(dy_tile, x_tile, lane, dw2, sel);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // d_hid = W2^T dh, gated by ReLU
    f32x4 dhid[4];
    {
      const X2 bd = chain_x2(dh, z4, sel);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) dhid[rb] = mfma_x2(w2t[rb], bd, z4);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dhid[rb][r] = acc1[rb][r] > 0.f ? dhid[rb][r] : 0.f;

    // dW1 += d_hid (x) x   (x tile in natural feature order 2*level + c)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) tile_store(dy_tile, g, j, rb, dhid[rb]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float2*>(x_tile + j * TILE_LD + 8 * q + 2 * g) =
          make_float2(live ? xin[2 * q] : 0.f, live ? xin[2 * q + 1] : 0.f);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    dw_accumulate_b2<4, 2>(dy_tile, x_tile, lane, dw1, sel);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // d_feat = W1^T d_hid
    const X2 d0 = chain_x2(dhid[0], dhid[1], sel), d1 = chain_x2(dhid[2], dhid[3], sel);
    f32x4 dx[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      dx[rb] = mfma_x2(w1t[2 * rb], d0, z4);
      dx[rb] = mfma_x2(w1t[2 * rb + 1], d1, dx[rb]);
    }
    if (live) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const uint32_t lv = 8 * rb + 2 * g;
        d_feat[(uint64_t)lv * M + m] = make_float2(dx[rb][0], dx[rb][1]);
        d_feat[(uint64_t)(lv + 1) * M + m] = make_float2(dx[rb][2], dx[rb][3]);
      }
    }
  }
  float* dst = partial + (size_t)wave * 3072;
  dw_store<4, 2>(dst, 32, lane, dw1);
  dw_store<1, 4>(dst + 2048, 64, lane, dw2);
}

// same partial-slot count as ucsa_sigma_mlp_bwd (ucsa_sigma_mlp_bwd_parts)
extern "C" int32_t ucsa_sigma_mlp_bwd_x2(const float* feat, const float* d_h,
                                         const void* packed_sigma_x3,
                                         const void* packed_sigma_t_x3, uint32_t M,
                                         uint32_t n_levels, float* d_feat,
                                         float* partial, void* stream) {
  UCSA_CHECK_ARG(feat, 0);
  UCSA_CHECK_ARG(d_h, 1);
  UCSA_CHECK_ARG(packed_sigma_x3 && packed_sigma_t_x3, 2);
  UCSA_CHECK_ARG(n_levels == 16, 5);
  UCSA_CHECK_ARG(d_feat && partial, 6);
  if (M == 0) return 0;
  const uint32_t blocks = ucsa_sigma_mlp_bwd_parts(M) / SIGX_WAVES;
  const size_t smem = (size_t)SIGX_WAVES * 2 * 16 * TILE_LD * sizeof(float);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sigma_mlp_bwd_x2, dim3(blocks), dim3(64 * SIGX_WAVES), smem,
                     (hipStream_t)stream, (const float2*)feat, d_h, packed_sigma_x3,
                     packed_sigma_t_x3, (uint64_t)M, (float2*)d_feat, partial);
  return ucsa_launch_status();
}
