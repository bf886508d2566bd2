// Backward of the sigma MLP (+ trunc_exp is applied upstream), transposed
// weight packing, and the deterministic reduction of per-wave dW partials.
// Restates the autograd of tcnn.Network as used by density()
// (reference nr4seg/nerf/network_tcnn_semantics.py:130-144).
#include "mfma_mlp.h"

__device__ __forceinline__ void wave_lds_sync_b() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t chain_col_b(uint32_t ks, uint32_t g) {
  return 16u * (ks >> 2) + 4u * g + (ks & 3u);
}

// ---------------------------------------------------------------------------
// ucsa_mlp_pack_t: A fragments of W^T for dX = W^T dY (see mfma_mlp.h).
//   sigma: [L2^T 16 frags | L1^T 32 frags]
//   color: [L3^T 16 | L2^T 64 | L1^T (h-slot rows only) 16]
//   sem  : [L2^T 16*nrb | L1^T (h-slot rows) 16]
// ---------------------------------------------------------------------------
__global__ void k_mlp_pack_t(int kind, const float* __restrict__ params,
                             float* __restrict__ packed, uint32_t n_total,
                             uint32_t nrb) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_total) return;
  uint32_t f = e >> 6;
  const uint32_t l = e & 63u, g = l >> 4, i = l & 15u;
  uint32_t src;
  if (kind == UCSA_MLP_SIGMA) {
    if (f < 16) {
      const uint32_t rb = f / 4, ks = f % 4;
      src = 64 * 32 + (4 * g + ks) * 64 + 16 * rb + i;
    } else {
      f -= 16;
      const uint32_t rb = f / 16, ks = f % 16;
      src = chain_col_b(ks, g) * 32 + 16 * rb + i;
    }
  } else if (kind == UCSA_MLP_COLOR) {
    if (f < 16) {
      const uint32_t rb = f / 4, ks = f % 4;
      src = 64 * 32 + 64 * 64 + (4 * g + ks) * 64 + 16 * rb + i;
    } else if (f < 80) {
      f -= 16;
      const uint32_t rb = f / 16, ks = f % 16;
      src = 64 * 32 + chain_col_b(ks, g) * 64 + 16 * rb + i;
    } else {
      f -= 80;
      const uint32_t col = i == 0 ? 31u : 15u + i;
      src = chain_col_b(f, g) * 32 + col;
    }
  } else {
    if (f < 16 * nrb) {
      const uint32_t rb = f / (4 * nrb), ks = f % (4 * nrb);
      src = 64 * 16 + chain_col_b(ks, g) * 64 + 16 * rb + i;
    } else {
      f -= 16 * nrb;
      const uint32_t col = i == 0 ? 15u : i - 1u;
      src = chain_col_b(f, g) * 16 + col;
    }
  }
  packed[e] = params[src];
}

static inline uint32_t pad16b(uint32_t n) { return (n + 15u) / 16u * 16u; }

extern "C" uint32_t ucsa_mlp_pack_t_size(int32_t kind, uint32_t n_classes) {
  if (kind == UCSA_MLP_SIGMA) return 48 * 64;
  if (kind == UCSA_MLP_COLOR) return 96 * 64;
  return (16 * (pad16b(n_classes) / 16) + 16) * 64;
}

extern "C" int32_t ucsa_mlp_pack_t(int32_t kind, const float* params,
                                   float* packed_t, uint32_t n_classes,
                                   void* stream) {
  UCSA_CHECK_ARG(kind >= 0 && kind <= 2, 0);
  UCSA_CHECK_ARG(params, 1);
  UCSA_CHECK_ARG(packed_t, 2);
  UCSA_CHECK_ARG(kind != UCSA_MLP_SEM || (n_classes >= 1 && n_classes <= 61), 3);
  const uint32_t n_total = ucsa_mlp_pack_t_size(kind, n_classes);
  const uint32_t nrb = pad16b(n_classes ? n_classes : 1) / 16;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_mlp_pack_t, dim3(ucsa_div_up(n_total, 256)), dim3(256),
                     0, (hipStream_t)stream, (int)kind, params, packed_t,
                     n_total, nrb);
  return ucsa_launch_status();
}

// rows sy, sy+8, sy+16, ... of column p, summed in that order; 8 loads are
// issued before their adds (the loop is latency-bound otherwise: 57 -> ~20 us
// for 1024 partial rows), the order of the adds -- hence the result -- is
// unchanged
__device__ __forceinline__ float column_sum(const float* __restrict__ partial,
                                            uint32_t n_parts, uint32_t n_params,
                                            uint32_t p, uint32_t sy) {
  float s = 0.0f;
  if (p >= n_params) return s;
  uint32_t w = sy;
  for (; w + 56 < n_parts; w += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(w + 8 * u) * n_params + p];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; w < n_parts; w += 8) s += partial[(size_t)w * n_params + p];
  return s;
}

// ---------------------------------------------------------------------------
// Deterministic reduction of per-wave partial gradients:
//   grad[p] = (accumulate ? grad[p] : 0) + sum_{w < n_parts} partial[w][p]
// summed in wave order (fixed), one thread per parameter.
// ---------------------------------------------------------------------------
// blockDim = (32 parameters, 8 slices of the part index): every thread sums
// parts slice, slice+8, ... in order, then the 8 slice sums are added in
// slice order -- a fixed tree, so the result is run-to-run identical.
__global__ void __launch_bounds__(256)
k_reduce_partials(const float* __restrict__ partial, uint32_t n_parts,
                  uint32_t n_params, int accumulate,
                  float* __restrict__ grad) {
  __shared__ float sm[8][33];
  const uint32_t px = threadIdx.x & 31u, sy = threadIdx.x >> 5;
  const uint32_t p = blockIdx.x * 32 + px;
  const float s = column_sum(partial, n_parts, n_params, p, sy);
  sm[sy][px] = s;
  __syncthreads();
  if (sy == 0 && p < n_params) {
    float t = accumulate ? grad[p] : 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sm[k][px];
    grad[p] = t;
  }
}

// Up to 4 independent reductions in one launch (the three MLP gradients of a
// training step: three ~50 us launches of 96 / 224 / 128 workgroups each).
// partial2 / n_parts2 (round 6): a SECOND array of partials added on top of the
// first one's sum, with the additions of two consecutive k_reduce_partials launches
// (the second with accumulate = 1) -- the sigma net's coarse and fine passes.
struct ReduceMulti {
  const float* partial[4];
  const float* partial2[4];
  float* grad[4];
  uint32_t n_parts[4], n_parts2[4], n_params[4], first_block[5];
  int accumulate;
};

__global__ void __launch_bounds__(256)
k_reduce_partials_multi(ReduceMulti a) {
  __shared__ float sm[8][33];
  uint32_t k = 0;
  while (k < 3 && blockIdx.x >= a.first_block[k + 1]) ++k;
  const float* partial = a.partial[k];
  const uint32_t n_parts = a.n_parts[k], n_params = a.n_params[k];
  const uint32_t px = threadIdx.x & 31u, sy = threadIdx.x >> 5;
  const uint32_t p = (blockIdx.x - a.first_block[k]) * 32 + px;
  const float s = column_sum(partial, n_parts, n_params, p, sy);
  sm[sy][px] = s;
  __syncthreads();
  float t = 0.0f;
  if (sy == 0 && p < n_params) {
    t = a.accumulate ? a.grad[k][p] : 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sm[q][px];
  }
  if (a.n_parts2[k] > 0) {   // (workgroup-uniform)
    __syncthreads();
    sm[sy][px] = column_sum(a.partial2[k], a.n_parts2[k], n_params, p, sy);
    __syncthreads();
    if (sy == 0 && p < n_params) {
#pragma unroll
      for (int q = 0; q < 8; ++q) t += sm[q][px];
    }
  }
  if (sy == 0 && p < n_params) a.grad[k][p] = t;
}

// (library-internal: ucsa_render_fused_bwd reduces the three nets' partials in one
// launch; partials2 / n_parts2 may be NULL)
int32_t ucsa_reduce_partials_chained(
    uint32_t count, const float* const* partials, const uint32_t* n_parts,
    const float* const* partials2, const uint32_t* n_parts2,
    const uint32_t* n_params, float* const* grads, int32_t accumulate,
    void* stream) {
  UCSA_CHECK_ARG(count >= 1 && count <= 4, 0);
  UCSA_CHECK_ARG(partials && n_parts && n_params && grads, 1);
  ReduceMulti a;
  uint32_t blocks = 0;
  for (uint32_t k = 0; k < 4; ++k) {
    const bool on = k < count;
    UCSA_CHECK_ARG(!on || (partials[k] && grads[k]), 1);
    a.partial[k] = on ? partials[k] : nullptr;
    a.grad[k] = on ? grads[k] : nullptr;
    a.n_parts[k] = on ? n_parts[k] : 0;
    const bool two = on && partials2 && n_parts2 && partials2[k] && n_parts2[k] > 0;
    a.partial2[k] = two ? partials2[k] : nullptr;
    a.n_parts2[k] = two ? n_parts2[k] : 0;
    a.n_params[k] = on ? n_params[k] : 0;
    a.first_block[k] = blocks;
    blocks += on ? ucsa_div_up(n_params[k], 32) : 0;
  }
  a.first_block[4] = blocks;
  a.accumulate = accumulate;
  if (blocks == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_reduce_partials_multi, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, a);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_reduce_partials_multi(
    uint32_t count, const float* const* partials, const uint32_t* n_parts,
    const uint32_t* n_params, float* const* grads, int32_t accumulate,
    void* stream) {
  return ucsa_reduce_partials_chained(count, partials, n_parts, nullptr, nullptr, n_params,
                                      grads, accumulate, stream);
}

extern "C" int32_t ucsa_reduce_partials(const float* partial, uint32_t n_parts,
                                        uint32_t n_params, int32_t accumulate,
                                        float* grad, void* stream) {
  UCSA_CHECK_ARG(partial, 0);
  UCSA_CHECK_ARG(grad, 4);
  if (n_params == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_reduce_partials, dim3(ucsa_div_up(n_params, 32)),
                     dim3(256), 0, (hipStream_t)stream, partial, n_parts,
                     n_params, (int)accumulate, grad);
  return ucsa_launch_status();
}

// ---------------------------------------------------------------------------
// sigma MLP backward.
//   in : feat[level][m] (saved), d_h[m][16] (grad wrt the RAW outputs: slot 0
//        already carries the trunc_exp backward), packed fwd + transposed W
//   out: d_feat[level][m] float2, per-wave dW partials [n_waves][3072]
// Per 16 samples: 32 (fwd L1) + 16 + 32 (dX) + 16 + 32 (dW) MFMAs.
// ---------------------------------------------------------------------------
#define SIGB_WAVES 4

extern __shared__ __attribute__((aligned(16))) float sigb_smem[];

// ROUND_H (train_precision "tcnn"): the recomputed hidden layer is rounded to
// fp16 before the ReLU gate and dW2 use it -- tiny-cuda-nn's forward keeps its
// hidden activations in fp16 (fp32 accumulation), and its backward reads those.
template <bool ROUND_H>
__global__ void __launch_bounds__(64 * SIGB_WAVES)
k_sigma_mlp_bwd(const float2* __restrict__ feat, const float* __restrict__ d_h,
                const float* __restrict__ packed,
                const float* __restrict__ packed_t, uint64_t M,
                float2* __restrict__ d_feat, float* __restrict__ partial) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint64_t wave = (uint64_t)blockIdx.x * SIGB_WAVES + wid;
  const uint64_t nwaves = (uint64_t)gridDim.x * SIGB_WAVES;
  float* dy_tile = sigb_smem + (size_t)wid * 2 * 16 * TILE_LD;
  float* x_tile = dy_tile + 16 * TILE_LD;

  float w1[4][8], w2t[4][4], w1t[2][16];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w1[rb][ks] = packed[(rb * 8 + ks) * 64 + lane];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w2t[rb][ks] = packed_t[(rb * 4 + ks) * 64 + lane];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      w1t[rb][ks] = packed_t[(16 + rb * 16 + ks) * 64 + lane];

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
    if (!live) dh = f32x4{0.f, 0.f, 0.f, 0.f};  // padded columns add nothing

    f32x4 acc1[4];
    mfma_layer<8, 4>(xin, [&](int rb, int ks) { return w1[rb][ks]; }, acc1);
    if constexpr (ROUND_H) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[rb][r] = (float)(_Float16)acc1[rb][r];
    }

    // dW2 += dh (x) relu(acc1)
    tile_store(dy_tile, g, j, 0, dh);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) tile_store(x_tile, g, j, rb, relu4(acc1[rb]));
    wave_lds_sync_b();
    dw_accumulate<1, 4>(dy_tile, x_tile, lane, dw2);
    wave_lds_sync_b();

    // d_hid = W2^T dh, gated by ReLU
    f32x4 dhid[4];
    {
      float b[4] = {dh[0], dh[1], dh[2], dh[3]};
      mfma_layer<4, 4>(b, [&](int rb, int ks) { return w2t[rb][ks]; }, dhid);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dhid[rb][r] = acc1[rb][r] > 0.f ? dhid[rb][r] : 0.f;

    // dW1 += d_hid (x) x   (x tile in natural feature order 2*level + c)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) tile_store(dy_tile, g, j, rb, dhid[rb]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float2*>(x_tile + j * TILE_LD + 8 * q + 2 * g) =
          make_float2(live ? xin[2 * q] : 0.f, live ? xin[2 * q + 1] : 0.f);
    wave_lds_sync_b();
    dw_accumulate<4, 2>(dy_tile, x_tile, lane, dw1);
    wave_lds_sync_b();

    // d_feat = W1^T d_hid
    float bh[16];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) bh[rb * 4 + r] = dhid[rb][r];
    f32x4 dx[2];
    mfma_layer<16, 2>(bh, [&](int rb, int ks) { return w1t[rb][ks]; }, dx);
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

extern "C" uint32_t ucsa_sigma_mlp_bwd_parts(uint32_t M) {
  const uint32_t need = ucsa_div_up(M, 16 * SIGB_WAVES * 4);
  const uint32_t blocks = need < 512u ? (need ? need : 1u) : 512u;
  return blocks * SIGB_WAVES;
}

extern "C" int32_t ucsa_sigma_mlp_bwd(const float* feat, const float* d_h,
                                      const float* packed_sigma,
                                      const float* packed_sigma_t, uint32_t M,
                                      uint32_t n_levels, float* d_feat,
                                      float* partial, void* stream) {
  UCSA_CHECK_ARG(feat, 0);
  UCSA_CHECK_ARG(d_h, 1);
  UCSA_CHECK_ARG(packed_sigma && packed_sigma_t, 2);
  UCSA_CHECK_ARG(n_levels == 16, 5);
  UCSA_CHECK_ARG(d_feat && partial, 6);
  if (M == 0) return 0;
  const uint32_t blocks = ucsa_sigma_mlp_bwd_parts(M) / SIGB_WAVES;
  const size_t smem = (size_t)SIGB_WAVES * 2 * 16 * TILE_LD * sizeof(float);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sigma_mlp_bwd<false>, dim3(blocks), dim3(64 * SIGB_WAVES), smem,
                     (hipStream_t)stream, (const float2*)feat, d_h,
                     packed_sigma, packed_sigma_t, (uint64_t)M,
                     (float2*)d_feat, partial);
  return ucsa_launch_status();
}

// The same with the recomputed hidden layer rounded to fp16 (k_sigma_mlp_bwd<true>):
// the backward of tiny-cuda-nn's fp16 sigma net, fed with fp16-rounded weights
// (ucsa_mlp_pack of the rounded parameters) and the fp16 features widened to fp32.
extern "C" int32_t ucsa_sigma_mlp_bwd_h16(const float* feat, const float* d_h,
                                          const float* packed_sigma,
                                          const float* packed_sigma_t, uint32_t M,
                                          uint32_t n_levels, float* d_feat,
                                          float* partial, void* stream) {
  UCSA_CHECK_ARG(feat, 0);
  UCSA_CHECK_ARG(d_h, 1);
  UCSA_CHECK_ARG(packed_sigma && packed_sigma_t, 2);
  UCSA_CHECK_ARG(n_levels == 16, 5);
  UCSA_CHECK_ARG(d_feat && partial, 6);
  if (M == 0) return 0;
  const uint32_t blocks = ucsa_sigma_mlp_bwd_parts(M) / SIGB_WAVES;
  const size_t smem = (size_t)SIGB_WAVES * 2 * 16 * TILE_LD * sizeof(float);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sigma_mlp_bwd<true>, dim3(blocks), dim3(64 * SIGB_WAVES), smem,
                     (hipStream_t)stream, (const float2*)feat, d_h,
                     packed_sigma, packed_sigma_t, (uint64_t)M,
                     (float2*)d_feat, partial);
  return ucsa_launch_status();
}
