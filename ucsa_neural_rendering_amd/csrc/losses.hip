// Losses, post-processing and the confusion matrix (SURVEY 8a rows a12, a15,
// a16, M).  All are per-ray / per-pixel elementwise work over C <= 64 classes
// plus small deterministic reductions (block partials -> one fixed-order pass;
// integer atomics only where order cannot matter).
#include <cmath>

#include "ucsa_common.h"
#include "wave_ops.h"

// ---------------------------------------------------------------------------
// block-level deterministic reduction of K floats per thread -> partial[blk][K]
// ---------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void block_partial(float (&v)[K], float* partial,
                                              float* smem) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < K; ++k) smem[wid * K + k] = v[k];
  __syncthreads();
  if (threadIdx.x < K) {
    float s = 0.f;
    for (uint32_t w = 0; w < nw; ++w) s += smem[w * K + threadIdx.x];
    partial[(size_t)blockIdx.x * K + threadIdx.x] = s;
  }
}

// ===========================================================================
// a12: NeRF losses.  reference joint_train_lightning_net.py:180-223, weights
// :44-45, :503-507.   stats[0..5] = {loss_color, loss_sem, loss_depth,
// n_invalid_sem, n_valid_depth, total}
// ===========================================================================
__global__ void __launch_bounds__(256)
k_nerf_loss_terms(const float* __restrict__ rgb, const float* __restrict__ sem,
                  const float* __restrict__ depth,
                  const float* __restrict__ gt_rgb,
                  const int64_t* __restrict__ labels,
                  const float* __restrict__ gt_depth, uint32_t N, uint32_t C,
                  float uom, float* __restrict__ partial) {
  __shared__ float smem[4 * 5];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // color, sem, depth, n_invalid, n_valid
  if (i < N) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float e = rgb[(size_t)i * 3 + c] - gt_rgb[(size_t)i * 3 + c];
      v[0] += e * e;
    }
    const float* s = sem + (size_t)i * C;
    float sum = 0.f;
    for (uint32_t k = 0; k < C; ++k) sum += s[k];
    const int64_t lab = labels[i];
    if (sum == 0.f) {
      v[3] = 1.f;
    } else if (lab >= 0 && lab < (int64_t)C) {
      v[1] = -logf(s[lab] / sum + 1e-15f);
    }
    const float gd = gt_depth[i];
    if (gd != 0.f) {
      v[2] = fabsf(depth[i] / uom - gd);
      v[4] = 1.f;
    }
  }
  block_partial<5>(v, partial, smem);
}

__global__ void k_nerf_loss_final(const float* __restrict__ partial,
                                  uint32_t n_blocks, uint32_t N,
                                  float w_sem, float w_depth,
                                  float* __restrict__ stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a[5] = {0, 0, 0, 0, 0};
  for (uint32_t b = 0; b < n_blocks; ++b)
    for (int k = 0; k < 5; ++k) a[k] += (double)partial[(size_t)b * 5 + k];
  const float lc = (float)(a[0] / (3.0 * N));
  const bool sem_ok = a[3] < (double)N;  // not every ray invalid
  const float ls = sem_ok ? (float)(a[1] / N) : 0.f;
  const float ld = a[4] > 0 ? (float)(a[2] / a[4]) : NAN;
  stats[0] = lc;
  stats[1] = sem_ok ? ls : NAN;  // "None" in the reference
  stats[2] = ld;
  stats[3] = (float)a[3];
  stats[4] = (float)a[4];
  float total = lc;
  if (sem_ok) total += ls * w_sem;
  total += ld * w_depth;  // NaN when no pixel has depth, like the reference
  stats[5] = total;
  stats[6] = ls;  // the semantics term with "None" as a zero (no read-back)
  stats[7] = 0.f;
}

// Backward of the loss node in ONE launch: the gradients ucsa_nerf_loss wrote
// (of w-weighted terms, i.e. of `total`) times the cotangents autograd hands
// back -- of the total (g_total) and, when a caller used the terms one by one,
// of the single terms (g_color / g_sem / g_depth; a term's own gradient is the
// stored one / its weight).  Device scalars, NULL = 0: no host read-back.
__global__ void __launch_bounds__(256)
k_nerf_loss_apply(const float* d_rgb, const float* d_sem, const float* d_depth,
                  float* o_rgb, float* o_sem, float* o_depth,  // may alias the inputs
                  uint32_t N, uint32_t C,
                  const float* __restrict__ g_total,
                  const float* __restrict__ g_color,
                  const float* __restrict__ g_sem,
                  const float* __restrict__ g_depth, float inv_w_sem,
                  float inv_w_depth) {
  const float gt = g_total ? g_total[0] : 0.f;
  const float kc = gt + (g_color ? g_color[0] : 0.f);
  const float ks = gt + (g_sem ? g_sem[0] * inv_w_sem : 0.f);
  const float kd = gt + (g_depth ? g_depth[0] * inv_w_depth : 0.f);
  const uint64_t n_rgb = (uint64_t)N * 3, n_sem = (uint64_t)N * C;
  const uint64_t total = n_rgb + n_sem + N;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * blockDim.x) {
    if (i < n_rgb) o_rgb[i] = d_rgb[i] * kc;
    else if (i < n_rgb + n_sem) o_sem[i - n_rgb] = d_sem[i - n_rgb] * ks;
    else o_depth[i - n_rgb - n_sem] = d_depth[i - n_rgb - n_sem] * kd;
  }
}

extern "C" int32_t ucsa_nerf_loss_apply(const float* d_rgb, const float* d_sem,
                                        const float* d_depth, float* out_rgb,
                                        float* out_sem, float* out_depth,
                                        uint32_t N, uint32_t C,
                                        const float* g_total,
                                        const float* g_color,
                                        const float* g_sem, const float* g_depth,
                                        float w_sem, float w_depth,
                                        void* stream) {
  UCSA_CHECK_ARG(d_rgb && d_sem && d_depth, 0);
  UCSA_CHECK_ARG(out_rgb && out_sem && out_depth, 3);
  UCSA_CHECK_ARG(N > 0 && C >= 1, 6);
  UCSA_CHECK_ARG(g_total || g_color || g_sem || g_depth, 8);
  UCSA_CHECK_ARG((!g_sem || w_sem != 0.f) && (!g_depth || w_depth != 0.f), 12);
  const uint64_t total = (uint64_t)N * (4 + C);
  uint32_t blocks = (uint32_t)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_nerf_loss_apply, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, d_rgb, d_sem, d_depth, out_rgb, out_sem,
                     out_depth, N, C, g_total,
                     g_color, g_sem, g_depth, g_sem ? 1.0f / w_sem : 0.f,
                     g_depth ? 1.0f / w_depth : 0.f);
  return ucsa_launch_status();
}

__global__ void __launch_bounds__(256)
k_nerf_loss_grad(const float* __restrict__ rgb, const float* __restrict__ sem,
                 const float* __restrict__ depth,
                 const float* __restrict__ gt_rgb,
                 const int64_t* __restrict__ labels,
                 const float* __restrict__ gt_depth,
                 const float* __restrict__ stats, uint32_t N, uint32_t C,
                 float uom, float w_sem, float w_depth, float scale,
                 float* __restrict__ d_rgb, float* __restrict__ d_sem,
                 float* __restrict__ d_depth) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const bool sem_ok = stats[3] < (float)N;
  const float n_valid = stats[4];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float e = rgb[(size_t)i * 3 + c] - gt_rgb[(size_t)i * 3 + c];
    d_rgb[(size_t)i * 3 + c] = scale * 2.0f * e / (3.0f * (float)N);
  }
  const float* s = sem + (size_t)i * C;
  float* ds = d_sem + (size_t)i * C;
  float sum = 0.f;
  for (uint32_t k = 0; k < C; ++k) sum += s[k];
  const int64_t lab = labels[i];
  if (sem_ok && sum != 0.f && lab >= 0 && lab < (int64_t)C) {
    const float pl = s[lab] / sum;
    const float coef = -scale * w_sem / ((float)N * (pl + 1e-15f) * sum);
    for (uint32_t k = 0; k < C; ++k)
      ds[k] = coef * (((int64_t)k == lab ? 1.0f : 0.0f) - pl);
  } else {
    for (uint32_t k = 0; k < C; ++k) ds[k] = 0.f;
  }
  // No pixel with depth: the reference's loss_depth is the mean of an EMPTY
  // selection = NaN (logged as such, stats[2] / stats[5]), but its gradient is
  // the scatter of an empty tensor = zeros, so the colour / semantics gradients
  // stay finite and GradScaler does NOT skip the step (checked against torch in
  // tests/test_gpu_losses_and_module.py).  Same here: d_depth = 0.
  const float gd = gt_depth[i];
  float dd = 0.f;
  if (gd != 0.f && n_valid > 0.f) {
    const float e = depth[i] / uom - gd;
    const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
    dd = scale * w_depth * sg / (uom * n_valid);
  }
  d_depth[i] = dd;
}

extern "C" uint32_t ucsa_loss_partial_floats(uint32_t n) {
  return ucsa_div_up(n ? n : 1, 256) * 8;
}

extern "C" int32_t ucsa_nerf_loss(const float* rgb, const float* sem,
                                  const float* depth, const float* gt_rgb,
                                  const int64_t* labels, const float* gt_depth,
                                  uint32_t N, uint32_t C, float uom,
                                  float w_sem, float w_depth, float grad_scale,
                                  float* stats, float* d_rgb, float* d_sem,
                                  float* d_depth, float* partial,
                                  void* stream) {
  UCSA_CHECK_ARG(rgb && sem && depth, 0);
  UCSA_CHECK_ARG(gt_rgb && labels && gt_depth, 3);
  UCSA_CHECK_ARG(N > 0, 6);
  UCSA_CHECK_ARG(C >= 1 && C <= 4096, 7);
  UCSA_CHECK_ARG(uom != 0.f, 8);
  UCSA_CHECK_ARG(stats, 12);
  UCSA_CHECK_ARG(partial, 16);
  hipStream_t s = (hipStream_t)stream;
  const uint32_t blocks = ucsa_div_up(N, 256);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_nerf_loss_terms, dim3(blocks), dim3(256), 0, s, rgb, sem,
                     depth, gt_rgb, labels, gt_depth, N, C, uom, partial);
  hipLaunchKernelGGL(k_nerf_loss_final, dim3(1), dim3(64), 0, s, partial, blocks,
                     N, w_sem, w_depth, stats);
  if (d_rgb && d_sem && d_depth)
    hipLaunchKernelGGL(k_nerf_loss_grad, dim3(blocks), dim3(256), 0, s, rgb, sem,
                       depth, gt_rgb, labels, gt_depth, stats, N, C, uom, w_sem,
                       w_depth, grad_scale, d_rgb, d_sem, d_depth);
  return ucsa_launch_status();
}

// ===========================================================================
// a16: composited probabilities -> normalised + argmax.
// reference joint_train_lightning_net.py:246-251
// ===========================================================================
__global__ void __launch_bounds__(256)
k_sem_postproc(const float* __restrict__ sem, uint32_t N, uint32_t C,
               float* __restrict__ norm_out, int64_t* __restrict__ arg_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float* s = sem + (size_t)i * C;
  float sum = 0.f;
  for (uint32_t k = 0; k < C; ++k) sum += s[k];
  const bool invalid = sum == 0.f;
  const float denom = invalid ? (float)C : sum;
  float best = -INFINITY;
  uint32_t arg = 0;
  for (uint32_t k = 0; k < C; ++k) {
    const float p = (invalid ? 1.0f : s[k]) / denom;
    if (norm_out) norm_out[(size_t)i * C + k] = p;
    if (p > best) {  // first maximum wins, like torch.argmax on the CPU
      best = p;
      arg = k;
    }
  }
  arg_out[i] = (int64_t)arg;
}

extern "C" int32_t ucsa_semantic_postproc(const float* sem, uint32_t N,
                                          uint32_t C, float* normalised,
                                          int64_t* argmax, void* stream) {
  UCSA_CHECK_ARG(sem, 0);
  UCSA_CHECK_ARG(C >= 1, 2);
  UCSA_CHECK_ARG(argmax, 4);
  if (N == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sem_postproc, dim3(ucsa_div_up(N, 256)), dim3(256), 0,
                     (hipStream_t)stream, sem, N, C, normalised, argmax);
  return ucsa_launch_status();
}

// ===========================================================================
// a15: segmentation tail on NCHW logits: softmax, argmax, and the reference's
// CrossEntropy-on-softmax loss (double softmax, joint_train_lightning_net.py
// :163, :456-458) with its gradient wrt the logits.
//   logits [B,C,P] (P = H*W), labels [B,P] int64 (-1 ignored; may be NULL)
// One lane per pixel: class planes are P-strided, so every class read/write
// is coalesced across the wave.
// ===========================================================================
#define SEG_MAX_C 4096

__global__ void __launch_bounds__(256)
k_seg_tail(const float* __restrict__ logits, const int64_t* __restrict__ labels,
           uint32_t B, uint32_t C, uint32_t P, float grad_scale,
           float* __restrict__ prob, int64_t* __restrict__ argmax,
           float* __restrict__ d_logits, float* __restrict__ partial) {
  __shared__ float smem[4];
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)B * P;
  float v[1] = {0.f};
  if (i < total) {
    const uint32_t b = (uint32_t)(i / P), px = (uint32_t)(i % P);
    const float* x = logits + (size_t)b * C * P + px;
    // No per-lane array (a runtime-indexed one would live in scratch): the
    // class planes are re-read per pass; they are coalesced and L2-resident.
    float mx = -INFINITY;
    uint32_t arg = 0;
    for (uint32_t k = 0; k < C; ++k) {
      const float xv = x[(size_t)k * P];
      if (xv > mx) { mx = xv; arg = k; }
    }
    float sum = 0.f;
    for (uint32_t k = 0; k < C; ++k) sum += expf(x[(size_t)k * P] - mx);
    if (prob)
      for (uint32_t k = 0; k < C; ++k)
        prob[(size_t)b * C * P + (size_t)k * P + px] =
            expf(x[(size_t)k * P] - mx) / sum;
    if (argmax) argmax[i] = (int64_t)arg;  // argmax(softmax) == argmax(logits)
    if (labels) {
      const int64_t lab = labels[i];
      const bool ok = lab >= 0 && lab < (int64_t)C;
      // second softmax over the probabilities; their maximum is p[arg]
      const float m2 = expf(x[(size_t)arg * P] - mx) / sum;
      float s2 = 0.f;
      for (uint32_t k = 0; k < C; ++k)
        s2 += expf(expf(x[(size_t)k * P] - mx) / sum - m2);
      const float lse = m2 + logf(s2);
      if (ok) v[0] = lse - expf(x[(size_t)lab * P] - mx) / sum;
      if (d_logits) {
        // g_k = dL/dp_k = (softmax(p)_k - [k==lab]) / total ; then through
        // the first softmax: dL/dx_j = p_j (g_j - sum_k p_k g_k)
        float dot = 0.f;
        const float inv = grad_scale / (float)total;
        for (uint32_t k = 0; k < C; ++k) {
          const float pk = expf(x[(size_t)k * P] - mx) / sum;
          const float q = expf(pk - m2) / s2;
          const float gk = ok ? (q - ((int64_t)k == lab ? 1.f : 0.f)) * inv : 0.f;
          dot += pk * gk;
        }
        for (uint32_t k = 0; k < C; ++k) {
          const float pk = expf(x[(size_t)k * P] - mx) / sum;
          const float q = expf(pk - m2) / s2;
          const float gk = ok ? (q - ((int64_t)k == lab ? 1.f : 0.f)) * inv : 0.f;
          d_logits[(size_t)b * C * P + (size_t)k * P + px] = pk * (gk - dot);
        }
      }
    }
  }
  if (partial) block_partial<1>(v, partial, smem);
}

__global__ void k_mean_final(const float* __restrict__ partial,
                             uint32_t n_blocks, double denom,
                             float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a = 0;
  for (uint32_t b = 0; b < n_blocks; ++b) a += (double)partial[b];
  out[0] = (float)(a / denom);
}

extern "C" int32_t ucsa_seg_tail(const float* logits, const int64_t* labels,
                                 uint32_t B, uint32_t C, uint32_t P,
                                 float grad_scale, float* prob, int64_t* argmax,
                                 float* loss, float* d_logits, float* partial,
                                 void* stream) {
  UCSA_CHECK_ARG(logits, 0);
  UCSA_CHECK_ARG(C >= 1 && C <= SEG_MAX_C, 3);
  UCSA_CHECK_ARG(!labels || (loss && partial), 8);
  const uint64_t total = (uint64_t)B * P;
  if (total == 0) return 0;
  const uint32_t blocks = ucsa_div_up(total, 256);
  hipStream_t s = (hipStream_t)stream;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_seg_tail, dim3(blocks), dim3(256), 0, s, logits, labels,
                     B, C, P, grad_scale, prob, argmax, d_logits,
                     labels ? partial : (float*)nullptr);
  if (labels)
    hipLaunchKernelGGL(k_mean_final, dim3(1), dim3(64), 0, s, partial, blocks,
                       (double)total, loss);
  return ucsa_launch_status();
}

// ===========================================================================
// M: confusion matrix (rows = truth), reference nr4seg/utils/metrics.py:31-46.
// Integer atomics: exact and order-independent.
// ===========================================================================
__global__ void __launch_bounds__(256)
k_confusion(const int64_t* __restrict__ preds,
            const int64_t* __restrict__ truths, uint64_t n, uint32_t C,
            unsigned long long* __restrict__ cm) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const int64_t t = truths[i], p = preds[i];
    if (t < 0 || t >= (int64_t)C || p < 0 || p >= (int64_t)C) continue;
    atomicAdd(cm + (size_t)t * C + (size_t)p, 1ull);
  }
}

extern "C" int32_t ucsa_confusion_matrix(const int64_t* preds,
                                         const int64_t* truths, uint64_t n,
                                         uint32_t C, int64_t* cm,
                                         void* stream) {
  UCSA_CHECK_ARG(preds, 0);
  UCSA_CHECK_ARG(truths, 1);
  UCSA_CHECK_ARG(C >= 1, 3);
  UCSA_CHECK_ARG(cm, 4);
  if (n == 0) return 0;
  uint32_t blocks = ucsa_div_up(n, 256);
  if (blocks > 2048) blocks = 2048;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_confusion, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, preds, truths, n, C,
                     (unsigned long long*)cm);
  return ucsa_launch_status();
}
