// Adam step: torch.optim.Adam semantics as configured at
// reference nr4seg/lightning/joint_train_lightning_net.py:897-919 --
// betas (0.9, 0.99), eps 1e-15, L2 weight decay folded into the gradient
// (1e-6 on the "net" group, 0 on "encoding"), non-AMSGrad.
// HBM-bound elementwise: 16 B read (p, g, m, v) + 12 B written per parameter,
// float4 vectorised, grid-stride.
#include <cmath>

#include "ucsa_common.h"

__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, const float* __restrict__ g,
       float* __restrict__ m, float* __restrict__ v, uint64_t n, float lr,
       float beta1, float beta2, float eps, float wd, float bc1,
       float bc2_sqrt, float inv_scale, bool vec) {
  // vec: all four base pointers are 16-byte aligned (always true for whole
  // torch tensors; a slice handed over by the sharded optimizer may not be)
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
  for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
       i < n; i += stride) {
    if (vec && i + 4 <= n) {
      float4 pp = *reinterpret_cast<float4*>(p + i);
      const float4 gg = *reinterpret_cast<const float4*>(g + i);
      float4 mm = *reinterpret_cast<float4*>(m + i);
      float4 vv = *reinterpret_cast<float4*>(v + i);
      float* pa = &pp.x;
      const float* ga = &gg.x;
      float* ma = &mm.x;
      float* va = &vv.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gr = ga[k] * inv_scale;
        if (wd != 0.0f) gr = gr + wd * pa[k];
        ma[k] = beta1 * ma[k] + (1.0f - beta1) * gr;
        va[k] = beta2 * va[k] + (1.0f - beta2) * gr * gr;
        const float denom = sqrtf(va[k]) / bc2_sqrt + eps;
        pa[k] = pa[k] - (lr / bc1) * (ma[k] / denom);
      }
      *reinterpret_cast<float4*>(p + i) = pp;
      *reinterpret_cast<float4*>(m + i) = mm;
      *reinterpret_cast<float4*>(v + i) = vv;
    } else {
      const uint64_t e = i + 4 < n ? i + 4 : n;
      for (uint64_t k = i; k < e; ++k) {
        float gr = g[k] * inv_scale;
        if (wd != 0.0f) gr = gr + wd * p[k];
        const float mk = beta1 * m[k] + (1.0f - beta1) * gr;
        const float vk = beta2 * v[k] + (1.0f - beta2) * gr * gr;
        m[k] = mk;
        v[k] = vk;
        const float denom = sqrtf(vk) / bc2_sqrt + eps;
        p[k] = p[k] - (lr / bc1) * (mk / denom);
      }
    }
  }
}

extern "C" int32_t ucsa_adam_step(float* params, const float* grads,
                                  float* exp_avg, float* exp_avg_sq,
                                  uint64_t n, uint32_t step, float lr,
                                  float beta1, float beta2, float eps,
                                  float weight_decay, float inv_grad_scale,
                                  void* stream) {
  UCSA_CHECK_ARG(params, 0);
  UCSA_CHECK_ARG(grads, 1);
  UCSA_CHECK_ARG(exp_avg && exp_avg_sq, 2);
  UCSA_CHECK_ARG(step >= 1, 5);
  if (n == 0) return 0;
  // bias corrections in double like torch's Python scalars, then cast
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  uint32_t blocks = ucsa_div_up(n, 256 * 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  const bool vec = (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg |
                     (uintptr_t)exp_avg_sq) & 15u) == 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                     eps, weight_decay, (float)bc1, (float)std::sqrt(bc2),
                     inv_grad_scale, vec);
  return ucsa_launch_status();
}

// ---------------------------------------------------------------------------
// The same step under torch.amp.GradScaler without a host read-back
// (GradScaler hands optimizers that declare _step_supports_amp_scaling the
// scale and the "found inf" flag as DEVICE tensors instead of synchronising
// on the flag, reference use: joint_train_lightning_net.py:509-513).
//   grad_scale[0]  : gradients are divided by it
//   found_inf[0]   : != 0 -> the step is skipped (parameters and moments
//                    untouched), and it does not count: skipped[0] holds the
//                    number of skipped steps so far, the bias corrections use
//                    step - skipped[0].  ucsa_adam_count_skipped advances it
//                    once per optimizer step, after all tensors were updated.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_adam_scaled(float* __restrict__ p, const float* __restrict__ g,
              float* __restrict__ m, float* __restrict__ v, uint64_t n,
              uint32_t step, float lr, float beta1, float beta2, float eps,
              float wd, const float* __restrict__ grad_scale,
              const float* __restrict__ found_inf,
              const uint32_t* __restrict__ skipped) {
  if (found_inf[0] != 0.0f) return;
  const float inv_scale = 1.0f / grad_scale[0];
  const uint32_t eff = step - skipped[0];
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)eff));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)eff));
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += stride) {
    float gr = g[k] * inv_scale;
    if (wd != 0.0f) gr = gr + wd * p[k];
    const float mk = beta1 * m[k] + (1.0f - beta1) * gr;
    const float vk = beta2 * v[k] + (1.0f - beta2) * gr * gr;
    m[k] = mk;
    v[k] = vk;
    const float denom = sqrtf(vk) / bc2_sqrt + eps;
    p[k] = p[k] - (lr / bc1) * (mk / denom);
  }
}

__global__ void k_adam_count_skipped(const float* __restrict__ found_inf,
                                     uint32_t* __restrict__ skipped) {
  if (found_inf[0] != 0.0f) skipped[0] += 1u;
}

extern "C" int32_t ucsa_adam_step_scaled(
    float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
    uint64_t n, uint32_t step, float lr, float beta1, float beta2, float eps,
    float weight_decay, const float* grad_scale, const float* found_inf,
    const uint32_t* skipped, void* stream) {
  UCSA_CHECK_ARG(params, 0);
  UCSA_CHECK_ARG(grads, 1);
  UCSA_CHECK_ARG(exp_avg && exp_avg_sq, 2);
  UCSA_CHECK_ARG(step >= 1, 5);
  UCSA_CHECK_ARG(grad_scale && found_inf && skipped, 11);
  if (n == 0) return 0;
  uint32_t blocks = ucsa_div_up(n, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_adam_scaled, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n,
                     step, lr, beta1, beta2, eps, weight_decay, grad_scale,
                     found_inf, skipped);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_adam_count_skipped(const float* found_inf,
                                           uint32_t* skipped, void* stream) {
  UCSA_CHECK_ARG(found_inf, 0);
  UCSA_CHECK_ARG(skipped, 1);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_adam_count_skipped, dim3(1), dim3(1), 0,
                     (hipStream_t)stream, found_inf, skipped);
  return ucsa_launch_status();
}
