// Pointwise colour / semantics queries: the reference's network.color() and
// network.semantics() (nr4seg/nerf/network_tcnn_semantics.py:147-207) for
// explicit per-point inputs.  Not on the render hot path (the renderer fuses
// these nets into k_composite); provided for API parity.
//   d [M,3] unit directions, geo_feat [M,15], mask [M] (uint8, may be NULL)
//   rgb [M,3] = sigmoid(colour net) ; probs [M,C] = softmax(semantics net)
//   rows with mask == 0 are written as zeros, like the reference's scatter
//   into a zero tensor (:155, :187-192).
#include "mfma_mlp.h"

extern __shared__ __attribute__((aligned(16))) float ps_smem[];

__device__ __forceinline__ void sh4_rows(float dx, float dy, float dz,
                                         uint32_t g, float (&o)[4]) {
  const float x = ((dx + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float y = ((dy + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float z = ((dz + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float xy = x * y, xz = x * z, yz = y * z;
  const float x2 = x * x, y2 = y * y, z2 = z * z;
  if (g == 0) {
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
  } else if (g == 1) {
    o[0] = 1.0925484305920792f * xy;
    o[1] = -1.0925484305920792f * yz;
    o[2] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[3] = -1.0925484305920792f * xz;
  } else if (g == 2) {
    o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[2] = 2.8906114426405538f * xy * z;
    o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  } else {
    o[0] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[1] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[2] = 1.4453057213202769f * z * (x2 - y2);
    o[3] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  }
}

template <int NRB>
__global__ void __launch_bounds__(256)
k_point_shade(const float* __restrict__ dirs, const float* __restrict__ geo,
              const uint8_t* __restrict__ mask,
              const float* __restrict__ packed_color,
              const float* __restrict__ packed_sem, uint32_t M, uint32_t C,
              uint32_t geo_stride, float* __restrict__ rgb,
              float* __restrict__ probs) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane >> 4, j = lane & 15u;
  float* w_color = ps_smem;
  float* w_sem = w_color + 7168;
  for (uint32_t i = threadIdx.x; i < 7168; i += blockDim.x) w_color[i] = packed_color[i];
  for (uint32_t i = threadIdx.x; i < 1024 + NRB * 1024; i += blockDim.x) w_sem[i] = packed_sem[i];
  __syncthreads();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t base = wave * 16; base < M; base += nwaves * 16) {
    uint32_t m = base + j;
    const bool live = m < M;
    if (!live) m = M - 1;
    float gf[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t slot = 4 * g + r;  // slot 0 = the "ones" pad column
      // geo_stride 15: geo_feat rows; 16: raw sigma-MLP rows h (slot 0 of h
      // is the log-density, which the pad column replaces)
      gf[r] = slot == 0 ? 1.0f : geo[(size_t)m * geo_stride + (geo_stride - 15) + (slot - 1)];
    }
    float out_rgb[3] = {0.f, 0.f, 0.f};
    if (rgb) {
      float sh[4];
      sh4_rows(dirs[(size_t)m * 3], dirs[(size_t)m * 3 + 1], dirs[(size_t)m * 3 + 2], g, sh);
      float xin[8] = {sh[0], sh[1], sh[2], sh[3], gf[0], gf[1], gf[2], gf[3]};
      f32x4 a1[4], a2[4], o3[1];
      mfma_layer<8, 4>(xin, [&](int rb, int ks) { return w_color[(rb * 8 + ks) * 64 + lane]; }, a1);
      float hid[16];
      chain_relu(a1, hid);
      mfma_layer<16, 4>(hid, [&](int rb, int ks) { return w_color[(COLOR_L1_FRAGS + rb * 16 + ks) * 64 + lane]; }, a2);
      chain_relu(a2, hid);
      mfma_layer<16, 1>(hid, [&](int, int ks) { return w_color[(COLOR_L1_FRAGS + COLOR_L2_FRAGS + ks) * 64 + lane]; }, o3);
#pragma unroll
      for (int c = 0; c < 3; ++c) out_rgb[c] = 1.0f / (1.0f + expf(-o3[0][c]));
    }
    const bool on = live && (!mask || mask[m] != 0);
    if (rgb && live && g == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[(size_t)m * 3 + c] = on ? out_rgb[c] : 0.0f;
    }
    if (probs) {
      f32x4 a1[4], lg[NRB];
      mfma_layer<4, 4>(gf, [&](int rb, int ks) { return w_sem[(rb * 4 + ks) * 64 + lane]; }, a1);
      float hid[16];
      chain_relu(a1, hid);
      mfma_layer<16, NRB>(hid, [&](int rb, int ks) { return w_sem[(SEM_L1_FRAGS + rb * 16 + ks) * 64 + lane]; }, lg);
      float mx = -INFINITY;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((uint32_t)(rb * 16 + 4 * g + r) < C) mx = fmaxf(mx, lg[rb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (uint32_t)(rb * 16 + 4 * g + r) < C;
          const float ex = ok ? expf(lg[rb][r] - mx) : 0.f;
          lg[rb][r] = ex;
          sum += ex;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (live) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t cls = rb * 16 + 4 * g + r;
            if (cls < C) probs[(size_t)m * C + cls] = on ? lg[rb][r] / sum : 0.0f;
          }
      }
    }
  }
}

static int32_t point_shade_launch(const float* dirs, const float* geo_feat,
                                  uint32_t geo_stride, const uint8_t* mask,
                                  const float* packed_color,
                                  const float* packed_sem, uint32_t M,
                                  uint32_t n_classes, float* rgb, float* probs,
                                  void* stream) {
  UCSA_CHECK_ARG(geo_feat, 1);
  UCSA_CHECK_ARG(!rgb || dirs, 0);
  UCSA_CHECK_ARG(packed_color && packed_sem, 3);
  UCSA_CHECK_ARG(n_classes >= 1 && n_classes <= 61, 6);
  UCSA_CHECK_ARG(rgb || probs, 7);
  if (M == 0) return 0;
  const uint32_t nrb = (n_classes + 15) / 16;
  const size_t smem = (7168 + 1024 + (size_t)nrb * 1024) * sizeof(float);
  uint32_t blocks = ucsa_div_up(M, 16 * 4 * 4);
  if (blocks > 512) blocks = 512;
  hipStream_t s = (hipStream_t)stream;
  UCSA_CLEAR_ERR();
  switch (nrb) {
    case 1: hipLaunchKernelGGL(k_point_shade<1>, dim3(blocks), dim3(256), smem, s, dirs, geo_feat, mask, packed_color, packed_sem, M, n_classes, geo_stride, rgb, probs); break;
    case 2: hipLaunchKernelGGL(k_point_shade<2>, dim3(blocks), dim3(256), smem, s, dirs, geo_feat, mask, packed_color, packed_sem, M, n_classes, geo_stride, rgb, probs); break;
    case 3: hipLaunchKernelGGL(k_point_shade<3>, dim3(blocks), dim3(256), smem, s, dirs, geo_feat, mask, packed_color, packed_sem, M, n_classes, geo_stride, rgb, probs); break;
    default: hipLaunchKernelGGL(k_point_shade<4>, dim3(blocks), dim3(256), smem, s, dirs, geo_feat, mask, packed_color, packed_sem, M, n_classes, geo_stride, rgb, probs); break;
  }
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_point_shade(const float* dirs, const float* geo_feat,
                                    const uint8_t* mask,
                                    const float* packed_color,
                                    const float* packed_sem, uint32_t M,
                                    uint32_t n_classes, float* rgb,
                                    float* probs, void* stream) {
  return point_shade_launch(dirs, geo_feat, 15, mask, packed_color, packed_sem,
                            M, n_classes, rgb, probs, stream);
}

extern "C" int32_t ucsa_point_shade_h(const float* dirs, const float* h,
                                      const uint8_t* mask,
                                      const float* packed_color,
                                      const float* packed_sem, uint32_t M,
                                      uint32_t n_classes, float* rgb,
                                      float* probs, void* stream) {
  return point_shade_launch(dirs, h, 16, mask, packed_color, packed_sem, M,
                            n_classes, rgb, probs, stream);
}
