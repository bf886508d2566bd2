// Rendered-image augmentation for the joint step:
// ColorJitter (brightness / contrast / saturation / hue in a drawn order),
// rotation (bilinear image, nearest label, fill 0), crop, horizontal flip --
// reference nr4seg/lightning/joint_train_lightning_net.py:259-302 with the
// transforms of :89-101.  The arithmetic is torchvision 0.12.0's tensor path
// (functional_tensor.py: _blend, rgb_to_grayscale, adjust_*, _rgb2hsv,
// _hsv2rgb, _gen_affine_grid, _apply_grid_transform over grid_sample with
// align_corners=False); the reference spends ~40 torch kernels per image on it.
// Here: one reduction (the contrast op needs the mean grey level of the image
// as it is when that op runs) and one gather kernel that jitters the four
// bilinear neighbours on the fly.  Every random draw is an input.
#include <cmath>

#include "ucsa_common.h"
#include "wave_ops.h"

#define AUG_BLOCK 256
#define AUG_MAX_BATCH 16

struct AugOne {
  int32_t order[4];
  float brightness, contrast, saturation, hue;
  float m[6];  // inverse rotation matrix (row-major 2x3), float32 of the doubles
  int32_t flip, crop_i, crop_j;
};

struct AugBatch {
  AugOne p[AUG_MAX_BATCH];
};

__device__ __forceinline__ float clamp01(float x) {
  return fminf(fmaxf(x, 0.0f), 1.0f);
}

__device__ __forceinline__ float blend(float a, float b, float ratio) {
  return clamp01(ratio * a + (1.0f - ratio) * b);
}

__device__ __forceinline__ float grey(float r, float g, float b) {
  return 0.2989f * r + 0.587f * g + 0.114f * b;
}

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b,
                                          float hue) {
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.0f : maxc);
  const float div = eqc ? 1.0f : cr;
  const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
  const float hr = (maxc == r) ? (bc - gc) : 0.0f;
  const float hg = (maxc == g && maxc != r) ? (2.0f + rc - bc) : 0.0f;
  const float hb = (maxc != g && maxc != r) ? (4.0f + gc - rc) : 0.0f;
  float h = hr + hg + hb;
  h = fmodf(h / 6.0f + 1.0f, 1.0f);
  h = h + hue;
  h = h - floorf(h);  // python-style % 1.0
  const float v = maxc;
  const float h6 = h * 6.0f;
  const float fi = floorf(h6);
  const float f = h6 - fi;
  const float p = clamp01(v * (1.0f - s));
  const float q = clamp01(v * (1.0f - s * f));
  const float t = clamp01(v * (1.0f - (s * (1.0f - f))));
  const int i = ((int)fi % 6 + 6) % 6;
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// apply the ops of `a.order`; with stop_at_contrast the pixel is returned as
// it enters the contrast op (for the mean grey level).
__device__ __forceinline__ void jitter(const AugOne& a, float mean,
                                       bool stop_at_contrast, float& r,
                                       float& g, float& b) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int op = a.order[k];
    if (op == 0) {
      r = blend(r, 0.0f, a.brightness);
      g = blend(g, 0.0f, a.brightness);
      b = blend(b, 0.0f, a.brightness);
    } else if (op == 1) {
      if (stop_at_contrast) return;
      r = blend(r, mean, a.contrast);
      g = blend(g, mean, a.contrast);
      b = blend(b, mean, a.contrast);
    } else if (op == 2) {
      const float l = grey(r, g, b);
      r = blend(r, l, a.saturation);
      g = blend(g, l, a.saturation);
      b = blend(b, l, a.saturation);
    } else if (op == 3) {
      hue_shift(r, g, b, a.hue);
    }
  }
}

__global__ void __launch_bounds__(AUG_BLOCK)
k_aug_grey_sum(const float* __restrict__ img, AugBatch batch, uint32_t b0,
               uint32_t H, uint32_t W, uint32_t n_blk,
               double* __restrict__ partial) {
  __shared__ double sm[AUG_BLOCK / 64];
  const uint32_t bi = blockIdx.y;
  const AugOne& a = batch.p[bi];
  const size_t P = (size_t)H * W;
  const float* im = img + (size_t)(b0 + bi) * 3 * P;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * AUG_BLOCK + threadIdx.x; i < P;
       i += (size_t)n_blk * AUG_BLOCK) {
    float r = im[i], g = im[P + i], b = im[2 * P + i];
    jitter(a, 0.0f, true, r, g, b);
    acc += (double)grey(r, g, b);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63u) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < AUG_BLOCK / 64; ++w) s += sm[w];
    partial[(size_t)bi * n_blk + blockIdx.x] = s;
  }
}

__global__ void __launch_bounds__(AUG_BLOCK)
k_aug_apply(const float* __restrict__ img, const int64_t* __restrict__ label,
            AugBatch batch, uint32_t b0, uint32_t H, uint32_t W, uint32_t oh,
            uint32_t ow, uint32_t n_blk, const double* __restrict__ partial,
            float* __restrict__ out_img, int64_t* __restrict__ out_label) {
  __shared__ float mean_s;
  const uint32_t bi = blockIdx.y;
  const AugOne& a = batch.p[bi];
  const size_t P = (size_t)H * W;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (uint32_t k = 0; k < n_blk; ++k) s += partial[(size_t)bi * n_blk + k];
    mean_s = (float)(s / (double)P);
  }
  __syncthreads();
  const float mean = mean_s;
  const uint32_t o = blockIdx.x * AUG_BLOCK + threadIdx.x;
  if (o >= oh * ow) return;
  const uint32_t oy = o / ow, ox = o % ow;
  // undo flip and crop: position in the rotated (H x W) image
  const uint32_t cx = a.flip ? ow - 1 - ox : ox;
  const uint32_t ry = (uint32_t)a.crop_i + oy, rx = (uint32_t)a.crop_j + cx;
  // _gen_affine_grid + grid_sample(align_corners=False) un-normalisation
  const float bx = -(float)W * 0.5f + 0.5f + (float)rx;
  const float by = -(float)H * 0.5f + 0.5f + (float)ry;
  const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
  const float gx = bx * (a.m[0] / hw) + by * (a.m[1] / hw) + (a.m[2] / hw);
  const float gy = bx * (a.m[3] / hh) + by * (a.m[4] / hh) + (a.m[5] / hh);
  const float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
  const float* im = img + (size_t)(b0 + bi) * 3 * P;
  // ---- image: bilinear, zero padding, times the interpolated ones-mask -----
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float wx1 = ix - x0f, wy1 = iy - y0f;
  const float wx0 = (x0f + 1.0f) - ix, wy0 = (y0f + 1.0f) - iy;  // as grid_sample
  float acc[3] = {0.f, 0.f, 0.f}, mask = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
    if (xx < 0 || yy < 0 || xx >= (int)W || yy >= (int)H) continue;
    const float w = ((c & 1) ? wx1 : wx0) * ((c >> 1) ? wy1 : wy0);
    const size_t p = (size_t)yy * W + xx;
    float r = im[p], g = im[P + p], b = im[2 * P + p];
    jitter(a, mean, false, r, g, b);
    acc[0] += r * w;
    acc[1] += g * w;
    acc[2] += b * w;
    mask += w;
  }
  const size_t OP = (size_t)oh * ow;
  float* oi = out_img + (size_t)(b0 + bi) * 3 * OP;
  oi[o] = acc[0] * mask;           // img * mask + (1 - mask) * fill, fill = 0
  oi[OP + o] = acc[1] * mask;
  oi[2 * OP + o] = acc[2] * mask;
  // ---- label: nearest (round half to even), fill "unknown" ------------------
  if (label) {
    const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
    int64_t l = -1;
    if (xn >= 0 && yn >= 0 && xn < (int)W && yn < (int)H)
      l = label[(size_t)(b0 + bi) * P + (size_t)yn * W + xn];
    out_label[(size_t)(b0 + bi) * OP + o] = l;
  }
}

extern "C" uint64_t ucsa_augment_workspace_bytes(uint32_t B, uint32_t H,
                                                 uint32_t W) {
  const uint32_t n_blk = ucsa_div_up((uint64_t)H * W, AUG_BLOCK * 4);
  return 8ull * AUG_MAX_BATCH * n_blk + 0 * (uint64_t)B;
}

extern "C" int32_t ucsa_augment(const float* img, const int64_t* label,
                                uint32_t B, uint32_t H, uint32_t W,
                                const ucsa_aug_params* params_host, uint32_t oh,
                                uint32_t ow, float* out_img, int64_t* out_label,
                                void* workspace, void* stream) {
  UCSA_CHECK_ARG(img, 0);
  UCSA_CHECK_ARG(H >= 1 && W >= 1 && (uint64_t)H * W < (1ull << 31), 3);
  UCSA_CHECK_ARG(params_host, 5);
  UCSA_CHECK_ARG(oh >= 1 && oh <= H && ow >= 1 && ow <= W, 6);
  UCSA_CHECK_ARG(out_img, 8);
  UCSA_CHECK_ARG(!label || out_label, 9);
  UCSA_CHECK_ARG(workspace, 10);
  if (B == 0) return 0;
  for (uint32_t b = 0; b < B; ++b) {
    const ucsa_aug_params& q = params_host[b];
    uint32_t seen = 0;
    for (int k = 0; k < 4; ++k) {
      UCSA_CHECK_ARG(q.order[k] >= 0 && q.order[k] <= 3, 5);
      seen |= 1u << q.order[k];
    }
    UCSA_CHECK_ARG(seen == 0xFu, 5);
    UCSA_CHECK_ARG(q.crop_i >= 0 && q.crop_j >= 0 &&
                       (uint32_t)q.crop_i + oh <= H && (uint32_t)q.crop_j + ow <= W, 5);
  }
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n_blk = ucsa_div_up((uint64_t)H * W, AUG_BLOCK * 4);
  double* partial = (double*)workspace;
  for (uint32_t b0 = 0; b0 < B; b0 += AUG_MAX_BATCH) {
    const uint32_t nb = B - b0 < AUG_MAX_BATCH ? B - b0 : AUG_MAX_BATCH;
    AugBatch batch;
    for (uint32_t k = 0; k < nb; ++k) {
      const ucsa_aug_params& q = params_host[b0 + k];
      AugOne& a = batch.p[k];
      for (int i = 0; i < 4; ++i) a.order[i] = q.order[i];
      a.brightness = q.brightness;
      a.contrast = q.contrast;
      a.saturation = q.saturation;
      a.hue = q.hue;
      // torchvision: _get_inverse_affine_matrix([0,0], -angle, ...) in double,
      // then torch.tensor(matrix, dtype=float32)
      const double rot = -(double)q.angle_deg * 3.14159265358979323846 / 180.0;
      const double ca = std::cos(rot), sa = std::sin(rot);
      a.m[0] = (float)ca;  a.m[1] = (float)sa;  a.m[2] = 0.0f;
      a.m[3] = (float)-sa; a.m[4] = (float)ca;  a.m[5] = 0.0f;
      a.flip = q.flip;
      a.crop_i = q.crop_i;
      a.crop_j = q.crop_j;
    }
    UCSA_CLEAR_ERR();
    hipLaunchKernelGGL(k_aug_grey_sum, dim3(n_blk, nb), dim3(AUG_BLOCK), 0, s,
                       img, batch, b0, H, W, n_blk, partial);
    hipLaunchKernelGGL(k_aug_apply, dim3(ucsa_div_up((uint64_t)oh * ow, AUG_BLOCK), nb),
                       dim3(AUG_BLOCK), 0, s, img, label, batch, b0, H, W, oh,
                       ow, n_blk, partial, out_img, out_label);
    const int32_t rc = ucsa_launch_status();
    if (rc) return rc;
  }
  return 0;
}
