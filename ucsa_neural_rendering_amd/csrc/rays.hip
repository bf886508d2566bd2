// Ray generation, AABB clipping and coarse depth sampling (SURVEY 8a rows
// a1-a3).  All three are trivially HBM-bound elementwise kernels: one lane
// per ray (or per sample), coalesced loads/stores.
#include <cfloat>
#include <cmath>

#include "ucsa_common.h"

// ---------------------------------------------------------------------------
// a1: pinhole rays.  reference nr4seg/dataset/ngp_utils.py:28-69 and
// joint_train_lightning_net.py:108-157 (inds variant).
// ---------------------------------------------------------------------------
__global__ void k_get_rays(const float* __restrict__ poses, uint32_t B,
                           float fx, float fy, float cx, float cy, uint32_t W,
                           const int64_t* __restrict__ inds, uint32_t n,
                           float* __restrict__ rays_o,
                           float* __restrict__ rays_d,
                           float* __restrict__ norms) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t b = blockIdx.y;
  if (i >= n) return;
  const uint32_t pix = inds ? (uint32_t)inds[i] : i;
  const float px = (float)(pix % W) + 0.5f;
  const float py = (float)(pix / W) + 0.5f;
  // same op order as the reference: (i - cx) / fx * 1
  const float x = (px - cx) / fx;
  const float y = (py - cy) / fy;
  const float z = 1.0f;
  const float nrm = sqrtf(x * x + y * y + z * z);
  const float dx = x / nrm, dy = y / nrm, dz = z / nrm;
  const float* P = poses + (size_t)b * 16;
  const size_t o = ((size_t)b * n + i) * 3;
  // rays_d = dir @ R^T  ->  d_r = sum_c dir_c * R[r][c]
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rays_d[o + r] = dx * P[r * 4 + 0] + dy * P[r * 4 + 1] + dz * P[r * 4 + 2];
    rays_o[o + r] = P[r * 4 + 3];
  }
  norms[(size_t)b * n + i] = nrm;
}

extern "C" int32_t ucsa_get_rays(const float* poses, uint32_t B, float fx,
                                 float fy, float cx, float cy, uint32_t H,
                                 uint32_t W, const int64_t* inds, uint32_t n,
                                 float* rays_o, float* rays_d, float* norms,
                                 void* stream) {
  UCSA_CHECK_ARG(poses, 0);
  UCSA_CHECK_ARG(B > 0, 1);
  UCSA_CHECK_ARG(H > 0 && W > 0, 6);
  UCSA_CHECK_ARG(inds || n == H * W, 9);
  UCSA_CHECK_ARG(rays_o && rays_d && norms, 10);
  if (n == 0) return 0;
  dim3 grid(ucsa_div_up(n, 256), B);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_get_rays, grid, dim3(256), 0, (hipStream_t)stream, poses,
                     B, fx, fy, cx, cy, W, inds, n, rays_o, rays_d, norms);
  return ucsa_launch_status();
}

// ---------------------------------------------------------------------------
// Training pixel indices (get_rays_train, reference
// joint_train_lightning_net.py:141) handed over tile by tile: the same multiset
// of indices, ordered by (tile row, tile column, row in tile, column in tile).
// The key is a bijection of the pixel index, so only 32-bit keys are sorted
// (one workgroup, bitonic network in LDS) and decoded afterwards.
// ---------------------------------------------------------------------------
#define TILE_ORDER_MAX 8192

__global__ void __launch_bounds__(1024)
k_tile_order(const int64_t* __restrict__ inds, uint32_t n, uint32_t n_pad,
             uint32_t W, uint32_t tile, int64_t* __restrict__ out) {
  __shared__ uint32_t keys[TILE_ORDER_MAX];
  const uint32_t tiles_x = (W + tile - 1u) / tile, tt = tile * tile;
  for (uint32_t i = threadIdx.x; i < n_pad; i += 1024) {
    uint32_t k = 0xFFFFFFFFu;
    if (i < n) {
      const uint32_t pix = (uint32_t)inds[i], y = pix / W, x = pix % W;
      k = ((y / tile) * tiles_x + x / tile) * tt + (y % tile) * tile + x % tile;
    }
    keys[i] = k;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= n_pad; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < n_pad; i += 1024) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const uint32_t a = keys[i], b = keys[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            keys[i] = b;
            keys[l] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = threadIdx.x; i < n; i += 1024) {
    const uint32_t k = keys[i], t_id = k / tt, r = k % tt;
    const uint32_t y = (t_id / tiles_x) * tile + r / tile,
                   x = (t_id % tiles_x) * tile + r % tile;
    out[i] = (int64_t)y * W + x;
  }
}

extern "C" int32_t ucsa_tile_order(const int64_t* inds, uint32_t n, uint32_t H,
                                   uint32_t W, uint32_t tile, int64_t* out,
                                   void* stream) {
  UCSA_CHECK_ARG(inds, 0);
  UCSA_CHECK_ARG(n <= TILE_ORDER_MAX, 1);
  UCSA_CHECK_ARG(H > 0 && W > 0 && (uint64_t)H * W < 0x7FFFFFFFull, 2);
  UCSA_CHECK_ARG(tile >= 1 && tile <= 1024, 4);
  // the key of the last pixel must fit 32 bits below the padding value
  UCSA_CHECK_ARG((uint64_t)((H + tile - 1) / tile) * ((W + tile - 1) / tile) *
                     tile * tile < 0xFFFFFFFFull, 4);
  UCSA_CHECK_ARG(out, 5);
  if (n == 0) return 0;
  uint32_t n_pad = 2;
  while (n_pad < n) n_pad <<= 1;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                     inds, n, n_pad, W, tile, out);
  return ucsa_launch_status();
}

// ---------------------------------------------------------------------------
// a2: slab test.  reference raymarching.cu:62-115.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void near_far_one(const float ox, const float oy,
                                             const float oz, const float dx,
                                             const float dy, const float dz,
                                             const Aabb bb,
                                             const float min_near,
                                             float& near_out, float& far_out) {
  const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;
  float near = (bb.lo[0] - ox) * rdx;
  float far = (bb.hi[0] - ox) * rdx;
  if (near > far) { float t = near; near = far; far = t; }
  float ny = (bb.lo[1] - oy) * rdy;
  float fy = (bb.hi[1] - oy) * rdy;
  if (ny > fy) { float t = ny; ny = fy; fy = t; }
  if (near > fy || ny > far) { near_out = far_out = FLT_MAX; return; }
  if (ny > near) near = ny;
  if (fy < far) far = fy;
  float nz = (bb.lo[2] - oz) * rdz;
  float fz = (bb.hi[2] - oz) * rdz;
  if (nz > fz) { float t = nz; nz = fz; fz = t; }
  if (near > fz || nz > far) { near_out = far_out = FLT_MAX; return; }
  if (nz > near) near = nz;
  if (fz < far) far = fz;
  if (near < min_near) near = min_near;
  near_out = near;
  far_out = far;
}

__global__ void k_near_far(const float* __restrict__ rays_o,
                           const float* __restrict__ rays_d, Aabb bb,
                           uint32_t N, float min_near,
                           float* __restrict__ nears,
                           float* __restrict__ fars) {
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* o = rays_o + (size_t)n * 3;
  const float* d = rays_d + (size_t)n * 3;
  float a, b;
  near_far_one(o[0], o[1], o[2], d[0], d[1], d[2], bb, min_near, a, b);
  nears[n] = a;
  fars[n] = b;
}

extern "C" int32_t ucsa_near_far_from_aabb(const float* rays_o,
                                           const float* rays_d,
                                           const float* aabb_host, uint32_t N,
                                           float min_near, float* nears,
                                           float* fars, void* stream) {
  UCSA_CHECK_ARG(rays_o, 0);
  UCSA_CHECK_ARG(rays_d, 1);
  UCSA_CHECK_ARG(aabb_host, 2);
  UCSA_CHECK_ARG(nears && fars, 5);
  if (N == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_near_far, dim3(ucsa_div_up(N, 256)), dim3(256), 0,
                     (hipStream_t)stream, rays_o, rays_d, ucsa_aabb(aabb_host),
                     N, min_near, nears, fars);
  return ucsa_launch_status();
}

// ---------------------------------------------------------------------------
// a3: coarse depths.  reference renderer_semantics.py:154-168.
// linspace(0,1,T) is evaluated like torch's kernel: step=(1-0)/(T-1) in fp32;
// the first half counts up from 0, the second half counts down from 1 with a
// fused multiply-add (verified bit-equal to torch.linspace on the CPU).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float linspace01(uint32_t i, uint32_t T) {
  if (T == 1) return 0.0f;
  const float step = 1.0f / (float)(T - 1);
  return (i < T / 2) ? step * (float)i : fmaf(-step, (float)(T - 1 - i), 1.0f);
}

__device__ __forceinline__ float coarse_z_at(float near, float far, uint32_t i,
                                             uint32_t T) {
  return near + (far - near) * linspace01(i, T);
}

__global__ void k_sample_coarse(const float* __restrict__ nears,
                                const float* __restrict__ fars,
                                const float* __restrict__ t_rand, uint32_t N,
                                uint32_t T, float* __restrict__ z) {
  const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= (uint64_t)N * T) return;
  const uint32_t r = (uint32_t)(m / T), i = (uint32_t)(m % T);
  const float near = nears[r], far = fars[r];
  float zi = coarse_z_at(near, far, i, T);
  if (t_rand) {
    const float zm = i > 0 ? coarse_z_at(near, far, i - 1, T) : 0.f;
    const float zp = i + 1 < T ? coarse_z_at(near, far, i + 1, T) : 0.f;
    const float lower = i > 0 ? 0.5f * (zi + zm) : zi;
    const float upper = i + 1 < T ? 0.5f * (zp + zi) : zi;
    zi = lower + (upper - lower) * t_rand[m];
  }
  z[m] = zi;
}

extern "C" int32_t ucsa_sample_coarse(const float* nears, const float* fars,
                                      const float* t_rand, uint32_t N,
                                      uint32_t T, float* z, void* stream) {
  UCSA_CHECK_ARG(nears, 0);
  UCSA_CHECK_ARG(fars, 1);
  UCSA_CHECK_ARG(T > 0, 4);
  UCSA_CHECK_ARG(z, 5);
  if (N == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_sample_coarse, dim3(ucsa_div_up((uint64_t)N * T, 256)),
                     dim3(256), 0, (hipStream_t)stream, nears, fars, t_rand, N,
                     T, z);
  return ucsa_launch_status();
}
