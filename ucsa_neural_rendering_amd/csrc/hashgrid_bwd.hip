// Hash-grid backward: scatter d_feat into the table gradient with fp32
// atomics (restates the autograd of tcnn's GridEncoding; call site reference
// nr4seg/nerf/network_tcnn_semantics.py:133-134).
//
// Level-major like the forward (one level's 4 MiB gradient slab is L2
// resident while it is being hit).  Lanes of a wave are consecutive samples
// of a ray: on coarse levels many of them add to the SAME entry, so equal
// (index) lanes are first combined inside the wave with a match-and-reduce
// over the ballot of each distinct index... kept simple here: plain atomics,
// the contention optimisation is DESIGN.md "next".
// Float atomics make the table gradient order-dependent at fp32 round-off
// (run-to-run ~1e-7 relative); see DESIGN.md "determinism".
#include "ucsa_common.h"

#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u

__device__ __forceinline__ uint32_t grid_index_b(uint32_t x, uint32_t y,
                                                 uint32_t z, uint32_t res,
                                                 uint32_t entries,
                                                 uint32_t hashed) {
  uint32_t idx = hashed ? (x ^ (y * PRIME_Y) ^ (z * PRIME_Z))
                        : (x + y * res + z * res * res);
  return hashed ? (idx & (entries - 1)) : (idx % entries);
}

__device__ __forceinline__ float clampf_b(float v, float lo, float hi) {
  return fminf(fmaxf(v, lo), hi);
}

// Run-combining: lanes of a wave are consecutive samples of a ray, so on the
// coarse levels consecutive lanes fall into the same cell and would hammer the
// same table entry (measured before this: 85 % of a training step).  Equal
// indices in CONSECUTIVE lanes are summed with a segmented wave scan and only
// the last lane of each run issues the atomic.  Non-adjacent duplicates just
// cost an extra atomic.
__device__ __forceinline__ void run_combine(uint32_t idx, float& vx, float& vy,
                                            uint32_t lane, bool& is_tail) {
  const uint32_t prev = (uint32_t)__shfl_up((int)idx, 1, 64);
  int f = (lane == 0 || prev != idx) ? 1 : 0;  // run head
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float ox = __shfl_up(vx, d, 64);
    const float oy = __shfl_up(vy, d, 64);
    const int of = __shfl_up(f, d, 64);
    if (lane >= (uint32_t)d && !f) {
      vx += ox;
      vy += oy;
      f |= of;
    }
  }
  const uint32_t next = (uint32_t)__shfl_down((int)idx, 1, 64);
  is_tail = (lane == 63) || (next != idx);
}

template <bool RUNRED>
__global__ void __launch_bounds__(256)
k_hashgrid_bwd(GridDev g, const float* __restrict__ rays_o,
               const float* __restrict__ rays_d, const float* __restrict__ zs,
               Aabb bb, uint32_t T, uint64_t M, uint32_t level0,
               const float2* __restrict__ d_feat,
               float* __restrict__ grad_table) {
  const uint32_t level = level0 + blockIdx.y;
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = m < M;
  if (!in_range) m = M - 1;
  float2 df = d_feat[(uint64_t)level * M + m];
  const bool act = in_range && !(df.x == 0.0f && df.y == 0.0f);
  if (!RUNRED && !act) return;
  const uint32_t r = (uint32_t)(m / T);
  const float zz = zs[m];
  const float* o = rays_o + (size_t)r * 3;
  const float* d = rays_d + (size_t)r * 3;
  const float px = clampf_b(o[0] + d[0] * zz, bb.lo[0], bb.hi[0]);
  const float py = clampf_b(o[1] + d[1] * zz, bb.lo[1], bb.hi[1]);
  const float pz = clampf_b(o[2] + d[2] * zz, bb.lo[2], bb.hi[2]);
  const float two_b = 2.0f * g.bound;
  const float scale = g.scale[level];
  const float x = (px + g.bound) / two_b * scale + 0.5f;
  const float y = (py + g.bound) / two_b * scale + 0.5f;
  const float z = (pz + g.bound) / two_b * scale + 0.5f;
  const float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
  const float wx = x - fx0, wy = y - fy0, wz = z - fz0;
  const uint32_t gx = (uint32_t)(int32_t)fx0, gy = (uint32_t)(int32_t)fy0,
                 gz = (uint32_t)(int32_t)fz0;
  float* gt = grad_table + (size_t)g.offset[level] * 2;
  const uint32_t res = g.res[level], entries = g.entries[level],
                 hashed = g.hashed[level];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w = (c & 1) ? wx : 1.0f - wx;
    w = w * ((c & 2) ? wy : 1.0f - wy);
    w = w * ((c & 4) ? wz : 1.0f - wz);
    uint32_t idx = grid_index_b(gx + (c & 1), gy + ((c >> 1) & 1),
                                gz + ((c >> 2) & 1), res, entries, hashed);
    float vx = w * df.x, vy = w * df.y;
    if (RUNRED) {
      if (!act) { idx = 0xFFFFFFFFu; vx = 0.f; vy = 0.f; }
      bool tail;
      run_combine(idx, vx, vy, lane, tail);
      if (tail && idx != 0xFFFFFFFFu) {
        atomicAdd(gt + (size_t)idx * 2, vx);
        atomicAdd(gt + (size_t)idx * 2 + 1, vy);
      }
    } else {
      atomicAdd(gt + (size_t)idx * 2, vx);
      atomicAdd(gt + (size_t)idx * 2 + 1, vy);
    }
  }
}

extern "C" int32_t ucsa_hashgrid_bwd_rays(const ucsa_grid* grid,
                                          const float* rays_o,
                                          const float* rays_d, const float* z,
                                          const float* aabb_host, uint32_t N,
                                          uint32_t T, const float* d_feat,
                                          float* grad_table, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 1);
  UCSA_CHECK_ARG(aabb_host, 4);
  UCSA_CHECK_ARG(d_feat, 7);
  UCSA_CHECK_ARG(grad_table, 8);
  const uint64_t M = (uint64_t)N * T;
  if (M == 0) return 0;
  // levels whose cells are wider than about one sample spacing get the
  // run-combining variant (cell = 2*bound/scale, spacing ~ 2*bound*sqrt(3)/T)
  uint32_t n_run = 0;
  while (n_run < grid->n_levels &&
         grid->level[n_run].scale < 0.6f * (float)T) ++n_run;
  const GridDev gd = ucsa_grid_dev(grid);
  const Aabb bb = ucsa_aabb(aabb_host);
  UCSA_CLEAR_ERR();
  if (n_run > 0)
    hipLaunchKernelGGL(k_hashgrid_bwd<true>, dim3(ucsa_div_up(M, 256), n_run),
                       dim3(256), 0, (hipStream_t)stream, gd, rays_o, rays_d, z,
                       bb, T, M, 0u, (const float2*)d_feat, grad_table);
  if (n_run < grid->n_levels)
    hipLaunchKernelGGL(k_hashgrid_bwd<false>,
                       dim3(ucsa_div_up(M, 256), grid->n_levels - n_run),
                       dim3(256), 0, (hipStream_t)stream, gd, rays_o, rays_d, z,
                       bb, T, M, n_run, (const float2*)d_feat, grad_table);
  return ucsa_launch_status();
}
