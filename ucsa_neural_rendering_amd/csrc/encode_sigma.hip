// Hash-grid encode + sigma MLP in ONE kernel for image-ordered rays
// (SURVEY 8a row a4 end to end: reference
// nr4seg/nerf/network_tcnn_semantics.py:130-144, density()).
//
// The staged pair (k_hashgrid_encode_tiled, then k_sigma_mlp) writes the 32
// features of every sample to HBM (128 B) and reads them back: 1.5 GB per
// 5.9 M-sample launch, the largest single intermediate of the render.  Here a
// workgroup gathers ALL 16 levels of its samples -- an 8x8 pixel tile at 8
// consecutive sample indices, 512 samples -- into LDS planes
// [level][component][sample] (66.6 KB, two workgroups per CU), then runs the
// sigma MLP straight from LDS and writes only h [M,16] and sigma [M].
//
// What this gives up ("why not one fused kernel"): the staged gather
// is level-major in time, so the 4 MiB table slice of the level in flight sits
// in every XCD's 4 MiB L2; here every resident workgroup walks all levels and
// the live table is the whole 52 MB (it fits the 256 MB Infinity Cache, not
// the L2s).  Workgroups that start together stay roughly level-synchronous,
// and the workgroup order is XCD-aware (an XCD works through neighbouring
// tiles / sample blocks), which is what the measurement in
// profiles/r02_encode_sigma_fused.txt is about.
//
// Arithmetic: encode_level / encode_level_hashed and the MFMA chain of
// k_sigma_mlp, unchanged -> h and sigma are bit-identical to the staged pair.
#include <cstdlib>

#include "hashgrid_common.h"
#include "mfma_mlp_f16.h"

// ES_S sample indices per workgroup (template: 8 -> 512 samples, 66.6 KB of
// LDS, 2 workgroups per CU; 4 -> 256 samples, 33.8 KB, 4 per CU);
// ES_PLANE = samples + 8: 2*ES_PLANE = 16 mod 64, so the four lane groups of
// an MFMA operand read hit disjoint bank windows
#define ES_UNROLL 4                 // column blocks in flight in the MLP phase

template <bool HALF, int ES_S>
__global__ void __launch_bounds__(256)
k_encode_sigma_tiled(GridDev g, const float2* __restrict__ table,
                     const float* __restrict__ rays_o,
                     const float* __restrict__ rays_d,
                     const float* __restrict__ zs, Aabb bb, uint32_t T,
                     uint32_t N, uint32_t W, uint32_t s_blocks,
                     uint32_t n_blocks, const void* __restrict__ packed,
                     float* __restrict__ h, float* __restrict__ sigma) {
  extern __shared__ __attribute__((aligned(16))) float es_smem[];
  constexpr uint32_t ES_N = 64 * ES_S, ES_PLANE = ES_N + 8;
  constexpr uint32_t KPW = ES_S / 4;      // sample indices per wave and level
  constexpr uint32_t CPW = ES_N / 64;     // column blocks per wave (MLP phase)
  float* f_s = es_smem;                                // [16][2][ES_PLANE]
  float* z_s = es_smem + 32 * ES_PLANE;                // [64][ES_S + 1]
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  // XCD-aware order: hardware block b runs on XCD b % 8; give XCD x the
  // contiguous logical range [x * n/8, (x+1) * n/8)
  const uint32_t per_xcd = (n_blocks + 7u) / 8u;
  const uint32_t bid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (bid >= n_blocks) return;
  const uint32_t sb = bid % s_blocks, tile = bid / s_blocks;
  const uint32_t tiles_x = (W + 7u) / 8u;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  const uint32_t s0 = sb * ES_S;
  auto ray_of = [&](uint32_t l) -> uint32_t {
    const uint32_t px = tx * 8 + (l & 7u), py = ty * 8 + (l >> 3);
    const uint64_t r = (uint64_t)py * W + px;
    return (px < W && r < N) ? (uint32_t)r : 0xFFFFFFFFu;
  };
  // depths of the tile, ray-major reads (32 B per ray)
#pragma unroll
  for (uint32_t k = 0; k < ES_N / 256; ++k) {
    const uint32_t e = threadIdx.x + 256u * k;
    const uint32_t r = ray_of(e / ES_S), ss = e % ES_S;
    z_s[(e / ES_S) * (ES_S + 1) + ss] =
        (r != 0xFFFFFFFFu && s0 + ss < T) ? zs[(uint64_t)r * T + s0 + ss] : 0.0f;
  }
  __syncthreads();
  // ---- gather: lane = pixel, wave = 2 sample indices, all 16 levels ---------
  const uint32_t ray = ray_of(lane);
  {
    const uint32_t rr = ray != 0xFFFFFFFFu ? ray : 0u;  // clamp loads
    const float* o = rays_o + (size_t)rr * 3;
    const float* d = rays_d + (size_t)rr * 3;
    const float ox = o[0], oy = o[1], oz = o[2];
    const float dx = d[0], dy = d[1], dz = d[2];
    const float two_b = 2.0f * g.bound, inv = unit_inv(two_b);
    float x01[KPW], y01[KPW], z01[KPW];
#pragma unroll
    for (uint32_t k = 0; k < KPW; ++k) {
      const uint32_t ss = wid * KPW + k;
      const float zz = z_s[lane * (ES_S + 1) + ss];
      const float px = clampf(ox + dx * zz, bb.lo[0], bb.hi[0]);
      const float py = clampf(oy + dy * zz, bb.lo[1], bb.hi[1]);
      const float pz = clampf(oz + dz * zz, bb.lo[2], bb.hi[2]);
      x01[k] = to_unit(px, g.bound, two_b, inv);
      y01[k] = to_unit(py, g.bound, two_b, inv);
      z01[k] = to_unit(pz, g.bound, two_b, inv);
    }
#pragma unroll 4
    for (uint32_t level = 0; level < 16; ++level) {
      const float2* tab = table + g.offset[level];
      const float scale = g.scale[level];
      const uint32_t res = g.res[level], entries = g.entries[level],
                     hashed = g.hashed[level];
#pragma unroll
      for (uint32_t k = 0; k < KPW; ++k) {
        const float2 f = hashed
            ? encode_level_hashed(tab, x01[k], y01[k], z01[k], scale, entries)
            : encode_level(tab, x01[k], y01[k], z01[k], scale, res, entries, 0u);
        const uint32_t sidx = lane * ES_S + wid * KPW + k;
        f_s[(level * 2 + 0) * ES_PLANE + sidx] = f.x;
        f_s[(level * 2 + 1) * ES_PLANE + sidx] = f.y;
      }
    }
  }
  __syncthreads();
  // ---- sigma MLP from LDS: wave = 8 column blocks of 16 samples -------------
  const uint32_t gq = lane >> 4, j = lane & 15u;
  if constexpr (!HALF) {
    const float* pk = reinterpret_cast<const float*>(packed);
    float w1[4][8], w2[16];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) w1[rb][ks] = pk[(rb * 8 + ks) * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w2[ks] = pk[(SIGMA_L1_FRAGS + ks) * 64 + lane];
    for (uint32_t c0 = wid * CPW; c0 < wid * CPW + CPW; c0 += ES_UNROLL) {
      float xin[ES_UNROLL][8];
#pragma unroll
      for (int u = 0; u < ES_UNROLL; ++u) {
        const uint32_t sidx = (c0 + u) * 16 + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xin[u][2 * q] = f_s[((4 * q + gq) * 2 + 0) * ES_PLANE + sidx];
          xin[u][2 * q + 1] = f_s[((4 * q + gq) * 2 + 1) * ES_PLANE + sidx];
        }
      }
#pragma unroll
      for (int u = 0; u < ES_UNROLL; ++u) {
        f32x4 acc[4];
        mfma_layer<8, 4>(xin[u], [&](int rb, int ks) { return w1[rb][ks]; }, acc);
        float hid[16];
        chain_relu(acc, hid);
        f32x4 out[1];
        mfma_layer<16, 1>(hid, [&](int, int ks) { return w2[ks]; }, out);
        const uint32_t sidx = (c0 + u) * 16 + j;
        const uint32_t r = ray_of(sidx / ES_S), ss = sidx % ES_S;
        if (r != 0xFFFFFFFFu && s0 + ss < T) {
          const uint64_t m = (uint64_t)r * T + s0 + ss;
          *reinterpret_cast<f32x4*>(h + m * 16 + 4 * gq) = out[0];
          if (gq == 0) sigma[m] = expf(out[0][0]);
        }
      }
    }
  } else {
    half8 w1[4], w2[2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) w1[rb] = frag_h(packed, rb, lane);
#pragma unroll
    for (int s = 0; s < 2; ++s) w2[s] = frag_h(packed, 4 + s, lane);
#pragma unroll
    for (uint32_t u = 0; u < CPW; ++u) {
      const uint32_t sidx = (wid * CPW + u) * 16 + j;
      half8 xin;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xin[2 * q] = (_Float16)f_s[((4 * q + gq) * 2 + 0) * ES_PLANE + sidx];
        xin[2 * q + 1] = (_Float16)f_s[((4 * q + gq) * 2 + 1) * ES_PLANE + sidx];
      }
      const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 a1[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a1[rb] = mfma_h(w1[rb], xin, z4);
      f32x4 out = mfma_h(w2[0], chain_relu_h(a1[0], a1[1]), z4);
      out = mfma_h(w2[1], chain_relu_h(a1[2], a1[3]), out);
      const uint32_t r = ray_of(sidx / ES_S), ss = sidx % ES_S;
      if (r != 0xFFFFFFFFu && s0 + ss < T) {
        const uint64_t m = (uint64_t)r * T + s0 + ss;
        *reinterpret_cast<f32x4*>(h + m * 16 + 4 * gq) = out;
        if (gq == 0) sigma[m] = expf(out[0]);
      }
    }
  }
}

static int32_t encode_sigma(bool half, const ucsa_grid* grid, const float* table,
                            const void* packed_sigma, const float* rays_o,
                            const float* rays_d, const float* z,
                            const float* aabb_host, uint32_t N, uint32_t T,
                            uint32_t image_width, float* h, float* sigma,
                            void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels == 16, 0);
  UCSA_CHECK_ARG(table, 1);
  UCSA_CHECK_ARG(packed_sigma, 2);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 3);
  UCSA_CHECK_ARG(aabb_host, 6);
  UCSA_CHECK_ARG(image_width >= 1, 9);
  UCSA_CHECK_ARG(h && sigma, 10);
  if ((uint64_t)N * T == 0) return 0;
  const GridDev gd = ucsa_grid_dev(grid);
  const uint32_t rows = ucsa_div_up(N, image_width);
  const uint32_t tiles = ((image_width + 7u) / 8u) * ((rows + 7u) / 8u);
  // sample indices per workgroup: 4 by default, UCSA_ES_S=8 for the other shape
  const char* ev = getenv("UCSA_ES_S");
  const uint32_t es = (ev && ev[0] == '8') ? 8u : 4u;
  const uint32_t s_blocks = ucsa_div_up(T, es);
  const uint32_t n_blocks = tiles * s_blocks;
  const uint32_t launch = (n_blocks + 7u) / 8u * 8u;
  const size_t smem = (32 * (size_t)(64 * es + 8) + 64 * (es + 1)) * 4;
  hipStream_t s = (hipStream_t)stream;
  const Aabb bb = ucsa_aabb(aabb_host);
#define ES_LAUNCH(H, SS)                                                      \
  do {                                                                        \
    hipError_t e = hipFuncSetAttribute(                                       \
        reinterpret_cast<const void*>(&k_encode_sigma_tiled<H, SS>),          \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);               \
    if (e != hipSuccess) return -(int32_t)e;                                  \
    UCSA_CLEAR_ERR();                                                         \
    hipLaunchKernelGGL((k_encode_sigma_tiled<H, SS>), dim3(launch), dim3(256), \
                       smem, s, gd, (const float2*)table, rays_o, rays_d, z,  \
                       bb, T, N, image_width, s_blocks, n_blocks,             \
                       packed_sigma, h, sigma);                               \
  } while (0)
  if (half) {
    if (es == 8) ES_LAUNCH(true, 8); else ES_LAUNCH(true, 4);
  } else {
    if (es == 8) ES_LAUNCH(false, 8); else ES_LAUNCH(false, 4);
  }
#undef ES_LAUNCH
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_encode_sigma_rays_image(
    const ucsa_grid* grid, const float* table, const float* packed_sigma,
    const float* rays_o, const float* rays_d, const float* z,
    const float* aabb_host, uint32_t N, uint32_t T, uint32_t image_width,
    float* h, float* sigma, void* stream) {
  return encode_sigma(false, grid, table, packed_sigma, rays_o, rays_d, z,
                      aabb_host, N, T, image_width, h, sigma, stream);
}

extern "C" int32_t ucsa_encode_sigma_rays_image_f16(
    const ucsa_grid* grid, const float* table, const void* packed_sigma_half,
    const float* rays_o, const float* rays_d, const float* z,
    const float* aabb_host, uint32_t N, uint32_t T, uint32_t image_width,
    float* h, float* sigma, void* stream) {
  return encode_sigma(true, grid, table, packed_sigma_half, rays_o, rays_d, z,
                      aabb_host, N, T, image_width, h, sigma, stream);
}
