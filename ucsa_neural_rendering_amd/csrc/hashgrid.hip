// Multiresolution hash-grid encoding (first half).
//
// Restates tiny-cuda-nn's GridEncoding (public algorithm; call site reference
// nr4seg/nerf/network_tcnn_semantics.py:36-46,133-134).
//
// MI355X layout decisions (DESIGN.md "hash-grid encode"):
//  * LEVEL-MAJOR launch: grid = (sample blocks, levels) so that all CUs work
//    on one level at a time; one level of the table (<= 4 MiB fp32) then lives
//    in each XCD's 4 MiB L2 instead of 52 MiB of table thrashing it.
//  * Output is level-major too: feat[level][sample] as float2, so a wave
//    stores 512 contiguous bytes; the sigma-MLP kernel reads the same way.
//  * One lane per sample; lanes of a wave are consecutive samples of one ray,
//    which share cells on the coarse levels (the TA coalesces equal lines).
#include <cstdlib>
#include <cstring>

#include "hashgrid_common.h"

template <bool FROM_RAYS>
__device__ __forceinline__ void sample_x01(const GridDev& g,
                                           const float* __restrict__ rays_o,
                                           const float* __restrict__ rays_d,
                                           const float* __restrict__ zs,
                                           const Aabb& bb, uint32_t T,
                                           uint64_t m, float& x01, float& y01,
                                           float& z01) {
  float px, py, pz;
  if (FROM_RAYS) {
    const uint32_t r = (uint32_t)(m / T);
    const float zz = zs[m];
    const float* o = rays_o + (size_t)r * 3;
    const float* d = rays_d + (size_t)r * 3;
    px = clampf(o[0] + d[0] * zz, bb.lo[0], bb.hi[0]);
    py = clampf(o[1] + d[1] * zz, bb.lo[1], bb.hi[1]);
    pz = clampf(o[2] + d[2] * zz, bb.lo[2], bb.hi[2]);
  } else {
    const float* x = rays_o + (size_t)m * 3;  // explicit points
    px = x[0];
    py = x[1];
    pz = x[2];
  }
  const float two_b = 2.0f * g.bound, inv = unit_inv(two_b);
  x01 = to_unit(px, g.bound, two_b, inv);
  y01 = to_unit(py, g.bound, two_b, inv);
  z01 = to_unit(pz, g.bound, two_b, inv);
}

// Coarse levels [0, n_coarse): cells span several samples of a ray, gathers
// hit L1/L2, and a per-level launch would be all fixed cost (~25 us each,
// measured) -- so one thread walks all of them.
template <bool FROM_RAYS, typename TT = float2, typename FT = float2>
__global__ void __launch_bounds__(256)
k_hashgrid_encode_coarse(GridDev g, uint32_t n_coarse,
                         const TT* __restrict__ table,
                         const float* __restrict__ rays_o,
                         const float* __restrict__ rays_d,
                         const float* __restrict__ zs, Aabb bb, uint32_t T,
                         uint64_t M, FT* __restrict__ feat) {
  const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float x01, y01, z01;
  sample_x01<FROM_RAYS>(g, rays_o, rays_d, zs, bb, T, m, x01, y01, z01);
  for (uint32_t level = 0; level < n_coarse; ++level) {
    const float2 f = encode_level(table + g.offset[level], x01, y01, z01,
                                  g.scale[level], g.res[level],
                                  g.entries[level], g.hashed[level]);
    feat_store(feat + (uint64_t)level * M + m, f);
  }
}

// Fine levels [level0, n_levels): level-major (see file header).
template <bool FROM_RAYS, typename TT = float2, typename FT = float2>
__global__ void __launch_bounds__(256)
k_hashgrid_encode(GridDev g, uint32_t level0, uint32_t simple_below,
                  const TT* __restrict__ table,
                  const float* __restrict__ rays_o,
                  const float* __restrict__ rays_d,
                  const float* __restrict__ zs, Aabb bb, uint32_t T,
                  uint64_t M, FT* __restrict__ feat) {
  const uint32_t level = level0 + blockIdx.y;
  const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float x01, y01, z01;
  sample_x01<FROM_RAYS>(g, rays_o, rays_d, zs, bb, T, m, x01, y01, z01);
  float2 f;
  if (g.hashed[level] && level >= simple_below)
    f = encode_level_hashed(table + g.offset[level], x01, y01, z01,
                            g.scale[level], g.entries[level]);
  else
    f = encode_level(table + g.offset[level], x01, y01, z01, g.scale[level],
                     g.res[level], g.entries[level], g.hashed[level]);
  feat_store(feat + (uint64_t)level * M + m, f);
}


// Fine levels for IMAGE-ORDERED rays (ray r = pixel (r / W, r % W) of full
// rows of an image W pixels wide): a wave covers an 8x8 pixel tile at ONE
// sample index instead of 64 consecutive samples of one ray.  Along a ray the
// samples are ~0.06 apart -- a new cell for every lane from level 7 up -- but
// the 64 pixels of a tile at equal depth span only ~0.04 x 0.04: on levels
// 6..11 most lanes fall into the same few cells, the TA coalesces their equal
// lines, and the gather stops being bound by the L2->L1 fill rate.  The block
// (4 waves) covers 16 sample indices of the tile; depths come in and features
// go out through LDS so that global accesses stay ray-major and contiguous
// (64 B of z, 128 B of features per ray).  Arithmetic per sample is unchanged:
// the features are bit-identical to k_hashgrid_encode's.
#define TILE_S 16
// Which level a workgroup works on: grid row y holds the `k` levels
// lv[y * k .. y * k + k - 1], interleaved along x (workgroup x -> level
// x % k).  k = 1 with the identity map is plain level-major order (the whole
// chip on one table slice at a time); k = 2 pairing a coarse level with a
// fine one puts a VALU-bound and a gather-bound workgroup on the same CU at
// the same time.
struct LevelMap {
  uint8_t lv[UCSA_MAX_LEVELS];
  uint32_t k;
  // hashed levels below this index take the plain 8-load gather instead of
  // the x-pair form (6 accesses + 24 selects per sample): on the coarse
  // hashed levels the lanes of a wave share their cells, the gathers hit the
  // L1 and the level is bound by VALU issue, not by the TA
  uint32_t simple_below;
};
template <typename TT = float2, typename FT = float2>
__global__ void __launch_bounds__(256)
k_hashgrid_encode_tiled(GridDev g, LevelMap lm,
                        const TT* __restrict__ table,
                        const float* __restrict__ rays_o,
                        const float* __restrict__ rays_d,
                        const float* __restrict__ zs, Aabb bb, uint32_t T,
                        uint32_t N, uint32_t W, uint32_t s_blocks,
                        FT* __restrict__ feat) {
  __shared__ float z_s[64][TILE_S + 1];
  __shared__ FT f_s[64][TILE_S + 1];
  const uint32_t level = lm.lv[blockIdx.y * lm.k + blockIdx.x % lm.k];
  const uint32_t bx = blockIdx.x / lm.k;
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t sb = bx % s_blocks, tile = bx / s_blocks;
  const uint32_t tiles_x = (W + 7u) / 8u;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  const uint32_t s0 = sb * TILE_S;
  const uint64_t M = (uint64_t)N * T;
  FT* feat_level = feat + (uint64_t)level * M;  // wave-uniform
  // (32-bit indices: the host checks N * T < 2^31; rows beyond the last one
  // give r >= N)
  auto ray_of = [&](uint32_t l) -> uint32_t {
    const uint32_t px = tx * 8 + (l & 7u), py = ty * 8 + (l >> 3);
    const uint32_t r = py * W + px;
    return (px < W && r < N) ? r : 0xFFFFFFFFu;
  };
  // depths of the tile, ray-major reads
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t e = threadIdx.x + 256u * k;
    const uint32_t r = ray_of(e / TILE_S), ss = e % TILE_S;
    if (r != 0xFFFFFFFFu && s0 + ss < T)
      z_s[e / TILE_S][ss] = zs[r * T + s0 + ss];
  }
  __syncthreads();
  const uint32_t ray = ray_of(lane);
  if (ray != 0xFFFFFFFFu) {
    const float* o = rays_o + ray * 3u;
    const float* d = rays_d + ray * 3u;
    const float ox = o[0], oy = o[1], oz = o[2];
    const float dx = d[0], dy = d[1], dz = d[2];
    const float two_b = 2.0f * g.bound, inv = unit_inv(two_b);
    const TT* tab = table + g.offset[level];
    const float scale = g.scale[level];
    const uint32_t res = g.res[level], entries = g.entries[level],
                   hashed = g.hashed[level];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t ss = wid + 4u * k;
      if (s0 + ss >= T) continue;
      const float zz = z_s[lane][ss];
      const float px = clampf(ox + dx * zz, bb.lo[0], bb.hi[0]);
      const float py = clampf(oy + dy * zz, bb.lo[1], bb.hi[1]);
      const float pz = clampf(oz + dz * zz, bb.lo[2], bb.hi[2]);
      const float x01 = to_unit(px, g.bound, two_b, inv),
                  y01 = to_unit(py, g.bound, two_b, inv),
                  z01 = to_unit(pz, g.bound, two_b, inv);
      to_feat(f_s[lane][ss],
              (hashed && level >= lm.simple_below)
                  ? encode_level_hashed(tab, x01, y01, z01, scale, entries)
                  : encode_level(tab, x01, y01, z01, scale, res, entries, hashed));
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t e = threadIdx.x + 256u * k;
    const uint32_t r = ray_of(e / TILE_S), ss = e % TILE_S;
    if (r != 0xFFFFFFFFu && s0 + ss < T)
      feat_store(feat_level + (r * T + s0 + ss), f_s[e / TILE_S][ss]);
  }
}

// ---------------------------------------------------------------------------
// Round 5: SEVERAL levels per workgroup for the levels that are bound by
// instruction issue, not by the gather (per-level times of the kernel above on
// the bench's chunk: levels 0-8 cost 31-37 us each in the coarse pass whatever
// their table -- ~205 VALU instructions per sample and level, a third of them
// the per-workgroup preamble (tile geometry, z staging, ray loads, position) and
// another ~20 the position -> unit-cube arithmetic, all of it repeated for every
// level).  Here a workgroup keeps its tile's 4 samples per lane in registers as
// unit-cube coordinates and walks levels [l_lo, l_hi): per level only cell /
// fraction, the 8 indices, the gather and the blend remain; the level's
// features leave through a double-buffered LDS tile (one barrier per level).
// Same arithmetic per sample (encode_cell below = encode_level's operations in
// encode_level's order): bit-identical features.  The fine levels stay with
// k_hashgrid_encode_tiled: they are bound by the L2 -> L1 line fills of their
// random corners and their time follows the L1 miss count, which every
// restructuring of the code around the gather made worse.
// ---------------------------------------------------------------------------
// (encode_cell: hashgrid_common.h)
template <typename TT = float2, typename FT = float2>
__global__ void __launch_bounds__(256)
k_hashgrid_encode_tiled_ml(GridDev g, uint32_t l_lo, uint32_t l_hi,
                           const TT* __restrict__ table,
                           const float* __restrict__ rays_o,
                           const float* __restrict__ rays_d,
                           const float* __restrict__ zs, Aabb bb, uint32_t T,
                           uint32_t N, uint32_t W, uint32_t s_blocks,
                           FT* __restrict__ feat) {
  __shared__ float z_s[64][TILE_S + 1];
  __shared__ FT f_s[2][64][TILE_S + 1];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t sb = blockIdx.x % s_blocks, tile = blockIdx.x / s_blocks;
  const uint32_t tiles_x = (W + 7u) / 8u;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  const uint32_t s0 = sb * TILE_S;
  const uint32_t M = N * T;   // the host checks N * T < 2^31
  auto ray_of = [&](uint32_t l) -> uint32_t {
    const uint32_t px = tx * 8 + (l & 7u), py = ty * 8 + (l >> 3);
    const uint32_t r = py * W + px;
    return (px < W && r < N) ? r : 0xFFFFFFFFu;
  };
  // ray-major slots of the tile (depths in, features out): element e of the
  // workgroup's 1024 = (pixel e / 16, sample index e % 16)
  uint32_t slot[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t e = threadIdx.x + 256u * k;
    const uint32_t r = ray_of(e / TILE_S), ss = e % TILE_S;
    slot[k] = (r != 0xFFFFFFFFu && s0 + ss < T) ? r * T + s0 + ss : 0xFFFFFFFFu;
    if (slot[k] != 0xFFFFFFFFu) z_s[e / TILE_S][ss] = zs[slot[k]];
  }
  __syncthreads();
  const uint32_t ray = ray_of(lane);
  float ux[4], uy[4], uz[4];
  uint32_t live = 0u;   // which of the lane's 4 samples exist
  if (ray != 0xFFFFFFFFu) {
    const float* o = rays_o + ray * 3u;
    const float* d = rays_d + ray * 3u;
    const float ox = o[0], oy = o[1], oz = o[2];
    const float dx = d[0], dy = d[1], dz = d[2];
    const float two_b = 2.0f * g.bound, inv = unit_inv(two_b);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t ss = wid + 4u * k;
      ux[k] = uy[k] = uz[k] = 0.f;
      if (s0 + ss >= T) continue;
      live |= 1u << k;
      const float zz = z_s[lane][ss];
      const float px = clampf(ox + dx * zz, bb.lo[0], bb.hi[0]);
      const float py = clampf(oy + dy * zz, bb.lo[1], bb.hi[1]);
      const float pz = clampf(oz + dz * zz, bb.lo[2], bb.hi[2]);
      ux[k] = to_unit(px, g.bound, two_b, inv);
      uy[k] = to_unit(py, g.bound, two_b, inv);
      uz[k] = to_unit(pz, g.bound, two_b, inv);
    }
  }
  uint32_t buf = 0u;
  for (uint32_t level = l_hi; level-- > l_lo; buf ^= 1u) {   // finest first
    const TT* tab = table + g.offset[level];
    const float scale = g.scale[level];
    const uint32_t res = g.res[level], entries = g.entries[level],
                   hashed = g.hashed[level];
    const uint32_t res2 = res * res;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!(live >> k & 1u)) continue;
      to_feat(f_s[buf][lane][wid + 4u * k],
              encode_cell(tab, ux[k], uy[k], uz[k], scale, res, res2, entries, hashed));
    }
    __syncthreads();
    // (no second barrier: the next level fills the other buffer, and a thread
    // reaches the level after that only past the next barrier, i.e. after
    // every thread has finished these reads)
    FT* feat_level = feat + (size_t)level * M;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t e = threadIdx.x + 256u * k;
      if (slot[k] != 0xFFFFFFFFu)
        feat_store(feat_level + slot[k], f_s[buf][e / TILE_S][e % TILE_S]);
    }
  }
}

// how many leading levels go to the fused coarse kernel: all dense levels plus
// hashed ones whose cells are still wider than ~2 sample spacings
static uint32_t coarse_levels(const ucsa_grid* grid) {
  uint32_t n = 0;
  while (n < grid->n_levels && (!grid->level[n].hashed ||
                                grid->level[n].scale < 160.0f)) ++n;
  return n;
}

// Hashed levels below this index take the plain 8-load gather
// (encode_level) instead of the x-pair form (encode_level_hashed: 6 accesses
// but 24 selects per sample).  Measured in round 3 on the tiled kernel with
// the fp32 table (tools/encode_order_sweep.py, 61 440-ray chunk): the plain
// gather is FASTER on every level -- coarse pass 0.80 -> 0.64 ms, fine pass
// 1.41 -> 1.23 ms -- the x-pair trick of rounds 1-2 (then a gain on the
// ray-ordered kernel) costs more VALU issue than the accesses it saves now
// that the lanes of a tile share their cache lines.  `env` overrides the
// default for experiments; results do not depend on it.
static uint32_t simple_gather_below(const char* env, uint32_t dflt) {
  const char* v = ucsa_getenv(env);
  return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

template <bool FROM_RAYS, typename TT = float2, typename FT = float2>
static int32_t launch_encode(const ucsa_grid* grid, const void* table,
                             const float* a, const float* b, const float* z,
                             Aabb bb, uint32_t T, uint64_t M, void* feat,
                             void* stream) {
  const GridDev gd = ucsa_grid_dev(grid);
  const uint32_t nc = coarse_levels(grid);
  const dim3 blk(256);
  UCSA_CLEAR_ERR();
  if (nc > 0)
    hipLaunchKernelGGL((k_hashgrid_encode_coarse<FROM_RAYS, TT, FT>),
                       dim3(ucsa_div_up(M, 256)), blk, 0, (hipStream_t)stream,
                       gd, nc, (const TT*)table, a, b, z, bb, T, M,
                       (FT*)feat);
  if (nc < grid->n_levels)
    hipLaunchKernelGGL((k_hashgrid_encode<FROM_RAYS, TT, FT>),
                       dim3(ucsa_div_up(M, 256), grid->n_levels - nc), blk, 0,
                       (hipStream_t)stream, gd, nc,
                       simple_gather_below("UCSA_ENC_SIMPLE_RAYS", 0u),
                       (const TT*)table, a, b, z, bb, T, M, (FT*)feat);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_hashgrid_encode_rays(
    const ucsa_grid* grid, const float* table, const float* rays_o,
    const float* rays_d, const float* z, const float* aabb_host, uint32_t N,
    uint32_t T, float* feat, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(table, 1);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 2);
  UCSA_CHECK_ARG(aabb_host, 5);
  UCSA_CHECK_ARG(feat, 8);
  const uint64_t M = (uint64_t)N * T;
  if (M == 0) return 0;
  return launch_encode<true>(grid, table, rays_o, rays_d, z,
                             ucsa_aabb(aabb_host), T, M, feat, stream);
}

// Level order of the tiled kernel (see LevelMap).  Default: level-major,
// FINEST level first -- one table slice at a time (it stays in each XCD's L2),
// and the launch ends on the cheap coarse levels instead of draining the
// slowest one: 0.80 vs 0.84 ms for the coarse pass of a 61 440-ray chunk,
// fine pass unchanged.  Interleaving levels (k > 1) is slower in every
// pairing tried (1.0 - 1.6 ms / 2.1 - 2.4 ms: two 4 MiB slices thrash the L2;
// profiles/r03_encode_level_order.txt).  UCSA_ENC_ORDER (experiments only;
// results do not depend on it): "k:l0,l1,..." = k levels per grid row in the
// given order.
static LevelMap level_map(uint32_t L) {
  LevelMap lm;
  lm.k = 1;
  lm.simple_below = 0u;   // set per table type by the caller
  for (uint32_t i = 0; i < UCSA_MAX_LEVELS; ++i) lm.lv[i] = (uint8_t)(i < L ? L - 1 - i : 0);
  const char* e = ucsa_getenv("UCSA_ENC_ORDER");
  if (e && *e) {
    const uint32_t k = (uint32_t)strtoul(e, nullptr, 10);
    const char* c = strchr(e, ':');
    if (k >= 1 && L % k == 0 && c) {
      uint32_t seen = 0, n = 0;
      uint8_t lv[UCSA_MAX_LEVELS];
      ++c;
      while (*c && n < L) {
        char* end;
        const unsigned long v = strtoul(c, &end, 10);
        if (end == c || v >= L || (seen >> v & 1u)) break;
        lv[n++] = (uint8_t)v;
        seen |= 1u << v;
        c = (*end == ',') ? end + 1 : end;
      }
      if (n == L) {  // a full permutation: accept
        lm.k = k;
        for (uint32_t i = 0; i < L; ++i) lm.lv[i] = lv[i];
      }
    }
  }
  return lm;
}

// image_width > 0: rays are the pixels of full rows of an image that wide
template <typename TT = float2, typename FT = float2>
static int32_t launch_encode_image(const ucsa_grid* grid, const void* table,
                                   const float* rays_o, const float* rays_d,
                                   const float* z, Aabb bb, uint32_t N,
                                   uint32_t T, uint32_t image_width,
                                   void* feat, void* stream) {
  const GridDev gd = ucsa_grid_dev(grid);
  // every level goes through the tiled kernel: on the coarse ones a whole tile
  // sits in one or two cells (measured 0.84 ms vs 0.93 ms with levels 0-5 in
  // k_hashgrid_encode_coarse, 5.9 M samples)
  const uint32_t nc = 0;
  const uint64_t M = (uint64_t)N * T;
  UCSA_CHECK_ARG(M < 0x80000000ull && N < 0x40000000u &&
                     (uint64_t)N + 8ull * image_width < 0xFFFFFFFFull, 6);
  UCSA_CLEAR_ERR();
  if (nc > 0)
    hipLaunchKernelGGL((k_hashgrid_encode_coarse<true, TT, FT>), dim3(ucsa_div_up(M, 256)),
                       dim3(256), 0, (hipStream_t)stream, gd, nc,
                       (const TT*)table, rays_o, rays_d, z, bb, T, M,
                       (FT*)feat);
  if (nc < grid->n_levels) {
    const uint32_t rows = ucsa_div_up(N, image_width);
    const uint32_t tiles = ((image_width + 7u) / 8u) * ((rows + 7u) / 8u);
    const uint32_t s_blocks = ucsa_div_up(T, TILE_S);
    LevelMap lm = level_map(grid->n_levels);
    // fp32 table: plain gather on every level (measured faster, see
    // simple_gather_below); half2 table: its group-of-four form stays
    lm.simple_below = sizeof(TT) == 8
                          ? simple_gather_below("UCSA_ENC_SIMPLE", UCSA_MAX_LEVELS)
                          : simple_gather_below("UCSA_ENC_SIMPLE_H", 0u);
    // UCSA_ENC_LDS_PAD (experiments only, tools/coresident_exp.py): extra
    // dynamic LDS per workgroup = fewer encoder workgroups per CU, i.e. free
    // registers / wave slots for a co-resident kernel.  Results do not change.
    static const uint32_t lds_pad = []() {
      const char* v = ucsa_getenv("UCSA_ENC_LDS_PAD");
      return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : 0u;
    }();
    // levels [0, n_ml) through the several-levels-per-workgroup kernel (see
    // k_hashgrid_encode_tiled_ml); UCSA_ENC_ML overrides (experiments only:
    // the features do not depend on it), 0 = every level through the
    // per-level kernel as in rounds 3-4.  Only for the default level order.
    const char* ml_v = ucsa_getenv("UCSA_ENC_ML");
    const int ml_env = ml_v && *ml_v ? (int)strtol(ml_v, nullptr, 10) : -1;
    // measured on the bench's chunk (tools/encode_ml_sweep.py, ms by n_ml):
    //   fp32 table  coarse pass 0: 0.702  4: 0.620  8: 0.607  9: 0.603  10: 0.600  12: 0.603  16: 0.824
    //               fine pass   0: 1.283  4: 1.259  8: 1.248  9: 1.247  10: 1.254  12: 1.298  16: 1.686
    //   fp16 table  coarse pass 0: 0.588  6: 0.512  10: 0.479 | fine pass 0: 1.035  4: 1.018  6: 1.044  10: 1.123
    //   (its per-level kernel has the group-of-four gather on the hashed levels)
    uint32_t n_ml = ml_env >= 0 ? (uint32_t)ml_env : (sizeof(TT) == 8 ? 9u : 6u);
    if (n_ml > grid->n_levels) n_ml = grid->n_levels;
    if (lm.k != 1 || ucsa_getenv("UCSA_ENC_ORDER")) n_ml = 0;
    const uint32_t n_fine = grid->n_levels - n_ml;
    if (n_fine > 0)   // lm.lv = finest level first: its first n_fine rows
      hipLaunchKernelGGL((k_hashgrid_encode_tiled<TT, FT>),
                         dim3(tiles * s_blocks * lm.k, n_fine / lm.k),
                         dim3(256), lds_pad, (hipStream_t)stream, gd, lm,
                         (const TT*)table, rays_o, rays_d, z, bb, T, N,
                         image_width, s_blocks, (FT*)feat);
    if (n_ml > 0)
      hipLaunchKernelGGL((k_hashgrid_encode_tiled_ml<TT, FT>),
                         dim3(tiles * s_blocks), dim3(256), 0,
                         (hipStream_t)stream, gd, 0u, n_ml, (const TT*)table,
                         rays_o, rays_d, z, bb, T, N, image_width, s_blocks,
                         (FT*)feat);
  }
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_hashgrid_encode_rays_image(
    const ucsa_grid* grid, const float* table, const float* rays_o,
    const float* rays_d, const float* z, const float* aabb_host, uint32_t N,
    uint32_t T, uint32_t image_width, float* feat, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(table, 1);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 2);
  UCSA_CHECK_ARG(aabb_host, 5);
  UCSA_CHECK_ARG(image_width >= 1, 8);
  UCSA_CHECK_ARG(feat, 9);
  if ((uint64_t)N * T == 0) return 0;
  return launch_encode_image(grid, table, rays_o, rays_d, z,
                             ucsa_aabb(aabb_host), N, T, image_width, feat,
                             stream);
}

// tiny-cuda-nn's all-half encoding: fp16 table (4-byte half2 entries,
// `table_half` = the fp32 table rounded to half, ucsa_cast_f32_to_f16) and
// fp16 features [L][N*T] half2 = the fp32 kernels' features on the rounded
// table, rounded to half (interpolation in fp32).  image_width = 0:
// ray-ordered samples.
extern "C" int32_t ucsa_hashgrid_encode_rays_h16(
    const ucsa_grid* grid, const void* table_half, const float* rays_o,
    const float* rays_d, const float* z, const float* aabb_host, uint32_t N,
    uint32_t T, uint32_t image_width, void* feat, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(table_half && ((uintptr_t)table_half & 15u) == 0, 1);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 2);
  UCSA_CHECK_ARG(aabb_host, 5);
  UCSA_CHECK_ARG(feat, 9);
  // the group-of-four gather reads 16 aligned bytes: hashed levels must start
  // on a multiple of four entries (ucsa_grid_init rounds every level to x8)
  for (uint32_t l = 0; l < grid->n_levels; ++l)
    UCSA_CHECK_ARG(!grid->level[l].hashed || (grid->level[l].offset & 3u) == 0, 0);
  if ((uint64_t)N * T == 0) return 0;
  if (image_width)
    return launch_encode_image<ucsa_half2, ucsa_half2>(
        grid, table_half, rays_o, rays_d, z, ucsa_aabb(aabb_host), N, T,
        image_width, feat, stream);
  return launch_encode<true, ucsa_half2, ucsa_half2>(
      grid, table_half, rays_o, rays_d, z, ucsa_aabb(aabb_host), T,
      (uint64_t)N * T, feat, stream);
}

// fp32 table, fp16 features: what ucsa_sigma_mlp_fwd_f16 would round on load,
// rounded at the source (half the feature round trip through HBM)
extern "C" int32_t ucsa_hashgrid_encode_rays_hf(
    const ucsa_grid* grid, const float* table, const float* rays_o,
    const float* rays_d, const float* z, const float* aabb_host, uint32_t N,
    uint32_t T, uint32_t image_width, void* feat_half, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(table, 1);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 2);
  UCSA_CHECK_ARG(aabb_host, 5);
  UCSA_CHECK_ARG(feat_half, 9);
  if ((uint64_t)N * T == 0) return 0;
  if (image_width)
    return launch_encode_image<float2, ucsa_half2>(
        grid, table, rays_o, rays_d, z, ucsa_aabb(aabb_host), N, T, image_width,
        feat_half, stream);
  return launch_encode<true, float2, ucsa_half2>(
      grid, table, rays_o, rays_d, z, ucsa_aabb(aabb_host), T, (uint64_t)N * T,
      feat_half, stream);
}

__global__ void k_cast_f32_to_f16(const float* __restrict__ src,
                                  _Float16* __restrict__ dst, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (_Float16)src[i];
}

extern "C" int32_t ucsa_cast_f32_to_f16(const float* src, void* dst_half,
                                        uint64_t n, void* stream) {
  UCSA_CHECK_ARG(src, 0);
  UCSA_CHECK_ARG(dst_half, 1);
  if (n == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_cast_f32_to_f16, dim3((uint32_t)((n + 255) / 256)), dim3(256),
                     0, (hipStream_t)stream, src, (_Float16*)dst_half, n);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_hashgrid_encode_points(const ucsa_grid* grid,
                                               const float* table,
                                               const float* x, uint32_t M,
                                               float* feat, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(table, 1);
  UCSA_CHECK_ARG(x, 2);
  UCSA_CHECK_ARG(feat, 4);
  if (M == 0) return 0;
  Aabb bb = {};
  return launch_encode<false>(grid, table, x, nullptr, nullptr, bb, 1u,
                              (uint64_t)M, feat, stream);
}

// ---------------------------------------------------------------------------
// Host: level table.  float64 evaluation of the scale, snapped when integral
// (SURVEY 8a caveat: levels 5/10/15 are exactly 127/1023/8191 for bound 4).
// ---------------------------------------------------------------------------
#include <cmath>
extern "C" int32_t ucsa_grid_init(ucsa_grid* grid, float bound,
                                  uint32_t n_levels,
                                  uint32_t log2_hashmap_size,
                                  uint32_t base_resolution,
                                  double per_level_scale) {
  UCSA_CHECK_ARG(grid, 0);
  UCSA_CHECK_ARG(bound > 0.f, 1);
  UCSA_CHECK_ARG(n_levels > 0 && n_levels <= UCSA_MAX_LEVELS, 2);
  UCSA_CHECK_ARG(log2_hashmap_size > 0 && log2_hashmap_size < 32, 3);
  UCSA_CHECK_ARG(per_level_scale > 0.0, 5);
  grid->n_levels = n_levels;
  grid->n_features = 2;
  grid->bound = bound;
  const double log2s = std::log2(per_level_scale);
  const uint64_t cap = 1ull << log2_hashmap_size;
  uint64_t offset = 0;
  for (uint32_t l = 0; l < UCSA_MAX_LEVELS; ++l) {
    ucsa_grid_level& lv = grid->level[l];
    if (l >= n_levels) {
      lv = ucsa_grid_level{0.f, 0, 0, 0, 0};
      continue;
    }
    double s = std::pow(2.0, l * log2s) * base_resolution - 1.0;
    const double r = std::round(s);
    if (std::fabs(s - r) < 1e-9 * std::fmax(1.0, std::fabs(s))) s = r;
    lv.scale = (float)s;
    lv.res = (uint32_t)std::ceil((double)lv.scale) + 1;
    uint64_t e = (uint64_t)lv.res * lv.res * lv.res;
    e = (e + 7) / 8 * 8;
    lv.hashed = e > cap ? 1u : 0u;
    if (e > cap) e = cap;
    lv.entries = (uint32_t)e;
    lv.offset = (uint32_t)offset;
    offset += e;
  }
  grid->total_entries = (uint32_t)offset;
  return 0;
}
