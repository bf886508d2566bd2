// density() of a depth-ordered sample array with the COARSE levels' features never
// written to HBM (round 6; SURVEY 8a row a4, reference
// nr4seg/nerf/network_tcnn_semantics.py:130-144: encoder -> sigma net -> trunc_exp).
//
// After hashgrid_sorted.hip a density pass writes 16 levels x 8 B of features per
// sample and the sigma MLP reads them back: 755 MB out + 755 MB in per 5.9 M samples,
// the largest HBM stream of a view that is not the table itself.  Levels 0-7 are cheap
// to compute -- dense, or hashed with cells wider than a depth slab of the tile, so
// their gathers hit the L1 -- and here the sigma MLP computes them itself (and, by
// default, levels 8-11 as well: NENC = 12, see ucsa_density_sorted):
//
//   * a wave owns 64 consecutive samples of a tile's depth order, lane = sample;
//   * the LEVEL is wave-uniform: for l = 0 .. 7 every lane gathers its sample at level
//     l (hashgrid_common.h encode_cell: the level's scale / resolution / base in SGPRs,
//     64 independent gathers in flight per instruction) and drops the feature pair
//     into a wave-private LDS tile [level][sample];
//   * then the four 16-sample column blocks of the wave go through the MLP exactly as
//     in k_sigma_mlp_h2 / _x3: lane (g, j) takes levels g and 4 + g of sample j from
//     the LDS tile and levels 8 + g, 12 + g from HBM (written by the per-level
//     kernel k_hashgrid_encode_sorted: those are the levels bound by L2 -> L1 line
//     fills, one table slice at a time), requested before the encoding starts.
//
// The features are encode_cell's and the MLP is k_sigma_mlp_h2 / _x3's: the same
// h / sigma BITS as ucsa_hashgrid_encode_sorted + ucsa_sigma_mlp_fwd_scatter
// (tests/test_gpu_parity.py::test_depth_ordered_density_is_bit_identical).  Round
// 5's attempt (level per LANE: two dependent gathers in front of an MFMA chain, one
// 16-sample block per iteration) was slower than the pair it replaced and is gone.
#include <cstdlib>

#include "hashgrid_sorted.h"
#include "mfma_mlp_h2.h"

#define DS_MAX_LEVELS 16u     // levels [0, NENC) are encoded here, NENC = 8, 12 or 16
#ifndef DS_UNROLL
#define DS_UNROLL 2           // levels per trip of the encoding loop
#endif
#define DS_PITCH 80u          // float2 per LDS row: 64 samples + 16 (rows g, g + 1 of a
                              // half-wave then sit 32 banks apart: conflict-free reads)

// PREC 2: bf16x3 (ucsa_mlp_pack_x3), 3: f16x2 (ucsa_mlp_pack_h2)
// NENC: 8 (levels g, 4 + g of a lane group from the tile) or 12 (8 + g as well: only
// levels 12-15 -- the ones bound by line fills -- come from the per-level encoder)
template <int PREC, uint32_t NENC>
__global__ void __launch_bounds__(256, 2)
k_density_sorted(GridDev g, const float2* __restrict__ table,
                 const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                 const float* __restrict__ z_sorted, const uint8_t* __restrict__ pix,
                 Aabb bb, uint32_t T, uint32_t rows, uint32_t W, uint32_t s_blocks,
                 uint32_t M, const float2* __restrict__ feat,   // levels NENC..15 valid
                 const void* __restrict__ packed, const uint32_t* __restrict__ slot,
                 float* __restrict__ h, float* __restrict__ sigma) {
  __shared__ __attribute__((aligned(16))) float ray_s[64][8];
  __shared__ __attribute__((aligned(16))) float2 ftile[4][NENC][DS_PITCH];
  const uint32_t bid = xcd_band(blockIdx.x, gridDim.x);
  const uint32_t sb = bid % s_blocks, tile = bid / s_blocks;
  const TileGeom tg = tile_geom(tile, rows, W, T);
  if (sb * 1024u >= tg.count) return;   // (workgroup-uniform)
  load_tile_rays(ray_s, tg, W, rays_o, rays_d);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t gq = lane >> 4, j = lane & 15u;
  const float two_b = 2.0f * g.bound, inv = unit_inv(two_b);

  H2W w1h[4], w2h[2];
  W3 w1x[4], w2x[2];
  if constexpr (PREC == 3) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) w1h[rb] = h2_frag(packed, rb, lane);
#pragma unroll
    for (int s = 0; s < 2; ++s) w2h[s] = h2_frag(packed, 4 + s, lane);
  } else {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) w1x[rb] = frag_x3(packed, rb, lane);
#pragma unroll
    for (int s = 0; s < 2; ++s) w2x[s] = frag_x3(packed, 4 + s, lane);
  }
  const H2Sel hsel = h2_selectors();
  const X3Sel xsel = x3_selectors();

  const float2* feat_hi0 = feat + (size_t)(8u + gq) * M + tg.base;
  const float2* feat_hi1 = feat + (size_t)(12u + gq) * M + tg.base;
  float2(*mine)[DS_PITCH] = ftile[wid];
  // the workgroup's 16 groups of 64 ranks, interleaved over its four waves (the
  // waves work on neighbouring depth slabs at the same time)
#pragma unroll 1
  for (uint32_t it = 0; it < 4u; ++it) {
    const uint32_t r0 = sb * 1024u + (it * 4u + wid) * 64u;
    if (r0 >= tg.count) break;            // (wave-uniform; later groups lie further out)
    const uint32_t last = tg.count - 1u;
    // levels 8 + g (NENC = 8 only), 12 + g of the four column blocks: requested first
    float2 hi0[4], hi1[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const uint32_t r = r0 + cb * 16u + j;
      const uint32_t rc = r < last ? r : last;    // clamp loads, predicate stores
      if constexpr (NENC <= 8u) hi0[cb] = feat_hi0[rc];
      if constexpr (NENC <= 12u) hi1[cb] = feat_hi1[rc];
    }
    {  // levels 0 .. 7 of sample r0 + lane -> the wave's LDS tile
      const uint32_t r = r0 + lane;
      const uint32_t rc = r < last ? r : last;
      float ux, uy, uz;
      unit_position(ray_s, pix[tg.base + rc], z_sorted[tg.base + rc], bb, g.bound, two_b,
                    inv, ux, uy, uz);
      // (two levels per trip: 16 gathers in flight per lane; fully unrolled the eight
      // levels' scalars spill out of the SGPR file)
#pragma unroll DS_UNROLL
      for (uint32_t level = 0; level < NENC; ++level) {
        const uint32_t res = g.res[level];
        mine[level][lane] = encode_cell(table + g.offset[level], ux, uy, uz, g.scale[level],
                                        res, res * res, g.entries[level], g.hashed[level]);
      }
    }
    // (one wave: its LDS instructions execute in order; the barrier only keeps the
    // compiler from moving the reads above the writes)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float2 raw[4];
      raw[0] = mine[gq][cb * 16u + j];
      raw[1] = mine[4u + gq][cb * 16u + j];
      if constexpr (NENC > 8u) raw[2] = mine[8u + gq][cb * 16u + j];
      else raw[2] = hi0[cb];
      if constexpr (NENC > 12u) raw[3] = mine[12u + gq][cb * 16u + j];
      else raw[3] = hi1[cb];
      f32x4 out;
      if constexpr (PREC == 3) {
        H2X xin;
#pragma unroll
        for (int q = 0; q < 4; ++q) h2_split_pair(raw[q].x, raw[q].y, xin, q, hsel);
        f32x4 a1[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a1[rb] = h2_mul1(w1h[rb], xin);
        out = h2_mul2(w2h[0], h2_chain_relu(a1[0], a1[1], hsel), w2h[1],
                      h2_chain_relu(a1[2], a1[3], hsel));
      } else {
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        X3 xin;
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair(raw[q].x, raw[q].y, xin, q, xsel);
        f32x4 a1[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a1[rb] = mfma_x3(w1x[rb], xin, z4);
        out = mfma_x3(w2x[0], chain_relu_x3(a1[0], a1[1], xsel), z4);
        out = mfma_x3(w2x[1], chain_relu_x3(a1[2], a1[3], xsel), out);
      }
      const uint32_t r = r0 + cb * 16u + j;
      if (r < tg.count) {
        const uint64_t mo = slot[tg.base + r];
        *reinterpret_cast<f32x4*>(h + mo * 16 + 4 * gq) = out;
        if (gq == 0) sigma[mo] = expf(out[0]);
      }
    }
    __builtin_amdgcn_wave_barrier();   // the next group's writes stay below these reads
  }
}

// levels [first_level, L) through k_hashgrid_encode_sorted (hashgrid_sorted.hip)
int32_t ucsa_hashgrid_encode_sorted_from(const ucsa_grid* grid, const float* table,
                                         const float* rays_o, const float* rays_d,
                                         const float* z_sorted, const uint8_t* pix,
                                         const float* aabb_host, uint32_t N, uint32_t T,
                                         uint32_t image_width, uint32_t first_level,
                                         float* feat, void* stream);

#ifndef UCSA_DENSITY_LEVELS_DEFAULT
#define UCSA_DENSITY_LEVELS_DEFAULT 12
#endif

extern "C" int32_t ucsa_density_sorted(
    int32_t mode, const ucsa_grid* grid, const float* table, const float* rays_o,
    const float* rays_d, const float* z_sorted, const uint8_t* pix,
    const uint32_t* slot, const float* aabb_host, uint32_t N, uint32_t T,
    uint32_t image_width, const void* packed_sigma, float* feat_ws, float* h,
    float* sigma, void* stream) {
  UCSA_CHECK_ARG(mode == 2 || mode == 3, 0);
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels == 16, 1);
  UCSA_CHECK_ARG(table, 2);
  UCSA_CHECK_ARG(rays_o && rays_d, 3);
  UCSA_CHECK_ARG(z_sorted && pix && slot, 5);
  UCSA_CHECK_ARG(aabb_host, 8);
  UCSA_CHECK_ARG(T >= 1 && T <= 1024 && (uint64_t)N * T < 0x80000000ull, 10);
  UCSA_CHECK_ARG(image_width >= 1 && N % image_width == 0, 11);
  UCSA_CHECK_ARG(packed_sigma, 12);
  UCSA_CHECK_ARG(feat_ws && h && sigma, 13);
  if (N == 0) return 0;
  // UCSA_DENSITY_LEVELS (lab switch): 8 or 12 (default) levels inside the sigma MLP --
  // measured on the cfg2 view: 15.68 ms with 8, 15.07 ms with 12 (20.4 M rays/s): with
  // levels 8-11 inside as well only the four levels bound by line fills keep their own
  // launch, and three quarters of the feature round trip are gone
  const char* lv = ucsa_getenv("UCSA_DENSITY_LEVELS");
  const int lvn = lv ? atoi(lv) : UCSA_DENSITY_LEVELS_DEFAULT;
  const uint32_t nenc = lvn >= 16 ? 16u : (lvn >= 12 ? 12u : 8u);
  if (nenc < 16u) {
    const int32_t rc = ucsa_hashgrid_encode_sorted_from(grid, table, rays_o, rays_d, z_sorted,
                                                        pix, aabb_host, N, T, image_width,
                                                        nenc, feat_ws, stream);
    if (rc != 0) return rc;
  }
  const GridDev gd = ucsa_grid_dev(grid);
  const uint32_t rows = N / image_width;
  const uint32_t tiles = ((image_width + 7u) / 8u) * ((rows + 7u) / 8u);
  const uint32_t s_blocks = ucsa_div_up(64u * T, 1024u);
  UCSA_CLEAR_ERR();
#define DS_GO(P, NE)                                                                        \
  hipLaunchKernelGGL((k_density_sorted<P, NE>), dim3(tiles * s_blocks), dim3(256), 0,          \
                     (hipStream_t)stream, gd, (const float2*)table, rays_o, rays_d, z_sorted, \
                     pix, ucsa_aabb(aabb_host), T, rows, image_width, s_blocks, N * T,        \
                     (const float2*)feat_ws, packed_sigma, slot, h, sigma)
  if (mode == 3) {
    if (nenc == 16u) DS_GO(3, 16u); else if (nenc == 12u) DS_GO(3, 12u); else DS_GO(3, 8u);
  } else {
    if (nenc == 16u) DS_GO(2, 16u); else if (nenc == 12u) DS_GO(2, 12u); else DS_GO(2, 8u);
  }
#undef DS_GO
  return ucsa_launch_status();
}
