// Depth-ordered density of the fine samples with the COARSE levels' features
// never written to HBM (round 5; SURVEY 8a row a4, reference
// nr4seg/nerf/network_tcnn_semantics.py:130-144 `density()`: encoder -> sigma net
// -> trunc_exp).
//
// After hashgrid_sorted.hip the fine pass spends 0.23 ms writing the features of
// levels 0-8 and the sigma MLP 0.28 ms, most of it reading all 16 levels back
// (755 MB out + 755 MB in per 5.9 M samples).  Levels 0-7 are cheap to compute
// (dense, or hashed with cells wider than a depth slab of the tile: their
// gathers hit the L1) -- so the sigma MLP computes them itself: in its operand
// layout lane (g, j) feeds sample j's levels g, 4 + g, 8 + g, 12 + g; here it
// ENCODES levels g and 4 + g (two encode_cell per lane and sample, no redundant
// work across the four lanes of a sample beyond the position) and reads levels
// 8 + g and 12 + g, which k_hashgrid_encode_sorted wrote (the levels bound by
// L2 -> L1 line fills, one table slice at a time).  Half of the feature round
// trip through HBM is gone.  The features are encode_cell's, the MLP is
// k_sigma_mlp_h2 / _x3's: the same h / sigma bits as the staged pair.
//
// STATUS (round 5): bit-identical, but SLOWER than the two calls it replaces --
// 0.58 ms against 0.23 (levels 0-8) + 0.27 (sigma MLP with scatter) on the
// bench's fine pass.  With the level per LANE the two gathers of a lane are a
// dependent chain in front of an MFMA chain, one 16-sample block per iteration at
// 128 VGPRs (four blocks unrolled: 256 + 60 AGPRs, one wave per SIMD); the staged
// kernels keep 4 samples' gathers resp. 4 blocks' MFMAs in flight.  Kept behind
// UCSA_ENC_FUSED_ML=1 (off) with its parity test; the form that could win --
// encode with the level uniform per wave into a wave-private LDS tile, then the
// MLP on four blocks -- is not built: the ceiling is ~0.1 ms of a 3.3 ms chunk.
#include <cstdlib>

#include "hashgrid_sorted.h"
#include "mfma_mlp_h2.h"


namespace {
struct LevelRegs {
  float scale;
  uint32_t res, res2, entries, hashed, boff;   // boff: byte offset of the level
};
}  // namespace

// PREC 2: bf16x3 (ucsa_mlp_pack_x3), 3: f16x2 (ucsa_mlp_pack_h2)
template <int PREC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
k_encode_sigma_sorted(GridDev g, const float2* __restrict__ table,
                      const float* __restrict__ rays_o,
                      const float* __restrict__ rays_d,
                      const float* __restrict__ z_sorted,
                      const uint8_t* __restrict__ pix, Aabb bb, uint32_t T,
                      uint32_t rows, uint32_t W, uint32_t s_blocks, uint32_t M,
                      const float2* __restrict__ feat,   // levels 8..15 valid
                      const void* __restrict__ packed,
                      const uint32_t* __restrict__ slot, float* __restrict__ h,
                      float* __restrict__ sigma) {
  __shared__ __attribute__((aligned(16))) float ray_s[64][8];
  __shared__ LevelRegs lev_s[8];
  const uint32_t sb = blockIdx.x % s_blocks, tile = blockIdx.x / s_blocks;
  const TileGeom tg = tile_geom(tile, rows, W, T);
  if (sb * 1024u >= tg.count) return;   // (workgroup-uniform)
  load_tile_rays(ray_s, tg, W, rays_o, rays_d);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int l = 0; l < 8; ++l)
      lev_s[l] = LevelRegs{g.scale[l], g.res[l], g.res[l] * g.res[l], g.entries[l],
                           g.hashed[l], g.offset[l] * (uint32_t)sizeof(float2)};
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t gq = lane >> 4, j = lane & 15u;
  const LevelRegs la = lev_s[gq], lb = lev_s[4 + gq];
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
  // the wave's 256 ranks of the workgroup's 1024, one 16-sample column block at
  // a time (rolled: the gathers of one block -- 2 levels x 8 corners per lane --
  // and the weight fragments are what the register file holds; four blocks
  // unrolled took 256 VGPRs + 60 AGPRs, one wave per SIMD)
#pragma unroll 1
  for (uint32_t cb = 0; cb < 16u; ++cb) {
    const uint32_t r0 = sb * 1024u + wid * 256u + cb * 16u;
    if (r0 >= tg.count) break;          // (wave-uniform)
    const uint32_t rank = r0 + j;
    const uint32_t r = rank < tg.count ? rank : tg.count - 1u;   // clamp loads, predicate stores
    float2 raw[4];
    raw[2] = feat_hi0[r];
    raw[3] = feat_hi1[r];
    float ux, uy, uz;
    unit_position(ray_s, pix[tg.base + r], z_sorted[tg.base + r], bb, g.bound, two_b,
                  inv, ux, uy, uz);
    raw[0] = encode_cell(table, ux, uy, uz, la.scale, la.res, la.res2, la.entries,
                         la.hashed, la.boff);
    raw[1] = encode_cell(table, ux, uy, uz, lb.scale, lb.res, lb.res2, lb.entries,
                         lb.hashed, lb.boff);
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
    if (rank < tg.count) {
      const uint64_t mo = slot[tg.base + rank];
      *reinterpret_cast<f32x4*>(h + mo * 16 + 4 * gq) = out;
      if (gq == 0) sigma[mo] = expf(out[0]);
    }
  }
}

// levels [8, 16) through k_hashgrid_encode_sorted (hashgrid_sorted.hip)
int32_t ucsa_hashgrid_encode_sorted_from(const ucsa_grid* grid, const float* table,
                                         const float* rays_o, const float* rays_d,
                                         const float* z_sorted, const uint8_t* pix,
                                         const float* aabb_host, uint32_t N, uint32_t T,
                                         uint32_t image_width, uint32_t first_level,
                                         float* feat, void* stream);

extern "C" int32_t ucsa_encode_sigma_sorted(
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
  const int32_t rc = ucsa_hashgrid_encode_sorted_from(grid, table, rays_o, rays_d, z_sorted,
                                                      pix, aabb_host, N, T, image_width, 8u,
                                                      feat_ws, stream);
  if (rc != 0) return rc;
  const GridDev gd = ucsa_grid_dev(grid);
  const uint32_t rows = N / image_width;
  const uint32_t tiles = ((image_width + 7u) / 8u) * ((rows + 7u) / 8u);
  const uint32_t s_blocks = ucsa_div_up(64u * T, 1024u);
  UCSA_CLEAR_ERR();
  if (mode == 3)
    hipLaunchKernelGGL(k_encode_sigma_sorted<3>, dim3(tiles * s_blocks), dim3(256), 0,
                       (hipStream_t)stream, gd, (const float2*)table, rays_o, rays_d, z_sorted,
                       pix, ucsa_aabb(aabb_host), T, rows, image_width, s_blocks, N * T,
                       (const float2*)feat_ws, packed_sigma, slot, h, sigma);
  else
    hipLaunchKernelGGL(k_encode_sigma_sorted<2>, dim3(tiles * s_blocks), dim3(256), 0,
                       (hipStream_t)stream, gd, (const float2*)table, rays_o, rays_d, z_sorted,
                       pix, ucsa_aabb(aabb_host), T, rows, image_width, s_blocks, N * T,
                       (const float2*)feat_ws, packed_sigma, slot, h, sigma);
  return ucsa_launch_status();
}
