// Tile geometry and per-sample position of the depth-ordered kernels
// (hashgrid_sorted.hip, encode_sigma_sorted.hip).
#pragma once
#include "hashgrid_common.h"

namespace {
struct TileGeom {
  uint32_t base;    // first sorted position of the tile (samples of earlier tiles)
  uint32_t count;   // samples of the tile
  uint32_t wt, ht;  // valid pixels of the tile in x / y
  uint32_t px0, py0;
};
}  // namespace

// XCD-aware work distribution (round 6).  The dispatcher hands consecutive workgroup
// ids to the 8 XCDs in turn, and every XCD has its own 4 MiB L2: with blockIdx ->
// (tile, sample block) taken literally, each XCD sees every 8th workgroup of the WHOLE
// chunk and all eight L2s end up caching the same table entries.  Here the blocks of
// one residue class (= one XCD, within a grid row) get a CONTIGUOUS band of the
// logical index space -- neighbouring tiles, the same part of the scene -- so that
// each L2 holds an eighth of the touched table instead of a copy of all of it.  A
// bijection of [0, n): results do not depend on it.
#ifndef UCSA_XCD_BAND
#define UCSA_XCD_BAND 1
#endif
__device__ __forceinline__ uint32_t xcd_band(uint32_t bid, uint32_t n) {
#if UCSA_XCD_BAND
  const uint32_t per = n >> 3, rem = n & 7u;
  const uint32_t r = bid & 7u, k = bid >> 3;
  return r * per + (r < rem ? r : rem) + k;
#else
  return bid;
#endif
}

// rays = the pixels of `rows` full image rows, W wide; tiles row-major
__device__ __forceinline__ TileGeom tile_geom(uint32_t tile, uint32_t rows,
                                              uint32_t W, uint32_t T) {
  const uint32_t tiles_x = (W + 7u) / 8u;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  TileGeom t;
  t.px0 = tx * 8u;
  t.py0 = ty * 8u;
  t.wt = W - t.px0 < 8u ? W - t.px0 : 8u;
  t.ht = rows - t.py0 < 8u ? rows - t.py0 : 8u;
  // every earlier tile of this band is 8 wide, every earlier band W x 8 pixels
  t.base = T * (t.py0 * W + t.px0 * t.ht);
  t.count = t.wt * t.ht * T;
  return t;
}


// the tile's rays in LDS: lane -> (origin, direction) by pixel id
__device__ __forceinline__ void load_tile_rays(float (*ray_s)[8], const TileGeom& tg,
                                               uint32_t W,
                                               const float* __restrict__ rays_o,
                                               const float* __restrict__ rays_d) {
  for (uint32_t e = threadIdx.x; e < 64u * 6u; e += blockDim.x) {
    const uint32_t p = e / 6u, c = e % 6u;
    const uint32_t lx = p & 7u, ly = p >> 3;
    float v = 0.f;
    if (lx < tg.wt && ly < tg.ht) {
      const uint32_t r = (tg.py0 + ly) * W + tg.px0 + lx;
      v = c < 3u ? rays_o[r * 3u + c] : rays_d[r * 3u + c - 3u];
    }
    ray_s[p][c < 3u ? c : c + 1u] = v;   // o in [0..2], d in [4..6]
  }
}

__device__ __forceinline__ void unit_position(const float (*ray_s)[8], uint32_t p,
                                              float zz, const Aabb& bb, float bound,
                                              float two_b, float inv, float& ux,
                                              float& uy, float& uz) {
  const float4 o = *reinterpret_cast<const float4*>(&ray_s[p][0]);
  const float4 d = *reinterpret_cast<const float4*>(&ray_s[p][4]);
  const float px = clampf(o.x + d.x * zz, bb.lo[0], bb.hi[0]);
  const float py = clampf(o.y + d.y * zz, bb.lo[1], bb.hi[1]);
  const float pz = clampf(o.z + d.z * zz, bb.lo[2], bb.hi[2]);
  ux = to_unit(px, bound, two_b, inv);
  uy = to_unit(py, bound, two_b, inv);
  uz = to_unit(pz, bound, two_b, inv);
}

