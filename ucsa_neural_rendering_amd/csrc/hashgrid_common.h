// Per-level trilinear gather of the multiresolution hash grid (tiny-cuda-nn's
// GridEncoding, restated): shared by hashgrid.hip and encode_sigma.hip so that
// both produce the same bits.
#pragma once
#include "ucsa_common.h"

#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u

__device__ __forceinline__ uint32_t grid_index(uint32_t x, uint32_t y,
                                               uint32_t z, uint32_t res,
                                               uint32_t entries,
                                               uint32_t hashed) {
  uint32_t idx = hashed ? (x ^ (y * PRIME_Y) ^ (z * PRIME_Z))
                        : (x + y * res + z * res * res);
  // entries is a power of two on hashed levels
  return hashed ? (idx & (entries - 1)) : (idx % entries);
}

// Table entry types: float2 (fp32 table, the parity default) or half2 (fp16
// table, what tiny-cuda-nn stores; fp32 master copy lives with the optimizer).
// Values are widened to fp32 on load; the interpolation arithmetic is the same.
typedef _Float16 ucsa_half2 __attribute__((ext_vector_type(2)));
typedef _Float16 ucsa_half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float2 tab_load(const float2* __restrict__ tab, uint32_t i) {
  return tab[i];
}
__device__ __forceinline__ float2 tab_load(const ucsa_half2* __restrict__ tab, uint32_t i) {
  const ucsa_half2 v = tab[i];
  return make_float2((float)v[0], (float)v[1]);
}

// Trilinear gather of one level at x01 (already in [0,1]).
template <typename TT>
__device__ __forceinline__ float2 encode_level(const TT* __restrict__ tab,
                                               float x, float y, float z,
                                               float scale, uint32_t res,
                                               uint32_t entries,
                                               uint32_t hashed) {
  const float px = x * scale + 0.5f, py = y * scale + 0.5f,
              pz = z * scale + 0.5f;
  const float fx0 = floorf(px), fy0 = floorf(py), fz0 = floorf(pz);
  const float wx = px - fx0, wy = py - fy0, wz = pz - fz0;
  const uint32_t gx = (uint32_t)(int32_t)fx0, gy = (uint32_t)(int32_t)fy0,
                 gz = (uint32_t)(int32_t)fz0;
  float2 v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint32_t ix = gx + (c & 1), iy = gy + ((c >> 1) & 1),
                   iz = gz + ((c >> 2) & 1);
    v[c] = tab_load(tab, grid_index(ix, iy, iz, res, entries, hashed));
  }
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    // weight = ((1*wx')*wy')*wz' in dimension order, as the oracle does
    float w = (c & 1) ? wx : 1.0f - wx;
    w = w * ((c & 2) ? wy : 1.0f - wy);
    w = w * ((c & 4) ? wz : 1.0f - wz);
    acc.x = acc.x + w * v[c].x;
    acc.y = acc.y + w * v[c].y;
  }
  return acc;
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) {
  // torch.min(torch.max(v, lo), hi)
  return fminf(fmaxf(v, lo), hi);
}

// Hashed level with x-pair loads.  Random 8-byte gathers run at the TCP's
// divergent-access rate (~0.46 lane-accesses/clk/CU measured, independent of
// the access width and of the cache policy; tools/ubench/gather.hip), so the
// lever is FEWER lane-accesses: for even x0 the two x-corners (x0, x0+1) hash
// to idx and idx^1, i.e. one aligned 16-byte pair -> one access instead of
// two.  Odd x0 issues the second (predicated) access.  6 instead of 8
// accesses per sample and level on average; arithmetic order unchanged.
__device__ __forceinline__ float2 encode_level_hashed(
    const float2* __restrict__ tab, float x, float y, float z, float scale,
    uint32_t entries) {
  const float px = x * scale + 0.5f, py = y * scale + 0.5f,
              pz = z * scale + 0.5f;
  const float fx0 = floorf(px), fy0 = floorf(py), fz0 = floorf(pz);
  const float wx = px - fx0, wy = py - fy0, wz = pz - fz0;
  const uint32_t gx = (uint32_t)(int32_t)fx0, gy = (uint32_t)(int32_t)fy0,
                 gz = (uint32_t)(int32_t)fz0;
  const uint32_t mask = entries - 1;
  const bool odd = (gx & 1u) != 0;
  float2 v[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = ((gy + (q & 1)) * PRIME_Y) ^ ((gz + (q >> 1)) * PRIME_Z);
    const uint32_t i0 = (gx ^ h) & mask;
    const float4 pr = *reinterpret_cast<const float4*>(tab + (i0 & ~1u));
    const bool hi = (i0 & 1u) != 0;
    v[2 * q] = hi ? make_float2(pr.z, pr.w) : make_float2(pr.x, pr.y);
    float2 other = hi ? make_float2(pr.x, pr.y) : make_float2(pr.z, pr.w);
    if (odd) other = tab[((gx + 1u) ^ h) & mask];
    v[2 * q + 1] = other;
  }
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w = (c & 1) ? wx : 1.0f - wx;
    w = w * ((c & 2) ? wy : 1.0f - wy);
    w = w * ((c & 4) ? wz : 1.0f - wz);
    acc.x = acc.x + w * v[c].x;
    acc.y = acc.y + w * v[c].y;
  }
  return acc;
}


// The same for an fp16 table: 4-byte entries, so one aligned 16-byte access
// holds the entries of an aligned group of FOUR x (the hash is x ^ h: XOR
// permutes within the group).  (x0, x0+1) share it unless x0 = 3 (mod 4):
// 5 instead of 6 accesses per sample and level on average.
__device__ __forceinline__ float2 encode_level_hashed(
    const ucsa_half2* __restrict__ tab, float x, float y, float z, float scale,
    uint32_t entries) {
  const float px = x * scale + 0.5f, py = y * scale + 0.5f,
              pz = z * scale + 0.5f;
  const float fx0 = floorf(px), fy0 = floorf(py), fz0 = floorf(pz);
  const float wx = px - fx0, wy = py - fy0, wz = pz - fz0;
  const uint32_t gx = (uint32_t)(int32_t)fx0, gy = (uint32_t)(int32_t)fy0,
                 gz = (uint32_t)(int32_t)fz0;
  const uint32_t mask = entries - 1;
  const bool split = (gx & 3u) == 3u;  // x0 + 1 starts the next group
  float2 v[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = ((gy + (q & 1)) * PRIME_Y) ^ ((gz + (q >> 1)) * PRIME_Z);
    const uint32_t i0 = (gx ^ h) & mask, i1 = ((gx + 1u) ^ h) & mask;
    const ucsa_half8 grp = *reinterpret_cast<const ucsa_half8*>(tab + (i0 & ~3u));
    auto pick = [&](uint32_t k) {
      const _Float16 a = k & 2u ? (k & 1u ? grp[6] : grp[4]) : (k & 1u ? grp[2] : grp[0]);
      const _Float16 b = k & 2u ? (k & 1u ? grp[7] : grp[5]) : (k & 1u ? grp[3] : grp[1]);
      return make_float2((float)a, (float)b);
    };
    v[2 * q] = pick(i0 & 3u);
    float2 other = pick(i1 & 3u);
    if (split) other = tab_load(tab, i1);
    v[2 * q + 1] = other;
  }
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w = (c & 1) ? wx : 1.0f - wx;
    w = w * ((c & 2) ? wy : 1.0f - wy);
    w = w * ((c & 4) ? wz : 1.0f - wz);
    acc.x = acc.x + w * v[c].x;
    acc.y = acc.y + w * v[c].y;
  }
  return acc;
}
