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

// entries i and i + 1 with one access (4-byte alignment is all the hardware
// asks of a global load)
typedef float ucsa_f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 ucsa_half4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void tab_load_pair(const float2* __restrict__ tab, uint32_t i,
                                              float2& a, float2& b) {
  const ucsa_f32x4_u p = *reinterpret_cast<const ucsa_f32x4_u*>(tab + i);
  a = make_float2(p[0], p[1]);
  b = make_float2(p[2], p[3]);
}
__device__ __forceinline__ void tab_load_pair(const ucsa_half2* __restrict__ tab,
                                              uint32_t i, float2& a, float2& b) {
  const ucsa_half4_u p = *reinterpret_cast<const ucsa_half4_u*>(tab + i);
  a = make_float2((float)p[0], (float)p[1]);
  b = make_float2((float)p[2], (float)p[3]);
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
  if (!hashed) {
    // dense level: idx(x0 + 1) = idx(x0) + 1, so the x-pair is one (possibly
    // unaligned) double-width access -- 4 instead of 8 per sample
    // `% entries` only bites on the far faces of the box (a corner index of
    // res - 1 + 1): the general unsigned modulo is a ~25-instruction sequence,
    // four of them per sample were the bulk of a dense level's VALU work, so it
    // sits behind a (practically never taken) branch.  Same indices.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t lin = gx + (gy + (q & 1)) * res + (gz + (q >> 1)) * res * res;
      uint32_t i0 = lin;
      if (__builtin_expect(i0 >= entries, 0)) i0 = lin % entries;
      if (__builtin_expect(i0 + 1u < entries, 1)) {
        tab_load_pair(tab, i0, v[2 * q], v[2 * q + 1]);
      } else {  // the pair wraps around the end of the level's slab:
        v[2 * q] = tab_load(tab, i0);          // i0 == entries - 1, so
        v[2 * q + 1] = tab_load(tab, 0u);      // (lin + 1) % entries == 0
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t ix = gx + (c & 1), iy = gy + ((c >> 1) & 1),
                     iz = gz + ((c >> 2) & 1);
      v[c] = tab_load(tab, grid_index(ix, iy, iz, res, entries, hashed));
    }
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
  // torch.min(torch.max(v, lo), hi) for lo <= hi and a finite v, as ONE
  // instruction (v_med3_f32); fminf(fmaxf()) is four (each first quiets a
  // possible signalling NaN)
  return __builtin_amdgcn_fmed3f(v, lo, hi);
}

// x01 = (p + bound) / (2 * bound) as the oracle computes it.  For a
// power-of-two 2 * bound (bound = 4 here) the division equals the
// multiplication by its reciprocal bit for bit, and an IEEE division is an
// 11-instruction sequence, three of them per sample and LEVEL (the level-major
// launch recomputes the position per level): `inv` is 1 / (2 * bound) then,
// 0 otherwise (wave-uniform).
__device__ __forceinline__ float unit_inv(float two_b) {
  return (__float_as_uint(two_b) & 0x007FFFFFu) == 0u ? 1.0f / two_b : 0.0f;
}
__device__ __forceinline__ float to_unit(float p, float bound, float two_b, float inv) {
  return inv != 0.0f ? (p + bound) * inv : (p + bound) / two_b;
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
  // all accesses of the sample are requested before the first one is used
  // (separate destination registers: no wait between them)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  uint32_t i0[4], i1[4];
  f32x4_t pr[4];
  f32x2_t ex[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = ((gy + (q & 1)) * PRIME_Y) ^ ((gz + (q >> 1)) * PRIME_Z);
    i0[q] = (gx ^ h) & mask;
    i1[q] = ((gx + 1u) ^ h) & mask;
    pr[q] = *reinterpret_cast<const f32x4_t*>(tab + (i0[q] & ~1u));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ex[q] = f32x2_t{0.f, 0.f};
    if (odd) ex[q] = *reinterpret_cast<const f32x2_t*>(tab + i1[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool hi = (i0[q] & 1u) != 0;
    v[2 * q] = hi ? make_float2(pr[q][2], pr[q][3]) : make_float2(pr[q][0], pr[q][1]);
    const float mx = hi ? pr[q][0] : pr[q][2], my = hi ? pr[q][1] : pr[q][3];
    v[2 * q + 1] = odd ? make_float2(ex[q][0], ex[q][1]) : make_float2(mx, my);
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
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  float2 v[8];
  uint32_t i0[4], i1[4], ex[4];
  u32x4_t grp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = ((gy + (q & 1)) * PRIME_Y) ^ ((gz + (q >> 1)) * PRIME_Z);
    i0[q] = (gx ^ h) & mask;
    i1[q] = ((gx + 1u) ^ h) & mask;
    grp[q] = *reinterpret_cast<const u32x4_t*>(tab + (i0[q] & ~3u));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ex[q] = 0u;
    if (split) ex[q] = *reinterpret_cast<const uint32_t*>(tab + i1[q]);
  }
  auto sel4 = [](const u32x4_t& g4, uint32_t k) {  // entry k of the group: 3 selects
    const uint32_t lo = (k & 1u) ? g4[1] : g4[0];
    const uint32_t hi = (k & 1u) ? g4[3] : g4[2];
    return (k & 2u) ? hi : lo;
  };
  auto widen = [](uint32_t u) {
    const ucsa_half2 h2 = __builtin_bit_cast(ucsa_half2, u);
    return make_float2((float)h2[0], (float)h2[1]);
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = widen(sel4(grp[q], i0[q] & 3u));
    const uint32_t mate = sel4(grp[q], i1[q] & 3u);
    v[2 * q + 1] = widen(split ? ex[q] : mate);
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

// ---------------------------------------------------------------------------
// Feature stores (shared by hashgrid.hip and hashgrid_sorted.hip)
// ---------------------------------------------------------------------------
// the features are written once and read once by the next kernel: streamed
// past the L2 (nontemporal) so that the level's table slab stays resident
// (measured: encode passes -2.5 ... -3.5 %, a 640x480 view -2.5 %)
#ifndef UCSA_NT_FEAT
#define UCSA_NT_FEAT 1
#endif
__device__ __forceinline__ void feat_store(float2* dst, float2 v) {
#if UCSA_NT_FEAT
  typedef float f32x2_nt __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(f32x2_nt{v.x, v.y}, reinterpret_cast<f32x2_nt*>(dst));
#else
  *dst = v;
#endif
}

// fp16 features (tiny-cuda-nn's all-half encoding): rounded where they are
// produced instead of where the f16 sigma MLP consumes them -- the same values
__device__ __forceinline__ void feat_store(ucsa_half2* dst, float2 v) {
  const ucsa_half2 h = {(_Float16)v.x, (_Float16)v.y};
  __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, h),
                              reinterpret_cast<uint32_t*>(dst));
}
__device__ __forceinline__ void feat_store(ucsa_half2* dst, ucsa_half2 h) {
  __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, h),
                              reinterpret_cast<uint32_t*>(dst));
}
__device__ __forceinline__ void to_feat(float2& d, float2 v) { d = v; }
__device__ __forceinline__ void to_feat(ucsa_half2& d, float2 v) {
  d = ucsa_half2{(_Float16)v.x, (_Float16)v.y};
}


// ---------------------------------------------------------------------------
// Round 5: the lean per-sample gather of the several-levels-per-workgroup and
// depth-ordered kernels (same operations in the same order as encode_level)
// ---------------------------------------------------------------------------
// table entries addressed as 32-bit BYTE offsets off the level's (wave-uniform)
// base: one `global_load ... v_off, s[base]` per corner instead of a 64-bit
// shift-and-add per address
__device__ __forceinline__ float2 tab_at(const float2* __restrict__ tab, uint32_t off) {
  return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(tab) + off);
}
__device__ __forceinline__ float2 tab_at(const ucsa_half2* __restrict__ tab, uint32_t off) {
  const ucsa_half2 v = *reinterpret_cast<const ucsa_half2*>(reinterpret_cast<const char*>(tab) + off);
  return make_float2((float)v[0], (float)v[1]);
}
__device__ __forceinline__ void tab_pair_at(const float2* __restrict__ tab, uint32_t off,
                                            float2& a, float2& b) {
  const ucsa_f32x4_u p = *reinterpret_cast<const ucsa_f32x4_u*>(reinterpret_cast<const char*>(tab) + off);
  a = make_float2(p[0], p[1]);
  b = make_float2(p[2], p[3]);
}
__device__ __forceinline__ void tab_pair_at(const ucsa_half2* __restrict__ tab, uint32_t off,
                                            float2& a, float2& b) {
  const ucsa_half4_u p = *reinterpret_cast<const ucsa_half4_u*>(reinterpret_cast<const char*>(tab) + off);
  a = make_float2((float)p[0], (float)p[1]);
  b = make_float2((float)p[2], (float)p[3]);
}

// the far faces of a dense level (tcnn's `% entries` wraps there): rare, kept
// out of line and rolled so that it costs no instruction-cache space
template <typename TT>
__device__ __noinline__ void dense_corners_wrapped(const TT* __restrict__ tab,
                                                   uint32_t gx, uint32_t gy, uint32_t gz,
                                                   uint32_t res, uint32_t entries,
                                                   float2* v) {
#pragma unroll 1
  for (int c = 0; c < 8; ++c)
    v[c] = tab_load(tab, grid_index(gx + (c & 1), gy + ((c >> 1) & 1),
                                    gz + ((c >> 2) & 1), res, entries, 0u));
}

// base_off: byte offset of the level inside `tab` when the level is not
// wave-uniform (encode_sigma_sorted.hip: one scalar base for the whole table,
// the level's offset per lane); 0 with `tab` already at the level.
template <typename TT>
__device__ __forceinline__ float2 encode_cell(const TT* __restrict__ tab,
                                              float x, float y, float z,
                                              float scale, uint32_t res,
                                              uint32_t res2, uint32_t entries,
                                              uint32_t hashed,
                                              uint32_t base_off = 0u) {
  constexpr uint32_t SH = sizeof(TT) == 8 ? 3u : 2u;   // log2 of the entry size
  const float px = x * scale + 0.5f, py = y * scale + 0.5f,
              pz = z * scale + 0.5f;
  const float fx0 = floorf(px), fy0 = floorf(py), fz0 = floorf(pz);
  const float wx = px - fx0, wy = py - fy0, wz = pz - fz0;
  const uint32_t gx = (uint32_t)(int32_t)fx0, gy = (uint32_t)(int32_t)fy0,
                 gz = (uint32_t)(int32_t)fz0;
  float2 v[8];
  if (!hashed) {   // wave-uniform
    const uint32_t b = gx + gy * res + gz * res2;
    // all four x-pairs inside the slab (false only on the far faces of the
    // box): one test instead of eight
    if (__builtin_expect(b + res + res2 + 1u < entries, 1)) {
      const uint32_t o = (b << SH) + base_off;
      tab_pair_at(tab, o, v[0], v[1]);
      tab_pair_at(tab, o + (res << SH), v[2], v[3]);
      tab_pair_at(tab, o + (res2 << SH), v[4], v[5]);
      tab_pair_at(tab, o + ((res + res2) << SH), v[6], v[7]);
    } else {
      dense_corners_wrapped(tab + (base_off >> SH), gx, gy, gz, res, entries, v);
    }
  } else {
    // byte offset of entry (ix ^ iy P_y ^ iz P_z) & (entries - 1): the shift
    // by SH commutes with the xor / and, and a product mod 2^32 shifted left
    // keeps the bits the mask reads -- the same entries as grid_index()
    const uint32_t mask = (entries - 1u) << SH;
    constexpr uint32_t PY = PRIME_Y << SH, PZ = PRIME_Z << SH;
    const uint32_t hy0 = gy * PY, hy1 = hy0 + PY;   // (gy + 1) P = gy P + P mod 2^32
    const uint32_t hz0 = gz * PZ, hz1 = hz0 + PZ;
    const uint32_t x0 = gx << SH, x1 = x0 + (1u << SH);
    const uint32_t h00 = hy0 ^ hz0, h10 = hy1 ^ hz0, h01 = hy0 ^ hz1, h11 = hy1 ^ hz1;
    v[0] = tab_at(tab, ((x0 ^ h00) & mask) + base_off);
    v[1] = tab_at(tab, ((x1 ^ h00) & mask) + base_off);
    v[2] = tab_at(tab, ((x0 ^ h10) & mask) + base_off);
    v[3] = tab_at(tab, ((x1 ^ h10) & mask) + base_off);
    v[4] = tab_at(tab, ((x0 ^ h01) & mask) + base_off);
    v[5] = tab_at(tab, ((x1 ^ h01) & mask) + base_off);
    v[6] = tab_at(tab, ((x0 ^ h11) & mask) + base_off);
    v[7] = tab_at(tab, ((x1 ^ h11) & mask) + base_off);
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

