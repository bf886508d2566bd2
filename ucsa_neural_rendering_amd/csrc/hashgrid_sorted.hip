// Depth-ordered hash-grid encoding of image-ordered rays (round 5; SURVEY 8a row
// a4, the density of the FINE samples, reference
// nr4seg/nerf/renderer_semantics.py:214-226 -> network_tcnn_semantics.py:130-144).
//
// Why: the tiled encoder (hashgrid.hip) gives a wave the 8x8 pixels of a tile at
// ONE sample index.  In the coarse pass that is one depth, the 64 lanes sit in a
// patch of a few cells and share their cache lines.  In the fine pass the i-th
// sample of neighbouring rays is anywhere (importance sampling with independent
// uniforms: the depth spread inside a wave is 1.2 scene units on the benchmark
// field against 0.066 between coarse samples), no two lanes share a cell from
// level 4 up, and every level costs up to twice its coarse-pass time
// (tools/encode_depth_coherence.py: fine pass 1.25 ms -> 0.74 ms when the same
// depths are dealt to the lanes in depth order).
//
// How: (1) k_tile_depth_order counting-sorts the 64 x T samples of every tile by
// (depth slab, pixel) -- about one slab per 64 samples; samples of one pixel in
// one slab land in whatever order the LDS atomics give: the features of a sample
// do not depend on where it is processed.  It writes, in that order and per tile back to back,
// the depth, the pixel inside the tile and the ray-major slot of each sample.
// (2) The encoders below give a wave 64 CONSECUTIVE samples of that order -- a
// thin depth slab of the tile -- and write the features in the same order:
// depths in and features out are plain coalesced accesses, no LDS transposes.
// (3) The sigma MLP reads the features in that order and scatters h / sigma to
// the ray-major slots (ucsa_sigma_mlp_fwd_scatter), so nothing downstream
// changes.  Per sample the arithmetic is encode_level's: bit-identical results.
#include <cstdlib>

#include "hashgrid_sorted.h"


// Sort key of a sample: (depth slab, pixel).  The tile's depth range is cut
// into `nbins` slabs (about one per 64 samples) and a slab holds its samples in
// PIXEL order: a wave's 64 consecutive samples are then a thin slab of the tile
// with neighbouring pixels in neighbouring lanes -- the arrangement of the
// coarse pass (measured: with the lanes of a slab in arbitrary order the fine
// levels gained half as much; equal lines merge best between adjacent lanes).
// Counting sort: one LDS counter per (slab, pixel), dynamic LDS nbins * 64 * 4 B.
#define SORT_THREADS 1024u
extern __shared__ uint32_t sort_hist[];   // [nbins][64]
__global__ void __launch_bounds__(SORT_THREADS)
k_tile_depth_order(const float* __restrict__ z, uint32_t rows, uint32_t T,
                   uint32_t W, float inv_T, uint32_t nbins,
                   float* __restrict__ z_sorted, uint8_t* __restrict__ pix,
                   uint32_t* __restrict__ slot) {
  __shared__ float red[32];
  __shared__ uint32_t part[SORT_THREADS / 64u];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
  const TileGeom tg = tile_geom(blockIdx.x, rows, W, T);
  const uint32_t n_el = 64u * T, n_ctr = nbins * 64u;
  // element e = (pixel p of the 8x8 tile, sample s): p = e / T exactly
  // ((e + 0.5) / T is at least 0.5 / T away from an integer; e < 2^16, T <= 1024)
  auto decode = [&](uint32_t e, uint32_t& p, uint32_t& src) -> bool {
    p = (uint32_t)(((float)e + 0.5f) * inv_T);
    const uint32_t s = e - p * T;
    const uint32_t lx = p & 7u, ly = p >> 3;
    src = ((tg.py0 + ly) * W + tg.px0 + lx) * T + s;
    return lx < tg.wt && ly < tg.ht;
  };
  float lo = __builtin_inff(), hi = -__builtin_inff();
#pragma unroll 4
  for (uint32_t e = tid; e < n_el; e += SORT_THREADS) {
    uint32_t p, src;
    if (decode(e, p, src)) {
      const float v = z[src];
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, d, 64));
    hi = fmaxf(hi, __shfl_xor(hi, d, 64));
  }
  if (lane == 0) {
    red[wid] = lo;
    red[16 + wid] = hi;
  }
  for (uint32_t b = tid; b < n_ctr; b += SORT_THREADS) sort_hist[b] = 0u;
  __syncthreads();
  lo = red[0];
  hi = red[16];
#pragma unroll
  for (int i = 1; i < 16; ++i) {
    lo = fminf(lo, red[i]);
    hi = fmaxf(hi, red[16 + i]);
  }
  const float scale = hi > lo ? (float)nbins / (hi - lo) : 0.f;
  auto ctr_of = [&](float v, uint32_t p) -> uint32_t {
    const uint32_t b = (uint32_t)((v - lo) * scale);   // NaN / negative -> 0
    return (b < nbins ? b : nbins - 1u) * 64u + p;
  };
#pragma unroll 4
  for (uint32_t e = tid; e < n_el; e += SORT_THREADS) {
    uint32_t p, src;
    if (decode(e, p, src)) atomicAdd(&sort_hist[ctr_of(z[src], p)], 1u);
  }
  __syncthreads();
  // exclusive scan of the counters: a contiguous run per thread, then the sums
  const uint32_t per = (n_ctr + SORT_THREADS - 1u) / SORT_THREADS;
  const uint32_t c0 = tid * per, c1 = c0 + per < n_ctr ? c0 + per : n_ctr;
  uint32_t sum = 0u;
  for (uint32_t c = c0; c < c1; ++c) sum += sort_hist[c];
  // (wave scan by shuffles, then the 16 wave totals)
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
    if (lane >= (uint32_t)d) incl += o;
  }
  if (lane == 63u) part[wid] = incl;
  __syncthreads();
  uint32_t run = incl - sum;
  for (uint32_t w = 0; w < wid; ++w) run += part[w];
  for (uint32_t c = c0; c < c1; ++c) {
    const uint32_t n = sort_hist[c];
    sort_hist[c] = run;
    run += n;
  }
  __syncthreads();
#pragma unroll 4
  for (uint32_t e = tid; e < n_el; e += SORT_THREADS) {
    uint32_t p, src;
    if (decode(e, p, src)) {
      const float v = z[src];
      const uint32_t pos = tg.base + atomicAdd(&sort_hist[ctr_of(v, p)], 1u);
      z_sorted[pos] = v;
      pix[pos] = (uint8_t)p;
      slot[pos] = src;
    }
  }
}

// The same with EQUAL-COUNT slabs (T <= 256): first an (almost) exact depth
// rank per sample -- counting sort over 4096 depth bins -- then the samples of
// every run of 64 consecutive ranks in pixel order (a second counting sort, one
// byte-wide counter per (run, pixel): four to a word).  A wave's 64 samples are
// then the thinnest slab the tile's samples allow, whatever the distribution of
// the depths (a trained scene puts most fine samples of a tile into a few
// centimetres: fixed-width slabs would hold thousands of samples there).
// The tile's depths live in LDS, and the three output arrays are written in
// output order (coalesced) through an inverse map: scattered global stores were
// a third of the kernel.
#define SORT_BINS1 4096u
__global__ void __launch_bounds__(SORT_THREADS)
k_tile_depth_order2(const float* __restrict__ z, uint32_t rows, uint32_t T,
                    uint32_t W, float inv_T, float* __restrict__ z_sorted,
                    uint8_t* __restrict__ pix, uint32_t* __restrict__ slot) {
  // dynamic LDS: hist [4096] u32 (reused as the byte counters [T][16] words of
  // the second sort) | z_l [64 T] f32 | rank_of [64 T] u16 | inv_of [64 T] u16
  const uint32_t n_el = 64u * T;
  uint32_t* hist = sort_hist;
  float* z_l = reinterpret_cast<float*>(sort_hist + SORT_BINS1);
  uint16_t* rank_of = reinterpret_cast<uint16_t*>(z_l + n_el);
  uint16_t* inv_of = rank_of + n_el;
  __shared__ float red[32];
  __shared__ uint32_t part[SORT_THREADS / 64u];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
  const TileGeom tg = tile_geom(blockIdx.x, rows, W, T);
  // element e = (pixel p of the 8x8 tile, sample s): p = e / T exactly
  auto pixel_of = [&](uint32_t e) -> uint32_t {
    return (uint32_t)(((float)e + 0.5f) * inv_T);
  };
  auto valid = [&](uint32_t p) -> bool { return (p & 7u) < tg.wt && (p >> 3) < tg.ht; };
  auto src_of = [&](uint32_t e, uint32_t p) -> uint32_t {
    return ((tg.py0 + (p >> 3)) * W + tg.px0 + (p & 7u)) * T + (e - p * T);
  };
  float lo = __builtin_inff(), hi = -__builtin_inff();
#pragma unroll 4
  for (uint32_t e = tid; e < n_el; e += SORT_THREADS) {
    const uint32_t p = pixel_of(e);
    if (valid(p)) {
      const float v = z[src_of(e, p)];
      z_l[e] = v;
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, d, 64));
    hi = fmaxf(hi, __shfl_xor(hi, d, 64));
  }
  if (lane == 0) {
    red[wid] = lo;
    red[16 + wid] = hi;
  }
  for (uint32_t b = tid; b < SORT_BINS1; b += SORT_THREADS) hist[b] = 0u;
  __syncthreads();
  lo = red[0];
  hi = red[16];
#pragma unroll
  for (int i = 1; i < 16; ++i) {
    lo = fminf(lo, red[i]);
    hi = fmaxf(hi, red[16 + i]);
  }
  const float scale = hi > lo ? (float)SORT_BINS1 / (hi - lo) : 0.f;
  auto bin_of = [&](float v) -> uint32_t {
    const uint32_t b = (uint32_t)((v - lo) * scale);   // NaN / negative -> 0
    return b < SORT_BINS1 ? b : SORT_BINS1 - 1u;
  };
#pragma unroll 4
  for (uint32_t e = tid; e < n_el; e += SORT_THREADS)
    if (valid(pixel_of(e))) atomicAdd(&hist[bin_of(z_l[e])], 1u);
  __syncthreads();
  {   // exclusive scan of the 4096 bins: 4 per thread, wave scan, wave totals
    const uint32_t c0 = tid * (SORT_BINS1 / SORT_THREADS);
    uint32_t n[SORT_BINS1 / SORT_THREADS], sum = 0u;
#pragma unroll
    for (uint32_t i = 0; i < SORT_BINS1 / SORT_THREADS; ++i) sum += (n[i] = hist[c0 + i]);
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63u) part[wid] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < wid; ++w) run += part[w];
#pragma unroll
    for (uint32_t i = 0; i < SORT_BINS1 / SORT_THREADS; ++i) {
      hist[c0 + i] = run;
      run += n[i];
    }
  }
  __syncthreads();
#pragma unroll 4
  for (uint32_t e = tid; e < n_el; e += SORT_THREADS)
    if (valid(pixel_of(e))) rank_of[e] = (uint16_t)atomicAdd(&hist[bin_of(z_l[e])], 1u);
  __syncthreads();
  // second sort: inside every run of 64 ranks by pixel.  Byte counters, four
  // pixels to a word: a count and its prefix are <= 64.
  const uint32_t n_words = T * 16u;   // (<= 4096: inside hist)
  for (uint32_t b = tid; b < n_words; b += SORT_THREADS) hist[b] = 0u;
  __syncthreads();
#pragma unroll 4
  for (uint32_t e = tid; e < n_el; e += SORT_THREADS) {
    const uint32_t p = pixel_of(e);
    if (valid(p))
      atomicAdd(&hist[((uint32_t)rank_of[e] >> 6) * 16u + (p >> 2)], 1u << (8u * (p & 3u)));
  }
  __syncthreads();
  for (uint32_t g = tid; g < T; g += SORT_THREADS) {   // one run per thread
    uint32_t run = 0u;
#pragma unroll
    for (uint32_t w = 0; w < 16u; ++w) {
      const uint32_t c = hist[g * 16u + w];
      const uint32_t b0 = c & 255u, b1 = (c >> 8) & 255u, b2 = (c >> 16) & 255u, b3 = c >> 24;
      hist[g * 16u + w] = run | ((run + b0) << 8) | ((run + b0 + b1) << 16) |
                          ((run + b0 + b1 + b2) << 24);
      run += b0 + b1 + b2 + b3;
    }
  }
  __syncthreads();
#pragma unroll 4
  for (uint32_t e = tid; e < n_el; e += SORT_THREADS) {
    const uint32_t p = pixel_of(e);
    if (valid(p)) {
      const uint32_t g = (uint32_t)rank_of[e] >> 6, sh = 8u * (p & 3u);
      const uint32_t old = atomicAdd(&hist[g * 16u + (p >> 2)], 1u << sh);
      inv_of[g * 64u + ((old >> sh) & 255u)] = (uint16_t)e;
    }
  }
  __syncthreads();
  for (uint32_t q = tid; q < tg.count; q += SORT_THREADS) {   // output order
    const uint32_t e = inv_of[q], p = pixel_of(e);
    z_sorted[tg.base + q] = z_l[e];
    pix[tg.base + q] = (uint8_t)p;
    slot[tg.base + q] = src_of(e, p);
  }
}

// One level per grid row (finest level first: level = l_top - blockIdx.y), a
// workgroup = 1024 consecutive samples of a tile's depth order.
template <typename TT, typename FT, bool LEAN>
__global__ void __launch_bounds__(256)
k_hashgrid_encode_sorted(GridDev g, uint32_t l_top, const TT* __restrict__ table,
                         const float* __restrict__ rays_o,
                         const float* __restrict__ rays_d,
                         const float* __restrict__ z_sorted,
                         const uint8_t* __restrict__ pix, Aabb bb, uint32_t T,
                         uint32_t rows, uint32_t W, uint32_t s_blocks,
                         uint32_t M, FT* __restrict__ feat) {
  __shared__ __attribute__((aligned(16))) float ray_s[64][8];
  const uint32_t level = l_top - blockIdx.y;
  const uint32_t bid = xcd_band(blockIdx.x, gridDim.x);
  const uint32_t sb = bid % s_blocks, tile = bid / s_blocks;
  const TileGeom tg = tile_geom(tile, rows, W, T);
  if (sb * 1024u >= tg.count) return;   // (workgroup-uniform)
  load_tile_rays(ray_s, tg, W, rays_o, rays_d);
  __syncthreads();
  const float two_b = 2.0f * g.bound, inv = unit_inv(two_b);
  const TT* tab = table + g.offset[level];
  const float scale = g.scale[level];
  const uint32_t res = g.res[level], entries = g.entries[level],
                 hashed = g.hashed[level];
  FT* feat_level = feat + (size_t)level * M + tg.base;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t rank = sb * 1024u + 256u * k + threadIdx.x;
    if (rank >= tg.count) continue;
    const float zz = z_sorted[tg.base + rank];
    const uint32_t p = pix[tg.base + rank];
    float ux, uy, uz;
    unit_position(ray_s, p, zz, bb, g.bound, two_b, inv, ux, uy, uz);
    const float2 f = LEAN ? encode_cell(tab, ux, uy, uz, scale, res, res * res, entries, hashed)
                          : encode_level(tab, ux, uy, uz, scale, res, entries, hashed);
    FT o;
    to_feat(o, f);
    feat_store(feat_level + rank, o);
  }
}

// Levels [l_lo, l_hi) in one workgroup (the issue-bound coarse levels; see
// k_hashgrid_encode_tiled_ml): unit-cube coordinates stay in registers.
template <typename TT, typename FT>
__global__ void __launch_bounds__(256)
k_hashgrid_encode_sorted_ml(GridDev g, uint32_t l_lo, uint32_t l_hi,
                            const TT* __restrict__ table,
                            const float* __restrict__ rays_o,
                            const float* __restrict__ rays_d,
                            const float* __restrict__ z_sorted,
                            const uint8_t* __restrict__ pix, Aabb bb, uint32_t T,
                            uint32_t rows, uint32_t W, uint32_t s_blocks,
                            uint32_t M, FT* __restrict__ feat) {
  __shared__ __attribute__((aligned(16))) float ray_s[64][8];
  const uint32_t bid = xcd_band(blockIdx.x, gridDim.x);
  const uint32_t sb = bid % s_blocks, tile = bid / s_blocks;
  const TileGeom tg = tile_geom(tile, rows, W, T);
  if (sb * 1024u >= tg.count) return;
  load_tile_rays(ray_s, tg, W, rays_o, rays_d);
  __syncthreads();
  const float two_b = 2.0f * g.bound, inv = unit_inv(two_b);
  float ux[4], uy[4], uz[4];
  uint32_t live = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t rank = sb * 1024u + 256u * k + threadIdx.x;
    ux[k] = uy[k] = uz[k] = 0.f;
    if (rank >= tg.count) continue;
    live |= 1u << k;
    unit_position(ray_s, pix[tg.base + rank], z_sorted[tg.base + rank], bb, g.bound,
                  two_b, inv, ux[k], uy[k], uz[k]);
  }
  for (uint32_t level = l_hi; level-- > l_lo;) {
    const TT* tab = table + g.offset[level];
    const float scale = g.scale[level];
    const uint32_t res = g.res[level], entries = g.entries[level],
                   hashed = g.hashed[level];
    const uint32_t res2 = res * res;
    FT* feat_level = feat + (size_t)level * M + tg.base + sb * 1024u + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!(live >> k & 1u)) continue;
      FT o;
      to_feat(o, encode_cell(tab, ux[k], uy[k], uz[k], scale, res, res2, entries, hashed));
      feat_store(feat_level + 256u * k, o);
    }
  }
}

// The COARSE samples of a tile need no sort: sample s of every ray sits at (nearly)
// one depth (z = near + (far - near) s / (T - 1), near / far vary slowly over 8x8
// pixels), so the order "sample index, then pixel" IS the tile's depth order up to
// that variation -- a wave of the encoders = the tile's pixels at one sample index,
// as in the image-ordered tiled kernel.  Writes the same three arrays as
// k_tile_depth_order2 (z_sorted, pix, slot) with a transpose instead of two LDS
// sorts.  Used for coarse passes of MORE than 128 samples per ray (the reference's
// native 256): there the sort (144 KiB of LDS per tile) costs more than the fused
// density kernel gains, while this order gets nearly all of that gain (cfg2, 96
// samples: 15.75 ms per view against 15.66 with the sort and 15.89 with an
// image-ordered coarse pass).
__global__ void __launch_bounds__(256)
k_tile_index_order(const float* __restrict__ z, uint32_t rows, uint32_t T, uint32_t W,
                   float* __restrict__ z_sorted, uint8_t* __restrict__ pix,
                   uint32_t* __restrict__ slot) {
  __shared__ float zt[64][33];     // 32 samples of the tile's 64 pixels, transposed
  const TileGeom tg = tile_geom(blockIdx.x, rows, W, T);
  const uint32_t np = tg.wt * tg.ht;
  for (uint32_t s0 = 0; s0 < T; s0 += 32u) {
    const uint32_t ns = T - s0 < 32u ? T - s0 : 32u;
    // read: consecutive threads = consecutive samples of a ray (coalesced)
    for (uint32_t e = threadIdx.x; e < 64u * 32u; e += 256u) {
      const uint32_t k = e >> 5, ds = e & 31u;       // k: valid-pixel index in row-major order
      if (k < np && ds < ns) {
        const uint32_t lx = k % tg.wt, ly = k / tg.wt;
        zt[k][ds] = z[((tg.py0 + ly) * W + tg.px0 + lx) * T + s0 + ds];
      }
    }
    __syncthreads();
    // write: consecutive threads = consecutive pixels at one sample index
    for (uint32_t e = threadIdx.x; e < ns * np; e += 256u) {
      const uint32_t ds = e / np, k = e - ds * np;
      const uint32_t lx = k % tg.wt, ly = k / tg.wt;
      const uint32_t q = tg.base + (s0 + ds) * np + k;
      z_sorted[q] = zt[k][ds];
      pix[q] = (uint8_t)(ly * 8u + lx);
      slot[q] = ((tg.py0 + ly) * W + tg.px0 + lx) * T + s0 + ds;
    }
    __syncthreads();
  }
}

extern "C" int32_t ucsa_tile_index_order(const float* z, uint32_t N, uint32_t T,
                                         uint32_t image_width, float* z_sorted,
                                         uint8_t* pix, uint32_t* slot, void* stream) {
  UCSA_CHECK_ARG(z, 0);
  UCSA_CHECK_ARG(T >= 1 && T <= 1024, 2);
  UCSA_CHECK_ARG(image_width >= 1 && N % image_width == 0, 3);
  UCSA_CHECK_ARG(z_sorted && pix && slot, 4);
  UCSA_CHECK_ARG((uint64_t)N * T < 0x80000000ull, 1);
  if (N == 0) return 0;
  const uint32_t rows = N / image_width;
  const uint32_t tiles = ((image_width + 7u) / 8u) * ((rows + 7u) / 8u);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_tile_index_order, dim3(tiles), dim3(256), 0, (hipStream_t)stream, z,
                     rows, T, image_width, z_sorted, pix, slot);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_tile_depth_order(const float* z, uint32_t N, uint32_t T,
                                         uint32_t image_width, float* z_sorted,
                                         uint8_t* pix, uint32_t* slot,
                                         void* stream) {
  UCSA_CHECK_ARG(z, 0);
  UCSA_CHECK_ARG(T >= 1 && T <= 1024, 2);
  UCSA_CHECK_ARG(image_width >= 1 && N % image_width == 0, 3);
  UCSA_CHECK_ARG(z_sorted && pix && slot, 4);
  UCSA_CHECK_ARG((uint64_t)N * T < 0x80000000ull, 1);
  if (N == 0) return 0;
  const uint32_t rows = N / image_width;
  const uint32_t tiles = ((image_width + 7u) / 8u) * ((rows + 7u) / 8u);
  // about one slab per 64 samples of a full tile, a power of two in [16, 128]
  // (128 slabs x 64 pixels x 4 B = 32 KiB of LDS)
  uint32_t nbins = 16u;
  while (nbins < T && nbins < 128u) nbins <<= 1;
  const char* nb = ucsa_getenv("UCSA_SORT_BINS");   // experiments only
  if (nb && *nb) {
    const uint32_t v = (uint32_t)strtoul(nb, nullptr, 10);
    if (v >= 1u && v <= 128u) nbins = v;
  }
  const char* sv = ucsa_getenv("UCSA_SORT_EXACT");   // experiments only; default on
  const bool exact = T <= 256u && !(sv && sv[0] == '0');
  UCSA_CLEAR_ERR();
  if (exact && !(nb && *nb)) {
    // hist + depths (4 B) + rank and inverse map (2 B each) per sample: 52 KiB at
    // T = 96, 144 KiB at T = 256 (a workgroup may hold all 160 KiB of a CU)
    const uint32_t lds = SORT_BINS1 * (uint32_t)sizeof(uint32_t) + 64u * T * 8u;
    if (lds > 65536u)
      (void)hipFuncSetAttribute((const void*)k_tile_depth_order2,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_tile_depth_order2, dim3(tiles), dim3(SORT_THREADS), lds,
                       (hipStream_t)stream, z, rows, T, image_width, 1.0f / (float)T,
                       z_sorted, pix, slot);
  }
  else
    hipLaunchKernelGGL(k_tile_depth_order, dim3(tiles), dim3(SORT_THREADS),
                       nbins * 64u * sizeof(uint32_t), (hipStream_t)stream, z, rows, T,
                       image_width, 1.0f / (float)T, nbins, z_sorted, pix, slot);
  return ucsa_launch_status();
}

template <typename FT>
static int32_t launch_sorted(const ucsa_grid* grid, const float* table,
                             const float* rays_o, const float* rays_d,
                             const float* z_sorted, const uint8_t* pix, Aabb bb,
                             uint32_t N, uint32_t T, uint32_t image_width,
                             void* feat, void* stream, uint32_t first_level = 0u) {
  const GridDev gd = ucsa_grid_dev(grid);
  const uint32_t rows = N / image_width;
  const uint32_t tiles = ((image_width + 7u) / 8u) * ((rows + 7u) / 8u);
  const uint32_t s_blocks = ucsa_div_up(64u * T, 1024u);
  const uint32_t M = N * T;
  // levels [0, n_ml) through the several-levels kernel (UCSA_ENC_SORTED_ML:
  // experiments only); UCSA_ENC_SORTED_LEAN=0: the fine levels' gather as
  // hashgrid.hip's encode_level instead of encode_cell.  Same features.
  auto env_u = [](const char* name, uint32_t dflt) {
    const char* v = ucsa_getenv(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
  };
  uint32_t n_ml = env_u("UCSA_ENC_SORTED_ML", 9u);
  if (n_ml > grid->n_levels) n_ml = grid->n_levels;
  // first_level > 0: only levels [first_level, L), one level per grid row (the
  // levels below are computed by the consumer, encode_sigma_sorted.hip)
  if (first_level > 0u) n_ml = first_level;
  const bool lean = env_u("UCSA_ENC_SORTED_LEAN", 1u) != 0u;
  UCSA_CLEAR_ERR();
  if (n_ml < grid->n_levels) {
    const dim3 g(tiles * s_blocks, grid->n_levels - n_ml);
    if (lean)
      hipLaunchKernelGGL((k_hashgrid_encode_sorted<float2, FT, true>), g, dim3(256), 0,
                         (hipStream_t)stream, gd, grid->n_levels - 1u,
                         (const float2*)table, rays_o, rays_d, z_sorted, pix, bb, T,
                         rows, image_width, s_blocks, M, (FT*)feat);
    else
      hipLaunchKernelGGL((k_hashgrid_encode_sorted<float2, FT, false>), g, dim3(256), 0,
                         (hipStream_t)stream, gd, grid->n_levels - 1u,
                         (const float2*)table, rays_o, rays_d, z_sorted, pix, bb, T,
                         rows, image_width, s_blocks, M, (FT*)feat);
  }
  if (n_ml > 0 && first_level == 0u)
    hipLaunchKernelGGL((k_hashgrid_encode_sorted_ml<float2, FT>), dim3(tiles * s_blocks),
                       dim3(256), 0, (hipStream_t)stream, gd, 0u, n_ml,
                       (const float2*)table, rays_o, rays_d, z_sorted, pix, bb, T,
                       rows, image_width, s_blocks, M, (FT*)feat);
  return ucsa_launch_status();
}

static int32_t check_sorted_args(const ucsa_grid* grid, const void* table,
                                 const float* rays_o, const float* rays_d,
                                 const float* z_sorted, const uint8_t* pix,
                                 const float* aabb_host, uint32_t N, uint32_t T,
                                 uint32_t image_width, const void* feat) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(table, 1);
  UCSA_CHECK_ARG(rays_o && rays_d, 2);
  UCSA_CHECK_ARG(z_sorted && pix, 4);
  UCSA_CHECK_ARG(aabb_host, 6);
  UCSA_CHECK_ARG(T >= 1 && T <= 1024 && (uint64_t)N * T < 0x80000000ull, 8);
  UCSA_CHECK_ARG(image_width >= 1 && N % image_width == 0, 9);
  UCSA_CHECK_ARG(feat, 10);
  return 0;
}

extern "C" int32_t ucsa_hashgrid_encode_sorted(
    const ucsa_grid* grid, const float* table, const float* rays_o,
    const float* rays_d, const float* z_sorted, const uint8_t* pix,
    const float* aabb_host, uint32_t N, uint32_t T, uint32_t image_width,
    float* feat, void* stream) {
  const int32_t rc = check_sorted_args(grid, table, rays_o, rays_d, z_sorted, pix,
                                       aabb_host, N, T, image_width, feat);
  if (rc != 0 || N == 0) return rc;
  return launch_sorted<float2>(grid, table, rays_o, rays_d, z_sorted, pix,
                               ucsa_aabb(aabb_host), N, T, image_width, feat, stream);
}

extern "C" int32_t ucsa_hashgrid_encode_sorted_hf(
    const ucsa_grid* grid, const float* table, const float* rays_o,
    const float* rays_d, const float* z_sorted, const uint8_t* pix,
    const float* aabb_host, uint32_t N, uint32_t T, uint32_t image_width,
    void* feat_half, void* stream) {
  const int32_t rc = check_sorted_args(grid, table, rays_o, rays_d, z_sorted, pix,
                                       aabb_host, N, T, image_width, feat_half);
  if (rc != 0 || N == 0) return rc;
  return launch_sorted<ucsa_half2>(grid, table, rays_o, rays_d, z_sorted, pix,
                                   ucsa_aabb(aabb_host), N, T, image_width, feat_half,
                                   stream);
}

// levels [first_level, L) only (see encode_sigma_sorted.hip)
int32_t ucsa_hashgrid_encode_sorted_from(const ucsa_grid* grid, const float* table,
                                         const float* rays_o, const float* rays_d,
                                         const float* z_sorted, const uint8_t* pix,
                                         const float* aabb_host, uint32_t N, uint32_t T,
                                         uint32_t image_width, uint32_t first_level,
                                         float* feat, void* stream) {
  const int32_t rc = check_sorted_args(grid, table, rays_o, rays_d, z_sorted, pix,
                                       aabb_host, N, T, image_width, feat);
  if (rc != 0 || N == 0) return rc;
  return launch_sorted<float2>(grid, table, rays_o, rays_d, z_sorted, pix,
                               ucsa_aabb(aabb_host), N, T, image_width, feat, stream,
                               first_level);
}
