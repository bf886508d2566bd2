// Merge of coarse+fine samples, final weights, masked colour / semantics MLPs
// and alpha compositing (SURVEY 8a rows a5 (second half), a6-a9).
// reference nr4seg/nerf/renderer_semantics.py:220-299 and
// nr4seg/nerf/network_tcnn_semantics.py:147-207.
//
// Structure (DESIGN.md "composite kernel"):
//  * every wave owns a contiguous range of rays and is independent of the
//    other waves of its workgroup after the weights are staged in LDS;
//  * phase A (per ray, VALU+LDS): stable merge by rank counting, alpha /
//    transmittance by wave scan, mask w > 1e-4, depth, ballot+popcount
//    compaction of the surviving samples into the wave's entry list;
//  * phase C (per 64 surviving samples, MFMA): colour and semantics nets on
//    4 column blocks of 16 samples, fp32 MFMA, A fragments from LDS,
//    softmax across the 4 lane groups by two xor-shuffles;
//  * phase D: w*rgb and w*p go through a 16 x 44 LDS tile and are summed per
//    ray in sample order by lane c = channel (deterministic, no atomics);
//    the running sums stay in registers across groups and are stored once
//    per ray.
// The semantic weights are the same numbers as the colour weights in the
// forward pass (they differ only in autograd: detached, :270).
#include "composite_common.h"
#include "mfma_mlp_h2.h"

#define CMP_MAX_WAVES 12
#define CMP_CBS 2    // column blocks of 16 samples in flight per wave (fp32)
#define CMP_CBS_H 2  // fp16 option (4 in flight measured slower: 9.9 vs 11.7 M rays/s)
#define CMP_H2_WAVES 12  // f16x2 (layer-major: 165 VGPRs, three waves per SIMD)

extern __shared__ __attribute__((aligned(16))) float cmp_smem[];

// Marched mode: slots (rays) handled by one wave, and segmented wave64 scans
// (a lane with `head` starts a new segment).
#define MARCH_RPW_MAX 128
#define MARCH_SLOT_FLOATS (5 * MARCH_RPW_MAX)

__device__ __forceinline__ float seg_incl_scan_mul(float v, bool head,
                                                   uint32_t lane) {
  int f = head ? 1 : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    const int of = __shfl_up(f, d, 64);
    if (lane >= (uint32_t)d && !f) {
      v = o * v;
      f = of;
    }
  }
  return v;
}

__device__ __forceinline__ float seg_incl_scan_add(float v, bool head,
                                                   uint32_t lane) {
  int f = head ? 1 : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    const int of = __shfl_up(f, d, 64);
    if (lane >= (uint32_t)d && !f) {
      v = o + v;
      f = of;
    }
  }
  return v;
}

struct CmpArgs {
  const float* rays_d;
  const float* norms;
  const float* z_c;
  const float* sigma_c;
  const float* h_c;
  const float* z_f;
  const float* sigma_f;
  const float* h_f;
  const float* packed_color;
  const float* packed_sem;
  uint32_t N, T, t, C;
  float density_scale;
  float* image;
  float* depth;
  float* semantics;
  int32_t* src_out;
  float* w_out;
  uint32_t rays_per_wave;
  uint32_t contrib_stride;
  // ---- marched spans (MARCH = true; see "marched mode" below) -------------
  const int32_t* rays_alive;
  float* rays_t;
  const int32_t* span;
  const float* deltas;
  const int32_t* n_alive_dev;
  float* weights_sum;
  uint32_t seg_cap;
  float w_min;
  // rays_alive / span may be columns of the training marcher's rays [N,3]
  uint32_t alive_stride, span_stride;
  // training pass over march_rays_train output: no early stop, rays_t is the
  // per-RAY start (nears, read-only), spans reaching n_points are skipped
  // (reference raymarching.cu:338), every sample's weight goes to w_out
  uint32_t train, n_points;
  float* t_out;  // training: ray parameter after each sample (depth backward)
};

// Marched mode (MARCH = true): the same shading machinery on
// the exact-size spans of the segmented marcher (raymarch.hip).  "Ray r" is
// alive slot r; its samples are rows span[r] = (first, count) of sigma_c /
// h_c / deltas; phase A is the early-stopping weight scan of
// k_seg_composite; samples with w > w_min are shaded; sums are ADDED to
// image / semantics / depth / weights_sum of ray rays_alive[r], and rays_t[r]
// is set for the next round.  List capacity is 64 + G: survivors are drained
// after every 64-sample trip.
// H2 (with HALF, marched mode; round 6): the nets in f16x2 -- two-term operands, three
// f16 MFMA passes per product (mfma_mlp_h2.h), fp32-grade like the f32-input chain at
// 72 instead of 160 MFMAs per column block; fragments as ucsa_mlp_pack_h2 writes them.
template <int NRB_SEM, int CBS, bool HALF, bool MARCH, bool H2 = false>
__global__ void __launch_bounds__(64 * (H2 ? CMP_H2_WAVES : CMP_MAX_WAVES))
k_composite(CmpArgs a) {
  static_assert(!H2 || HALF, "f16x2 is a mode of the 16-bit fragment path");
  constexpr uint32_t TERMS = H2 ? 2u : 1u;
  constexpr uint32_t G = 16u * CBS;  // survivors shaded per MFMA group
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t nw_block = blockDim.x >> 6;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint32_t T = a.T, t = a.t, S = a.T + a.t, C = a.C;
  const uint32_t cstride = a.contrib_stride;

  // ---- LDS carve ------------------------------------------------------
  // fp32: A fragments as floats; fp16 option: 16-byte half8 fragments
  constexpr uint32_t WC_FLOATS = HALF ? COLOR_H_FRAGS * 256 * TERMS : 7168;
  constexpr uint32_t WS_FLOATS = HALF ? SEM_H_FRAGS(NRB_SEM) * 256 * TERMS
                                      : 1024 + NRB_SEM * 1024;
  float* w_color = cmp_smem;
  float* w_sem = w_color + WC_FLOATS;
  float* per_wave = w_sem + WS_FLOATS;
  // entry list capacity: survivors are drained after every 64-sample trip
  // (fewer than G wait afterwards), so 64 + G entries always suffice -- a list
  // of S + G used to cap 512-sample frames at 6 waves per workgroup
  const uint32_t cap = 64u + G;
  // per wave: zraw[S] (later weights), zm[S], sgm[S], srcs[S],
  //           lw[cap], lrow[cap], lray[cap], contrib[16*cstride]
  const uint32_t per_wave_floats =
      4 * S + 3 * cap + 16 * cstride + 64 + (MARCH ? MARCH_SLOT_FLOATS : 0);
  float* base = per_wave + (size_t)wid * per_wave_floats;
  float* zraw = base;
  float* zm = zraw + S;
  float* sgm = zm + S;
  uint32_t* srcs = reinterpret_cast<uint32_t*>(sgm + S);
  float* lw = reinterpret_cast<float*>(srcs + S);
  uint32_t* lrow = reinterpret_cast<uint32_t*>(lw + cap);
  uint32_t* lray = lrow + cap;
  float* contrib = reinterpret_cast<float*>(lray + cap);
  float* shpart = contrib + 16 * cstride;  // [64] colour-L1 SH part of one ray
  // marched mode: per-slot tables of the wave's rays
  uint32_t* slot_off = reinterpret_cast<uint32_t*>(shpart + 64);
  uint32_t* slot_cnt = slot_off + MARCH_RPW_MAX;
  uint32_t* slot_idx = slot_cnt + MARCH_RPW_MAX;
  float* slot_T0 = reinterpret_cast<float*>(slot_idx + MARCH_RPW_MAX);
  float* slot_t0 = slot_T0 + MARCH_RPW_MAX;

  for (uint32_t i = threadIdx.x; i < WC_FLOATS; i += blockDim.x)
    w_color[i] = a.packed_color[i];
  for (uint32_t i = threadIdx.x; i < WS_FLOATS; i += blockDim.x)
    w_sem[i] = a.packed_sem[i];
  __syncthreads();

  uint32_t n_rays = a.N;
  if constexpr (MARCH) {
    if (a.n_alive_dev) {
      const uint32_t live = (uint32_t)a.n_alive_dev[0];
      if (live < n_rays) n_rays = live;
    }
  }
  const uint64_t gwave = (uint64_t)blockIdx.x * nw_block + wid;
  const uint64_t r_begin64 = gwave * a.rays_per_wave;
  if (r_begin64 >= n_rays) return;
  const uint32_t r_begin = (uint32_t)r_begin64;
  const uint32_t r_end = (r_begin + a.rays_per_wave < n_rays)
                             ? r_begin + a.rays_per_wave : n_rays;

  uint32_t cnt = 0;            // entries waiting in the list (wave-uniform)
  uint32_t cur_ray = 0xFFFFFFFFu;  // ray whose sums sit in `acc`
  uint32_t sh_ray = 0xFFFFFFFFu;   // ray whose SH part sits in `shpart`
  float acc = 0.0f;            // lane c: running sum of channel c

  auto flush_ray = [&](uint32_t ray) {
    if constexpr (MARCH) {  // a ray is owned by one wave per round: plain RMW
      if (lane < 3) a.image[(size_t)ray * 3 + lane] += acc;
      else if (lane < 3 + C) a.semantics[(size_t)ray * C + (lane - 3)] += acc;
    } else {
      if (lane < 3) a.image[(size_t)ray * 3 + lane] = acc;
      else if (lane < 3 + C) a.semantics[(size_t)ray * C + (lane - 3)] = acc;
    }
  };

  // ---- phase C + D on the first `n` (<= 64) entries of the list ---------
  auto shade = [&](uint32_t n) {
    float ew[CBS], geo[CBS][4];
    uint32_t eray[CBS];
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      uint32_t e = cb * 16 + j;
      const bool live = e < n;
      if (!live) e = n - 1;  // pad with the last real entry, weight 0
      ew[cb] = live ? lw[e] : 0.0f;
      const uint32_t row = lrow[e];
      eray[cb] = lray[e];
      const float* hp = ((row & ROW_FINE) ? a.h_f : a.h_c) +
                        (size_t)(row & ~ROW_FINE) * 16 + 4 * g;
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hp);
      geo[cb][0] = (g == 0) ? 1.0f : hv[0];  // slot m==0 -> the "ones" pad
      geo[cb][1] = hv[1];
      geo[cb][2] = hv[2];
      geo[cb][3] = hv[3];
    }

    float rgb[CBS][3];
    f32x4 lg[CBS][NRB_SEM];
    if constexpr (!HALF) {
    // ---------------- colour net: 32 -> 64 -> 64 -> 16 -------------------
    {
      f32x4 acc1[CBS][4];
      // First layer, SH half (k-steps 0..3).  The direction -- hence this
      // partial sum -- is the same for every sample of a ray, and a column
      // block usually holds 16 samples of ONE ray: compute it once per ray,
      // keep it in LDS, and start the accumulators from it.  Same k order as
      // the plain chain, so results are bit-identical.
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        const uint32_t ray0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)eray[cb]);
        const bool uniform = __all(eray[cb] == ray0);
        float sh[4];
        if (!uniform || ray0 != sh_ray) {  // SH values only when (re)computed
          const float* dd = a.rays_d + (size_t)eray[cb] * 3;
          sh4_select(dd[0], dd[1], dd[2], g, sh);
        }
        if (uniform && ray0 != sh_ray) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
              p = mfma16(w_color[(rb * 8 + ks) * 64 + lane], sh[ks], p);
            if (j == 0) *reinterpret_cast<f32x4*>(shpart + 16 * rb + 4 * g) = p;
          }
          sh_ray = ray0;
          wave_lds_sync();
        }
        if (uniform) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
            acc1[cb][rb] = *reinterpret_cast<const f32x4*>(shpart + 16 * rb + 4 * g);
        } else {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
              p = mfma16(w_color[(rb * 8 + ks) * 64 + lane], sh[ks], p);
            acc1[cb][rb] = p;
          }
        }
      }
      // geo half (k-steps 4..7) for all column blocks
#pragma unroll
      for (int ks = 4; ks < 8; ++ks) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const float wa = w_color[(rb * 8 + ks) * 64 + lane];
#pragma unroll
          for (int cb = 0; cb < CBS; ++cb)
            acc1[cb][rb] = mfma16(wa, geo[cb][ks - 4], acc1[cb][rb]);
        }
      }
      float hid[CBS][16];
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) chain_relu(acc1[cb], hid[cb]);
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc1[cb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const float wa = w_color[(COLOR_L1_FRAGS + rb * 16 + ks) * 64 + lane];
#pragma unroll
          for (int cb = 0; cb < CBS; ++cb)
            acc1[cb][rb] = mfma16(wa, hid[cb][ks], acc1[cb][rb]);
        }
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) chain_relu(acc1[cb], hid[cb]);
      f32x4 o3[CBS];
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) o3[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float wa =
            w_color[(COLOR_L1_FRAGS + COLOR_L2_FRAGS + ks) * 64 + lane];
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) o3[cb] = mfma16(wa, hid[cb][ks], o3[cb]);
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          rgb[cb][c] = fast_sigmoid(o3[cb][c]);  // rows 0..2 live in g==0
    }

    // ---------------- semantics net: 16 -> 64 -> 16*NRB_SEM --------------
    {
      f32x4 acc1[CBS][4];
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc1[cb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const float wa = w_sem[(rb * 4 + ks) * 64 + lane];
#pragma unroll
          for (int cb = 0; cb < CBS; ++cb)
            acc1[cb][rb] = mfma16(wa, geo[cb][ks], acc1[cb][rb]);
        }
      }
      float hid[CBS][16];
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) chain_relu(acc1[cb], hid[cb]);
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb)
#pragma unroll
        for (int rb = 0; rb < NRB_SEM; ++rb) lg[cb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int rb = 0; rb < NRB_SEM; ++rb) {
          const float wa = w_sem[(SEM_L1_FRAGS + rb * 16 + ks) * 64 + lane];
#pragma unroll
          for (int cb = 0; cb < CBS; ++cb)
            lg[cb][rb] = mfma16(wa, hid[cb][ks], lg[cb][rb]);
        }
      }
    }

    } else if constexpr (H2) {
      // f16x2: the layer structure of the fp16 branch below with two-term operands,
      // layer-major as in k_shade16 (composite_split.hip): a weight fragment is read
      // from LDS once per group and used by all its column blocks.  `wl`: the lane
      // plus an opaque zero re-made per group, so that the fragment loads stay
      // inside the group instead of ~190 pinned VGPRs.
      const H2Sel hsel = h2_selectors();
      uint32_t zoff = 0;
      asm volatile("" : "+v"(zoff));
      const uint32_t wl = lane + zoff;
      H2X b1[CBS];
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        float sh[4];
        const float* dd = a.rays_d + (size_t)eray[cb] * 3;
        sh4_select(dd[0], dd[1], dd[2], g, sh);
        h2_split_pair(sh[0], sh[1], b1[cb], 0, hsel);
        h2_split_pair(sh[2], sh[3], b1[cb], 1, hsel);
        h2_split_pair(geo[cb][0], geo[cb][1], b1[cb], 2, hsel);
        h2_split_pair(geo[cb][2], geo[cb][3], b1[cb], 3, hsel);
      }
      f32x4 a1[CBS][4], a2[CBS][4];
      H2X h0[CBS], h1[CBS];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const H2W w = h2_frag(w_color, rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) a1[cb][rb] = h2_mul1(w, b1[cb]);
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        h0[cb] = h2_chain_relu(a1[cb][0], a1[cb][1], hsel);
        h1[cb] = h2_chain_relu(a1[cb][2], a1[cb][3], hsel);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const H2W wa = h2_frag(w_color, 4 + 2 * rb, wl), wb = h2_frag(w_color, 5 + 2 * rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) a2[cb][rb] = h2_mul2(wa, h0[cb], wb, h1[cb]);
      }
      // semantics L1 reads the h-row slots of b1 (zeros in the SH slots' place)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const H2W w = h2_frag(w_sem, rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) {
          H2X bs;
#pragma unroll
          for (int term = 0; term < 2; ++term)
            bs.t[term] = u32x4{b1[cb].t[term][2], b1[cb].t[term][3], 0u, 0u};
          a1[cb][rb] = h2_mul1(w, bs);
        }
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        h0[cb] = h2_chain_relu(a2[cb][0], a2[cb][1], hsel);
        h1[cb] = h2_chain_relu(a2[cb][2], a2[cb][3], hsel);
      }
      {
        const H2W wa = h2_frag(w_color, 12, wl), wb = h2_frag(w_color, 13, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) {
          const f32x4 o3 = h2_mul2(wa, h0[cb], wb, h1[cb]);
#pragma unroll
          for (int c = 0; c < 3; ++c) rgb[cb][c] = fast_sigmoid(o3[c]);
        }
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        h0[cb] = h2_chain_relu(a1[cb][0], a1[cb][1], hsel);
        h1[cb] = h2_chain_relu(a1[cb][2], a1[cb][3], hsel);
      }
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb) {
        const H2W wa = h2_frag(w_sem, 4 + 2 * rb, wl), wb = h2_frag(w_sem, 5 + 2 * rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) lg[cb][rb] = h2_mul2(wa, h0[cb], wb, h1[cb]);
      }
    } else {
      // fp16 option: 24 MFMAs (16x16x32) per column block instead of 160
      const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        // (the lane plus an opaque zero per column block: the fragment loads stay
        // next to their MFMAs instead of being hoisted into pinned VGPRs, which
        // spilled up to 74 of them)
        uint32_t zoff = 0;
        asm volatile("" : "+v"(zoff));
        const uint32_t wl = lane + zoff;
        float sh[4];
        const float* dd = a.rays_d + (size_t)eray[cb] * 3;
        sh4_select(dd[0], dd[1], dd[2], g, sh);
        half8 b1, bs;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          b1[r] = (_Float16)sh[r];
          b1[4 + r] = (_Float16)geo[cb][r];
          bs[r] = (_Float16)geo[cb][r];
          bs[4 + r] = (_Float16)0.f;
        }
        f32x4 a1[4], a2[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a1[rb] = mfma_h(frag_h(w_color, rb, wl), b1, z4);
        half8 h0 = chain_relu_h(a1[0], a1[1]), h1 = chain_relu_h(a1[2], a1[3]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          a2[rb] = mfma_h(frag_h(w_color, 4 + 2 * rb, wl), h0, z4);
          a2[rb] = mfma_h(frag_h(w_color, 5 + 2 * rb, wl), h1, a2[rb]);
        }
        h0 = chain_relu_h(a2[0], a2[1]);
        h1 = chain_relu_h(a2[2], a2[3]);
        f32x4 o3 = mfma_h(frag_h(w_color, 12, wl), h0, z4);
        o3 = mfma_h(frag_h(w_color, 13, wl), h1, o3);
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[cb][c] = fast_sigmoid(o3[c]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a1[rb] = mfma_h(frag_h(w_sem, rb, wl), bs, z4);
        h0 = chain_relu_h(a1[0], a1[1]);
        h1 = chain_relu_h(a1[2], a1[3]);
#pragma unroll
        for (int rb = 0; rb < NRB_SEM; ++rb) {
          lg[cb][rb] = mfma_h(frag_h(w_sem, 4 + 2 * rb, wl), h0, z4);
          lg[cb][rb] = mfma_h(frag_h(w_sem, 5 + 2 * rb, wl), h1, lg[cb][rb]);
        }
      }
    }

    // ---------------- softmax + contributions + per-ray sums -------------
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      float mx = -INFINITY;
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((uint32_t)(rb * 16 + 4 * g + r) < C) mx = fast_max(mx, lg[cb][rb][r]);
      mx = fast_max(mx, __shfl_xor(mx, 16, 64));
      mx = fast_max(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.0f;
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (uint32_t)(rb * 16 + 4 * g + r) < C;
          const float ex = ok ? fast_exp(lg[cb][rb][r] - mx) : 0.0f;
          lg[cb][rb][r] = ex;
          sum += ex;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float wgt = ew[cb];
      const float ws = wgt * fast_rcp(sum);
      float* crow = contrib + j * cstride;
      if (g == 0) {
        crow[0] = wgt * rgb[cb][0];
        crow[1] = wgt * rgb[cb][1];
        crow[2] = wgt * rgb[cb][2];
      }
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t cls = rb * 16 + 4 * g + r;
          if (cls < C) crow[3 + cls] = lg[cb][rb][r] * ws;
        }
      wave_lds_sync();
      const uint32_t nb = (n > (uint32_t)cb * 16) ? ((n - cb * 16 < 16) ? n - cb * 16 : 16) : 0;
      // Per-ray sums in sample order (the order makes results independent of
      // how rays are chunked / sharded).  All 16 tile rows and ray ids are
      // fetched first so the sequential adds do not each wait on an LDS read.
      float cv[16];
      uint32_t rv[16];
      const uint32_t ch = lane < 3 + C ? lane : 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        cv[e] = contrib[e * cstride + ch];
        rv[e] = lray[cb * 16 + e];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if ((uint32_t)e < nb) {
          const uint32_t ray = (uint32_t)__builtin_amdgcn_readfirstlane((int)rv[e]);
          if (ray != cur_ray) {
            if (cur_ray != 0xFFFFFFFFu) flush_ray(cur_ray);
            cur_ray = ray;
            acc = 0.0f;
          }
          acc = acc + cv[e];
        }
      }
      wave_lds_sync();
    }
  };

  // ---- C/D: shade the full groups waiting in the list -------------------
  auto drain = [&]() {
    uint32_t head = 0;
    while (cnt - head >= G) {
      // shade() reads entries [0,G): move the window down first if needed
      if (head) {
        for (uint32_t i = lane; i < G; i += 64) {
          lw[i] = lw[head + i];
          lrow[i] = lrow[head + i];
          lray[i] = lray[head + i];
        }
        wave_lds_sync();
      }
      shade(G);
      head += G;
    }
    if (head) {  // compact the tail [head, cnt) to the front
      const uint32_t rem = cnt - head;  // < G <= 64
      float tw = 0.f;
      uint32_t trow = 0, tray = 0;
      if (lane < rem) {
        tw = lw[head + lane];
        trow = lrow[head + lane];
        tray = lray[head + lane];
      }
      wave_lds_sync();
      if (lane < rem) {
        lw[lane] = tw;
        lrow[lane] = trow;
        lray[lane] = tray;
      }
      cnt = rem;
      wave_lds_sync();
    }
  };

  if constexpr (MARCH) {
    // ================= marched spans, streamed 64 points at a time ==========
    // Marched rays are short (tens of samples a round, ~12 in the first one):
    // a loop over rays leaves most lanes idle and pays a chain of dependent
    // loads per ray.  The spans of a wave's slots are CONTIGUOUS in the point
    // buffers (exclusive prefix sums in slot order), so the wave streams its
    // points 64 at a time -- sigma / deltas fully coalesced -- whatever rays
    // they belong to, with segmented scans: a lane that is the first sample of
    // its ray starts a segment; the ray cut by a chunk boundary continues from
    // wave-uniform carries.
    const uint32_t n_slots = r_end - r_begin;  // <= MARCH_RPW_MAX
    uint32_t p_end = 0;
    for (uint32_t i = lane; i < n_slots; i += 64) {
      const uint32_t r = r_begin + i;
      const uint32_t index = (uint32_t)a.rays_alive[(size_t)r * a.alive_stride];
      const uint32_t off = (uint32_t)a.span[(size_t)r * a.span_stride];
      uint32_t c = (uint32_t)a.span[(size_t)r * a.span_stride + 1];
      if (a.train && off + c >= a.n_points) c = 0;
      slot_off[i] = off;
      slot_cnt[i] = c;
      slot_idx[i] = index;
      slot_T0[i] = 1.0f - a.weights_sum[index];
      slot_t0[i] = a.rays_t[a.train ? index : r];
      if (c) p_end = off + c;  // offsets ascend: the last non-empty slot wins
      else if (!a.train) a.rays_t[r] = -1.0f;  // nothing marched: finished
    }
#pragma unroll
    for (int d2 = 32; d2 >= 1; d2 >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)p_end, d2, 64);
      p_end = o > p_end ? o : p_end;
    }
    wave_lds_sync();
    const uint32_t p_begin = slot_off[0];
    // carries of the ray cut by the previous chunk boundary (wave-uniform)
    float carry_T = 1.0f, carry_Tprev = 1.0f, carry_t = 0.0f, carry_sw = 0.0f,
          carry_sd = 0.0f;
    for (uint32_t p0 = p_begin; p0 < p_end; p0 += 64) {
      const uint32_t p = p0 + lane;
      const bool live = p < p_end;
      // slot of the point: last slot whose first point is <= p
      uint32_t lo = 0, hi = n_slots;
      const uint32_t pq = live ? p : p_end - 1;
      while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (slot_off[mid] <= pq) lo = mid; else hi = mid;
      }
      const uint32_t si = lo;
      const uint32_t off = slot_off[si], c = slot_cnt[si];
      const uint32_t index = slot_idx[si];
      const bool head = live && p == off;
      const bool tail = live && p == off + c - 1u;
      const size_t m = live ? p : p_end - 1;
      const float sg = a.sigma_c[m] * a.density_scale;
      const float2 dl = *reinterpret_cast<const float2*>(a.deltas + 2 * m);
      const float alpha = live ? 1.0f - __expf(-sg * dl.x) : 0.0f;
      const unsigned long long heads = __ballot(head);
      const uint32_t first_head =
          heads ? (uint32_t)__ffsll((long long)heads) - 1u : 64u;
      const bool cont = lane < first_head;  // part of the ray cut by the boundary
      const float Tin = seg_incl_scan_mul(1.0f - alpha, head, lane);
      float Tex = __shfl_up(Tin, 1, 64);
      if (head || lane == 0) Tex = 1.0f;
      const float Tr = (cont ? carry_T : slot_T0[si]) * Tex;
      // reference raymarching.cu:693-706: take the sample, then stop if its
      // incoming T < 1e-4 (T never grows: "some earlier sample saw T <= 1e-4"
      // is "the previous one did")
      float Tprev = __shfl_up(Tr, 1, 64);
      if (lane == 0) Tprev = carry_Tprev;
      const bool use = live && (a.train || head || !(Tprev <= 1e-4f));
      const float w = use ? alpha * Tr : 0.0f;
      const float tt = (cont ? carry_t : slot_t0[si]) +
                       seg_incl_scan_add(live ? dl.y : 0.0f, head, lane);
      const bool keep = use && (w > a.w_min);
      if (a.w_out && live) a.w_out[m] = w;
      if (a.t_out && live) a.t_out[m] = tt;
      // per-ray sums: weights over every used sample, depth over the kept ones
      const float sw = seg_incl_scan_add(w, head, lane) + (cont ? carry_sw : 0.0f);
      const float sd = seg_incl_scan_add(keep ? w * tt : 0.0f, head, lane) +
                       (cont ? carry_sd : 0.0f);
      if (tail) {
        a.weights_sum[index] += sw;
        a.depth[index] += sd;
        if (!a.train)
          a.rays_t[r_begin + si] = (Tr <= 1e-4f || c < a.seg_cap) ? -1.0f : tt;
      }
      const unsigned long long bal = __ballot(keep);
      if (keep) {
        const uint32_t pos =
            cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        lw[pos] = w;
        lrow[pos] = (uint32_t)m;
        lray[pos] = index;
      }
      cnt += (uint32_t)__popcll(bal);
      // the ray running past lane 63 (if any) continues in the next chunk
      carry_T = wave_bcast(Tr * (1.0f - alpha), 63);
      carry_Tprev = wave_bcast(Tr, 63);
      carry_t = wave_bcast(tt, 63);
      carry_sw = wave_bcast(sw, 63);
      carry_sd = wave_bcast(sd, 63);
      wave_lds_sync();
      drain();
    }
  } else {
  // ======================= per-ray loop ==================================
  for (uint32_t r = r_begin; r < r_end; ++r) {
    // ---- A1: load raw z (coarse then fine) ------------------------------
    const float* zc = a.z_c + (size_t)r * T;
    const float* zf = a.z_f + (size_t)r * t;
    for (uint32_t e = lane; e < S; e += 64) zraw[e] = e < T ? zc[e] : zf[e - T];
    wave_lds_sync();
    // ---- A2: rank of every element in the stable sort of [coarse|fine] --
    // Normal case: coarse depths ascend (near < far) and ucsa_resample emits
    // ascending fine samples, so both ranks are binary searches.  Anything
    // else (near clamped above far gives DESCENDING coarse depths; arbitrary
    // callers may pass unsorted z_f) takes the general path: stable rank by
    // counting over all S elements, exactly torch.sort(stable) semantics.
    bool sorted_in = true;
    for (uint32_t k = lane; k + 1 < T; k += 64)
      sorted_in = sorted_in && (zraw[k] <= zraw[k + 1]);
    for (uint32_t k = lane; k + 1 < t; k += 64)
      sorted_in = sorted_in && (zraw[T + k] <= zraw[T + k + 1]);
    sorted_in = __all(sorted_in);
    for (uint32_t e = lane; e < S; e += 64) {
      const float ze = zraw[e];
      uint32_t rank;
      if (!sorted_in) {
        uint32_t c = 0;
        for (uint32_t k = 0; k < S; ++k) {
          const float zk = zraw[k];
          c += (zk < ze || (zk == ze && k < e)) ? 1u : 0u;
        }
        rank = c;
      } else if (e < T) {
        uint32_t lo = 0, hi = t;  // #fine strictly below ze
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (zraw[T + mid] < ze) lo = mid + 1; else hi = mid;
        }
        rank = e + lo;
      } else {
        uint32_t lo = 0, hi = T;  // #coarse <= ze (coarse first on ties)
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (zraw[mid] <= ze) lo = mid + 1; else hi = mid;
        }
        rank = lo + (e - T);  // equal fine neighbours keep index order
      }
      const float sg = e < T ? a.sigma_c[(size_t)r * T + e]
                             : a.sigma_f[(size_t)r * t + (e - T)];
      zm[rank] = ze;
      sgm[rank] = sg;
      srcs[rank] = e;
    }
    wave_lds_sync();
    // ---- A3: weights, mask, depth, compaction ---------------------------
    float carry = 1.0f, dsum = 0.0f;
    uint32_t kept = 0;  // survivors of this ray (wave-uniform)
    for (uint32_t sbase = 0; sbase < S; sbase += 64) {
      const uint32_t s = sbase + lane;
      float alpha = 0.0f, zi = 0.0f;
      if (s < S) {
        zi = zm[s];
        const float delta = (s + 1 < S) ? zm[s + 1] - zi : 1e10f;
        alpha = 1.0f - expf(-delta * a.density_scale * sgm[s]);
      }
      const float fac = (s < S) ? (1.0f - alpha + 1e-15f) : 1.0f;
      const float incl = wave_incl_scan_mul_dpp(fac);
      const float excl = wave_shift_up1(incl, 1.0f);
      const float w = alpha * (carry * excl);
      carry = carry * wave_last(incl);
      const bool keep = (s < S) && (w > 1e-4f);
      if (keep) dsum += w * zi;
      if (s < S) {
        if (a.w_out) a.w_out[(size_t)r * S + s] = w;
        if (a.src_out) a.src_out[(size_t)r * S + s] = (int32_t)srcs[s];
      }
      const unsigned long long bal = __ballot(keep);
      if (keep) {
        const uint32_t pos =
            cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        const uint32_t e = srcs[s];
        lw[pos] = w;
        lrow[pos] = e < T ? (r * T + e) : (ROW_FINE | (r * t + (e - T)));
        lray[pos] = r;
      }
      cnt += (uint32_t)__popcll(bal);
      kept += (uint32_t)__popcll(bal);
      wave_lds_sync();
      drain();
    }
    dsum = wave_sum(dsum);
    if (lane == 0) a.depth[r] = dsum / a.norms[r];
    if (kept == 0) {  // nothing survived the mask: all-zero outputs
      if (lane < 3) a.image[(size_t)r * 3 + lane] = 0.0f;
      else if (lane < 3 + C) a.semantics[(size_t)r * C + (lane - 3)] = 0.0f;
    }
  }
  }
  if (cnt) shade(cnt);
  if (cur_ray != 0xFFFFFFFFu) flush_ray(cur_ray);
}

static inline uint32_t cmp_pad16(uint32_t n) { return (n + 15u) / 16u * 16u; }

static int32_t composite_launch(
    bool half, const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const float* packed_color,
    const float* packed_sem, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, int32_t* src, float* weights, void* stream) {
  UCSA_CHECK_ARG(rays_d, 0);
  UCSA_CHECK_ARG(norms, 1);
  UCSA_CHECK_ARG(z_c && sigma_c && h_c, 2);
  UCSA_CHECK_ARG(t == 0 || (z_f && sigma_f && h_f), 5);
  UCSA_CHECK_ARG(packed_color && packed_sem, 8);
  UCSA_CHECK_ARG(T >= 1 && (uint64_t)N * T < 0x80000000ull, 11);
  UCSA_CHECK_ARG((uint64_t)N * t < 0x80000000ull && T + t <= 8192, 12);
  UCSA_CHECK_ARG(n_classes >= 1 && n_classes <= 61, 13);
  UCSA_CHECK_ARG(image && depth && semantics, 15);
  if (N == 0) return 0;
  const uint32_t S = T + t;
  const uint32_t nrb = cmp_pad16(n_classes) / 16;
  uint32_t cstride = 3 + n_classes;
  if ((cstride & 1u) == 0) cstride += 1;  // odd stride: conflict-free rows
  const size_t w_floats = half ? (size_t)(COLOR_H_FRAGS + SEM_H_FRAGS(nrb)) * 256
                               : 7168 + 1024 + (size_t)nrb * 1024;
  const uint32_t cbs = half ? CMP_CBS_H : CMP_CBS;
  const size_t per_wave = 4 * (size_t)S + 3 * (size_t)(64 + 16 * cbs) + 16 * cstride + 64;
  // as many waves per workgroup as fit in LDS (one workgroup per CU)
  uint32_t waves = CMP_MAX_WAVES;
  while (waves > 1 && (w_floats + waves * per_wave) * 4 > 158 * 1024) --waves;
  const size_t smem = (w_floats + waves * per_wave) * 4;
  UCSA_CHECK_ARG(smem <= 160 * 1024, 12);
  // rays per wave: spread over the chip, with enough rays per wave that the
  // survivor groups stay dense across ray boundaries ...
  const uint64_t total_waves = 256ull * waves;
  uint32_t rpw = (uint32_t)((N + total_waves - 1) / total_waves);
  // ... but not more than needed for that: a ray of S >= 128 samples fills
  // its groups of 32 by itself.  With the old minimum of 4 a training batch
  // (4096 rays x 512 samples) ran on 114 workgroups -- less than half the
  // chip -- and its forward composite took 0.95 instead of 0.60 ms.
  const uint32_t rpw_min = S >= 128 ? 1u : (S >= 64 ? 2u : 4u);
  if (rpw < rpw_min) rpw = rpw_min;
  const uint32_t n_waves = ucsa_div_up(N, rpw);
  const uint32_t blocks = ucsa_div_up(n_waves, waves);
  CmpArgs a{rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f,
            packed_color, packed_sem, N, T, t, n_classes, density_scale,
            image, depth, semantics, src, weights, rpw, cstride,
            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0.0f,
            1u, 2u, 0u, 0u, nullptr};
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(NRB, H)                                                        \
  do {                                                                        \
    constexpr int CB = H ? CMP_CBS_H : CMP_CBS;                               \
    hipError_t e = hipFuncSetAttribute(                                       \
        reinterpret_cast<const void*>(&k_composite<NRB, CB, H, false>),              \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);               \
    if (e != hipSuccess) return -(int32_t)e;                                  \
    UCSA_CLEAR_ERR();                                                         \
    hipLaunchKernelGGL((k_composite<NRB, CB, H, false>), dim3(blocks),               \
                       dim3(64 * waves), smem, s, a);                         \
  } while (0)
  if (half) {
    switch (nrb) {
      case 1: LAUNCH(1, true); break;
      case 2: LAUNCH(2, true); break;
      case 3: LAUNCH(3, true); break;
      default: LAUNCH(4, true); break;
    }
  } else {
    switch (nrb) {
      case 1: LAUNCH(1, false); break;
      case 2: LAUNCH(2, false); break;
      case 3: LAUNCH(3, false); break;
      default: LAUNCH(4, false); break;
    }
  }
#undef LAUNCH
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_composite_fwd(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const float* packed_color,
    const float* packed_sem, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, int32_t* src, float* weights, void* stream) {
  return composite_launch(false, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f,
                          h_f, packed_color, packed_sem, N, T, t, n_classes,
                          density_scale, image, depth, semantics, src, weights,
                          stream);
}

extern "C" int32_t ucsa_composite_fwd_f16(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_half,
    const void* packed_sem_half, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, int32_t* src, float* weights, void* stream) {
  return composite_launch(true, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f,
                          h_f, (const float*)packed_color_half,
                          (const float*)packed_sem_half, N, T, t, n_classes,
                          density_scale, image, depth, semantics, src, weights,
                          stream);
}

// ---------------------------------------------------------------------------
// Marched spans: weights with early stop -> ballot compaction of the samples
// with w > w_min -> colour + semantics nets on the survivors -> per-ray sums
// added in place.  One kernel instead of point_shade + segment_composite, and
// rgb / class probabilities never travel through HBM.
// ---------------------------------------------------------------------------
static int32_t march_shade_launch(
    int half /* 0 f32-input MFMA, 1 f16, 2 f16x2 */, bool train, uint32_t n_points, float* w_out, float* t_out,
    uint32_t n_cap,
    const int32_t* n_alive_dev, uint32_t cap, const int32_t* rays_alive,
    uint32_t alive_stride, float* rays_t, const int32_t* span,
    uint32_t span_stride,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const float* packed_color, const float* packed_sem,
    uint32_t n_classes, float w_min, float* weights_sum, float* depth,
    float* image, float* semantics, void* stream) {
  UCSA_CHECK_ARG(cap >= 1, 2);
  UCSA_CHECK_ARG(rays_alive, 3);
  UCSA_CHECK_ARG(rays_t, 4);
  UCSA_CHECK_ARG(span, 5);
  UCSA_CHECK_ARG(rays_d, 6);
  UCSA_CHECK_ARG(sigmas, 7);
  UCSA_CHECK_ARG(h, 9);
  UCSA_CHECK_ARG(deltas, 10);
  UCSA_CHECK_ARG(packed_color && packed_sem, 11);
  UCSA_CHECK_ARG(n_classes >= 1 && n_classes <= 61, 13);
  UCSA_CHECK_ARG(w_min >= 0.f, 14);
  UCSA_CHECK_ARG(weights_sum && depth && image && semantics, 15);
  if (n_cap == 0) return 0;
  const uint32_t nrb = cmp_pad16(n_classes) / 16;
  uint32_t cstride = 3 + n_classes;
  if ((cstride & 1u) == 0) cstride += 1;
  const size_t w_floats = half ? (size_t)(COLOR_H_FRAGS + SEM_H_FRAGS(nrb)) * 256 * (half == 2 ? 2 : 1)
                               : 7168 + 1024 + (size_t)nrb * 1024;
  const uint32_t cbs = half ? CMP_CBS_H : CMP_CBS;
  const size_t per_wave =
      3 * (size_t)(64 + 16 * cbs) + 16 * cstride + 64 + MARCH_SLOT_FLOATS;
  const uint32_t waves = half == 2 ? CMP_H2_WAVES : CMP_MAX_WAVES;
  const size_t smem = (w_floats + waves * per_wave) * 4;
  const uint64_t total_waves = 256ull * waves;
  uint32_t rpw = (uint32_t)((n_cap + total_waves - 1) / total_waves);
  // at least 2 slots per wave (marched spans are short: groups of 32 fill
  // across slots); 4 left a 4096-ray training batch on 86 workgroups
  // (measured 1.53 -> 1.47 ms per marched step with 2, 1.51 with 1)
  if (rpw < 2) rpw = 2;
  if (rpw > MARCH_RPW_MAX) rpw = MARCH_RPW_MAX;  // slot tables live in LDS
  const uint32_t blocks = ucsa_div_up(ucsa_div_up(n_cap, rpw), waves);
  CmpArgs a{rays_d, nullptr, nullptr, sigmas, h, nullptr, nullptr, nullptr,
            packed_color, packed_sem, n_cap, 0u, 0u, n_classes, sigma_scale,
            image, depth, semantics, nullptr, w_out, rpw, cstride,
            rays_alive, rays_t, span, deltas, n_alive_dev, weights_sum, cap,
            w_min, alive_stride, span_stride, train ? 1u : 0u, n_points,
            t_out};
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_M(NRB, H)                                                      \
  do {                                                                        \
    constexpr int CB = H ? CMP_CBS_H : CMP_CBS;                               \
    hipError_t e = hipFuncSetAttribute(                                       \
        reinterpret_cast<const void*>(&k_composite<NRB, CB, H, true>),        \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);               \
    if (e != hipSuccess) return -(int32_t)e;                                  \
    UCSA_CLEAR_ERR();                                                         \
    hipLaunchKernelGGL((k_composite<NRB, CB, H, true>), dim3(blocks),         \
                       dim3(64 * waves), smem, s, a);                         \
  } while (0)
#define LAUNCH_M2(NRB)                                                        \
  do {                                                                        \
    hipError_t e = hipFuncSetAttribute(                                       \
        reinterpret_cast<const void*>(&k_composite<NRB, CMP_CBS_H, true, true, true>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);               \
    if (e != hipSuccess) return -(int32_t)e;                                  \
    UCSA_CLEAR_ERR();                                                         \
    hipLaunchKernelGGL((k_composite<NRB, CMP_CBS_H, true, true, true>),       \
                       dim3(blocks), dim3(64 * waves), smem, s, a);           \
  } while (0)
  if (half == 2) {
    switch (nrb) {
      case 1: LAUNCH_M2(1); break;
      case 2: LAUNCH_M2(2); break;
      case 3: LAUNCH_M2(3); break;
      default: LAUNCH_M2(4); break;
    }
  } else if (half) {
    switch (nrb) {
      case 1: LAUNCH_M(1, true); break;
      case 2: LAUNCH_M(2, true); break;
      case 3: LAUNCH_M(3, true); break;
      default: LAUNCH_M(4, true); break;
    }
  } else {
    switch (nrb) {
      case 1: LAUNCH_M(1, false); break;
      case 2: LAUNCH_M(2, false); break;
      case 3: LAUNCH_M(3, false); break;
      default: LAUNCH_M(4, false); break;
    }
  }
#undef LAUNCH_M2
#undef LAUNCH_M
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_march_segment_shade(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, float* rays_t, const int32_t* span,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const float* packed_color, const float* packed_sem,
    uint32_t n_classes, float w_min, float* weights_sum, float* depth,
    float* image, float* semantics, void* stream) {
  return march_shade_launch(false, false, 0u, nullptr, nullptr, n_cap,
                            n_alive_dev, cap,
                            rays_alive, 1u, rays_t, span, 2u, rays_d, sigmas,
                            sigma_scale, h, deltas,
                            packed_color, packed_sem, n_classes, w_min,
                            weights_sum, depth, image, semantics, stream);
}

extern "C" int32_t ucsa_march_segment_shade_f16(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, float* rays_t, const int32_t* span,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const void* packed_color_half,
    const void* packed_sem_half, uint32_t n_classes, float w_min,
    float* weights_sum, float* depth, float* image, float* semantics,
    void* stream) {
  return march_shade_launch(true, false, 0u, nullptr, nullptr, n_cap,
                            n_alive_dev, cap,
                            rays_alive, 1u, rays_t, span, 2u, rays_d, sigmas,
                            sigma_scale, h, deltas,
                            (const float*)packed_color_half,
                            (const float*)packed_sem_half, n_classes, w_min,
                            weights_sum, depth, image, semantics, stream);
}

extern "C" int32_t ucsa_march_segment_shade_h2(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, float* rays_t, const int32_t* span,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const void* packed_color_h2, const void* packed_sem_h2,
    uint32_t n_classes, float w_min, float* weights_sum, float* depth, float* image,
    float* semantics, void* stream) {
  return march_shade_launch(2, false, 0u, nullptr, nullptr, n_cap,
                            n_alive_dev, cap,
                            rays_alive, 1u, rays_t, span, 2u, rays_d, sigmas,
                            sigma_scale, h, deltas,
                            (const float*)packed_color_h2,
                            (const float*)packed_sem_h2, n_classes, w_min,
                            weights_sum, depth, image, semantics, stream);
}

// Training forward over the output of ucsa_march_rays_train: rays [N,3] =
// (ray id, first point, count).  Outputs by ray id; the caller zero-fills
// weights_sum / depth / image / semantics and w_out [M].
extern "C" int32_t ucsa_march_train_fwd(
    const int32_t* rays, uint32_t N, uint32_t M, const float* nears,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const float* packed_color, const float* packed_sem,
    uint32_t n_classes, float w_min, float* weights_sum, float* depth,
    float* image, float* semantics, float* w_out, float* t_out,
    void* stream) {
  UCSA_CHECK_ARG(rays, 0);
  UCSA_CHECK_ARG(nears, 3);
  UCSA_CHECK_ARG(rays_d, 4);
  UCSA_CHECK_ARG(M == 0 || (sigmas && h && deltas), 5);
  UCSA_CHECK_ARG(packed_color && packed_sem, 9);
  UCSA_CHECK_ARG(n_classes >= 1 && n_classes <= 61, 11);
  UCSA_CHECK_ARG(w_min >= 0.f, 12);
  UCSA_CHECK_ARG(weights_sum && depth && image && semantics, 13);
  UCSA_CHECK_ARG(w_out && t_out, 17);
  if (M == 0 || N == 0) return 0;
  return march_shade_launch(false, true, M, w_out, t_out, N, nullptr, 1u, rays,
                            3u,
                            const_cast<float*>(nears), rays + 1, 3u, rays_d,
                            sigmas, sigma_scale, h, deltas, packed_color,
                            packed_sem, n_classes, w_min, weights_sum, depth,
                            image, semantics, stream);
}
