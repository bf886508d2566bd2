// Inference composite as TWO dense kernels (SURVEY 8a rows a5 second half,
// a6-a9; reference nr4seg/nerf/renderer_semantics.py:220-299 and
// nr4seg/nerf/network_tcnn_semantics.py:147-207).
//
// Why split.  k_composite (composite.hip) runs per-ray sampling logic and the
// MFMA shading in ONE wave: every group of 32 survivors waits on a chain of
// dependent latencies (LDS list -> global h row -> three dependent MFMA layers
// -> shuffles -> LDS tile -> ordered sums) with 3 waves per SIMD to hide them.
// Measured (round 1): in fp16 mode the MFMA work falls 13x and the kernel only
// 25-35 %: ~30 k cycles per wave and group, almost all of it waiting.  Here:
//
//  k_weights_compact  one wave per RAY, no MFMA state: merge of coarse+fine
//      depths (binary-search ranks), alpha / transmittance by wave scan, mask
//      w > 1e-4, depth, ballot+popcount compaction of the survivors into the
//      ray's own region of a global list (w, row, ray) -- every access
//      coalesced; also the 16 SH values of the ray's direction, once per ray.
//  k_shade_dense      each wave owns a range of rays whose survivors form one
//      virtual list; it walks that list G entries at a time, REQUESTS the next
//      group's entries / h rows / SH values before shading the current one
//      (software pipeline), runs the colour + semantics nets -- PREC 0: f32
//      16x16x4 MFMA (vector-ALU work on gfx950), 1: f16 16x16x32 MFMA, 2:
//      bf16x3 (six bf16 16x16x32 MFMAs per product, fp32-grade,
//      mfma_mlp_x3.h) -- and sums w*rgb, w*p per ray in sample order through a
//      16-row LDS tile.  The kernel is bound by instruction issue (MFMA cycles
//      + VALU cycles): every VALU instruction in the loop counts.
//
// The list costs 8 B per survivor each way (<= 0.1 ms per 61 440-ray chunk at
// HBM rates).  The arithmetic -- k order of every contraction, expf vs
// v_exp_f32, order of the per-ray sums -- is the fused kernel's, statement
// for statement, so both paths give the same bits (tested).
#include <cstdlib>

#include "composite_common.h"
#include "mfma_mlp_x3.h"

extern __shared__ __attribute__((aligned(16))) float cs_smem[];

struct WcArgs {
  const float* rays_d;
  const float* norms;
  const float* z_c;
  const float* sigma_c;
  const float* z_f;
  const float* sigma_f;
  uint32_t N, T, t;
  float density_scale;
  float* depth;
  float* list_w;       // [N*S]; wave w's entries start at w * rays_per_wave * S
  uint32_t* list_row;  // [N*S]
  uint32_t* list_ray;  // [N*S]
  uint32_t* counts;    // [number of waves]
  uint32_t rays_per_wave;
  float* image;        // rays without a survivor get zeros here
  float* semantics;
  uint32_t C;
  float* sh;           // [N,16] SH basis of the ray direction (null: not wanted)
};

#define WC_WAVES 4

__global__ void __launch_bounds__(64 * WC_WAVES) k_weights_compact(WcArgs a) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t T = a.T, t = a.t, S = a.T + a.t;
  const uint32_t gwave = blockIdx.x * WC_WAVES + wid;
  const uint64_t r_begin64 = (uint64_t)gwave * a.rays_per_wave;
  if (r_begin64 >= a.N) return;
  const uint32_t r_begin = (uint32_t)r_begin64;
  const uint32_t r_end = (r_begin + a.rays_per_wave < a.N)
                             ? r_begin + a.rays_per_wave : a.N;
  float* zraw = cs_smem + (size_t)wid * 4 * S;
  float* zm = zraw + S;
  float* sgm = zm + S;
  uint32_t* srcs = reinterpret_cast<uint32_t*>(sgm + S);
  const size_t base = (size_t)r_begin * S;
  uint32_t cnt = 0;  // entries written by this wave so far (wave-uniform)
  for (uint32_t r = r_begin; r < r_end; ++r) {
    // ---- A1: raw depths (coarse then fine) -------------------------------
    const float* zc = a.z_c + (size_t)r * T;
    const float* zf = a.z_f + (size_t)r * t;
    for (uint32_t e = lane; e < S; e += 64) zraw[e] = e < T ? zc[e] : zf[e - T];
    wave_lds_sync();
    // ---- A2: rank in the stable sort of [coarse|fine] (composite.hip A2) --
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
        rank = lo + (e - T);
      }
      const float sg = e < T ? a.sigma_c[(size_t)r * T + e]
                             : a.sigma_f[(size_t)r * t + (e - T)];
      zm[rank] = ze;
      sgm[rank] = sg;
      srcs[rank] = e;
    }
    wave_lds_sync();
    // ---- A3: weights, mask, depth, compaction -----------------------------
    float carry = 1.0f, dsum = 0.0f;
    uint32_t kept = 0;
    for (uint32_t sbase = 0; sbase < S; sbase += 64) {
      const uint32_t s = sbase + lane;
      float alpha = 0.0f, zi = 0.0f;
      if (s < S) {
        zi = zm[s];
        const float delta = (s + 1 < S) ? zm[s + 1] - zi : 1e10f;
        alpha = 1.0f - expf(-delta * a.density_scale * sgm[s]);
      }
      const float fac = (s < S) ? (1.0f - alpha + 1e-15f) : 1.0f;
      const float incl = wave_incl_scan_mul(fac, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.0f;
      const float w = alpha * (carry * excl);
      carry = carry * wave_bcast(incl, 63);
      const bool keep = (s < S) && (w > 1e-4f);
      if (keep) dsum += w * zi;
      const unsigned long long bal = __ballot(keep);
      if (keep) {
        const size_t pos =
            base + cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        const uint32_t e = srcs[s];
        a.list_w[pos] = w;
        a.list_row[pos] = e < T ? (r * T + e) : (ROW_FINE | (r * t + (e - T)));
        a.list_ray[pos] = r;
      }
      cnt += (uint32_t)__popcll(bal);
      kept += (uint32_t)__popcll(bal);
    }
    dsum = wave_sum(dsum);
    if (lane == 0) a.depth[r] = dsum / a.norms[r];
    if (a.sh && lane < 16) {  // the 16 SH values of the ray, once per ray
      const float* dd = a.rays_d + (size_t)r * 3;
      float sh[4];
      sh4_select(dd[0], dd[1], dd[2], lane >> 2, sh);
      const uint32_t q = lane & 3u;
      a.sh[(size_t)r * 16 + lane] = q == 0 ? sh[0] : (q == 1 ? sh[1] : (q == 2 ? sh[2] : sh[3]));
    }
    if (kept == 0) {  // nothing survived the mask: all-zero outputs
      if (lane < 3) a.image[(size_t)r * 3 + lane] = 0.0f;
      else if (lane < 3 + a.C) a.semantics[(size_t)r * a.C + (lane - 3)] = 0.0f;
    }
    wave_lds_sync();  // the next ray overwrites zraw / zm / sgm / srcs
  }
  if (lane == 0) a.counts[gwave] = cnt;
}

struct ShArgs {
  const float* rays_d;
  const float* h_c;
  const float* h_f;
  const float* packed_color;
  const float* packed_sem;
  const float* list_w;
  const uint32_t* list_row;
  const uint32_t* list_ray;
  const uint32_t* counts;
  uint32_t N, S, C;
  uint32_t rays_per_wave;
  float* image;
  float* semantics;
  const float* sh;  // [N,16] from k_weights_compact (16-bit MFMA modes)
};

template <int CBS>
struct Ent {  // one group's list entries, requested two groups ahead
  float w[CBS];
  uint32_t row[CBS];
  uint32_t ray[CBS];
};

template <int CBS>
struct Pre {  // one group's operands, requested one group ahead
  float ew[CBS];
  uint32_t eray[CBS];
  f32x4 hv[CBS];
  float d[CBS][3];  // f32-input MFMA mode: the ray direction
  f32x4 shv[CBS];   // 16-bit MFMA modes: this lane's four SH values of the ray
};

// PREC: 0 = f32-input MFMA, 1 = f16 MFMA (tiny-cuda-nn's numerics), 2 = bf16x3
// (fp32-grade on the bf16 MFMA pipe, mfma_mlp_x3.h)
template <int NRB_SEM, int CBS, int PREC, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_shade_dense(ShArgs a) {
  constexpr bool HALF = PREC == 1;
  constexpr uint32_t G = 16u * CBS;
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint32_t S = a.S, C = a.C;
  constexpr uint32_t CSTRIDE = 16 * NRB_SEM + 4;

  constexpr uint32_t WC_FLOATS = PREC == 2   ? COLOR_H_FRAGS * 768
                                 : PREC == 1 ? COLOR_H_FRAGS * 256
                                             : 7168;
  constexpr uint32_t WS_FLOATS = PREC == 2   ? SEM_H_FRAGS(NRB_SEM) * 768
                                 : PREC == 1 ? SEM_H_FRAGS(NRB_SEM) * 256
                                             : 1024 + NRB_SEM * 1024;
  float* w_color = cs_smem;
  float* w_sem = w_color + WC_FLOATS;
  constexpr uint32_t per_wave_floats = 16 * CSTRIDE + 64;
  float* base = w_sem + WS_FLOATS + (size_t)wid * per_wave_floats;
  float* contrib = base;                      // [16][CSTRIDE]
  float* shpart = contrib + 16 * CSTRIDE;     // [64] colour-L1 SH part of a ray

  for (uint32_t i = threadIdx.x; i < WC_FLOATS; i += blockDim.x)
    w_color[i] = a.packed_color[i];
  for (uint32_t i = threadIdx.x; i < WS_FLOATS; i += blockDim.x)
    w_sem[i] = a.packed_sem[i];
  __syncthreads();

  const uint64_t gwave = (uint64_t)blockIdx.x * WAVES + wid;
  const uint64_t r_begin64 = gwave * a.rays_per_wave;
  if (r_begin64 >= a.N) return;
  const uint32_t total =
      (uint32_t)__builtin_amdgcn_readfirstlane((int)a.counts[gwave]);
  if (total == 0) return;
  const size_t lbase = (size_t)r_begin64 * S;

  uint32_t cur_ray = 0xFFFFFFFFu;  // ray whose sums sit in `acc`
  uint32_t sh_ray = 0xFFFFFFFFu;   // ray whose SH part sits in `shpart`
  float acc = 0.0f;                // lane c: running sum of channel c

  // lane c < C sums class c, lanes C..C+2 sum r, g, b
  const uint32_t choff = lane < C ? lane : (lane < C + 3 ? 16 * NRB_SEM + (lane - C) : 0u);
  auto flush_ray = [&](uint32_t ray) {
    if (lane < C) a.semantics[(size_t)ray * C + lane] = acc;
    else if (lane < C + 3) a.image[(size_t)ray * 3 + (lane - C)] = acc;
  };

  // stage 1: the list entries of the group starting at gb (entries past
  // `total` are padded with the last real one at weight 0)
  auto load_entries = [&](uint32_t gb, Ent<CBS>& en) {
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      uint32_t e = gb + cb * 16 + j;
      const bool live = e < total;
      if (!live) e = total - 1;
      const float w = a.list_w[lbase + e];
      en.w[cb] = live ? w : 0.0f;
      en.row[cb] = a.list_row[lbase + e];
      en.ray[cb] = a.list_ray[lbase + e];
    }
  };
  // stage 2: h rows and ray directions of entries that have arrived
  auto load_operands = [&](const Ent<CBS>& en, Pre<CBS>& p) {
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      p.ew[cb] = en.w[cb];
      p.eray[cb] = en.ray[cb];
      const uint32_t row = en.row[cb];
      const float* hp = ((row & ROW_FINE) ? a.h_f : a.h_c) +
                        (size_t)(row & ~ROW_FINE) * 16 + 4 * g;
      p.hv[cb] = *reinterpret_cast<const f32x4*>(hp);
      if constexpr (PREC == 0) {
        const float* dd = a.rays_d + (size_t)en.ray[cb] * 3;
        p.d[cb][0] = dd[0];
        p.d[cb][1] = dd[1];
        p.d[cb][2] = dd[2];
      } else {
        p.shv[cb] = *reinterpret_cast<const f32x4*>(
            a.sh + (size_t)en.ray[cb] * 16 + 4 * g);
      }
    }
  };

  // ---- nets + softmax + ordered per-ray sums on one fetched group ---------
  auto shade = [&](const Pre<CBS>& p, uint32_t n) {
    float geo[CBS][4];
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      geo[cb][0] = (g == 0) ? 1.0f : p.hv[cb][0];  // slot m==0 -> the "ones" pad
      geo[cb][1] = p.hv[cb][1];
      geo[cb][2] = p.hv[cb][2];
      geo[cb][3] = p.hv[cb][3];
    }
    float rgb[CBS][3];
    f32x4 lg[CBS][NRB_SEM];
    if constexpr (PREC == 2) {
      // six bf16 partial products per fp32 product: 144 MFMAs (16 cycles
      // each) per column block instead of 176 f32-input ones (32 cycles).
      // Layer-major: a weight fragment (three terms, 3 ds_read_b128) is read
      // once per group and used by all its column blocks.
      uint32_t zoff = 0;
      asm volatile("" : "+v"(zoff));  // keep the 72 fragments in LDS
      const uint32_t wl = lane + zoff;
      const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
      X3 b1[CBS];
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        split_pair(p.shv[cb][0], p.shv[cb][1], b1[cb], 0);
        split_pair(p.shv[cb][2], p.shv[cb][3], b1[cb], 1);
        split_pair(geo[cb][0], geo[cb][1], b1[cb], 2);
        split_pair(geo[cb][2], geo[cb][3], b1[cb], 3);
      }
      f32x4 a1[CBS][4], a2[CBS][4];
      X3 h0[CBS], h1[CBS];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const W3 w = frag_x3(w_color, rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) a1[cb][rb] = mfma_x3(w, b1[cb], z4);
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        h0[cb] = chain_relu_x3(a1[cb][0], a1[cb][1]);
        h1[cb] = chain_relu_x3(a1[cb][2], a1[cb][3]);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const W3 wa = frag_x3(w_color, 4 + 2 * rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) a2[cb][rb] = mfma_x3(wa, h0[cb], z4);
        const W3 wb = frag_x3(w_color, 5 + 2 * rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) a2[cb][rb] = mfma_x3(wb, h1[cb], a2[cb][rb]);
      }
      // semantics L1 reads the h-row slots of b1: its fragments carry zeros
      // in the SH slots' place (k-slots e >= 4 of the f16 layout <-> e < 4)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const W3 w = frag_x3(w_sem, rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) {
          X3 bs;
#pragma unroll
          for (int term = 0; term < 3; ++term)
            bs.t[term] = u32x4{b1[cb].t[term][2], b1[cb].t[term][3], 0u, 0u};
          a1[cb][rb] = mfma_x3(w, bs, z4);
        }
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        h0[cb] = chain_relu_x3(a2[cb][0], a2[cb][1]);
        h1[cb] = chain_relu_x3(a2[cb][2], a2[cb][3]);
      }
      {
        const W3 wa = frag_x3(w_color, 12, wl), wb = frag_x3(w_color, 13, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) {
          f32x4 o3 = mfma_x3(wa, h0[cb], z4);
          o3 = mfma_x3(wb, h1[cb], o3);
#pragma unroll
          for (int c = 0; c < 3; ++c) rgb[cb][c] = fast_sigmoid(o3[c]);
        }
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        h0[cb] = chain_relu_x3(a1[cb][0], a1[cb][1]);
        h1[cb] = chain_relu_x3(a1[cb][2], a1[cb][3]);
      }
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb) {
        const W3 wa = frag_x3(w_sem, 4 + 2 * rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) lg[cb][rb] = mfma_x3(wa, h0[cb], z4);
        const W3 wb = frag_x3(w_sem, 5 + 2 * rb, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) lg[cb][rb] = mfma_x3(wb, h1[cb], lg[cb][rb]);
      }
    } else if constexpr (!HALF) {
      {  // colour net 32 -> 64 -> 64 -> 16 (composite.hip, same k order)
        f32x4 acc1[CBS][4];
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) {
          const uint32_t ray0 =
              (uint32_t)__builtin_amdgcn_readfirstlane((int)p.eray[cb]);
          const bool uniform = __all(p.eray[cb] == ray0);
          float sh[4];
          if (!uniform || ray0 != sh_ray)
            sh4_select(p.d[cb][0], p.d[cb][1], p.d[cb][2], g, sh);
          if (uniform && ray0 != sh_ray) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
              f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int ks = 0; ks < 4; ++ks)
                q = mfma16(w_color[(rb * 8 + ks) * 64 + lane], sh[ks], q);
              if (j == 0) *reinterpret_cast<f32x4*>(shpart + 16 * rb + 4 * g) = q;
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
              f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int ks = 0; ks < 4; ++ks)
                q = mfma16(w_color[(rb * 8 + ks) * 64 + lane], sh[ks], q);
              acc1[cb][rb] = q;
            }
          }
        }
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
            rgb[cb][c] = fast_sigmoid(o3[cb][c]);
      }
      {  // semantics net 16 -> 64 -> 16*NRB_SEM
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
    } else {
      // fp16 weights / layer inputs, fp32 accumulate: 24 MFMAs per column block.
      // The 24 A fragments (96 VGPRs) are loop-invariant, and the compiler
      // would keep them in registers for the whole kernel -- 216 VGPRs, two
      // waves per SIMD, or spills under a smaller budget.  Reading them from
      // LDS per group costs 24 ds_read_b128 (shared by the group's column
      // blocks) and lets four waves per SIMD hide this kernel's latency
      // chains; `zoff` (an opaque zero, re-made per group) keeps the loads
      // inside the loop.
      uint32_t zoff = 0;
      if (WAVES > 8) asm volatile("" : "+v"(zoff));
      const uint32_t wl = lane + zoff;
      const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        half8 b1, bs;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          b1[r] = (_Float16)p.shv[cb][r];
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

    // softmax + contributions + per-ray sums in sample order.  Row of entry j
    // in the LDS tile: [16*NRB_SEM class slots | r g b | pad], CSTRIDE floats
    // (an odd number of 16-byte chunks: the b128 stores of 8 lanes land in
    // distinct bank groups); padded classes carry zeros.
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
      const float wgt = p.ew[cb];
      const float ws = wgt * fast_rcp(sum);
      float* crow = contrib + j * CSTRIDE;
      if (g == 0) {
        crow[16 * NRB_SEM + 0] = wgt * rgb[cb][0];
        crow[16 * NRB_SEM + 1] = wgt * rgb[cb][1];
        crow[16 * NRB_SEM + 2] = wgt * rgb[cb][2];
      }
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = lg[cb][rb][r] * ws;
        *reinterpret_cast<f32x4*>(crow + rb * 16 + 4 * g) = v;
      }
      wave_lds_sync();
      const uint32_t nb =
          (n > (uint32_t)cb * 16) ? ((n - cb * 16 < 16) ? n - cb * 16 : 16) : 0;
      float cv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) cv[e] = contrib[e * CSTRIDE + choff];
      // the ray of entry e of this column block: lane e (g == 0) holds it; the
      // list is sorted by ray, so first == last means one ray for all 16
      const uint32_t ray_a = (uint32_t)__builtin_amdgcn_readlane((int)p.eray[cb], 0);
      const uint32_t ray_z = (uint32_t)__builtin_amdgcn_readlane((int)p.eray[cb], 15);
      if (nb == 16 && ray_a == ray_z) {
        if (ray_a != cur_ray) {
          if (cur_ray != 0xFFFFFFFFu) flush_ray(cur_ray);
          cur_ray = ray_a;
          acc = 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = acc + cv[e];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if ((uint32_t)e < nb) {
            const uint32_t ray =
                (uint32_t)__builtin_amdgcn_readlane((int)p.eray[cb], e);
            if (ray != cur_ray) {
              if (cur_ray != 0xFFFFFFFFu) flush_ray(cur_ray);
              cur_ray = ray;
              acc = 0.0f;
            }
            acc = acc + cv[e];
          }
        }
      }
      wave_lds_sync();
    }
  };

  // three-stage software pipeline: entries two groups ahead, operands one
  // group ahead, so neither level of the dependent loads (list -> h row)
  // stalls the in-order instruction stream of the wave
  // (two groups per trip, the two register sets alternating: no copies)
  Ent<CBS> e1, e2;
  Pre<CBS> p0, p1;
  load_entries(0u, e1);
  load_operands(e1, p0);
  load_entries(G, e1);
  for (uint32_t gb = 0; gb < total; gb += 2 * G) {
    load_entries(gb + 2 * G, e2);
    load_operands(e1, p1);
    shade(p0, (total - gb < G) ? total - gb : G);
    if (gb + G >= total) break;
    load_entries(gb + 3 * G, e1);
    load_operands(e2, p0);
    shade(p1, (total - gb - G < G) ? total - gb - G : G);
  }
  if (cur_ray != 0xFFFFFFFFu) flush_ray(cur_ray);
}

static inline uint32_t cs_pad16(uint32_t n) { return (n + 15u) / 16u * 16u; }

extern "C" uint64_t ucsa_composite_infer_workspace_bytes(uint32_t N, uint32_t T,
                                                         uint32_t t) {
  const uint64_t S = (uint64_t)T + t;
  return (((uint64_t)N * S * 12 + 255) & ~255ull) + (((uint64_t)N * 4 + 255) & ~255ull) +
         (uint64_t)N * 64;
}

template <int NRB, int CBS, int H, int WAVES>
static int32_t launch_shade(const ShArgs& a, uint32_t blocks, size_t smem,
                            hipStream_t s) {
  hipError_t e = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&k_shade_dense<NRB, CBS, H, WAVES>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return -(int32_t)e;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL((k_shade_dense<NRB, CBS, H, WAVES>), dim3(blocks),
                     dim3(64 * WAVES), smem, s, a);
  return ucsa_launch_status();
}

// Launch shapes (waves per workgroup, column blocks per group).  The register
// budget follows from the workgroup size: 16 waves -> 128 VGPRs, 12 -> 168,
// 8 -> 256.  UCSA_SHADE_VARIANT selects another shape for experiments
// (tools/composite_split_bench.py); results do not depend on it.
static int shade_variant() {
  const char* v = getenv("UCSA_SHADE_VARIANT");
  return v ? atoi(v) : 0;
}

static int32_t composite_infer(int prec, const float* rays_d,
                               const float* norms, const float* z_c,
                               const float* sigma_c, const float* h_c,
                               const float* z_f, const float* sigma_f,
                               const float* h_f, const void* packed_color,
                               const void* packed_sem, uint32_t N, uint32_t T,
                               uint32_t t, uint32_t n_classes,
                               float density_scale, float* image, float* depth,
                               float* semantics, void* ws, void* stream) {
  UCSA_CHECK_ARG(rays_d, 0);
  UCSA_CHECK_ARG(norms, 1);
  UCSA_CHECK_ARG(z_c && sigma_c && h_c, 2);
  UCSA_CHECK_ARG(t == 0 || (z_f && sigma_f && h_f), 5);
  UCSA_CHECK_ARG(packed_color && packed_sem, 8);
  UCSA_CHECK_ARG(T >= 1 && (uint64_t)N * T < 0x80000000ull, 11);
  UCSA_CHECK_ARG((uint64_t)N * t < 0x80000000ull && T + t <= 8192, 12);
  UCSA_CHECK_ARG(n_classes >= 1 && n_classes <= 61, 13);
  UCSA_CHECK_ARG(image && depth && semantics, 15);
  UCSA_CHECK_ARG(ws, 18);
  if (N == 0) return 0;
  const uint32_t S = T + t;
  const bool half = prec == 1;
  hipStream_t s = (hipStream_t)stream;
  char* wp = (char*)ws;
  float* list_w = (float*)wp;
  uint32_t* list_row = (uint32_t*)(wp + (size_t)N * S * 4);
  uint32_t* list_ray = (uint32_t*)(wp + (size_t)N * S * 8);
  uint32_t* counts = (uint32_t*)(wp + (((size_t)N * S * 12 + 255) & ~(size_t)255));
  float* sh = prec == 0 ? nullptr
                        : (float*)((char*)counts + (((size_t)N * 4 + 255) & ~(size_t)255));
  const uint32_t nrb = cs_pad16(n_classes) / 16;
  const uint32_t cstride = 16 * nrb + 4;  // k_shade_dense's CSTRIDE
  const int variant = shade_variant();
  // (waves per workgroup, column blocks per group); measured on the bench's
  // 61 440-ray chunk, k_weights_compact (0.13 ms) included:
  // fp16: 0 = (16, 1): 0.94 ms  <- default: 120 VGPRs, 4 waves per SIMD
  //       1 = (8, 2): 1.17 (216 VGPRs: the compiler keeps the 24 weight
  //           fragments in registers), 2 = (8, 4): 1.31, 3 = (12, 2): 2.30
  //           (spills), 4 = (16, 2): 3.7 (spills)
  // fp32: 0 = (16, 1): 2.49, 1 = (12, 2): 2.85, 2 = (8, 2): 2.63 -- all behind
  //       the fused k_composite (2.33), which ucsa_render_fwd keeps for fp32
  // bf16x3: 0 = (8, 2): 1.57 ms  <- default (a weight fragment, three terms,
  //           read once for both column blocks), 1 = (16, 1): 1.60,
  //           2 = (12, 2): 1.80, 3 = (12, 1): 1.74, 4 = (8, 4): 1.90
  const uint32_t waves =
      prec == 2 ? (variant == 1 ? 16u : ((variant == 2 || variant == 3) ? 12u : 8u))
      : half    ? ((variant == 0 || variant == 4) ? 16u : (variant == 3 ? 12u : 8u))
                : (variant == 0 ? 16u : (variant == 1 ? 12u : 8u));
  // both kernels use the same ranges of whole rays per wave: enough waves to
  // fill the chip twice over
  const uint64_t total_waves = 256ull * 16 * 2;
  uint32_t rpw = (uint32_t)((N + total_waves - 1) / total_waves);
  const uint32_t rpw_min = S >= 128 ? 1u : (S >= 64 ? 2u : 4u);
  if (rpw < rpw_min) rpw = rpw_min;
  const uint32_t n_waves = ucsa_div_up(N, rpw);
  // ---- A: weights + compaction ---------------------------------------------
  {
    WcArgs a{rays_d, norms, z_c, sigma_c, z_f, sigma_f, N, T, t, density_scale,
             depth, list_w, list_row, list_ray, counts, rpw, image, semantics,
             n_classes, sh};
    const size_t smem = (size_t)WC_WAVES * 4 * S * 4;
    UCSA_CHECK_ARG(smem <= 160 * 1024, 12);
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_weights_compact),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return -(int32_t)e;
    UCSA_CLEAR_ERR();
    hipLaunchKernelGGL(k_weights_compact, dim3(ucsa_div_up(n_waves, WC_WAVES)),
                       dim3(64 * WC_WAVES), smem, s, a);
    const int32_t rc = ucsa_launch_status();
    if (rc != 0) return rc;
  }
  // ---- B: dense shading of the survivor lists -------------------------------
  const size_t w_floats =
      prec == 2 ? (size_t)(COLOR_H_FRAGS + SEM_H_FRAGS(nrb)) * 768
      : half    ? (size_t)(COLOR_H_FRAGS + SEM_H_FRAGS(nrb)) * 256
                : 7168 + 1024 + (size_t)nrb * 1024;
  const size_t per_wave = 16 * (size_t)cstride + 64;
  const size_t smem = (w_floats + waves * per_wave) * 4;
  const uint32_t blocks = ucsa_div_up(n_waves, waves);
  ShArgs b{rays_d, h_c, h_f, (const float*)packed_color, (const float*)packed_sem,
           list_w, list_row, list_ray, counts, N, S, n_classes, rpw, image,
           semantics, sh};
#define SH_GO(NRB)                                                             \
  do {                                                                         \
    if (prec == 2) {                                                           \
      if (variant == 0) return launch_shade<NRB, 2, 2, 8>(b, blocks, smem, s);  \
      if (variant == 1) return launch_shade<NRB, 1, 2, 16>(b, blocks, smem, s); \
      if (variant == 2) return launch_shade<NRB, 2, 2, 12>(b, blocks, smem, s); \
      if (variant == 3) return launch_shade<NRB, 1, 2, 12>(b, blocks, smem, s); \
      return launch_shade<NRB, 4, 2, 8>(b, blocks, smem, s);                   \
    }                                                                          \
    if (half) {                                                                \
      if (variant == 0) return launch_shade<NRB, 1, 1, 16>(b, blocks, smem, s); \
      if (variant == 1) return launch_shade<NRB, 2, 1, 8>(b, blocks, smem, s);  \
      if (variant == 2) return launch_shade<NRB, 4, 1, 8>(b, blocks, smem, s);  \
      if (variant == 3) return launch_shade<NRB, 2, 1, 12>(b, blocks, smem, s); \
      return launch_shade<NRB, 2, 1, 16>(b, blocks, smem, s);                  \
    }                                                                          \
    if (variant == 0) return launch_shade<NRB, 1, 0, 16>(b, blocks, smem, s);  \
    if (variant == 1) return launch_shade<NRB, 2, 0, 12>(b, blocks, smem, s);  \
    return launch_shade<NRB, 2, 0, 8>(b, blocks, smem, s);                     \
  } while (0)
  switch (nrb) {
    case 1: SH_GO(1);
    case 2: SH_GO(2);
    case 3: SH_GO(3);
    default: SH_GO(4);
  }
#undef SH_GO
}

extern "C" int32_t ucsa_composite_infer(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const float* packed_color,
    const float* packed_sem, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, void* workspace, void* stream) {
  return composite_infer(0, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f,
                         h_f, packed_color, packed_sem, N, T, t, n_classes,
                         density_scale, image, depth, semantics, workspace,
                         stream);
}

extern "C" int32_t ucsa_composite_infer_f16(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_half,
    const void* packed_sem_half, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, void* workspace, void* stream) {
  return composite_infer(1, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f,
                         h_f, packed_color_half, packed_sem_half, N, T, t,
                         n_classes, density_scale, image, depth, semantics,
                         workspace, stream);
}

extern "C" int32_t ucsa_composite_infer_x3(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_x3,
    const void* packed_sem_x3, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, void* workspace, void* stream) {
  return composite_infer(2, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f,
                         packed_color_x3, packed_sem_x3, N, T, t, n_classes,
                         density_scale, image, depth, semantics, workspace,
                         stream);
}
