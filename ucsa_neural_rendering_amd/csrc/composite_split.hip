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
//  k_shade_dense / k_shade16   each wave owns a range of rays whose survivors
//      form one virtual list; it walks that list 16 entries at a time, REQUESTS
//      the next group's entries / h rows / SH values before shading the current
//      one (software pipeline) and runs the colour + semantics nets:
//      k_shade_dense (precision "fp32"): f32 16x16x4 MFMA (vector-ALU work on
//      gfx950), per-ray sums in sample order through a 16-row LDS tile;
//      k_shade16 (round 3; "fp16": f16 16x16x32 MFMA; "bf16x3", the default:
//      six bf16 16x16x32 MFMAs per product, fp32-grade, mfma_mlp_x3.h): per-ray
//      sums in REGISTERS, the SH operand per ray from k_weights_compact in
//      operand form, 32-bit offsets -- see the comment above k_shade16.
//      Both are bound by instruction issue (MFMA cycles + VALU cycles add up,
//      DESIGN 5): every VALU instruction in the loop counts.
//
// The list costs 8 B per survivor each way (<= 0.1 ms per 61 440-ray chunk at
// HBM rates).  The arithmetic -- k order of every contraction, expf vs
// v_exp_f32, order of the per-ray sums -- is the fused kernel's, statement
// for statement, so both paths give the same bits (tested).
#include <cstdlib>

#include "composite_common.h"
#include "mfma_mlp_h2.h"

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
  void* sh;            // per-ray SH basis of the direction in the form the
                       // shading kernel's MFMA operand wants (null: not wanted)
  uint32_t sh_mode;    // 1: [N][16] half; 2: [N][3 terms][16] bf16 (bf16x3 split); 3: [N][2 terms][16] half (f16x2)
  int32_t* src_out;    // training forward: source index of sorted sample s (or null)
  float* w_out;        // training forward: weight of sorted sample s (or null)
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
  // A wave walks its rays one after the other and every ray starts with a dependent
  // global load (69 % of the wave cycles were waits): for S <= 256 the NEXT ray's
  // depths and densities are requested while the current ray is processed (round 6;
  // four elements per lane, same values, same order of every later operation).
  constexpr uint32_t PF = 4;
#ifndef WC_PREFETCH
#define WC_PREFETCH 1
#endif
  const bool prefetch = WC_PREFETCH && S <= 64u * PF;
  float zpre[PF], spre[PF];
  auto fetch = [&](uint32_t r) {
    const float* zc = a.z_c + (size_t)r * T;
    const float* zf = a.z_f + (size_t)r * t;
    const float* sc = a.sigma_c + (size_t)r * T;
    const float* sf = a.sigma_f + (size_t)r * t;
#pragma unroll
    for (uint32_t k = 0; k < PF; ++k) {
      const uint32_t e = lane + 64u * k;
      zpre[k] = e < S ? (e < T ? zc[e] : zf[e - T]) : 0.0f;
      spre[k] = e < S ? (e < T ? sc[e] : sf[e - T]) : 0.0f;
    }
  };
  if (prefetch) fetch(r_begin);
  for (uint32_t r = r_begin; r < r_end; ++r) {
    // ---- A1: raw depths (coarse then fine) -------------------------------
    float sg_mine[PF];
    if (prefetch) {
#pragma unroll
      for (uint32_t k = 0; k < PF; ++k) {
        const uint32_t e = lane + 64u * k;
        if (e < S) zraw[e] = zpre[k];
        sg_mine[k] = spre[k];
      }
      if (r + 1 < r_end) fetch(r + 1);      // in flight during this ray's ranks / scans
    } else {
      const float* zc = a.z_c + (size_t)r * T;
      const float* zf = a.z_f + (size_t)r * t;
      for (uint32_t e = lane; e < S; e += 64) zraw[e] = e < T ? zc[e] : zf[e - T];
    }
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
      float sg;
      if (prefetch) {   // (e - lane) / 64 is wave-uniform: selects, no indexed register array
        const uint32_t kq = (e - lane) >> 6;
        sg = kq == 0u ? sg_mine[0] : (kq == 1u ? sg_mine[1] : (kq == 2u ? sg_mine[2] : sg_mine[3]));
      } else {
        sg = e < T ? a.sigma_c[(size_t)r * T + e] : a.sigma_f[(size_t)r * t + (e - T)];
      }
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
      const float incl = wave_incl_scan_mul_dpp(fac);
      const float excl = wave_shift_up1(incl, 1.0f);
      const float w = alpha * (carry * excl);
      carry = carry * wave_last(incl);
      const bool keep = (s < S) && (w > 1e-4f);
      if (a.w_out && s < S) {  // what the backward needs (composite.hip phase A)
        a.w_out[(size_t)r * S + s] = w;
        a.src_out[(size_t)r * S + s] = (int32_t)srcs[s];
      }
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
      const float v = q == 0 ? sh[0] : (q == 1 ? sh[1] : (q == 2 ? sh[2] : sh[3]));
      if (a.sh_mode == 1) {
        reinterpret_cast<_Float16*>(a.sh)[(size_t)r * 16 + lane] = (_Float16)v;
      } else if (a.sh_mode == 3) {  // the two-term f16 split of mfma_mlp_h2.h
        _Float16* o = reinterpret_cast<_Float16*>(a.sh) + (size_t)r * 32 + lane;
        const _Float16 hi = (_Float16)v;
        o[0] = hi;
        o[16] = (_Float16)((v - (float)hi) * H2_LO_SCALE);
      } else {  // the exact three-term bf16 split of mfma_mlp_x3.h, per value
        uint16_t* o = reinterpret_cast<uint16_t*>(a.sh) + (size_t)r * 48 + lane;
        const uint32_t p0 = bf16_pair(v, 0.0f);
        const float r1 = v - pair_lo(p0);
        const uint32_t p1 = bf16_pair(r1, 0.0f);
        const float r2 = r1 - pair_lo(p1);
        o[0] = (uint16_t)p0;
        o[16] = (uint16_t)p1;
        o[32] = (uint16_t)bf16_pair(r2, 0.0f);
      }
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
  const void* sh;   // per-ray SH operands from k_weights_compact (16-bit MFMA modes)
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
  float d[CBS][3];  // the ray direction
};

// f32-input MFMA shading (bit-identical to the fused k_composite).  The 16-bit
// MFMA modes have their own kernel, k_shade16 below.
template <int NRB_SEM, int CBS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_shade_dense(ShArgs a) {
  constexpr uint32_t G = 16u * CBS;
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint32_t S = a.S, C = a.C;
  constexpr uint32_t CSTRIDE = 16 * NRB_SEM + 4;

  constexpr uint32_t WC_FLOATS = 7168;
  constexpr uint32_t WS_FLOATS = 1024 + NRB_SEM * 1024;
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
      const float* dd = a.rays_d + (size_t)en.ray[cb] * 3;
      p.d[cb][0] = dd[0];
      p.d[cb][1] = dd[1];
      p.d[cb][2] = dd[2];
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
    {
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


// ---------------------------------------------------------------------------
// k_shade16: the colour + semantics nets on the 16-bit matrix pipe -- PREC 1:
// f16 16x16x32 MFMA (tiny-cuda-nn's numerics), PREC 2: bf16x3 (six bf16 MFMAs
// per fp32 product, fp32-grade, mfma_mlp_x3.h) -- and the per-ray sums of
// w*rgb, w*p.  Same list walk and software pipeline as k_shade_dense.
//
// The kernel is bound by instruction issue: MFMA cycles and VALU cycles add up
// on a SIMD, and on gfx950 only the plain fp32 add / mul / fma and the bitwise
// ops issue at 2 cycles per wave; conversions, shifts, permutes, selects and
// integer min/max take 4, v_exp / v_rcp 8, and packed fp32 (v_pk_*) costs
// exactly two plain ones (tools/ubench/valu_rates.hip,
// profiles/r03_valu_rates.txt).  So every instruction counts:
//
//  * per-ray sums in REGISTERS: lane (g, j) keeps partial sums of its own
//    samples (entry j of every group) for its classes 16 rb + 4 g + r and, in
//    row g = 0, for r, g, b; they are added over the 16 lanes of a row (four
//    DPP row rotations per value) once per RAY.  The LDS tile of
//    k_shade_dense costs 4 wide stores, 16 reads, 16 adds and two waits per
//    16 samples.  The sums associate differently from the sample-ordered ones
//    of the fused kernel (ordinary fp32 round-off, ~1e-7).
//  * the SH operand arrives in MFMA operand form (k_weights_compact converts /
//    splits it once per ray), 32-bit byte offsets off scalar bases where the
//    sizes allow (OFF32), padded classes masked once by -inf logits, and the
//    logits multiplied by log2(e) FIRST: exp2(l' - max') needs no further
//    multiply, and fmaxf of a product is one v_max_f32 (of an MFMA result the
//    compiler first quiets a possible signalling NaN: two more per max).
//    No inline-asm VALU next to MFMAs: the hazard recogniser does not see it.
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(
                 0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float x) {
  x = dpp_add<0x128>(x);  // row_ror:8
  x = dpp_add<0x124>(x);  // row_ror:4
  x = dpp_add<0x122>(x);  // row_ror:2
  x = dpp_add<0x121>(x);  // row_ror:1
  return x;
}
template <typename T>
__device__ __forceinline__ T ld_off32(const void* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }
#define UCSA_LOG2E 1.4426950408889634f

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int CBS>
struct Ent16 {
  float w[CBS];
  uint32_t row[CBS];
  uint32_t ray[CBS];
};
template <int CBS, int PREC>
struct Pre16 {
  float ew[CBS];
  uint32_t eray[CBS];
  f32x4 hv[CBS];
  u32x2 sh[CBS][PREC == 2 ? 3 : (PREC == 3 ? 2 : 1)];  // this lane's 4 SH k-slots, operand form
};

// The sums of one ray: add the partial sums of the 16 lanes of every row, lane
// j == 0 of row g stores classes 16 rb + 4 g + r (and row 0 r, g, b).  Once per
// ray and out of line: the shading loop stays small.
template <int NRB_SEM>
struct Acc16 {  // by value: stays in VGPRs across the call
  f32x4 s[NRB_SEM];
  float c[3];
};
template <int NRB_SEM>
__device__ __attribute__((noinline)) void shade16_flush(
    Acc16<NRB_SEM> acc, uint32_t ray, uint32_t C, float* __restrict__ semantics,
    float* __restrict__ image, uint32_t lane) {
  const uint32_t g = lane >> 4, j = lane & 15u;
  f32x4 o[NRB_SEM];
#pragma unroll
  for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[rb][r] = row16_sum(acc.s[rb][r]);
  float c3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) c3[c] = row16_sum(acc.c[c]);
  if (j == 0) {
    float* dst = semantics + (size_t)ray * C;
#pragma unroll
    for (int rb = 0; rb < NRB_SEM; ++rb) {
      const uint32_t c0 = rb * 16 + 4 * g;
      if ((C & 3u) == 0u) {  // rows are 16-byte aligned: whole quads
        if (c0 < C) *reinterpret_cast<f32x4*>(dst + c0) = o[rb];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c0 + r < C) dst[c0 + r] = o[rb][r];
      }
    }
    if (g == 0) {
      float* im = image + (size_t)ray * 3;
      im[0] = c3[0];
      im[1] = c3[1];
      im[2] = c3[2];
    }
  }
}

template <int NRB_SEM, int CBS, int PREC, int WAVES, bool OFF32>
__global__ void __launch_bounds__(64 * WAVES) k_shade16(ShArgs a) {
  static_assert(PREC >= 1 && PREC <= 3, "16-bit MFMA modes");
  constexpr int TERMS = PREC == 2 ? 3 : (PREC == 3 ? 2 : 1);
  constexpr uint32_t G = 16u * CBS;
  const X3Sel sel = x3_selectors();  // bf16x3 split constants, once per kernel
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint32_t S = a.S, C = a.C;

  constexpr uint32_t WC_FLOATS = COLOR_H_FRAGS * 256 * TERMS;
  constexpr uint32_t WS_FLOATS = SEM_H_FRAGS(NRB_SEM) * 256 * TERMS;
  float* w_color = cs_smem;
  float* w_sem = w_color + WC_FLOATS;
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
  // the wave's region of the survivor lists: scalar bases, 32-bit offsets
  const size_t lbase = (size_t)r_begin64 * S;
  const float* lw = a.list_w + lbase;
  const uint32_t* lrow = a.list_row + lbase;
  const uint32_t* lray = a.list_ray + lbase;
  constexpr uint32_t SH_BYTES = 32u * TERMS;  // per ray

  uint32_t cur_ray = 0xFFFFFFFFu;  // ray whose partial sums sit in accS / accC
  f32x4 accS[NRB_SEM];             // classes 16 rb + 4 g + r, entries j, j+16, ...
  float accC[3];                   // r, g, b (row g == 0)
  auto zero_acc = [&]() {
#pragma unroll
    for (int rb = 0; rb < NRB_SEM; ++rb) accS[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    accC[0] = accC[1] = accC[2] = 0.0f;
  };
  zero_acc();
  auto flush_ray = [&](uint32_t ray) {  // also zeroes the accumulators
    Acc16<NRB_SEM> v;
#pragma unroll
    for (int rb = 0; rb < NRB_SEM; ++rb) v.s[rb] = accS[rb];
    v.c[0] = accC[0];
    v.c[1] = accC[1];
    v.c[2] = accC[2];
    shade16_flush<NRB_SEM>(v, ray, C, a.semantics, a.image, lane);
    zero_acc();
  };

  // stage 1: list entries (past `total`: the last real one at weight 0)
  auto load_entries = [&](uint32_t gb, Ent16<CBS>& en) {
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      uint32_t e = gb + cb * 16 + j;
      const bool live = e < total;
      if (!live) e = total - 1;
      const uint32_t off = e * 4u;
      const float w = ld_off32<float>(lw, off);
      en.w[cb] = live ? w : 0.0f;
      en.row[cb] = ld_off32<uint32_t>(lrow, off);
      en.ray[cb] = ld_off32<uint32_t>(lray, off);
    }
  };
  // stage 2: h rows and SH operands of entries that have arrived
  auto load_operands = [&](const Ent16<CBS>& en, Pre16<CBS, PREC>& p) {
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      p.ew[cb] = en.w[cb];
      p.eray[cb] = en.ray[cb];
      const uint32_t row = en.row[cb];
      const bool fine = (row & ROW_FINE) != 0u;
      const uint32_t r0 = row & ~ROW_FINE;
      if constexpr (OFF32) {
        const char* hb = reinterpret_cast<const char*>(fine ? a.h_f : a.h_c);
        p.hv[cb] = *reinterpret_cast<const f32x4*>(hb + ((r0 << 6) | (g << 4)));
        const uint32_t so = en.ray[cb] * SH_BYTES + 8u * g;
#pragma unroll
        for (int term = 0; term < TERMS; ++term)
          p.sh[cb][term] = ld_off32<u32x2>(a.sh, so + 32u * term);
      } else {
        const float* hp = (fine ? a.h_f : a.h_c) + (size_t)r0 * 16 + 4 * g;
        p.hv[cb] = *reinterpret_cast<const f32x4*>(hp);
        const char* sp = reinterpret_cast<const char*>(a.sh) +
                         (size_t)en.ray[cb] * SH_BYTES + 8u * g;
#pragma unroll
        for (int term = 0; term < TERMS; ++term)
          p.sh[cb][term] = *reinterpret_cast<const u32x2*>(sp + 32 * term);
      }
    }
  };

  // classes of the last row block beyond C (per lane, loop-invariant)
  bool pad[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) pad[r] = (uint32_t)((NRB_SEM - 1) * 16 + 4 * g + r) >= C;

  auto shade = [&](const Pre16<CBS, PREC>& p) {
    float geo[CBS][4];
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      geo[cb][0] = (g == 0) ? 1.0f : p.hv[cb][0];  // slot m == 0 -> the "ones" pad
      geo[cb][1] = p.hv[cb][1];
      geo[cb][2] = p.hv[cb][2];
      geo[cb][3] = p.hv[cb][3];
    }
    f32x4 o3[CBS];
    f32x4 lg[CBS][NRB_SEM];
    // the weight fragments stay in LDS (`zoff`: an opaque zero re-made per
    // group keeps the loads inside the loop instead of 96+ pinned VGPRs)
    uint32_t zoff = 0;
    asm volatile("" : "+v"(zoff));
    const uint32_t wl = lane + zoff;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PREC == 2) {
      // layer-major: a weight fragment (three terms, 3 ds_read_b128) is read
      // once per group and used by all its column blocks
      X3 b1[CBS];
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
#pragma unroll
        for (int term = 0; term < 3; ++term) {
          b1[cb].t[term][0] = p.sh[cb][term][0];
          b1[cb].t[term][1] = p.sh[cb][term][1];
        }
        split_pair(geo[cb][0], geo[cb][1], b1[cb], 2, sel);
        split_pair(geo[cb][2], geo[cb][3], b1[cb], 3, sel);
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
        h0[cb] = chain_relu_x3(a1[cb][0], a1[cb][1], sel);
        h1[cb] = chain_relu_x3(a1[cb][2], a1[cb][3], sel);
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
        h0[cb] = chain_relu_x3(a2[cb][0], a2[cb][1], sel);
        h1[cb] = chain_relu_x3(a2[cb][2], a2[cb][3], sel);
      }
      {
        const W3 wa = frag_x3(w_color, 12, wl), wb = frag_x3(w_color, 13, wl);
#pragma unroll
        for (int cb = 0; cb < CBS; ++cb) {
          o3[cb] = mfma_x3(wa, h0[cb], z4);
          o3[cb] = mfma_x3(wb, h1[cb], o3[cb]);
        }
      }
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        h0[cb] = chain_relu_x3(a1[cb][0], a1[cb][1], sel);
        h1[cb] = chain_relu_x3(a1[cb][2], a1[cb][3], sel);
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
    } else if constexpr (PREC == 3) {
      // f16x2 (mfma_mlp_h2.h): the layer structure of the bf16x3 branch with
      // two-term operands -- 72 MFMAs per column block instead of 144
      const H2Sel hsel = h2_selectors();
      H2X b1[CBS];
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
#pragma unroll
        for (int term = 0; term < 2; ++term) {
          b1[cb].t[term][0] = p.sh[cb][term][0];
          b1[cb].t[term][1] = p.sh[cb][term][1];
        }
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
        for (int cb = 0; cb < CBS; ++cb) o3[cb] = h2_mul2(wa, h0[cb], wb, h1[cb]);
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
      // fp16 weights / layer inputs, fp32 accumulate: 24 MFMAs per column block
#pragma unroll
      for (int cb = 0; cb < CBS; ++cb) {
        const uint32_t g01 = __builtin_bit_cast(uint32_t, cvt_pk_h(geo[cb][0], geo[cb][1]));
        const uint32_t g23 = __builtin_bit_cast(uint32_t, cvt_pk_h(geo[cb][2], geo[cb][3]));
        const half8 b1 = __builtin_bit_cast(half8, u32x4{p.sh[cb][0][0], p.sh[cb][0][1], g01, g23});
        const half8 bs = __builtin_bit_cast(half8, u32x4{g01, g23, 0u, 0u});
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
        o3[cb] = mfma_h(frag_h(w_color, 12, wl), h0, z4);
        o3[cb] = mfma_h(frag_h(w_color, 13, wl), h1, o3[cb]);
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

    // sigmoid, softmax, weighted partial sums
#pragma unroll
    for (int cb = 0; cb < CBS; ++cb) {
      float rgb[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        rgb[c] = fast_rcp(1.0f + exp2_hw(o3[cb][c] * -UCSA_LOG2E));
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (pad[r]) lg[cb][NRB_SEM - 1][r] = -INFINITY;
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) lg[cb][rb][r] = lg[cb][rb][r] * UCSA_LOG2E;
      float mx = lg[cb][0][0];
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (rb + r > 0) mx = fmaxf(mx, lg[cb][rb][r]);
      mx = rows4_max(mx);
      float sum = 0.0f;
#pragma unroll
      for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ex = exp2_hw(lg[cb][rb][r] - mx);
          lg[cb][rb][r] = ex;
          sum += ex;
        }
      sum = rows4_sum(sum);
      const float wgt = p.ew[cb];
      const float ws = wgt * fast_rcp(sum);
      // the list is sorted by ray: first == last means one ray for all 16
      const uint32_t ray_a = (uint32_t)__builtin_amdgcn_readlane((int)p.eray[cb], 0);
      const uint32_t ray_z = (uint32_t)__builtin_amdgcn_readlane((int)p.eray[cb], 15);
      if (ray_a == ray_z) {
        if (ray_a != cur_ray) {
          if (cur_ray != 0xFFFFFFFFu) flush_ray(cur_ray);
          cur_ray = ray_a;
        }
#pragma unroll
        for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            accS[rb][r] = __builtin_fmaf(lg[cb][rb][r], ws, accS[rb][r]);
#pragma unroll
        for (int c = 0; c < 3; ++c) accC[c] = __builtin_fmaf(rgb[c], wgt, accC[c]);
      } else {
        uint32_t ray = ray_a;
        for (;;) {
          if (ray != cur_ray) {
            if (cur_ray != 0xFFFFFFFFu) flush_ray(cur_ray);
            cur_ray = ray;
          }
          const bool mine = p.eray[cb] == ray;
          const float wsm = mine ? ws : 0.0f, wgm = mine ? wgt : 0.0f;
#pragma unroll
          for (int rb = 0; rb < NRB_SEM; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              accS[rb][r] = __builtin_fmaf(lg[cb][rb][r], wsm, accS[rb][r]);
#pragma unroll
          for (int c = 0; c < 3; ++c) accC[c] = __builtin_fmaf(rgb[c], wgm, accC[c]);
          const unsigned long long later = __ballot(p.eray[cb] > ray) & 0xFFFFull;
          if (later == 0ull) break;
          ray = (uint32_t)__builtin_amdgcn_readlane((int)p.eray[cb],
                                                    (int)__builtin_ctzll(later));
        }
      }
    }
  };

  Ent16<CBS> e1, e2;
  Pre16<CBS, PREC> p0, p1;
  load_entries(0u, e1);
  load_operands(e1, p0);
  load_entries(G, e1);
  for (uint32_t gb = 0; gb < total; gb += 2 * G) {
    load_entries(gb + 2 * G, e2);
    load_operands(e1, p1);
    shade(p0);
    if (gb + G >= total) break;
    load_entries(gb + 3 * G, e1);
    load_operands(e2, p0);
    shade(p1);
  }
  if (cur_ray != 0xFFFFFFFFu) flush_ray(cur_ray);
}

static inline uint32_t cs_pad16(uint32_t n) { return (n + 15u) / 16u * 16u; }

extern "C" uint64_t ucsa_composite_infer_workspace_bytes(uint32_t N, uint32_t T,
                                                         uint32_t t) {
  const uint64_t S = (uint64_t)T + t;
  return (((uint64_t)N * S * 12 + 255) & ~255ull) + (((uint64_t)N * 4 + 255) & ~255ull) +
         (uint64_t)N * 96;  // per-ray SH operands: 32 B (f16) or 3 x 32 B (bf16x3)
}

template <int NRB, int CBS, int WAVES>
static int32_t launch_shade(const ShArgs& a, uint32_t blocks, size_t smem,
                            hipStream_t s) {
  hipError_t e = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&k_shade_dense<NRB, CBS, WAVES>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return -(int32_t)e;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL((k_shade_dense<NRB, CBS, WAVES>), dim3(blocks),
                     dim3(64 * WAVES), smem, s, a);
  return ucsa_launch_status();
}

template <int NRB, int CBS, int PREC, int WAVES>
static int32_t launch_shade16(const ShArgs& a, uint32_t n_waves, bool off32,
                              hipStream_t s) {
  const size_t smem = (size_t)(COLOR_H_FRAGS + SEM_H_FRAGS(NRB)) *
                      (PREC == 2 ? 768 : (PREC == 3 ? 512 : 256)) * 4;
  const uint32_t blocks = ucsa_div_up(n_waves, WAVES);
  const void* fn = off32
      ? reinterpret_cast<const void*>(&k_shade16<NRB, CBS, PREC, WAVES, true>)
      : reinterpret_cast<const void*>(&k_shade16<NRB, CBS, PREC, WAVES, false>);
  hipError_t e = hipFuncSetAttribute(
      fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return -(int32_t)e;
  UCSA_CLEAR_ERR();
  if (off32)
    hipLaunchKernelGGL((k_shade16<NRB, CBS, PREC, WAVES, true>), dim3(blocks),
                       dim3(64 * WAVES), smem, s, a);
  else
    hipLaunchKernelGGL((k_shade16<NRB, CBS, PREC, WAVES, false>), dim3(blocks),
                       dim3(64 * WAVES), smem, s, a);
  return ucsa_launch_status();
}

// Launch shapes (waves per workgroup, column blocks per group).  The register
// budget follows from the workgroup size: 16 waves -> 128 VGPRs, 12 -> 168,
// 8 -> 256.  UCSA_SHADE_VARIANT selects another shape for experiments
// (tools/composite_split_bench.py); results do not depend on it.
static int shade_variant() {
  const char* v = ucsa_getenv("UCSA_SHADE_VARIANT");
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
                               float* semantics, void* ws, void* stream,
                               int32_t* src_out = nullptr, float* w_out = nullptr) {
  UCSA_CHECK_ARG(rays_d, 0);
  UCSA_CHECK_ARG((src_out == nullptr) == (w_out == nullptr), 19);
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
  hipStream_t s = (hipStream_t)stream;
  char* wp = (char*)ws;
  float* list_w = (float*)wp;
  uint32_t* list_row = (uint32_t*)(wp + (size_t)N * S * 4);
  uint32_t* list_ray = (uint32_t*)(wp + (size_t)N * S * 8);
  uint32_t* counts = (uint32_t*)(wp + (((size_t)N * S * 12 + 255) & ~(size_t)255));
  void* sh = prec == 0 ? nullptr
                       : (void*)((char*)counts + (((size_t)N * 4 + 255) & ~(size_t)255));
  const uint32_t nrb = cs_pad16(n_classes) / 16;
  const uint32_t cstride = 16 * nrb + 4;  // k_shade_dense's CSTRIDE
  const int variant = shade_variant();
  // (waves per workgroup, column blocks per group) of the shading kernel;
  // UCSA_SHADE_VARIANT picks another shape for experiments
  // (tools/composite_split_bench.py), results do not depend on it.
  //   fp32 (k_shade_dense): 0 = (16, 1), 1 = (12, 2), 2 = (8, 2) -- all behind
  //     the fused k_composite, which ucsa_render_fwd keeps for fp32
  //   f16 / bf16x3 (k_shade16): see launch table below
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
             n_classes, sh, (uint32_t)prec, src_out, w_out};
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
  ShArgs b{rays_d, h_c, h_f, (const float*)packed_color, (const float*)packed_sem,
           list_w, list_row, list_ray, counts, N, S, n_classes, rpw, image,
           semantics, sh};
  if (prec == 0) {
    const uint32_t waves = variant == 0 ? 16u : (variant == 1 ? 12u : 8u);
    const size_t w_floats = 7168 + 1024 + (size_t)nrb * 1024;
    const size_t per_wave = 16 * (size_t)cstride + 64;
    const size_t smem = (w_floats + waves * per_wave) * 4;
    const uint32_t blocks = ucsa_div_up(n_waves, waves);
#define SH_GO(NRB)                                                         \
  do {                                                                     \
    if (variant == 0) return launch_shade<NRB, 1, 16>(b, blocks, smem, s); \
    if (variant == 1) return launch_shade<NRB, 2, 12>(b, blocks, smem, s); \
    return launch_shade<NRB, 2, 8>(b, blocks, smem, s);                    \
  } while (0)
    switch (nrb) {
      case 1: SH_GO(1);
      case 2: SH_GO(2);
      case 3: SH_GO(3);
      default: SH_GO(4);
    }
#undef SH_GO
  }
  // 32-bit byte offsets into h (64 B per sample) and the SH operands (<= 96 B
  // per ray) when the launch is small enough -- it is for every chunk render()
  // makes; the 64-bit form covers the rest of the C API's range
  const uint64_t rows = (uint64_t)N * (T > t ? T : t);
  const bool off32 = rows * 64 < (1ull << 32) && (uint64_t)N * 96 < (1ull << 32);
  // measured on the bench's 61 440-ray chunk (ms, k_weights_compact's 0.14
  // included; round 3, tools/x3_bench.py):
  //   f16:    0 = (16 waves, 1 block) 0.55 <- default   1 = (8, 2) 0.61
  //           2 = (8, 1) 0.55   3 = (12, 1) 0.57
  //   bf16x3: 0 = (8 waves, 1 block) 1.43 <- default   1 = (16, 1) 1.50
  //           2 = (12, 1) 1.60   3 = (8, 2) 1.63
  //   f16x2 (round 4, M rays/s of the whole cfg2 view): 0 = (8 waves, 2 blocks)
  //           17.87 <- default   1 = (16, 1) 17.25   2 = (12, 1) 17.31
  //           3 = (8, 1) 17.25: with half the MFMAs a weight fragment read from
  //           LDS is worth sharing between two column blocks
#define SH_GO16(NRB)                                                                \
  do {                                                                              \
    if (prec == 3) {                                                                \
      if (variant == 1) return launch_shade16<NRB, 1, 3, 16>(b, n_waves, off32, s); \
      if (variant == 2) return launch_shade16<NRB, 1, 3, 12>(b, n_waves, off32, s); \
      if (variant == 3) return launch_shade16<NRB, 1, 3, 8>(b, n_waves, off32, s);  \
      return launch_shade16<NRB, 2, 3, 8>(b, n_waves, off32, s);                    \
    }                                                                               \
    if (prec == 2) {                                                                \
      if (variant == 1) return launch_shade16<NRB, 1, 2, 16>(b, n_waves, off32, s); \
      if (variant == 2) return launch_shade16<NRB, 1, 2, 12>(b, n_waves, off32, s); \
      if (variant == 3) return launch_shade16<NRB, 2, 2, 8>(b, n_waves, off32, s);  \
      return launch_shade16<NRB, 1, 2, 8>(b, n_waves, off32, s);                    \
    }                                                                               \
    if (variant == 1) return launch_shade16<NRB, 2, 1, 8>(b, n_waves, off32, s);    \
    if (variant == 2) return launch_shade16<NRB, 1, 1, 8>(b, n_waves, off32, s);    \
    if (variant == 3) return launch_shade16<NRB, 1, 1, 12>(b, n_waves, off32, s);   \
    return launch_shade16<NRB, 1, 1, 16>(b, n_waves, off32, s);                     \
  } while (0)
  switch (nrb) {
    case 1: SH_GO16(1);
    case 2: SH_GO16(2);
    case 3: SH_GO16(3);
    default: SH_GO16(4);
  }
#undef SH_GO16
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

extern "C" int32_t ucsa_composite_infer_h2(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_h2,
    const void* packed_sem_h2, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, void* workspace, void* stream) {
  return composite_infer(3, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f,
                         packed_color_h2, packed_sem_h2, N, T, t, n_classes,
                         density_scale, image, depth, semantics, workspace,
                         stream);
}

// Training forward of the colour / semantics stage on the split pair with the
// bf16x3 nets (fp32-grade): same outputs as ucsa_composite_fwd plus its aux
// arrays -- src [N, T + t] (source index of every sorted sample: < T coarse,
// else T + fine index) and w [N, T + t] (its weight, masked or not) -- which
// ucsa_composite_bwd consumes.  Reference: renderer_semantics.py:220-299.
extern "C" int32_t ucsa_composite_train_fwd_x3(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_x3,
    const void* packed_sem_x3, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, int32_t* src, float* w, void* workspace, void* stream) {
  UCSA_CHECK_ARG(src && w, 18);
  return composite_infer(2, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f,
                         packed_color_x3, packed_sem_x3, N, T, t, n_classes,
                         density_scale, image, depth, semantics, workspace,
                         stream, src, w);
}

// ... and with the f16x2 nets (mfma_mlp_h2.h): what the default training mode's
// forward runs since round 4 (the backward recomputes with its own bf16x2 packs)
extern "C" int32_t ucsa_composite_train_fwd_h2(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_h2,
    const void* packed_sem_h2, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, int32_t* src, float* w, void* workspace, void* stream) {
  UCSA_CHECK_ARG(src && w, 18);
  return composite_infer(3, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f, h_f,
                         packed_color_h2, packed_sem_h2, N, T, t, n_classes,
                         density_scale, image, depth, semantics, workspace,
                         stream, src, w);
}
