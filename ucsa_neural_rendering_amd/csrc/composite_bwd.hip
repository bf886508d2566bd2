// Backward of merge/weights/masked MLPs/compositing (the autograd of
// reference nr4seg/nerf/renderer_semantics.py:238-299 and
// nr4seg/nerf/network_tcnn_semantics.py:147-207).
//
// Two kernels:
//  k_shade_bwd   : for every sample that passed the w > 1e-4 mask, recompute
//                  the colour / semantics nets (fp32 MFMA), push d_image and
//                  d_semantics back through them, write d(geo_feat) into the
//                  sample's d_h row, the colour/depth gradient wrt its weight
//                  into G[N,S], and accumulate dW for both nets in registers
//                  (per-wave partials, reduced deterministically afterwards).
//                  The semantic weights are detached (:270): no d_w from them.
//  k_weights_bwd : per ray, d_sigma from G by a forward transmittance scan and
//                  a reverse suffix scan, times the trunc_exp backward
//                  (reference nr4seg/nerf/activation.py:17-21), into d_h[:,0].
#include "composite_common.h"
#include "mfma_mlp_x3.h"

#define CB_WAVES 4
// Waves per workgroup of the f16 variant.  Measured: 8 waves (2 per SIMD, a
// 256-register budget) spill ~100-160 registers into scratch inside the main
// loop -- the 176 fp32 dW accumulators plus the activations of both nets do
// not fit -- and end no faster than the fp32 kernel (1.43 vs 1.48 ms); 4 waves
// (1 per SIMD, no spills) it is.
#define CB_WAVES_H 4
#define CB_WAVES_SPLIT 8  // per-net kernels (NET = 1, 2): two waves per SIMD
#define CB_CAP 80  // 15 pending + one 64-sample chunk
#define ROW_FINE 0x80000000u

extern __shared__ __attribute__((aligned(16))) float cb_smem[];

__device__ __forceinline__ void cb_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct ShadeBwdArgs {
  const float* rays_d;
  const float* norms;
  const float* z_c;
  const float* z_f;
  const float* h_c;
  const float* h_f;
  const int32_t* src;
  const float* weights;
  const float* d_image;
  const float* d_depth;
  const float* d_sem;
  const float* packed_color;
  const float* packed_sem;
  const float* packed_color_t;
  const float* packed_sem_t;
  uint32_t N, T, t, C;
  float* G;
  float* d_h_c;
  float* d_h_f;
  float* partial_color;
  float* partial_sem;
  uint32_t rays_per_wave;
  // marched spans (MARCH = true): rays [N,3] = (ray id, first point, count)
  // of ucsa_march_rays_train; weights / G are [M], h_c / d_h_c are [M,16],
  // t_all [M] is the ray parameter of each sample (ucsa_march_train_fwd)
  const int32_t* rays;
  const float* t_all;
  uint32_t n_points;
  float w_min;
  // HALF = true: gradients entering the f16 MFMAs are multiplied by f16_scale
  // (a power of two) and everything that leaves the kernel is divided by it
  float f16_scale;
};

// two accumulator blocks -> one 32-wide k-step operand, no ReLU (gradients)
__device__ __forceinline__ half8 chain_h(f32x4 lo, f32x4 hi) {
  half8 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] = (_Float16)lo[r];
    v[4 + r] = (_Float16)hi[r];
  }
  return v;
}

template <int OB, int IB>
__device__ __forceinline__ void dw_scale(f32x4 (&dw)[OB][IB], float s) {
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) dw[ob][ib][r] = dw[ob][ib][r] * s;
}

// what the f16 forward fed into the next layer: relu, rounded to fp16
__device__ __forceinline__ f32x4 relu_q4(f32x4 v) {
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = (float)(_Float16)relu1(v[k]);
  return r;
}

__device__ __forceinline__ f32x4 q4(f32x4 v) {
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = (float)(_Float16)v[k];
  return r;
}

__device__ __forceinline__ void sh4_select_b(float dx, float dy, float dz,
                                             uint32_t g, f32x4& o) {
  const float x = ((dx + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float y = ((dy + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float z = ((dz + 1.0f) / 2.0f) * 2.0f - 1.0f;
  const float xy = x * y, xz = x * z, yz = y * z;
  const float x2 = x * x, y2 = y * y, z2 = z * z;
  if (g == 0) {
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
  } else if (g == 1) {
    o[0] = 1.0925484305920792f * xy;
    o[1] = -1.0925484305920792f * yz;
    o[2] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[3] = -1.0925484305920792f * xz;
  } else if (g == 2) {
    o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[2] = 2.8906114426405538f * xy * z;
    o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  } else {
    o[0] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[1] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[2] = 1.4453057213202769f * z * (x2 - y2);
    o[3] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  }
}

__device__ __forceinline__ f32x4 gate4(f32x4 pre, f32x4 v) {
  f32x4 r;
  r[0] = pre[0] > 0.f ? v[0] : 0.f;
  r[1] = pre[1] > 0.f ? v[1] : 0.f;
  r[2] = pre[2] > 0.f ? v[2] : 0.f;
  r[3] = pre[3] > 0.f ? v[3] : 0.f;
  return r;
}

// HALF: the colour / semantics nets and their dX contractions run on
// 16x16x32 f16 MFMA (fp16 weights and layer inputs, fp32 accumulate) -- the
// forward recompute reproduces k_composite<.., HALF = true> exactly, so the
// ReLU gates and the activations entering dW are the forward's.  dW runs on
// the f16 pipe too (dw_accumulate_h: one 16x16x16 f16 MFMA per weight tile and
// 16 samples; gradients under the loss scale and the forward's fp16 layer
// inputs, accumulated in fp32 registers for the whole kernel) -- as f32-input
// MFMAs the 176 weight-gradient k-steps per 16 samples were four fifths of
// this variant's matrix work and ran on the vector ALU.
//
// NET: 0 = both nets in one kernel (the dW accumulators of both, 176 fp32x4
// registers, pin it to ONE wave per SIMD); 1 = the colour net only, 2 = the
// semantics net only -- launched as a pair (colour first: it stores its part
// of d(geo_feat), the semantics kernel adds its own) each holds only its own
// dW accumulators (112 / 64 registers) and weights (52 / 32 KB of LDS), so
// CB_WAVES_SPLIT = 8 waves per workgroup = TWO waves per SIMD fit, and the LDS
// round trips / global loads of one wave hide behind the MFMAs of the other.
//
// B2 (round 4, with NET = 1 / 2; "bf16x2"): the same structure as HALF -- fragment
// layout, chaining, transposed weight fragments -- with every f16 operand
// replaced by a two-term bf16 split (mfma_mlp_x3.h "bf16x2": 2^-16 per
// product, fp32 range, no loss scale) and every f16 MFMA by three bf16 passes:
// 512 f32-input MFMAs (32 cycles each, vector-ALU rate) per 16 samples become
// 276 bf16 ones (16 cycles).  The weights come as ucsa_mlp_pack_x3 /
// ucsa_mlp_pack_t_x3 buffers; LDS keeps their terms 0 and 1.
template <int NRB, bool MARCH, bool HALF, int NET = 0, bool B2 = false>
__global__ void __launch_bounds__(64 * (NET ? CB_WAVES_SPLIT : (HALF ? CB_WAVES_H : CB_WAVES)))
k_shade_bwd(ShadeBwdArgs a) {
  static_assert(!(B2 && HALF), "one reduced-precision mode at a time");
  constexpr uint32_t NW = NET ? CB_WAVES_SPLIT : (HALF ? CB_WAVES_H : CB_WAVES);
  constexpr bool DO_C = NET != 2, DO_S = NET != 1;
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t g = lane >> 4, j = lane & 15u;
  const uint32_t T = a.T, t = a.t, S = a.T + a.t, C = a.C;
  constexpr int NS = (NRB + 1) / 2;  // half8 k-steps covering the class logits

  constexpr uint32_t XT = B2 ? 2 : 1;   // bf16 terms kept per fragment
  constexpr uint32_t WC = !DO_C ? 0 : (HALF || B2) ? COLOR_H_FRAGS * 256 * XT : 7168;
  constexpr uint32_t WS = !DO_S ? 0 : (HALF || B2) ? SEM_H_FRAGS(NRB) * 256 * XT : 1024 + NRB * 1024;
  constexpr uint32_t WTC = !DO_C ? 0 : (HALF || B2) ? 14 * 256 * XT : 6144;
  constexpr uint32_t WTS = !DO_S ? 0 : (HALF || B2) ? (4 * NS + 2) * 256 * XT : (16 * NRB + 16) * 64;
  float* w_color = cb_smem;
  float* w_sem = w_color + WC;
  float* wt_color = w_sem + WS;
  float* wt_sem = wt_color + WTC;
  float* per_wave = wt_sem + WTS;
  const uint32_t per_wave_floats = 5 * CB_CAP + 2 * 16 * TILE_LD;
  float* base = per_wave + (size_t)wid * per_wave_floats;
  float* lw = base;
  uint32_t* lrow = reinterpret_cast<uint32_t*>(lw + CB_CAP);
  uint32_t* lray = lrow + CB_CAP;
  uint32_t* lsmp = lray + CB_CAP;
  float* lz = reinterpret_cast<float*>(lsmp + CB_CAP);
  float* dy_tile = lz + CB_CAP;
  float* x_tile = dy_tile + 16 * TILE_LD;

  if constexpr (B2) {
    // terms 0 and 1 of the three-term global pack: [(f*3+term)*64+lane] (16 B)
    auto copy2 = [&](float* dst, const float* src, uint32_t n_floats) {
      const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
      u32x4* d4 = reinterpret_cast<u32x4*>(dst);
      for (uint32_t i = threadIdx.x; i < n_floats / 4; i += blockDim.x) {
        const uint32_t f = i >> 7, term = (i >> 6) & 1u, l = i & 63u;
        d4[i] = s4[(f * 3 + term) * 64 + l];
      }
    };
    copy2(w_color, a.packed_color, WC);
    copy2(w_sem, a.packed_sem, WS);
    copy2(wt_color, a.packed_color_t, WTC);
    copy2(wt_sem, a.packed_sem_t, WTS);
  } else {
    for (uint32_t i = threadIdx.x; i < WC; i += blockDim.x) w_color[i] = a.packed_color[i];
    for (uint32_t i = threadIdx.x; i < WS; i += blockDim.x) w_sem[i] = a.packed_sem[i];
    for (uint32_t i = threadIdx.x; i < WTC; i += blockDim.x) wt_color[i] = a.packed_color_t[i];
    for (uint32_t i = threadIdx.x; i < WTS; i += blockDim.x) wt_sem[i] = a.packed_sem_t[i];
  }
  __syncthreads();
  const float gs = HALF ? a.f16_scale : 1.0f, inv_gs = 1.0f / gs;
  const X3Sel sel = x3_selectors();

  const uint64_t gwave = (uint64_t)blockIdx.x * NW + wid;
  f32x4 dwc1[4][2], dwc2[4][4], dwc3[1][4], dws1[4][1], dws2[NRB][4];
  if constexpr (DO_C) {
    dw_zero(dwc1);
    dw_zero(dwc2);
    dw_zero(dwc3);
  }
  if constexpr (DO_S) {
    dw_zero(dws1);
    dw_zero(dws2);
  }

  uint32_t cnt = 0;

  // Operands of one block of <= 16 surviving samples.  With one wave per SIMD
  // nothing hides a global load, so the NEXT block's operands (h row,
  // direction, upstream gradients of its ray) are requested before the
  // current block is shaded (same values, same arithmetic).
  struct BlockOps {
    bool live, fine;
    float wgt, zz;
    uint32_t ray, smp;
    size_t hoff;
    f32x4 geo;
    float dir[3], di[3], dd, nrm;
    float dsem[NRB * 4];
  };
  auto load_ops = [&](uint32_t first, uint32_t n, BlockOps& b) {
    uint32_t e = j;
    b.live = e < n;
    if (!b.live) e = n - 1;
    e += first;
    b.wgt = b.live ? lw[e] : 0.0f;
    const uint32_t row = lrow[e];
    b.ray = lray[e];
    b.smp = lsmp[e];
    b.zz = lz[e];
    b.fine = (row & ROW_FINE) != 0;
    b.hoff = (size_t)(row & ~ROW_FINE) * 16 + 4 * g;
    b.geo = *reinterpret_cast<const f32x4*>((b.fine ? a.h_f : a.h_c) + b.hoff);
    if constexpr (DO_C) {
      const float* dptr = a.rays_d + (size_t)b.ray * 3;
      b.dir[0] = dptr[0];
      b.dir[1] = dptr[1];
      b.dir[2] = dptr[2];
      const float* di = a.d_image + (size_t)b.ray * 3;
      b.di[0] = di[0];
      b.di[1] = di[1];
      b.di[2] = di[2];
      b.dd = a.d_depth[b.ray];
      b.nrm = a.norms[b.ray];
    }
    if constexpr (DO_S) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t cls = rb * 16 + 4 * g + r;
          b.dsem[rb * 4 + r] = cls < C ? a.d_sem[(size_t)b.ray * C + cls] : 0.0f;
        }
    }
  };

  // ---- one block of up to 16 surviving samples ---------------------------
  auto shade16 = [&](const BlockOps& b) {
    const bool live = b.live, fine = b.fine;
    const float wgt = b.wgt, zz = b.zz;
    const uint32_t ray = b.ray, smp = b.smp;
    const size_t hoff = b.hoff;
    f32x4 geo = b.geo;
    if (g == 0) geo[0] = 1.0f;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dslot[1] = {z4}, dslot_s[1] = {z4};

    // =========================== colour net ===============================
    if constexpr (DO_C) {
      f32x4 sh;
      sh4_select_b(b.dir[0], b.dir[1], b.dir[2], g, sh);
      // forward (recompute)
      f32x4 a1c[4], a2c[4], o3[1];
      f32x4 geo_c = geo;
      if constexpr (B2) {
        X2 b1;
        split2_pair(sh[0], sh[1], b1, 0, sel);
        split2_pair(sh[2], sh[3], b1, 1, sel);
        split2_pair(geo[0], geo[1], b1, 2, sel);
        split2_pair(geo[2], geo[3], b1, 3, sel);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a1c[rb] = mfma_x2(frag_x2(w_color, rb, lane), b1, z4);
        X2 h0 = chain_relu_x2(a1c[0], a1c[1], sel), h1 = chain_relu_x2(a1c[2], a1c[3], sel);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          a2c[rb] = mfma_x2(frag_x2(w_color, 4 + 2 * rb, lane), h0, z4);
          a2c[rb] = mfma_x2(frag_x2(w_color, 5 + 2 * rb, lane), h1, a2c[rb]);
        }
        h0 = chain_relu_x2(a2c[0], a2c[1], sel);
        h1 = chain_relu_x2(a2c[2], a2c[3], sel);
        o3[0] = mfma_x2(frag_x2(w_color, 12, lane), h0, z4);
        o3[0] = mfma_x2(frag_x2(w_color, 13, lane), h1, o3[0]);
      } else if constexpr (HALF) {
        half8 b1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          b1[r] = (_Float16)sh[r];
          b1[4 + r] = (_Float16)geo[r];
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a1c[rb] = mfma_h(frag_h(w_color, rb, lane), b1, z4);
        half8 h0 = chain_relu_h(a1c[0], a1c[1]), h1 = chain_relu_h(a1c[2], a1c[3]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          a2c[rb] = mfma_h(frag_h(w_color, 4 + 2 * rb, lane), h0, z4);
          a2c[rb] = mfma_h(frag_h(w_color, 5 + 2 * rb, lane), h1, a2c[rb]);
        }
        h0 = chain_relu_h(a2c[0], a2c[1]);
        h1 = chain_relu_h(a2c[2], a2c[3]);
        o3[0] = mfma_h(frag_h(w_color, 12, lane), h0, z4);
        o3[0] = mfma_h(frag_h(w_color, 13, lane), h1, o3[0]);
        // the layer inputs the forward actually used (fp16-rounded): dW sees them
        sh = q4(sh);
        geo_c = q4(geo);
      } else {
        float xin[8] = {sh[0], sh[1], sh[2], sh[3], geo[0], geo[1], geo[2], geo[3]};
        mfma_layer<8, 4>(xin, [&](int rb, int ks) { return w_color[(rb * 8 + ks) * 64 + lane]; }, a1c);
        float hid[16];
        chain_relu(a1c, hid);
        mfma_layer<16, 4>(hid, [&](int rb, int ks) { return w_color[(COLOR_L1_FRAGS + rb * 16 + ks) * 64 + lane]; }, a2c);
        chain_relu(a2c, hid);
        mfma_layer<16, 1>(hid, [&](int, int ks) { return w_color[(COLOR_L1_FRAGS + COLOR_L2_FRAGS + ks) * 64 + lane]; }, o3);
      }
      // upstream gradients
      const float* di = b.di;
      f32x4 dy3 = z4;
      if (g == 0) {
        float dwsum = b.dd * zz / b.nrm;
        // image = sum_s w*rgb  ->  d_rgb = w*d_image, d_w += d_image . rgb
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float rgb = fast_sigmoid(o3[0][c]);  // as the forward
          dwsum += di[c] * rgb;
          dy3[c] = gs * (wgt * di[c] * rgb * (1.0f - rgb));
        }
        if (live) a.G[MARCH ? (size_t)smp : (size_t)ray * S + smp] = dwsum;
      }
      // L3: dW3 += dy3 (x) relu(a2c);  d_hid2 = W3^T dy3
      tile_store(dy_tile, g, j, 0, dy3);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        tile_store(x_tile, g, j, rb, HALF ? relu_q4(a2c[rb]) : relu4(a2c[rb]));
      cb_sync();
      if constexpr (B2) dw_accumulate_b2<1, 4>(dy_tile, x_tile, lane, dwc3, sel);
      else if constexpr (HALF) dw_accumulate_h<1, 4>(dy_tile, x_tile, lane, dwc3);
      else dw_accumulate<1, 4>(dy_tile, x_tile, lane, dwc3);
      cb_sync();
      f32x4 dh2[4];
      if constexpr (B2) {
        const X2 bd = chain_x2(dy3, z4, sel);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) dh2[rb] = mfma_x2(frag_x2(wt_color, rb, lane), bd, z4);
      } else if constexpr (HALF) {
        const half8 bd = chain_h(dy3, z4);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) dh2[rb] = mfma_h(frag_h(wt_color, rb, lane), bd, z4);
      } else {
        float bb[4] = {dy3[0], dy3[1], dy3[2], dy3[3]};
        mfma_layer<4, 4>(bb, [&](int rb, int ks) { return wt_color[(rb * 4 + ks) * 64 + lane]; }, dh2);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) dh2[rb] = gate4(a2c[rb], dh2[rb]);
      // L2: dW2 += d_hid2 (x) relu(a1c);  d_hid1 = W2^T d_hid2
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        tile_store(dy_tile, g, j, rb, dh2[rb]);
        tile_store(x_tile, g, j, rb, HALF ? relu_q4(a1c[rb]) : relu4(a1c[rb]));
      }
      cb_sync();
      if constexpr (B2) dw_accumulate_b2<4, 4>(dy_tile, x_tile, lane, dwc2, sel);
      else if constexpr (HALF) dw_accumulate_h<4, 4>(dy_tile, x_tile, lane, dwc2);
      else dw_accumulate<4, 4>(dy_tile, x_tile, lane, dwc2);
      cb_sync();
      f32x4 dh1[4];
      if constexpr (B2) {
        const X2 d0 = chain_x2(dh2[0], dh2[1], sel), d1 = chain_x2(dh2[2], dh2[3], sel);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          dh1[rb] = mfma_x2(frag_x2(wt_color, 4 + 2 * rb, lane), d0, z4);
          dh1[rb] = mfma_x2(frag_x2(wt_color, 5 + 2 * rb, lane), d1, dh1[rb]);
        }
      } else if constexpr (HALF) {
        const half8 d0 = chain_h(dh2[0], dh2[1]), d1 = chain_h(dh2[2], dh2[3]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          dh1[rb] = mfma_h(frag_h(wt_color, 4 + 2 * rb, lane), d0, z4);
          dh1[rb] = mfma_h(frag_h(wt_color, 5 + 2 * rb, lane), d1, dh1[rb]);
        }
      } else {
        float bb[16];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) bb[rb * 4 + r] = dh2[rb][r];
        mfma_layer<16, 4>(bb, [&](int rb, int ks) { return wt_color[(16 + rb * 16 + ks) * 64 + lane]; }, dh1);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) dh1[rb] = gate4(a1c[rb], dh1[rb]);
      // L1: dW1 += d_hid1 (x) [SH16 | geo15 | 1];  d_slot = W1^T(slot rows) d_hid1
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) tile_store(dy_tile, g, j, rb, dh1[rb]);
      tile_store(x_tile, g, j, 0, sh);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t m = 4 * g + r;
        x_tile[j * TILE_LD + (m == 0 ? 31u : 15u + m)] = geo_c[r];
      }
      cb_sync();
      if constexpr (B2) dw_accumulate_b2<4, 2>(dy_tile, x_tile, lane, dwc1, sel);
      else if constexpr (HALF) dw_accumulate_h<4, 2>(dy_tile, x_tile, lane, dwc1);
      else dw_accumulate<4, 2>(dy_tile, x_tile, lane, dwc1);
      cb_sync();
      if constexpr (B2) {
        dslot[0] = mfma_x2(frag_x2(wt_color, 12, lane), chain_x2(dh1[0], dh1[1], sel), z4);
        dslot[0] = mfma_x2(frag_x2(wt_color, 13, lane), chain_x2(dh1[2], dh1[3], sel), dslot[0]);
      } else if constexpr (HALF) {
        dslot[0] = mfma_h(frag_h(wt_color, 12, lane), chain_h(dh1[0], dh1[1]), z4);
        dslot[0] = mfma_h(frag_h(wt_color, 13, lane), chain_h(dh1[2], dh1[3]), dslot[0]);
      } else {
        float bb[16];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) bb[rb * 4 + r] = dh1[rb][r];
        mfma_layer<16, 1>(bb, [&](int, int ks) { return wt_color[(80 + ks) * 64 + lane]; }, dslot);
      }
    }

    // ========================== semantics net =============================
    if constexpr (DO_S) {
      f32x4 a1s[4], lg[NRB];
      f32x4 geo_s = geo;
      if constexpr (B2) {
        X2 bs;
        split2_pair(geo[0], geo[1], bs, 0, sel);
        split2_pair(geo[2], geo[3], bs, 1, sel);
        bs.t[0][2] = bs.t[0][3] = bs.t[1][2] = bs.t[1][3] = 0u;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a1s[rb] = mfma_x2(frag_x2(w_sem, rb, lane), bs, z4);
        const X2 h0 = chain_relu_x2(a1s[0], a1s[1], sel), h1 = chain_relu_x2(a1s[2], a1s[3], sel);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          lg[rb] = mfma_x2(frag_x2(w_sem, 4 + 2 * rb, lane), h0, z4);
          lg[rb] = mfma_x2(frag_x2(w_sem, 5 + 2 * rb, lane), h1, lg[rb]);
        }
      } else if constexpr (HALF) {
        half8 bs;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bs[r] = (_Float16)geo[r];
          bs[4 + r] = (_Float16)0.f;
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a1s[rb] = mfma_h(frag_h(w_sem, rb, lane), bs, z4);
        const half8 h0 = chain_relu_h(a1s[0], a1s[1]), h1 = chain_relu_h(a1s[2], a1s[3]);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          lg[rb] = mfma_h(frag_h(w_sem, 4 + 2 * rb, lane), h0, z4);
          lg[rb] = mfma_h(frag_h(w_sem, 5 + 2 * rb, lane), h1, lg[rb]);
        }
        geo_s = q4(geo);
      } else {
        float xs[4] = {geo[0], geo[1], geo[2], geo[3]};
        mfma_layer<4, 4>(xs, [&](int rb, int ks) { return w_sem[(rb * 4 + ks) * 64 + lane]; }, a1s);
        float hid[16];
        chain_relu(a1s, hid);
        mfma_layer<16, NRB>(hid, [&](int rb, int ks) { return w_sem[(SEM_L1_FRAGS + rb * 16 + ks) * 64 + lane]; }, lg);
      }
      // softmax
      float mx = -INFINITY;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((uint32_t)(rb * 16 + 4 * g + r) < C) mx = fast_max(mx, lg[rb][r]);
      mx = fast_max(mx, __shfl_xor(mx, 16, 64));
      mx = fast_max(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.0f;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (uint32_t)(rb * 16 + 4 * g + r) < C;
          const float ex = ok ? __expf(lg[rb][r] - mx) : 0.0f;  // as the forward (composite.hip fast_exp)
          lg[rb][r] = ex;
          sum += ex;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv_sum = fast_rcp(sum);  // as the forward
      // semantics = sum_s w_detached * p ; p = softmax(logits)
      float dot = 0.0f;
      f32x4 dlg[NRB];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t cls = rb * 16 + 4 * g + r;
          const float p = lg[rb][r] * inv_sum;
          const float dp = cls < C ? wgt * b.dsem[rb * 4 + r] : 0.0f;
          lg[rb][r] = p;
          dlg[rb][r] = dp;
          dot += p * dp;
        }
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dlg[rb][r] = gs * (lg[rb][r] * (dlg[rb][r] - dot));
      // L2: dW2 += dlogits (x) relu(a1s);  d_hid = W2^T dlogits
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) tile_store(dy_tile, g, j, rb, dlg[rb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        tile_store(x_tile, g, j, rb, HALF ? relu_q4(a1s[rb]) : relu4(a1s[rb]));
      cb_sync();
      if constexpr (B2) dw_accumulate_b2<NRB, 4>(dy_tile, x_tile, lane, dws2, sel);
      else if constexpr (HALF) dw_accumulate_h<NRB, 4>(dy_tile, x_tile, lane, dws2);
      else dw_accumulate<NRB, 4>(dy_tile, x_tile, lane, dws2);
      cb_sync();
      f32x4 dhs[4];
      if constexpr (B2) {
        X2 ld[NS];
#pragma unroll
        for (int sx = 0; sx < NS; ++sx)
          ld[sx] = chain_x2(dlg[2 * sx], (2 * sx + 1 < NRB) ? dlg[(2 * sx + 1 < NRB) ? 2 * sx + 1 : 0] : z4, sel);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          dhs[rb] = z4;
#pragma unroll
          for (int sx = 0; sx < NS; ++sx)
            dhs[rb] = mfma_x2(frag_x2(wt_sem, NS * rb + sx, lane), ld[sx], dhs[rb]);
        }
      } else if constexpr (HALF) {
        half8 ld[NS];
#pragma unroll
        for (int sx = 0; sx < NS; ++sx)
          ld[sx] = chain_h(dlg[2 * sx], (2 * sx + 1 < NRB) ? dlg[(2 * sx + 1 < NRB) ? 2 * sx + 1 : 0] : z4);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          dhs[rb] = z4;
#pragma unroll
          for (int sx = 0; sx < NS; ++sx)
            dhs[rb] = mfma_h(frag_h(wt_sem, NS * rb + sx, lane), ld[sx], dhs[rb]);
        }
      } else {
        float bb[4 * NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) bb[rb * 4 + r] = dlg[rb][r];
        mfma_layer<4 * NRB, 4>(bb, [&](int rb, int ks) { return wt_sem[(rb * 4 * NRB + ks) * 64 + lane]; }, dhs);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        dhs[rb] = gate4(a1s[rb], dhs[rb]);
        tile_store(dy_tile, g, j, rb, dhs[rb]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t m = 4 * g + r;
        x_tile[j * TILE_LD + (m == 0 ? 15u : m - 1u)] = geo_s[r];
      }
      cb_sync();
      if constexpr (B2) dw_accumulate_b2<4, 1>(dy_tile, x_tile, lane, dws1, sel);
      else if constexpr (HALF) dw_accumulate_h<4, 1>(dy_tile, x_tile, lane, dws1);
      else dw_accumulate<4, 1>(dy_tile, x_tile, lane, dws1);
      cb_sync();
      if constexpr (B2) {
        dslot_s[0] = mfma_x2(frag_x2(wt_sem, 4 * NS, lane), chain_x2(dhs[0], dhs[1], sel), z4);
        dslot_s[0] = mfma_x2(frag_x2(wt_sem, 4 * NS + 1, lane), chain_x2(dhs[2], dhs[3], sel), dslot_s[0]);
      } else if constexpr (HALF) {
        dslot_s[0] = mfma_h(frag_h(wt_sem, 4 * NS, lane), chain_h(dhs[0], dhs[1]), z4);
        dslot_s[0] = mfma_h(frag_h(wt_sem, 4 * NS + 1, lane), chain_h(dhs[2], dhs[3]), dslot_s[0]);
      } else {
        float bb[16];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) bb[rb * 4 + r] = dhs[rb][r];
        mfma_layer<16, 1>(bb, [&](int, int ks) { return wt_sem[(16 * NRB + ks) * 64 + lane]; }, dslot_s);
      }
    }

    if (live) {
      f32x4* dst = reinterpret_cast<f32x4*>((fine ? a.d_h_f : a.d_h_c) + hoff);
      f32x4 out;
      if constexpr (NET == 2) {
        // the colour kernel (launched first on the same stream) stored its
        // part; slot 0 stays what it is (k_weights_bwd writes it afterwards)
        const f32x4 prev = *dst;
        out[0] = (g == 0) ? prev[0] : prev[0] + inv_gs * dslot_s[0][0];
        out[1] = prev[1] + inv_gs * dslot_s[0][1];
        out[2] = prev[2] + inv_gs * dslot_s[0][2];
        out[3] = prev[3] + inv_gs * dslot_s[0][3];
      } else {
        out[0] = (g == 0) ? 0.0f : inv_gs * (dslot[0][0] + dslot_s[0][0]);  // slot 0: k_weights_bwd
        out[1] = inv_gs * (dslot[0][1] + dslot_s[0][1]);
        out[2] = inv_gs * (dslot[0][2] + dslot_s[0][2]);
        out[3] = inv_gs * (dslot[0][3] + dslot_s[0][3]);
      }
      *dst = out;
    }
  };

  // shade the full blocks of 16 waiting in the list (operands one block
  // ahead), keep the tail
  auto drain16 = [&]() {
    uint32_t head = 0;
    BlockOps cur, nxt;
    bool have = false;
    while (cnt - head >= 16) {
      if (!have) load_ops(head, 16u, cur);
      // one block ahead only in the f16 variant: measured 0.93 -> 0.89 ms there,
      // but 1.46 -> 1.61 ms for the fp32 kernel, whose 256 VGPR + 244 AGPR leave
      // no room for a second operand set
      const bool more = HALF && cnt - head >= 32;
      if (more) load_ops(head + 16, 16u, nxt);
      shade16(cur);
      if (more) cur = nxt;
      have = more;
      head += 16;
    }
    if (head) {
      const uint32_t rem = cnt - head;  // < 16
      float tw = 0.f, tz = 0.f;
      uint32_t tr = 0, ty = 0, ts = 0;
      if (lane < rem) {
        tw = lw[head + lane]; tr = lrow[head + lane]; ty = lray[head + lane];
        ts = lsmp[head + lane]; tz = lz[head + lane];
      }
      cb_sync();
      if (lane < rem) {
        lw[lane] = tw; lrow[lane] = tr; lray[lane] = ty; lsmp[lane] = ts; lz[lane] = tz;
      }
      cnt = rem;
      cb_sync();
    }
  };

  const uint64_t r_begin64 = gwave * a.rays_per_wave;
  if (r_begin64 < a.N) {
    const uint32_t r_begin = (uint32_t)r_begin64;
    const uint32_t r_end = (r_begin + a.rays_per_wave < a.N) ? r_begin + a.rays_per_wave : a.N;
    for (uint32_t r = r_begin; r < r_end; ++r) {
      if constexpr (MARCH) {
        const uint32_t index = (uint32_t)a.rays[3 * (size_t)r];
        const uint32_t offset = (uint32_t)a.rays[3 * (size_t)r + 1];
        uint32_t count = (uint32_t)a.rays[3 * (size_t)r + 2];
        if (offset + count >= a.n_points) count = 0;
        for (uint32_t sbase = 0; sbase < count; sbase += 64) {
          const uint32_t s = sbase + lane;
          const bool in = s < count;
          const size_t m = (size_t)offset + (in ? s : sbase);
          const float w = in ? a.weights[m] : 0.0f;
          const bool keep = in && w > a.w_min;
          if (DO_C && in && !keep) a.G[m] = 0.0f;
          const unsigned long long bal = __ballot(keep);
          if (keep) {
            const uint32_t pos = cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            lw[pos] = w;
            lrow[pos] = (uint32_t)m;
            lray[pos] = index;
            lsmp[pos] = (uint32_t)m;
            lz[pos] = a.t_all[m];
          }
          cnt += (uint32_t)__popcll(bal);
          cb_sync();
          drain16();
        }
      } else {
      for (uint32_t sbase = 0; sbase < S; sbase += 64) {
        const uint32_t s = sbase + lane;
        float w = 0.0f;
        bool keep = false;
        if (s < S) {
          w = a.weights[(size_t)r * S + s];
          keep = w > 1e-4f;
          if (DO_C && !keep) a.G[(size_t)r * S + s] = 0.0f;
        }
        const unsigned long long bal = __ballot(keep);
        if (keep) {
          const uint32_t pos = cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
          const uint32_t e = (uint32_t)a.src[(size_t)r * S + s];
          lw[pos] = w;
          lrow[pos] = e < T ? (r * T + e) : (ROW_FINE | (r * t + (e - T)));
          lray[pos] = r;
          lsmp[pos] = s;
          lz[pos] = e < T ? a.z_c[(size_t)r * T + e] : a.z_f[(size_t)r * t + (e - T)];
        }
        cnt += (uint32_t)__popcll(bal);
        cb_sync();
        drain16();
      }
      }
    }
    if (cnt) {
      BlockOps last;
      load_ops(0u, cnt, last);
      shade16(last);
    }
  }
  // per-wave partial gradients, tcnn layout
  if constexpr (DO_C) {
    if constexpr (HALF) {
      dw_scale(dwc1, inv_gs);
      dw_scale(dwc2, inv_gs);
      dw_scale(dwc3, inv_gs);
    }
    float* pc = a.partial_color + (size_t)gwave * 7168;
    dw_store<4, 2>(pc, 32, lane, dwc1);
    dw_store<4, 4>(pc + 2048, 64, lane, dwc2);
    dw_store<1, 4>(pc + 6144, 64, lane, dwc3);
  }
  if constexpr (DO_S) {
    if constexpr (HALF) {
      dw_scale(dws1, inv_gs);
      dw_scale(dws2, inv_gs);
    }
    float* ps = a.partial_sem + (size_t)gwave * (1024 + NRB * 1024);
    dw_store<4, 1>(ps, 16, lane, dws1);
    dw_store<NRB, 4>(ps + 1024, 64, lane, dws2);
  }
}

static inline uint32_t cb_pad16(uint32_t n) { return (n + 15u) / 16u * 16u; }

static size_t shade_bwd_weight_floats(bool half, uint32_t nrb) {
  if (half)
    return (size_t)(COLOR_H_FRAGS + SEM_H_FRAGS(nrb) + 14 + 4 * ((nrb + 1) / 2) + 2) * 256;
  return 7168 + 1024 + (size_t)nrb * 1024 + 6144 + (16 * (size_t)nrb + 16) * 64;
}

static void shade_bwd_geometry(uint32_t N, uint32_t& rpw, uint32_t& blocks,
                               uint32_t waves = CB_WAVES) {
  const uint64_t total_waves = 256ull * waves;
  rpw = (uint32_t)((N + total_waves - 1) / total_waves);
  if (rpw < 2) rpw = 2;
  const uint32_t n_waves = ucsa_div_up(N, rpw);
  blocks = ucsa_div_up(n_waves, waves);
}

// fp32 kernel as a per-net pair (NET = 1 then 2, two waves per SIMD each):
// UCSA_SHADE_BWD_SPLIT=0 keeps the single kernel (one wave per SIMD).
static bool shade_bwd_split() {
  static int v = -1;
  if (v < 0) {
    const char* e = ucsa_getenv("UCSA_SHADE_BWD_SPLIT");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v != 0;
}

// number of per-wave partial slots the caller must provide
extern "C" uint32_t ucsa_composite_bwd_parts(uint32_t N) {
  uint32_t rpw, blocks;
  const uint32_t waves = shade_bwd_split() ? CB_WAVES_SPLIT : CB_WAVES;
  shade_bwd_geometry(N ? N : 1, rpw, blocks, waves);
  return blocks * waves;
}

extern "C" uint32_t ucsa_composite_bwd_parts_f16(uint32_t N) {
  uint32_t rpw, blocks;
  const uint32_t waves = shade_bwd_split() ? CB_WAVES_SPLIT : CB_WAVES_H;
  shade_bwd_geometry(N ? N : 1, rpw, blocks, waves);
  return blocks * waves;
}

// ---------------------------------------------------------------------------
// k_weights_bwd: one wave per ray.
// ---------------------------------------------------------------------------
#define WB_WAVES 4
extern __shared__ __attribute__((aligned(16))) float wb_smem[];

__global__ void __launch_bounds__(64 * WB_WAVES)
k_weights_bwd(const float* __restrict__ z_c, const float* __restrict__ z_f,
              const float* __restrict__ sigma_c,
              const float* __restrict__ sigma_f,
              const int32_t* __restrict__ src, const float* __restrict__ weights,
              const float* __restrict__ G, uint32_t N, uint32_t T, uint32_t t,
              float density_scale, float* __restrict__ d_h_c,
              float* __restrict__ d_h_f) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t r = blockIdx.x * WB_WAVES + wid;
  if (r >= N) return;
  const uint32_t S = T + t;
  float* zm = wb_smem + (size_t)wid * 3 * S;
  float* sg = zm + S;
  float* suf = sg + S;
  for (uint32_t s = lane; s < S; s += 64) {
    const uint32_t e = (uint32_t)src[(size_t)r * S + s];
    zm[s] = e < T ? z_c[(size_t)r * T + e] : z_f[(size_t)r * t + (e - T)];
    sg[s] = e < T ? sigma_c[(size_t)r * T + e] : sigma_f[(size_t)r * t + (e - T)];
  }
  cb_sync();
  // reverse exclusive suffix sums of q_k = G_k * w_k
  float carry = 0.0f;
  for (uint32_t rb = 0; rb < S; rb += 64) {
    const uint32_t k = rb + lane;
    const bool ok = k < S;
    const uint32_t i = ok ? S - 1 - k : 0;
    const float q = ok ? G[(size_t)r * S + i] * weights[(size_t)r * S + i] : 0.0f;
    const float incl = wave_incl_scan_add_dpp(q);
    const float excl = wave_shift_up1(incl, 0.0f);
    if (ok) suf[i] = carry + excl;
    carry = carry + wave_last(incl);
  }
  cb_sync();
  float tcarry = 1.0f;
  const float lo = expf(-15.0f), hi = expf(15.0f);
  for (uint32_t sb = 0; sb < S; sb += 64) {
    const uint32_t s = sb + lane;
    float ex = 1.0f, alpha = 0.0f, delta = 0.0f;
    if (s < S) {
      delta = (s + 1 < S) ? zm[s + 1] - zm[s] : 1e10f;
      ex = expf(-delta * density_scale * sg[s]);
      alpha = 1.0f - ex;
    }
    const float fac = (s < S) ? (1.0f - alpha + 1e-15f) : 1.0f;
    const float incl = wave_incl_scan_mul_dpp(fac);
    const float excl = wave_shift_up1(incl, 1.0f);
    const float Ti = tcarry * excl;
    tcarry = tcarry * wave_last(incl);
    if (s < S) {
      const float Gi = G[(size_t)r * S + s];
      const float dalpha = Gi * Ti - suf[s] / fac;
      const float dsigma = dalpha * (ex * delta * density_scale);
      const float sig = sg[s];
      const float dh0 = dsigma * fminf(fmaxf(sig, lo), hi);
      const uint32_t e = (uint32_t)src[(size_t)r * S + s];
      if (e < T) d_h_c[((size_t)r * T + e) * 16] = dh0;
      else d_h_f[((size_t)r * t + (e - T)) * 16] = dh0;
    }
  }
}

// ---------------------------------------------------------------------------
static int32_t composite_bwd_impl(
    int mode, float f16_scale, const float* rays_d, const float* norms,
    const float* z_c, const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const int32_t* src,
    const float* weights, const float* packed_color, const float* packed_sem,
    const float* packed_color_t, const float* packed_sem_t,
    const float* d_image, const float* d_depth, const float* d_sem, uint32_t N,
    uint32_t T, uint32_t t, uint32_t n_classes, float density_scale, float* G,
    float* d_h_c, float* d_h_f, float* partial_color, float* partial_sem,
    void* stream) {
  UCSA_CHECK_ARG(rays_d && norms, 0);
  UCSA_CHECK_ARG(z_c && sigma_c && h_c, 2);
  UCSA_CHECK_ARG(t == 0 || (z_f && sigma_f && h_f), 5);
  UCSA_CHECK_ARG(src && weights, 8);
  UCSA_CHECK_ARG(packed_color && packed_sem && packed_color_t && packed_sem_t, 10);
  UCSA_CHECK_ARG(d_image && d_depth && d_sem, 14);
  UCSA_CHECK_ARG(T >= 1 && (uint64_t)N * T < 0x80000000ull, 18);
  UCSA_CHECK_ARG((uint64_t)N * t < 0x80000000ull && T + t <= 8192, 19);
  UCSA_CHECK_ARG(n_classes >= 1 && n_classes <= 61, 20);
  UCSA_CHECK_ARG(G && d_h_c && (t == 0 || d_h_f), 22);
  UCSA_CHECK_ARG(partial_color && partial_sem, 25);
  const bool half = mode == 1, x2 = mode == 2;
  UCSA_CHECK_ARG(!half || f16_scale > 0.f, 28);
  UCSA_CHECK_ARG(!x2 || shade_bwd_split(), 0);   // bf16x2 exists as the per-net pair only
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nrb = cb_pad16(n_classes) / 16;
  uint32_t rpw, blocks;
  const bool split = shade_bwd_split();
  const uint32_t waves = split ? CB_WAVES_SPLIT : half ? CB_WAVES_H : CB_WAVES;
  shade_bwd_geometry(N, rpw, blocks, waves);
  // d_h rows of samples outside the mask get no geo gradient
  hipError_t e = hipMemsetAsync(d_h_c, 0, (size_t)N * T * 16 * sizeof(float), s);
  if (e != hipSuccess) return -(int32_t)e;
  if (t) {
    e = hipMemsetAsync(d_h_f, 0, (size_t)N * t * 16 * sizeof(float), s);
    if (e != hipSuccess) return -(int32_t)e;
  }
  ShadeBwdArgs a{rays_d, norms, z_c, z_f, h_c, h_f, src, weights, d_image,
                 d_depth, d_sem, packed_color, packed_sem, packed_color_t,
                 packed_sem_t, N, T, t, n_classes, G, d_h_c, d_h_f,
                 partial_color, partial_sem, rpw, nullptr, nullptr, 0u, 0.0f,
                 half ? f16_scale : 1.0f};
  const size_t w_floats = shade_bwd_weight_floats(half, nrb);
  const size_t per_wave_b = (size_t)waves * (5 * CB_CAP + 2 * 16 * TILE_LD) * 4;
  const size_t smem = w_floats * 4 + per_wave_b;
  if (split) {
    const size_t ns = (nrb + 1) / 2;
    const size_t xt = x2 ? 2 : 1;
    const size_t smem_c = ((half || x2) ? (size_t)(COLOR_H_FRAGS + 14) * 256 * xt : 7168 + 6144) * 4 +
                          per_wave_b;
    const size_t smem_s = ((half || x2) ? (size_t)(SEM_H_FRAGS(nrb) + 4 * ns + 2) * 256 * xt
                                        : 1024 + (size_t)nrb * 1024 + (16 * (size_t)nrb + 16) * 64) * 4 +
                          per_wave_b;
#define LAUNCH_NET(NRB, H, NET, XX, SM)                                       \
  do {                                                                        \
    hipError_t e2 = hipFuncSetAttribute(                                      \
        reinterpret_cast<const void*>(&k_shade_bwd<NRB, false, H, NET, XX>),  \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SM));               \
    if (e2 != hipSuccess) return -(int32_t)e2;                                \
    UCSA_CLEAR_ERR();                                                         \
    hipLaunchKernelGGL((k_shade_bwd<NRB, false, H, NET, XX>), dim3(blocks),   \
                       dim3(64 * waves), (SM), s, a);                         \
    int32_t rc2 = ucsa_launch_status();                                       \
    if (rc2) return rc2;                                                      \
  } while (0)
#define LAUNCH_PAIR(NRB)                                                      \
  do {                                                                        \
    if (x2) { LAUNCH_NET(NRB, false, 1, true, smem_c); LAUNCH_NET(NRB, false, 2, true, smem_s); } \
    else if (half) { LAUNCH_NET(NRB, true, 1, false, smem_c); LAUNCH_NET(NRB, true, 2, false, smem_s); } \
    else { LAUNCH_NET(NRB, false, 1, false, smem_c); LAUNCH_NET(NRB, false, 2, false, smem_s); }    \
  } while (0)
    switch (nrb) {
      case 1: LAUNCH_PAIR(1); break;
      case 2: LAUNCH_PAIR(2); break;
      case 3: LAUNCH_PAIR(3); break;
      default: LAUNCH_PAIR(4); break;
    }
#undef LAUNCH_PAIR
#undef LAUNCH_NET
  } else {
#define LAUNCH(NRB, H)                                                        \
  do {                                                                        \
    hipError_t e2 = hipFuncSetAttribute(                                      \
        reinterpret_cast<const void*>(&k_shade_bwd<NRB, false, H>),           \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);               \
    if (e2 != hipSuccess) return -(int32_t)e2;                                \
    UCSA_CLEAR_ERR();                                                         \
    hipLaunchKernelGGL((k_shade_bwd<NRB, false, H>), dim3(blocks),            \
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
  }
  int32_t rc = ucsa_launch_status();
  if (rc) return rc;
  const size_t smem2 = (size_t)WB_WAVES * 3 * (T + t) * sizeof(float);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_weights_bwd, dim3(ucsa_div_up(N, WB_WAVES)),
                     dim3(64 * WB_WAVES), smem2, s, z_c, z_f, sigma_c, sigma_f,
                     src, weights, G, N, T, t, density_scale, d_h_c, d_h_f);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_composite_bwd(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const int32_t* src,
    const float* weights, const float* packed_color, const float* packed_sem,
    const float* packed_color_t, const float* packed_sem_t,
    const float* d_image, const float* d_depth, const float* d_sem, uint32_t N,
    uint32_t T, uint32_t t, uint32_t n_classes, float density_scale, float* G,
    float* d_h_c, float* d_h_f, float* partial_color, float* partial_sem,
    void* stream) {
  return composite_bwd_impl(0, 1.0f, rays_d, norms, z_c, sigma_c, h_c, z_f,
                            sigma_f, h_f, src, weights, packed_color,
                            packed_sem, packed_color_t, packed_sem_t, d_image,
                            d_depth, d_sem, N, T, t, n_classes, density_scale,
                            G, d_h_c, d_h_f, partial_color, partial_sem, stream);
}

extern "C" int32_t ucsa_composite_bwd_f16(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const int32_t* src,
    const float* weights, const void* packed_color_half,
    const void* packed_sem_half, const void* packed_color_t_half,
    const void* packed_sem_t_half, const float* d_image, const float* d_depth,
    const float* d_sem, uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes,
    float density_scale, float f16_scale, float* G, float* d_h_c, float* d_h_f,
    float* partial_color, float* partial_sem, void* stream) {
  return composite_bwd_impl(1, f16_scale, rays_d, norms, z_c, sigma_c, h_c,
                            z_f, sigma_f, h_f, src, weights,
                            (const float*)packed_color_half,
                            (const float*)packed_sem_half,
                            (const float*)packed_color_t_half,
                            (const float*)packed_sem_t_half, d_image, d_depth,
                            d_sem, N, T, t, n_classes, density_scale, G, d_h_c,
                            d_h_f, partial_color, partial_sem, stream);
}

// The colour / semantics backward with every contraction on the bf16 MFMA pipe
// as two-term splits (k_shade_bwd<.., B2>): fp32 range, 2^-16 per product.
// Weights: ucsa_mlp_pack_x3 / ucsa_mlp_pack_t_x3 buffers.  Needs the per-net
// kernel pair (UCSA_SHADE_BWD_SPLIT != 0).
extern "C" int32_t ucsa_composite_bwd_x2(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const int32_t* src,
    const float* weights, const void* packed_color_x3, const void* packed_sem_x3,
    const void* packed_color_t_x3, const void* packed_sem_t_x3,
    const float* d_image, const float* d_depth, const float* d_sem, uint32_t N,
    uint32_t T, uint32_t t, uint32_t n_classes, float density_scale, float* G,
    float* d_h_c, float* d_h_f, float* partial_color, float* partial_sem,
    void* stream) {
  return composite_bwd_impl(2, 1.0f, rays_d, norms, z_c, sigma_c, h_c, z_f, sigma_f,
                            h_f, src, weights, (const float*)packed_color_x3,
                            (const float*)packed_sem_x3,
                            (const float*)packed_color_t_x3,
                            (const float*)packed_sem_t_x3, d_image, d_depth, d_sem,
                            N, T, t, n_classes, density_scale, G, d_h_c, d_h_f,
                            partial_color, partial_sem, stream);
}

// ===========================================================================
// Marched training backward: the autograd of
// ucsa_march_train_fwd.
//   k_shade_bwd<MARCH>   : as above on the spans of rays [N,3]; G is [M]
//   k_march_weights_bwd  : one wave per ray:
//       d_sigma_i = delta_i * scale * (G_i * T_{i+1} - sum_{j>i} G_j w_j)
//     (the reference's composite_rays_train backward, raymarching.cu:468-474,
//     with G_i = dL/dw_i), times the trunc_exp backward, into d_h[:,0].
// ===========================================================================
#define MW_WAVES 4

__global__ void __launch_bounds__(64 * MW_WAVES)
k_march_weights_bwd(const int32_t* __restrict__ rays, uint32_t N, uint32_t M,
                    const float* __restrict__ sigmas, float sigma_scale,
                    const float* __restrict__ deltas,
                    const float* __restrict__ w_all, const float* __restrict__ G,
                    float* __restrict__ d_h) {
  __shared__ float suf_s[MW_WAVES][1024];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t r = blockIdx.x * MW_WAVES + wid;
  if (r >= N) return;
  const uint32_t offset = (uint32_t)rays[3 * (size_t)r + 1];
  const uint32_t count = (uint32_t)rays[3 * (size_t)r + 2];
  if (count == 0 || offset + count >= M) return;
  float* suf = suf_s[wid];
  float carry = 0.0f;
  for (uint32_t rb = 0; rb < count; rb += 64) {
    const uint32_t k = rb + lane;
    const bool ok = k < count;
    const uint32_t i = ok ? count - 1 - k : 0;
    const float q = ok ? G[(size_t)offset + i] * w_all[(size_t)offset + i] : 0.0f;
    const float incl = wave_incl_scan_add(q, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.0f;
    if (ok) suf[i] = carry + excl;
    carry = carry + wave_bcast(incl, 63);
  }
  cb_sync();
  float tcarry = 1.0f;
  const float lo = expf(-15.0f), hi = expf(15.0f);
  for (uint32_t sb = 0; sb < count; sb += 64) {
    const uint32_t s = sb + lane;
    const bool ok = s < count;
    const size_t m = (size_t)offset + (ok ? s : sb);
    const float sig = sigmas[m];
    const float delta = deltas[2 * m];
    const float ex = ok ? __expf(-sig * sigma_scale * delta) : 1.0f;
    const float incl = wave_incl_scan_mul(ex, lane);
    const float Tnext = tcarry * incl;  // transmittance after this sample
    tcarry = tcarry * wave_bcast(incl, 63);
    if (ok) {
      const float dsigma = delta * sigma_scale * (G[m] * Tnext - suf[s]);
      d_h[m * 16] = dsigma * fminf(fmaxf(sig, lo), hi);
    }
  }
}

extern "C" int32_t ucsa_march_train_bwd(
    const int32_t* rays, uint32_t N, uint32_t M, const float* rays_d,
    const float* norms, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const float* w_all, const float* t_all,
    const float* packed_color, const float* packed_sem,
    const float* packed_color_t, const float* packed_sem_t, uint32_t n_classes,
    float w_min, const float* d_image, const float* d_depth, const float* d_sem,
    float* G, float* d_h, float* partial_color, float* partial_sem,
    void* stream) {
  UCSA_CHECK_ARG(rays, 0);
  UCSA_CHECK_ARG(rays_d && norms, 3);
  UCSA_CHECK_ARG(M == 0 || (sigmas && h && deltas && w_all && t_all), 5);
  UCSA_CHECK_ARG(packed_color && packed_sem && packed_color_t && packed_sem_t, 11);
  UCSA_CHECK_ARG(n_classes >= 1 && n_classes <= 61, 15);
  UCSA_CHECK_ARG(w_min >= 0.f, 16);
  UCSA_CHECK_ARG(d_image && d_depth && d_sem, 17);
  UCSA_CHECK_ARG(M == 0 || (G && d_h), 20);
  UCSA_CHECK_ARG(partial_color && partial_sem, 22);
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nrb = cb_pad16(n_classes) / 16;
  uint32_t rpw, blocks;
  // same geometry (and so the same number of partial slots,
  // ucsa_composite_bwd_parts) as the live path: the per-net pair at two waves
  // per SIMD, or the single kernel under UCSA_SHADE_BWD_SPLIT=0
  const bool split = shade_bwd_split();
  const uint32_t waves = split ? CB_WAVES_SPLIT : CB_WAVES;
  shade_bwd_geometry(N, rpw, blocks, waves);
  if (M) {
    hipError_t e = hipMemsetAsync(d_h, 0, (size_t)M * 16 * sizeof(float), s);
    if (e != hipSuccess) return -(int32_t)e;
  }
  ShadeBwdArgs a{rays_d, norms, nullptr, nullptr, h, nullptr, nullptr, w_all,
                 d_image, d_depth, d_sem, packed_color, packed_sem,
                 packed_color_t, packed_sem_t, N, 0u, 0u, n_classes, G, d_h,
                 nullptr, partial_color, partial_sem, rpw, rays, t_all, M, w_min,
                 1.0f};
  const size_t per_wave_b = (size_t)waves * (5 * CB_CAP + 2 * 16 * TILE_LD) * 4;
  const size_t smem_c = (7168 + 6144) * 4 + per_wave_b;
  const size_t smem_s = (1024 + (size_t)nrb * 1024 + (16 * (size_t)nrb + 16) * 64) * 4 + per_wave_b;
  const size_t smem = smem_c + smem_s - per_wave_b;
#define LAUNCH_M(NRB, NET, SM)                                                \
  do {                                                                        \
    hipError_t e2 = hipFuncSetAttribute(                                      \
        reinterpret_cast<const void*>(&k_shade_bwd<NRB, true, false, NET>),   \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SM));               \
    if (e2 != hipSuccess) return -(int32_t)e2;                                \
    UCSA_CLEAR_ERR();                                                         \
    hipLaunchKernelGGL((k_shade_bwd<NRB, true, false, NET>), dim3(blocks),    \
                       dim3(64 * waves), (SM), s, a);                         \
    int32_t rc2 = ucsa_launch_status();                                       \
    if (rc2) return rc2;                                                      \
  } while (0)
#define LAUNCH_MM(NRB)                                                        \
  do {                                                                        \
    if (split) { LAUNCH_M(NRB, 1, smem_c); LAUNCH_M(NRB, 2, smem_s); }        \
    else LAUNCH_M(NRB, 0, smem);                                              \
  } while (0)
  switch (nrb) {
    case 1: LAUNCH_MM(1); break;
    case 2: LAUNCH_MM(2); break;
    case 3: LAUNCH_MM(3); break;
    default: LAUNCH_MM(4); break;
  }
#undef LAUNCH_MM
#undef LAUNCH_M
  int32_t rc = ucsa_launch_status();
  if (rc) return rc;
  if (M == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_march_weights_bwd, dim3(ucsa_div_up(N, MW_WAVES)),
                     dim3(64 * MW_WAVES), 0, s, rays, N, M, sigmas, sigma_scale,
                     deltas, w_all, G, d_h);
  return ucsa_launch_status();
}
