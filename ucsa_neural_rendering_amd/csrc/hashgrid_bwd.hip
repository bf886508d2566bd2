// Hash-grid backward: d_feat -> table gradient (restates the autograd of
// tcnn's GridEncoding; call site reference
// nr4seg/nerf/network_tcnn_semantics.py:133-134).
//
// Two regimes, picked per level on the host (ucsa_hashgrid_bwd_rays):
//  * coarse levels (dense, or cells wider than ~2 sample spacings): lanes of
//    a wave are consecutive, ASCENDING samples of a ray (ucsa_resample sorts
//    its uniforms), so neighbours fall into the same cell; equal indices in
//    consecutive lanes are summed by a segmented wave scan and only the last
//    lane of each run issues fp32 atomics (k_hashgrid_bwd<true>);
//  * fine hashed levels: updates are spread over the whole 4 MiB slab and
//    random fp32 atomics retire at only ~21 G ops/s chip-wide, so records are
//    binned by table slice and summed in LDS (k_grid_bwd_bin / _accum below).
// Both use float atomics somewhere (global or LDS), so the table gradient is
// order-dependent at fp32 round-off (run-to-run ~1e-7 relative); the MLP
// gradients are reduced in a fixed order and are bit-reproducible.
#include <cstdlib>
#include <mutex>
#include "ucsa_common.h"
#include "wave_ops.h"

#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u

__device__ __forceinline__ uint32_t grid_index_b(uint32_t x, uint32_t y,
                                                 uint32_t z, uint32_t res,
                                                 uint32_t entries,
                                                 uint32_t hashed) {
  uint32_t idx = hashed ? (x ^ (y * PRIME_Y) ^ (z * PRIME_Z))
                        : (x + y * res + z * res * res);
  return hashed ? (idx & (entries - 1)) : (idx % entries);
}

// (same one-instruction clamp and exact-reciprocal position as the forward,
// hashgrid_common.h: the backward must land in the forward's cells)
__device__ __forceinline__ float clampf_b(float v, float lo, float hi) {
  return __builtin_amdgcn_fmed3f(v, lo, hi);
}
__device__ __forceinline__ float unit_inv_b(float two_b) {
  return (__float_as_uint(two_b) & 0x007FFFFFu) == 0u ? 1.0f / two_b : 0.0f;
}
__device__ __forceinline__ float to_unit_b(float p, float bound, float two_b, float inv) {
  return inv != 0.0f ? (p + bound) * inv : (p + bound) / two_b;
}

// Where a kernel's sample m comes from.  Single pass (src == nullptr): sample
// m of z [N*T] / d_feat [L][N*T]; z == nullptr: explicit points (rays_o = x).
// MERGED (src != nullptr, round 4): m = r * S + s walks the S = Tc + Tf samples
// of ray r in SORTED depth order -- src[m] = e < Tc: coarse sample e of the
// ray (z, d_feat), else fine sample e - Tc (z_f, d_feat_f) -- the order the
// forward composited them in.  Consecutive lanes are then consecutive samples
// ALONG the ray whichever pass they came from: on the coarse levels the fine
// samples fall into cells the coarse samples of the same ray already visit, so
// the merged walk combines them into the same runs and the fine pass adds
// (almost) no accumulator traffic of its own; one launch instead of two.
struct MergedSrc {
  const int32_t* src;
  const float* z_f;
  const float2* d_feat_f;
  uint32_t Tc, Tf, N;
};

__device__ __forceinline__ void sample_ref(const MergedSrc& mg, uint32_t T, uint64_t M,
                                           uint64_t m, uint32_t level,
                                           const float* __restrict__ zs,
                                           const float2* __restrict__ d_feat,
                                           uint32_t& r, float& zz, float2& df) {
  if (mg.src) {
    r = (uint32_t)(m / T);  // T = S here
    const uint32_t e = (uint32_t)mg.src[m];
    if (e < mg.Tc) {
      const uint64_t i = (uint64_t)r * mg.Tc + e;
      zz = zs[i];
      df = d_feat[(uint64_t)level * ((uint64_t)mg.N * mg.Tc) + i];
    } else {
      const uint64_t i = (uint64_t)r * mg.Tf + (e - mg.Tc);
      zz = mg.z_f[i];
      df = mg.d_feat_f[(uint64_t)level * ((uint64_t)mg.N * mg.Tf) + i];
    }
  } else {
    r = zs ? (uint32_t)(m / T) : 0u;
    zz = zs ? zs[m] : 0.0f;
    df = d_feat[(uint64_t)level * M + m];
  }
}

// Run-combining: lanes of a wave are consecutive samples of a ray, so on the
// coarse levels consecutive lanes fall into the same cell and would hammer the
// same table entry (measured before this: 85 % of a training step).  Equal
// indices in CONSECUTIVE lanes are summed with a segmented wave scan and only
// the last lane of each run issues the atomic.  Non-adjacent duplicates just
// cost an extra atomic.
// Run structure of a wave: all 8 corners of a sample share the sample's cell,
// so the run heads (cell differs from the previous lane's) and the scan's
// "may add from lane-d" predicates are computed once and reused by the 16
// value scans.
struct RunPlan {
  bool add[6];
  bool tail;
  uint32_t live;  // bit s: some lane of the wave adds at step s (wave-uniform)
};

// The scans run on the DPP data path (round 6; rounds 2-5 used __shfl_up, i.e.
// ds_bpermute_b32 -- an LDS instruction with its latency in a dependent chain, two
// per live step for each of the 16 values of a sample; a DPP step is one VALU
// instruction.  Measured in round 6, tools/bwd_switches_ab.py: merged grid backward
// 1.420 -> 1.361 ms with the 8-byte records, 1.205 -> 1.162 ms with x-pair records;
// rel L2 5e-8 against the shuffle ladder, tests/test_dpp_run_scan_model_cpu.py is
// its numpy model).  Ladder: row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, row_bcast15 into
// rows 1 and 3, row_bcast31 into rows 2 and 3 -- a segmented scan under the
// associative operator (f1, v1) o (f2, v2) = (f1 | f2, f2 ? v2 : v1 + v2), so the
// tail lane of a run ends up with the run's sum as with the shuffle ladder (other
// association of the fp32 additions).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_src_i(int v, int identity) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_src_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ RunPlan run_plan(uint32_t cx, uint32_t cy, uint32_t cz,
                                                bool act, uint32_t lane) {
  const uint32_t kx = act ? cx : 0xFFFFFFFFu;
  // previous lane's cell (wave_shr:1; lane 0 is a head anyway)
  const uint32_t px = (uint32_t)dpp_src_i<0x138, 0xf>((int)kx, 0);
  const uint32_t py = (uint32_t)dpp_src_i<0x138, 0xf>((int)cy, 0);
  const uint32_t pz = (uint32_t)dpp_src_i<0x138, 0xf>((int)cz, 0);
  const bool head = lane == 0 || px != kx || py != cy || pz != cz || !act;
  int f = head ? 1 : 0;
  RunPlan p;
  p.live = 0;
  const uint32_t in_row = lane & 15u;
  auto step = [&](int s, int of, bool has_src) {
    p.add[s] = has_src && !f;
    if (p.add[s]) f |= of;
    if (__any(p.add[s])) p.live |= 1u << s;
  };
  step(0, dpp_src_i<0x111, 0xf>(f, 1), in_row >= 1u);          // row_shr:1
  step(1, dpp_src_i<0x112, 0xf>(f, 1), in_row >= 2u);          // row_shr:2
  step(2, dpp_src_i<0x114, 0xf>(f, 1), in_row >= 4u);          // row_shr:4
  step(3, dpp_src_i<0x118, 0xf>(f, 1), in_row >= 8u);          // row_shr:8
  step(4, dpp_src_i<0x142, 0xa>(f, 1), ((lane >> 4) & 1u) != 0u);  // row_bcast:15 -> rows 1, 3
  step(5, dpp_src_i<0x143, 0xc>(f, 1), lane >= 32u);           // row_bcast:31 -> rows 2, 3
  const int next_head = __shfl_down(head ? 1 : 0, 1, 64);
  p.tail = act && (lane == 63 || next_head);
  return p;
}

__device__ __forceinline__ void run_sum(const RunPlan& p, float& vx, float& vy) {
#define UCSA_RUN_STEP(S, CTRL, MASK)                         \
  if ((p.live >> S) & 1u) {                                  \
    const float ox = dpp_src_f<CTRL, MASK>(vx);              \
    const float oy = dpp_src_f<CTRL, MASK>(vy);              \
    if (p.add[S]) {                                          \
      vx += ox;                                              \
      vy += oy;                                              \
    }                                                        \
  }
  UCSA_RUN_STEP(0, 0x111, 0xf)
  UCSA_RUN_STEP(1, 0x112, 0xf)
  UCSA_RUN_STEP(2, 0x114, 0xf)
  UCSA_RUN_STEP(3, 0x118, 0xf)
  UCSA_RUN_STEP(4, 0x142, 0xa)
  UCSA_RUN_STEP(5, 0x143, 0xc)
#undef UCSA_RUN_STEP
}

// (x, y) += (vx, vy) on an 8-byte aligned pair of LDS floats.  ds_add_f32 is
// serialised per lane on gfx950 (measured, tools/ubench/lds_atomic.hip: two of
// them per record retire at 100 G records/s chip-wide, integer LDS atomics at
// 1670 G/s); one 64-bit compare-and-swap loop on the pair does 800 G/s and is
// the same arithmetic (plain fp32 adds in arrival order).
__device__ __forceinline__ void lds_add_pair(float* pair, float vx, float vy) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(pair);
  unsigned long long seen = *reinterpret_cast<volatile unsigned long long*>(q);
  unsigned long long expect;
  do {
    expect = seen;
    const float sx = __uint_as_float((uint32_t)expect) + vx;
    const float sy = __uint_as_float((uint32_t)(expect >> 32)) + vy;
    seen = atomicCAS(q, expect,
                     (unsigned long long)__float_as_uint(sx) |
                         ((unsigned long long)__float_as_uint(sy) << 32));
  } while (seen != expect);
}

// Workgroup-private LDS accumulator (open addressing, keyed by table index).
// All rays of a training batch start in the camera's cell, so on the coarse
// levels a handful of table entries receive an update from every ray: as
// global atomics those serialise on one L2 line (measured: 1.2 ms for 393 k
// samples, independent of the sample count).  Here every workgroup first sums
// its ~8 k samples into LDS (same-address LDS atomics are cheap) and flushes
// each distinct entry once; entries that do not fit go straight to memory.
// 2^11 slots = 24 KiB of LDS: six workgroups per CU instead of three with 2^12
// (the step -1.2 %; 2^10 and fewer tiles per workgroup measured no better)
#ifndef ACC_BITS
#define ACC_BITS 11
#endif
#define ACC_SLOTS (1u << ACC_BITS)
#define ACC_EMPTY 0xFFFFFFFFu
#define ACC_TILES 32  // 256-sample tiles per workgroup

__device__ __forceinline__ void lds_accumulate(uint32_t* keys, float* vals,
                                               uint32_t idx, float vx, float vy,
                                               float* gt) {
  uint32_t slot = (idx * 2654435761u) >> (32 - ACC_BITS);
#pragma unroll 1
  for (int probe = 0; probe < 4; ++probe) {
    const uint32_t old = atomicCAS(&keys[slot], ACC_EMPTY, idx);
    if (old == ACC_EMPTY || old == idx) {
      lds_add_pair(&vals[2 * slot], vx, vy);
      return;
    }
    slot = (slot + 1) & (ACC_SLOTS - 1);
  }
  atomicAdd(gt + (size_t)idx * 2, vx);  // table crowded: not a hot entry
  atomicAdd(gt + (size_t)idx * 2 + 1, vy);
}

template <bool RUNRED>
__global__ void __launch_bounds__(256)
k_hashgrid_bwd(GridDev g, const float* __restrict__ rays_o,
               const float* __restrict__ rays_d, const float* __restrict__ zs,
               Aabb bb, uint32_t T, uint64_t M, uint32_t level0,
               uint32_t tiles, const float2* __restrict__ d_feat,
               float* __restrict__ grad_table, MergedSrc mg) {
  __shared__ uint32_t acc_keys[RUNRED ? ACC_SLOTS : 1];
  __shared__ __attribute__((aligned(8))) float acc_vals[RUNRED ? 2 * ACC_SLOTS : 2];
  const uint32_t level = level0 + blockIdx.y;
  const uint32_t lane = threadIdx.x & 63u;
  float* gt = grad_table + (size_t)g.offset[level] * 2;
  const uint32_t res = g.res[level], entries = g.entries[level],
                 hashed = g.hashed[level];
  const float two_b = 2.0f * g.bound, inv_b = unit_inv_b(two_b);
  const float scale = g.scale[level];
  if (RUNRED) {
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256) {
      acc_keys[i] = ACC_EMPTY;
      acc_vals[2 * i] = 0.f;
      acc_vals[2 * i + 1] = 0.f;
    }
    __syncthreads();
  }
  const int n_tiles = RUNRED ? (int)tiles : 1;
  for (int tile = 0; tile < n_tiles; ++tile) {
    uint64_t m = ((uint64_t)blockIdx.x * n_tiles + tile) * 256 + threadIdx.x;
    const bool in_range = m < M;
    if (RUNRED && __syncthreads_or(in_range) == 0) break;
    if (!in_range) m = M - 1;
    float2 df;
    uint32_t r;
    float zz;
    sample_ref(mg, T, M, m, level, zs, d_feat, r, zz, df);
    const bool act = in_range && !(df.x == 0.0f && df.y == 0.0f);
    if (!RUNRED && !act) return;
    float px, py, pz;
    if (zs) {
      const float* o = rays_o + (size_t)r * 3;
      const float* d = rays_d + (size_t)r * 3;
      px = clampf_b(o[0] + d[0] * zz, bb.lo[0], bb.hi[0]);
      py = clampf_b(o[1] + d[1] * zz, bb.lo[1], bb.hi[1]);
      pz = clampf_b(o[2] + d[2] * zz, bb.lo[2], bb.hi[2]);
    } else {  // explicit points (ucsa_hashgrid_bwd_points): rays_o = x [M,3]
      const float* xp = rays_o + (size_t)m * 3;
      px = xp[0];
      py = xp[1];
      pz = xp[2];
    }
    const float x = to_unit_b(px, g.bound, two_b, inv_b) * scale + 0.5f;
    const float y = to_unit_b(py, g.bound, two_b, inv_b) * scale + 0.5f;
    const float z = to_unit_b(pz, g.bound, two_b, inv_b) * scale + 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
    const float wx = x - fx0, wy = y - fy0, wz = z - fz0;
    const uint32_t gx = (uint32_t)(int32_t)fx0, gy = (uint32_t)(int32_t)fy0,
                   gz = (uint32_t)(int32_t)fz0;
    RunPlan plan;
    if (RUNRED) plan = run_plan(gx, gy, gz, act, lane);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float w = (c & 1) ? wx : 1.0f - wx;
      w = w * ((c & 2) ? wy : 1.0f - wy);
      w = w * ((c & 4) ? wz : 1.0f - wz);
      const uint32_t idx = grid_index_b(gx + (c & 1), gy + ((c >> 1) & 1),
                                        gz + ((c >> 2) & 1), res, entries, hashed);
      float vx = w * df.x, vy = w * df.y;
      if (RUNRED) {
        if (!act) { vx = 0.f; vy = 0.f; }
        run_sum(plan, vx, vy);
        if (plan.tail) lds_accumulate(acc_keys, acc_vals, idx, vx, vy, gt);
      } else {
        atomicAdd(gt + (size_t)idx * 2, vx);
        atomicAdd(gt + (size_t)idx * 2 + 1, vy);
      }
    }
  }
  if (RUNRED) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ACC_SLOTS; i += 256) {
      const uint32_t k = acc_keys[i];
      if (k != ACC_EMPTY) {
        atomicAdd(gt + (size_t)k * 2, acc_vals[2 * i]);
        atomicAdd(gt + (size_t)k * 2 + 1, acc_vals[2 * i + 1]);
      }
    }
  }
}

// ===========================================================================
// Binned two-pass backward (default when the caller provides a workspace).
//
// Measured on MI355X (tools/ubench/atomic_scatter.hip): random fp32 atomics
// retire at ~21 G ops/s chip-wide whatever their scope -> 10.5 G table entries
// per second (two floats each), i.e. ~0.8 ms per level for the 8.4 M corner
// updates of a 4096 x 256 training batch; 85 % of a training step.
// Instead:
//   pass 1  every workgroup turns its 2048 samples of one level into
//           (entry, vx, vy) records and appends them to 256 per-level bins
//           (bin = contiguous slice of the level's table): LDS histogram, one
//           global reservation per (workgroup, bin), 16-byte record stores;
//   pass 2  one workgroup per (level, bin) streams its records and sums them
//           into the bin's slice held in LDS (lds_add_pair), then adds the slice
//           to the gradient with plain stores -- the slice is owned by exactly
//           one workgroup, no global float atomics at all.
// A bin that overflows its capacity falls back to direct atomics for the
// excess records, so results never depend on the capacity guess.
// ===========================================================================
#define BIN_COUNT 256
#define BIN_TILE 8  // samples per thread in pass 1

struct BinGeom {
  uint32_t bin_size[UCSA_MAX_LEVELS];  // table entries per bin
  uint32_t bin_shift[UCSA_MAX_LEVELS]; // log2(bin_size) when a power of two, else 32
  uint32_t loc_bits[UCSA_MAX_LEVELS];  // bits that hold an entry index inside a bin
  uint32_t cap;                        // records per bin
};

// Packed ("p64") record words: one 64-bit word = entry index inside the bin (L = loc_bits)
// | vx | vy, each value the fp32 rounded (to nearest even) to its top
// V = min(32, (64 - L) / 2) bits.  For the 2^19-entry levels of the reference's
// grid L = 11, V = 26: sign, exponent and 17 mantissa bits, a relative error of
// 2^-18 per record -- finer than the two-term bf16 split (2^-16) of the MLP
// backward that produces the values; the sums stay fp32.  Half the bytes of the
// 16-byte records through HBM in both passes.  Non-finite values stay
// non-finite (a NaN may become an infinity), so overflow detection still sees them.
__device__ __forceinline__ uint32_t p64_value_bits(uint32_t L) {
  const uint32_t v = (64u - L) >> 1;
  return v > 32u ? 32u : v;
}
__device__ __forceinline__ uint32_t p64_round(float f, uint32_t V) {
  uint32_t b = __float_as_uint(f);
  const uint32_t drop = 32u - V;
  if (drop) {
    // non-finite values are truncated, not rounded: the rounding add would
    // carry a NaN's all-ones payload into the sign bit (0x7FFFFFFF -> -0.0) or
    // wrap it around (0xFFFFFFFF -> +0.0) -- ADVICE r4.  A finite value that
    // rounds past the largest one becomes an infinity, as it should.
    const bool nonfinite = (b & 0x7F800000u) == 0x7F800000u;
    const uint32_t r = nonfinite ? b : b + ((1u << (drop - 1u)) - 1u) + ((b >> drop) & 1u);
    b = r >> drop;
  }
  return b;
}
__device__ __forceinline__ uint64_t p64_pack(uint32_t loc, float vx, float vy, uint32_t L) {
  const uint32_t V = p64_value_bits(L);
  return (uint64_t)loc | ((uint64_t)p64_round(vx, V) << L) |
         ((uint64_t)p64_round(vy, V) << (L + V));
}
__device__ __forceinline__ void p64_unpack(uint64_t w, uint32_t L, uint32_t& loc,
                                           float& vx, float& vy) {
  const uint32_t V = p64_value_bits(L), drop = 32u - V;
  const uint64_t vmask = V == 32u ? 0xFFFFFFFFull : ((1ull << V) - 1ull);
  loc = (uint32_t)(w & ((1ull << L) - 1ull));
  vx = __uint_as_float((uint32_t)((w >> L) & vmask) << drop);
  vy = __uint_as_float((uint32_t)((w >> (L + V)) & vmask) << drop);
}

__device__ __forceinline__ void sample_cell(const GridDev& g, uint32_t level,
                                            const float* __restrict__ rays_o,
                                            const float* __restrict__ rays_d,
                                            bool from_rays, uint32_t r, float zz,
                                            const Aabb& bb, uint64_t m,
                                            uint32_t (&gi)[3], float (&wf)[3]) {
  const float two_b = 2.0f * g.bound, inv_b = unit_inv_b(two_b);
  const float scale = g.scale[level];
  float pos[3];
  if (from_rays) {
    const float* o = rays_o + (size_t)r * 3;
    const float* d = rays_d + (size_t)r * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      pos[a] = clampf_b(o[a] + d[a] * zz, bb.lo[a], bb.hi[a]);
  } else {  // explicit points: rays_o = x [M,3]
    const float* xp = rays_o + (size_t)m * 3;
    pos[0] = xp[0];
    pos[1] = xp[1];
    pos[2] = xp[2];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = pos[a];
    const float x = to_unit_b(p, g.bound, two_b, inv_b) * scale + 0.5f;
    const float f0 = floorf(x);
    wf[a] = x - f0;
    gi[a] = (uint32_t)(int32_t)f0;
  }
}

// HREC: 8-byte records (entry index inside the bin | the value pair as half2,
// multiplied by rec_scale) instead of 16-byte ones -- half the record traffic
// of both passes.  Used by the f16 training mode only (train_precision="fp16":
// tiny-cuda-nn itself accumulates its grid gradient from half2 values); the
// sums in the accumulate pass stay fp32.
// REC: 0 = 16-byte records, REC_H16 = HREC above.  (The packed 64-bit words of
// round 4, "REC_P64", travel as x-PAIR records since round 6: k_grid_bwd_bin_xpair.)
#define REC_F32 0
#define REC_H16 1
template <int REC>
__global__ void __launch_bounds__(256)
k_grid_bwd_bin(GridDev g, BinGeom bg, uint32_t level0,
               const float* __restrict__ rays_o,
               const float* __restrict__ rays_d, const float* __restrict__ zs,
               Aabb bb, uint32_t T, uint64_t M,
               const float2* __restrict__ d_feat,
               uint32_t* __restrict__ gcount, void* __restrict__ records_v,
               float* __restrict__ grad_table, float rec_scale, MergedSrc mg) {
  constexpr bool HREC = REC == REC_H16;
  float4* records = reinterpret_cast<float4*>(records_v);
  uint2* records_h = reinterpret_cast<uint2*>(records_v);
  static_assert(BIN_COUNT == 256, "one thread per bin");
  __shared__ uint32_t hist[BIN_COUNT], base[BIN_COUNT], cursor[BIN_COUNT];
  __shared__ uint32_t it_cnt[BIN_COUNT], it_off[BIN_COUNT], wave_tot[4];
  // one iteration's records, bin-sorted: float4 (loc, vx, vy, bin)
  __shared__ float4 stage[256 * 8];
  const uint32_t level = level0 + blockIdx.y;
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t res = g.res[level], entries = g.entries[level],
                 hashed = g.hashed[level], bsz = bg.bin_size[level],
                 bshift = bg.bin_shift[level];
  auto bin_of = [&](uint32_t idx) -> uint32_t {
    return bshift < 32 ? idx >> bshift : idx / bsz;
  };
  hist[threadIdx.x] = 0;
  cursor[threadIdx.x] = 0;
  it_cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t m0 = (uint64_t)blockIdx.x * (256 * BIN_TILE) + threadIdx.x;
  // pass 1: the workgroup's records per bin -> one global reservation per bin
  // Consecutive threads are consecutive samples of a ray (or consecutive
  // marched points): where they fall into the same cell -- importance samples
  // clustered on a surface, the small steps of a marcher near the camera --
  // their updates are summed by a segmented wave scan first and only the last
  // lane of a run emits records (same plan in both passes).
  // (not unrolled: the fully unrolled pass made the kernel ~80 KB of code, more
  // than the 64 KB instruction cache two CUs share; rolled: -1 % of the step.
  // A version with BOTH 32- and 64-bit index paths in the code was +3 %.)
#pragma unroll 1
  for (int it = 0; it < BIN_TILE; ++it) {
    const uint64_t m = m0 + (uint64_t)it * 256;
    bool act = m < M;
    uint32_t r = 0;
    float zz = 0.f;
    if (act) {
      float2 df;
      sample_ref(mg, T, M, m, level, zs, d_feat, r, zz, df);
      act = !(df.x == 0.0f && df.y == 0.0f);
    }
    uint32_t gi[3] = {0u, 0u, 0u};
    float wf[3];
    if (act) sample_cell(g, level, rays_o, rays_d, zs != nullptr, r, zz, bb, m, gi, wf);
    const RunPlan plan = run_plan(gi[0], gi[1], gi[2], act, lane);
    if (!(act && plan.tail)) continue;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t idx = grid_index_b(gi[0] + (c & 1), gi[1] + ((c >> 1) & 1),
                                        gi[2] + ((c >> 2) & 1), res, entries, hashed);
      atomicAdd(&hist[bin_of(idx)], 1u);
    }
  }
  __syncthreads();
  {
    const uint32_t h = hist[threadIdx.x];
    base[threadIdx.x] = h ? atomicAdd(&gcount[level * BIN_COUNT + threadIdx.x], h) : 0u;
  }
  __syncthreads();
  float* gt = grad_table + (size_t)g.offset[level] * 2;
  float4* rec_level = records + (size_t)level * BIN_COUNT * bg.cap;
  // pass 2, 256 samples at a time: the 2048 records are counting-sorted by bin
  // in LDS and stored from there, so that consecutive lanes write consecutive
  // records of a bin (scattered 16-byte stores retire at ~180 G/s chip-wide,
  // like divergent gathers: 0.47 of this kernel's 0.64 ms before).
#pragma unroll 1
  for (int it = 0; it < BIN_TILE; ++it) {
    const uint64_t m = m0 + (uint64_t)it * 256;
    bool act = m < M;
    float2 df = make_float2(0.f, 0.f);
    uint32_t r = 0;
    float zz = 0.f;
    if (act) {
      sample_ref(mg, T, M, m, level, zs, d_feat, r, zz, df);
      act = !(df.x == 0.0f && df.y == 0.0f);
    }
    uint32_t key[8];   // bin << 16 | rank inside the bin (this iteration)
    uint32_t loc[8];
    float valx[8], valy[8];
    uint32_t gi[3] = {0u, 0u, 0u};
    float wf[3] = {0.f, 0.f, 0.f};
    if (act) sample_cell(g, level, rays_o, rays_d, zs != nullptr, r, zz, bb, m, gi, wf);
    const RunPlan plan = run_plan(gi[0], gi[1], gi[2], act, lane);
    const bool has_runs = plan.live != 0;  // some lane continues a run
    const bool emit = act && plan.tail;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float w = (c & 1) ? wf[0] : 1.0f - wf[0];
      w = w * ((c & 2) ? wf[1] : 1.0f - wf[1]);
      w = w * ((c & 4) ? wf[2] : 1.0f - wf[2]);
      float vx = act ? w * df.x : 0.0f, vy = act ? w * df.y : 0.0f;
      if (has_runs) {
        run_sum(plan, vx, vy);
      }
      valx[c] = vx;
      valy[c] = vy;
      if (emit) {
        const uint32_t idx = grid_index_b(gi[0] + (c & 1), gi[1] + ((c >> 1) & 1),
                                          gi[2] + ((c >> 2) & 1), res, entries, hashed);
        const uint32_t bin = bin_of(idx);
        loc[c] = idx - bin * bsz;
        key[c] = (bin << 16) | atomicAdd(&it_cnt[bin], 1u);
      }
    }
    __syncthreads();
    {  // exclusive prefix of it_cnt over the bins (thread = bin)
      const uint32_t v = it_cnt[threadIdx.x];
      const uint32_t inc = wave_incl_scan_add_u32(v, lane);
      if (lane == 63) wave_tot[wid] = inc;
      __syncthreads();
      uint32_t before = 0;
      for (uint32_t w = 0; w < wid; ++w) before += wave_tot[w];
      it_off[threadIdx.x] = before + inc - v;
    }
    __syncthreads();
    if (emit) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t bin = key[c] >> 16;
        const uint32_t slot = it_off[bin] + (key[c] & 0xFFFFu);
        stage[slot] = make_float4(__uint_as_float(loc[c]), valx[c], valy[c],
                                  __uint_as_float(bin));
      }
    }
    __syncthreads();
    const uint32_t total = it_off[BIN_COUNT - 1] + it_cnt[BIN_COUNT - 1];
    for (uint32_t sidx = threadIdx.x; sidx < total; sidx += 256) {
      const float4 r = stage[sidx];
      const uint32_t bin = __float_as_uint(r.w);
      const uint32_t pos = base[bin] + cursor[bin] + (sidx - it_off[bin]);
      if (pos < bg.cap) {
        if constexpr (HREC) {
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          h2 hv;
          hv[0] = (_Float16)(r.y * rec_scale);
          hv[1] = (_Float16)(r.z * rec_scale);
          records_h[(size_t)level * BIN_COUNT * bg.cap + (size_t)bin * bg.cap + pos] =
              make_uint2(__float_as_uint(r.x), __builtin_bit_cast(uint32_t, hv));
        } else {
          rec_level[(size_t)bin * bg.cap + pos] = make_float4(r.x, r.y, r.z, 0.f);
        }
      } else {  // bin full: direct atomics keep the result exact
        const size_t idx = (size_t)bin * bsz + __float_as_uint(r.x);
        atomicAdd(gt + idx * 2, r.y);
        atomicAdd(gt + idx * 2 + 1, r.z);
      }
    }
    __syncthreads();
    cursor[threadIdx.x] += it_cnt[threadIdx.x];
    it_cnt[threadIdx.x] = 0;
    __syncthreads();
  }
}

#ifndef ACC_INFLIGHT
#define ACC_INFLIGHT 8
#endif
extern __shared__ __attribute__((aligned(16))) float binacc_smem[];

template <int REC>
__global__ void __launch_bounds__(512)
k_grid_bwd_accum(GridDev g, BinGeom bg, uint32_t level0,
                 const uint32_t* __restrict__ gcount,
                 const void* __restrict__ records_v,
                 float* __restrict__ grad_table, float inv_rec_scale) {
  const float4* records = reinterpret_cast<const float4*>(records_v);
  const uint32_t level = level0 + blockIdx.y, bin = blockIdx.x;
  const uint32_t bsz = bg.bin_size[level];
  uint32_t n = gcount[level * BIN_COUNT + bin];
  if (n == 0) return;
  if (n > bg.cap) n = bg.cap;
  constexpr bool HREC = REC == REC_H16;
  float* acc = binacc_smem;  // [bsz][2]
  for (uint32_t e = threadIdx.x; e < 2 * bsz; e += 512) acc[e] = 0.f;
  __syncthreads();
  if constexpr (HREC) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const uint2* rec = reinterpret_cast<const uint2*>(records_v) +
                       ((size_t)level * BIN_COUNT + bin) * bg.cap;
    uint32_t i = threadIdx.x;
    for (; i + 7 * 512 < n; i += 8 * 512) {   // 8 record loads in flight
      uint2 r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = rec[i + k * 512];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const h2 hv = __builtin_bit_cast(h2, r[k].y);
        lds_add_pair(&acc[2 * r[k].x], (float)hv[0] * inv_rec_scale,
                     (float)hv[1] * inv_rec_scale);
      }
    }
    for (; i < n; i += 512) {
      const uint2 r = rec[i];
      const h2 hv = __builtin_bit_cast(h2, r.y);
      lds_add_pair(&acc[2 * r.x], (float)hv[0] * inv_rec_scale,
                   (float)hv[1] * inv_rec_scale);
    }
  } else {
  const float4* rec = records + ((size_t)level * BIN_COUNT + bin) * bg.cap;
  // ACC_INFLIGHT record loads in flight per thread before the LDS adds
  uint32_t i = threadIdx.x;
  for (; i + (ACC_INFLIGHT - 1) * 512 < n; i += ACC_INFLIGHT * 512) {
    float4 r[ACC_INFLIGHT];
#pragma unroll
    for (int k = 0; k < ACC_INFLIGHT; ++k) r[k] = rec[i + k * 512];
#pragma unroll
    for (int k = 0; k < ACC_INFLIGHT; ++k)
      lds_add_pair(&acc[2 * __float_as_uint(r[k].x)], r[k].y, r[k].z);
  }
  for (; i < n; i += 512) {
    const float4 r = rec[i];
    lds_add_pair(&acc[2 * __float_as_uint(r.x)], r.y, r.z);
  }
  }
  __syncthreads();
  const uint32_t first = bin * bsz;
  const uint32_t entries = g.entries[level];
  float* gt = grad_table + (size_t)g.offset[level] * 2;
  for (uint32_t e = threadIdx.x; e < 2 * bsz; e += 512) {
    const uint32_t ent = first + (e >> 1);
    if (ent < entries) {
      const float v = acc[e];
      if (v != 0.f) gt[(size_t)ent * 2 + (e & 1)] += v;  // exclusive owner
    }
  }
}

// ---------------------------------------------------------------------------
// x-PAIR records: the packed ("p64") records of ucsa_hashgrid_bwd_rays_p64 /
// _merged_p64 (written at the end of round 5, timed and made the default in round 6:
// merged grid backward 1.420 -> 1.162 ms, training step 3.41 -> 3.16 ms,
// tools/bwd_switches_ab.py; the one-record-per-corner kernels they replace are gone).
//
// The bin kernel is issue- and LDS-atomic-bound (~82 VALU instructions per
// record, profiles/r04_train_sq_counters.txt), not byte-bound: what costs is the
// NUMBER of records.  The corners x and x + 1 of a cell differ in the low bits of
// the index only (idx = x ^ y P1 ^ z P2: x ^ (x + 1) = 2^(k+1) - 1 with k the
// number of trailing ones of x; dense rows: idx + 1), so with 2048-entry bins
// they fall into the SAME bin unless x ends in eleven 1-bits -- never below
// resolution 2048, 2^-11 of the pairs at the finest level (checked on 4 M random
// cells per level).  One 16-byte record per x-pair = two packed 64-bit words,
// p64(loc0, v0) and p64(loc0 ^ loc1, v1): 4 hashes' worth of bin counters, LDS
// ranks and stage slots per sample and level instead of 8, the same bytes, the
// 26-bit values (p64_round: 2^-18 per record), fp32 sums.  A pair that straddles two bins goes to the table by direct
// atomics (unrounded).  The workspace is the REC_F32 one (16 bytes x cap per bin).
__global__ void __launch_bounds__(256)
k_grid_bwd_bin_xpair(GridDev g, BinGeom bg, uint32_t level0,
                     const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                     const float* __restrict__ zs, Aabb bb, uint32_t T, uint64_t M,
                     const float2* __restrict__ d_feat, uint32_t* __restrict__ gcount,
                     ulonglong2* __restrict__ records, float* __restrict__ grad_table,
                     MergedSrc mg) {
  __shared__ uint32_t hist[BIN_COUNT], base[BIN_COUNT], cursor[BIN_COUNT];
  __shared__ uint32_t it_cnt[BIN_COUNT], it_off[BIN_COUNT], wave_tot[4];
  __shared__ ulonglong2 stage_w[256 * 4];   // one iteration's records, bin-sorted
  __shared__ uint16_t stage_b[256 * 4];
  const uint32_t level = level0 + blockIdx.y;
  const uint32_t lbits = bg.loc_bits[level];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t res = g.res[level], entries = g.entries[level],
                 hashed = g.hashed[level], bsz = bg.bin_size[level],
                 bshift = bg.bin_shift[level];
  auto bin_of = [&](uint32_t idx) -> uint32_t {
    return bshift < 32 ? idx >> bshift : idx / bsz;
  };
  hist[threadIdx.x] = 0;
  cursor[threadIdx.x] = 0;
  it_cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t m0 = (uint64_t)blockIdx.x * (256 * BIN_TILE) + threadIdx.x;
  // pass 1: the workgroup's x-pair records per bin
#pragma unroll 1
  for (int it = 0; it < BIN_TILE; ++it) {
    const uint64_t m = m0 + (uint64_t)it * 256;
    bool act = m < M;
    uint32_t r = 0;
    float zz = 0.f;
    if (act) {
      float2 df;
      sample_ref(mg, T, M, m, level, zs, d_feat, r, zz, df);
      act = !(df.x == 0.0f && df.y == 0.0f);
    }
    uint32_t gi[3] = {0u, 0u, 0u};
    float wf[3];
    if (act) sample_cell(g, level, rays_o, rays_d, zs != nullptr, r, zz, bb, m, gi, wf);
    const RunPlan plan = run_plan(gi[0], gi[1], gi[2], act, lane);
    if (!(act && plan.tail)) continue;
    // index of the corner at x + 1 from the corner at x: one xor on a hashed level
    // ((x ^ h) & (E - 1) with E a power of two), the next entry on a dense one
    const uint32_t xflip = (gi[0] ^ (gi[0] + 1u)) & (entries - 1u);
    auto x_neighbour = [&](uint32_t i0) -> uint32_t {
      return hashed ? (i0 ^ xflip) : (i0 + 1u == entries ? 0u : i0 + 1u);
    };
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint32_t i0 = grid_index_b(gi[0], gi[1] + (p & 1), gi[2] + (p >> 1), res, entries, hashed);
      const uint32_t i1 = x_neighbour(i0);
      const uint32_t b0 = bin_of(i0);
      if (b0 == bin_of(i1)) atomicAdd(&hist[b0], 1u);
    }
  }
  __syncthreads();
  {
    const uint32_t h = hist[threadIdx.x];
    base[threadIdx.x] = h ? atomicAdd(&gcount[level * BIN_COUNT + threadIdx.x], h) : 0u;
  }
  __syncthreads();
  float* gt = grad_table + (size_t)g.offset[level] * 2;
  ulonglong2* rec_level = records + (size_t)level * BIN_COUNT * bg.cap;
  // pass 2, 256 samples at a time (as k_grid_bwd_bin: counting sort by bin in
  // LDS, consecutive lanes store consecutive records of a bin)
#pragma unroll 1
  for (int it = 0; it < BIN_TILE; ++it) {
    const uint64_t m = m0 + (uint64_t)it * 256;
    bool act = m < M;
    float2 df = make_float2(0.f, 0.f);
    uint32_t r = 0;
    float zz = 0.f;
    if (act) {
      sample_ref(mg, T, M, m, level, zs, d_feat, r, zz, df);
      act = !(df.x == 0.0f && df.y == 0.0f);
    }
    uint32_t gi[3] = {0u, 0u, 0u};
    float wf[3] = {0.f, 0.f, 0.f};
    if (act) sample_cell(g, level, rays_o, rays_d, zs != nullptr, r, zz, bb, m, gi, wf);
    const RunPlan plan = run_plan(gi[0], gi[1], gi[2], act, lane);
    const bool has_runs = plan.live != 0;
    const bool emit = act && plan.tail;
    float valx[8], valy[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {   // c = x | y << 1 | z << 2, as in k_grid_bwd_bin
      float w = (c & 1) ? wf[0] : 1.0f - wf[0];
      w = w * ((c & 2) ? wf[1] : 1.0f - wf[1]);
      w = w * ((c & 4) ? wf[2] : 1.0f - wf[2]);
      float vx = act ? w * df.x : 0.0f, vy = act ? w * df.y : 0.0f;
      if (has_runs) {
        run_sum(plan, vx, vy);
      }
      valx[c] = vx;
      valy[c] = vy;
    }
    uint32_t key[4];       // bin << 16 | rank inside the bin; ~0u: no record
    uint64_t w0[4], w1[4];
    const uint32_t xflip = (gi[0] ^ (gi[0] + 1u)) & (entries - 1u);
    auto x_neighbour = [&](uint32_t i0) -> uint32_t {
      return hashed ? (i0 ^ xflip) : (i0 + 1u == entries ? 0u : i0 + 1u);
    };
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      key[p] = 0xFFFFFFFFu;
      w0[p] = w1[p] = 0ull;
      if (emit) {
        const uint32_t i0 = grid_index_b(gi[0], gi[1] + (p & 1), gi[2] + (p >> 1), res, entries, hashed);
        const uint32_t i1 = x_neighbour(i0);
        const uint32_t b0 = bin_of(i0);
        if (b0 == bin_of(i1)) {
          const uint32_t l0 = i0 - b0 * bsz, l1 = i1 - b0 * bsz;
          w0[p] = p64_pack(l0, valx[2 * p], valy[2 * p], lbits);
          w1[p] = p64_pack(l0 ^ l1, valx[2 * p + 1], valy[2 * p + 1], lbits);
          key[p] = (b0 << 16) | atomicAdd(&it_cnt[b0], 1u);
        } else {   // the pair straddles two bins: rare, straight to the table
          atomicAdd(gt + (size_t)i0 * 2, valx[2 * p]);
          atomicAdd(gt + (size_t)i0 * 2 + 1, valy[2 * p]);
          atomicAdd(gt + (size_t)i1 * 2, valx[2 * p + 1]);
          atomicAdd(gt + (size_t)i1 * 2 + 1, valy[2 * p + 1]);
        }
      }
    }
    __syncthreads();
    {  // exclusive prefix of it_cnt over the bins (thread = bin)
      const uint32_t v = it_cnt[threadIdx.x];
      const uint32_t inc = wave_incl_scan_add_u32(v, lane);
      if (lane == 63) wave_tot[wid] = inc;
      __syncthreads();
      uint32_t before = 0;
      for (uint32_t w = 0; w < wid; ++w) before += wave_tot[w];
      it_off[threadIdx.x] = before + inc - v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (key[p] != 0xFFFFFFFFu) {
        const uint32_t bin = key[p] >> 16;
        const uint32_t slot = it_off[bin] + (key[p] & 0xFFFFu);
        stage_w[slot] = make_ulonglong2(w0[p], w1[p]);
        stage_b[slot] = (uint16_t)bin;
      }
    }
    __syncthreads();
    const uint32_t total = it_off[BIN_COUNT - 1] + it_cnt[BIN_COUNT - 1];
    for (uint32_t sidx = threadIdx.x; sidx < total; sidx += 256) {
      const ulonglong2 w = stage_w[sidx];
      const uint32_t bin = stage_b[sidx];
      const uint32_t pos = base[bin] + cursor[bin] + (sidx - it_off[bin]);
      if (pos < bg.cap) {
        rec_level[(size_t)bin * bg.cap + pos] = w;
      } else {  // bin full: direct atomics (of the rounded values: same sum)
        uint32_t l0, mk;
        float ax, ay, bx, by;
        p64_unpack(w.x, lbits, l0, ax, ay);
        p64_unpack(w.y, lbits, mk, bx, by);
        const size_t e0 = (size_t)bin * bsz + l0, e1 = (size_t)bin * bsz + (l0 ^ mk);
        atomicAdd(gt + e0 * 2, ax);
        atomicAdd(gt + e0 * 2 + 1, ay);
        atomicAdd(gt + e1 * 2, bx);
        atomicAdd(gt + e1 * 2 + 1, by);
      }
    }
    __syncthreads();
    cursor[threadIdx.x] += it_cnt[threadIdx.x];
    it_cnt[threadIdx.x] = 0;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(512)
k_grid_bwd_accum_xpair(GridDev g, BinGeom bg, uint32_t level0,
                       const uint32_t* __restrict__ gcount,
                       const ulonglong2* __restrict__ records,
                       float* __restrict__ grad_table) {
  const uint32_t level = level0 + blockIdx.y, bin = blockIdx.x;
  const uint32_t bsz = bg.bin_size[level];
  uint32_t n = gcount[level * BIN_COUNT + bin];
  if (n == 0) return;
  if (n > bg.cap) n = bg.cap;
  float* acc = binacc_smem;  // [bsz][2]
  for (uint32_t e = threadIdx.x; e < 2 * bsz; e += 512) acc[e] = 0.f;
  __syncthreads();
  const ulonglong2* rec = records + ((size_t)level * BIN_COUNT + bin) * bg.cap;
  const uint32_t lbits = bg.loc_bits[level];
  auto add = [&](const ulonglong2& w) {
    uint32_t l0, mk;
    float ax, ay, bx, by;
    p64_unpack(w.x, lbits, l0, ax, ay);
    p64_unpack(w.y, lbits, mk, bx, by);
    lds_add_pair(&acc[2 * l0], ax, ay);
    lds_add_pair(&acc[2 * (l0 ^ mk)], bx, by);
  };
  constexpr int INF = ACC_INFLIGHT / 2;   // 16-byte loads: the same bytes in flight
  uint32_t i = threadIdx.x;
  for (; i + (INF - 1) * 512 < n; i += INF * 512) {
    ulonglong2 r[INF];
#pragma unroll
    for (int k = 0; k < INF; ++k) r[k] = rec[i + k * 512];
#pragma unroll
    for (int k = 0; k < INF; ++k) add(r[k]);
  }
  for (; i < n; i += 512) add(rec[i]);
  __syncthreads();
  const uint32_t first = bin * bsz;
  const uint32_t entries = g.entries[level];
  float* gt = grad_table + (size_t)g.offset[level] * 2;
  for (uint32_t e = threadIdx.x; e < 2 * bsz; e += 512) {
    const uint32_t ent = first + (e >> 1);
    if (ent < entries) {
      const float v = acc[e];
      if (v != 0.f) gt[(size_t)ent * 2 + (e & 1)] += v;  // exclusive owner
    }
  }
}

static BinGeom bin_geometry(const ucsa_grid* grid, uint64_t M) {
  BinGeom bg;
  for (uint32_t l = 0; l < UCSA_MAX_LEVELS; ++l) {
    const uint32_t e = l < grid->n_levels ? grid->level[l].entries : 0;
    bg.bin_size[l] = e ? (e + BIN_COUNT - 1) / BIN_COUNT : 1;
    bg.bin_shift[l] = 32;
    for (uint32_t sh = 0; sh < 32; ++sh)
      if (bg.bin_size[l] == (1u << sh)) bg.bin_shift[l] = sh;
    bg.loc_bits[l] = 0;
    while ((1ull << bg.loc_bits[l]) < bg.bin_size[l]) ++bg.loc_bits[l];
  }
  uint64_t cap = 2 * (8 * M / BIN_COUNT + 1);
  if (cap < 4096) cap = 4096;
  bg.cap = (uint32_t)cap;
  return bg;
}

// workspace: [16 x 256 counters (u32) = 16 KiB] [records]
#define BIN_HDR_BYTES (UCSA_MAX_LEVELS * BIN_COUNT * 4)
extern "C" uint64_t ucsa_hashgrid_bwd_workspace_bytes(uint32_t N, uint32_t T,
                                                      uint32_t n_levels) {
  ucsa_grid tmp;
  tmp.n_levels = 0;
  const BinGeom bg = bin_geometry(&tmp, (uint64_t)N * T);
  return (uint64_t)BIN_HDR_BYTES +
         (uint64_t)n_levels * BIN_COUNT * bg.cap * sizeof(float4);
}

// The coarse levels (LDS hash accumulators, k_hashgrid_bwd) and the fine levels
// (bin records through HBM, k_grid_bwd_bin / _accum) write disjoint slices of
// grad_table and wait on different resources, so the coarse kernel runs on a
// side stream forked from / joined to the caller's stream by events
// (capturable; UCSA_BWD_OVERLAP=0 keeps everything on the caller's stream).
// One (side stream, fork event, join event) triple per CALLER STREAM: two
// caller streams (or two devices) never share events, so their fork / join
// pairs cannot mis-order each other.  The device is the caller stream's
// (hipStreamGetDevice; the null stream falls back to the current device), not
// whatever device happens to be current.  Two HOST THREADS launching on the
// SAME caller stream concurrently are not supported (a stream is an ordered
// queue: the caller serialises its own stream).
struct BwdSide {
  hipStream_t caller = nullptr;
  int dev = -1;
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false, used = false;
};
static BwdSide* bwd_side(hipStream_t caller) {
  static std::mutex mu;
  static BwdSide slots[64];
  static int enabled = -1;
  std::lock_guard<std::mutex> lk(mu);
  if (enabled < 0) {
    const char* e = ucsa_getenv("UCSA_BWD_OVERLAP");
    enabled = !(e && e[0] == '0');
  }
  if (!enabled) return nullptr;
  int dev = -1;
  hipDevice_t sdev;
  if (caller && hipStreamGetDevice(caller, &sdev) == hipSuccess) dev = (int)sdev;
  else if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  BwdSide* free_slot = nullptr;
  for (BwdSide& b : slots) {
    if (b.used && b.caller == caller && b.dev == dev) return b.ok ? &b : nullptr;
    if (!b.used && !free_slot) free_slot = &b;
  }
  if (!free_slot) return nullptr;   // > 64 caller streams: no overlap, still correct
  BwdSide& b = *free_slot;
  b.used = true;
  b.caller = caller;
  b.dev = dev;
  int cur = -1;
  const bool sw = hipGetDevice(&cur) == hipSuccess && cur != dev;
  if (sw && hipSetDevice(dev) != hipSuccess) return nullptr;
  b.ok = hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&b.fork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&b.join, hipEventDisableTiming) == hipSuccess;
  if (sw) (void)hipSetDevice(cur);
  return b.ok ? &b : nullptr;
}

// z == nullptr: rays_o holds M = N explicit points (T = 1), aabb unused.
static int32_t hashgrid_bwd_launch(const ucsa_grid* grid, const float* rays_o,
                                   const float* rays_d, const float* z,
                                   const float* aabb_host, uint32_t N,
                                   uint32_t T, const float* d_feat,
                                   float* grad_table, void* workspace,
                                   void* stream,
                                   float rec_scale = 0.0f,  // > 0: REC_H16, < 0: packed x-pair records
                                   MergedSrc mg = MergedSrc{nullptr, nullptr, nullptr, 0u, 0u, 0u}) {
  const uint64_t M = (uint64_t)N * T;
  if (M == 0) return 0;
  // Binning pays where updates are spread over the whole slab (hashed levels
  // with cells finer than ~2 sample spacings: 170 us vs 800 us per level and
  // million samples).  On the dense coarse levels the records pile up in a
  // few bins, so those keep the run-combining atomics kernels below.
  uint32_t n_lo = grid->n_levels;
  if (workspace) {
    n_lo = 0;
    // UCSA_BWD_BIN_SCALE (tuning only; the result does not depend on it):
    // hashed levels with scale >= this go through the bins
    static const float bin_scale = []() {
      const char* v = ucsa_getenv("UCSA_BWD_BIN_SCALE");
      return v && *v ? (float)atof(v) : 100.0f;  // measured: 160 -> 4.81, 100 -> 4.76, 60 -> 5.04 ms per step
    }();
    while (n_lo < grid->n_levels && (!grid->level[n_lo].hashed ||
                                     grid->level[n_lo].scale < bin_scale)) ++n_lo;
  }
  // both halves present: fork the coarse half onto the side stream
  BwdSide* side = (workspace && n_lo > 0 && n_lo < grid->n_levels) ? bwd_side((hipStream_t)stream) : nullptr;
  hipStream_t coarse_stream = (hipStream_t)stream;
  if (side) {
    if (hipEventRecord(side->fork, (hipStream_t)stream) == hipSuccess &&
        hipStreamWaitEvent(side->side, side->fork, 0) == hipSuccess)
      coarse_stream = side->side;
    else
      side = nullptr;
  }
  auto join = [&](int32_t rc) -> int32_t {
    if (side) {
      const hipError_t e1 = hipEventRecord(side->join, side->side);
      const hipError_t e2 = hipStreamWaitEvent((hipStream_t)stream, side->join, 0);
      if (rc == 0 && e1 != hipSuccess) rc = -(int32_t)e1;
      if (rc == 0 && e2 != hipSuccess) rc = -(int32_t)e2;
    }
    return rc;
  };
  if (workspace && n_lo < grid->n_levels) {
    const BinGeom bg = bin_geometry(grid, M);
    const GridDev gd = ucsa_grid_dev(grid);
    const Aabb bb = ucsa_aabb(aabb_host);
    uint32_t* gcount = (uint32_t*)workspace;
    float4* records = (float4*)((char*)workspace + BIN_HDR_BYTES);
    hipError_t e = hipMemsetAsync(gcount, 0, BIN_HDR_BYTES, (hipStream_t)stream);
    if (e != hipSuccess) return -(int32_t)e;
    uint32_t max_bsz = 1;
    for (uint32_t l = n_lo; l < grid->n_levels; ++l)
      if (bg.bin_size[l] > max_bsz) max_bsz = bg.bin_size[l];
    const uint32_t nl = grid->n_levels - n_lo;
    UCSA_CLEAR_ERR();
    if (rec_scale < 0.0f) {  // packed records: one 16-byte record per x-pair of corners
      hipLaunchKernelGGL(k_grid_bwd_bin_xpair, dim3(ucsa_div_up(M, 256 * BIN_TILE), nl),
                         dim3(256), 0, (hipStream_t)stream, gd, bg, n_lo, rays_o, rays_d, z,
                         bb, T, M, (const float2*)d_feat, gcount, (ulonglong2*)records,
                         grad_table, mg);
      hipLaunchKernelGGL(k_grid_bwd_accum_xpair, dim3(BIN_COUNT, nl), dim3(512),
                         (size_t)max_bsz * 2 * sizeof(float), (hipStream_t)stream, gd, bg,
                         n_lo, gcount, (const ulonglong2*)records, grad_table);
    } else if (rec_scale > 0.0f) {
      hipLaunchKernelGGL(k_grid_bwd_bin<REC_H16>,
                         dim3(ucsa_div_up(M, 256 * BIN_TILE), nl), dim3(256), 0,
                         (hipStream_t)stream, gd, bg, n_lo, rays_o, rays_d, z, bb,
                         T, M, (const float2*)d_feat, gcount, (void*)records,
                         grad_table, rec_scale, mg);
      hipLaunchKernelGGL(k_grid_bwd_accum<REC_H16>, dim3(BIN_COUNT, nl), dim3(512),
                         (size_t)max_bsz * 2 * sizeof(float), (hipStream_t)stream,
                         gd, bg, n_lo, gcount, (const void*)records, grad_table,
                         1.0f / rec_scale);
    } else {
      hipLaunchKernelGGL(k_grid_bwd_bin<REC_F32>,
                         dim3(ucsa_div_up(M, 256 * BIN_TILE), nl), dim3(256), 0,
                         (hipStream_t)stream, gd, bg, n_lo, rays_o, rays_d, z, bb,
                         T, M, (const float2*)d_feat, gcount, (void*)records,
                         grad_table, 1.0f, mg);
      hipLaunchKernelGGL(k_grid_bwd_accum<REC_F32>, dim3(BIN_COUNT, nl), dim3(512),
                         (size_t)max_bsz * 2 * sizeof(float), (hipStream_t)stream,
                         gd, bg, n_lo, gcount, (const void*)records, grad_table,
                         1.0f);
    }
    const int32_t rc = ucsa_launch_status();
    if (rc) return join(rc);
  }
  if (n_lo == 0) return 0;
  // (no workspace: direct-atomics path)
  // levels whose cells are wider than about one sample spacing get the
  // run-combining variant (cell = 2*bound/scale, spacing ~ 2*bound*sqrt(3)/T)
  // without a workspace only levels with cells wider than ~one sample spacing
  // combine runs; with one, every level handled here is coarse enough
  uint32_t n_run = 0;
  while (n_run < n_lo && (workspace != nullptr ||
                          grid->level[n_run].scale < 0.6f * (float)T)) ++n_run;
  const GridDev gd = ucsa_grid_dev(grid);
  const Aabb bb = ucsa_aabb(aabb_host);
  UCSA_CLEAR_ERR();
  // 256-sample tiles per workgroup: ACC_TILES for big batches (fewer flushes
  // of the LDS accumulator), fewer when that would leave the chip under-filled
  // (a marched training batch has ~0.3 M points: 34 workgroups per level)
  uint32_t tiles = ACC_TILES;
  while (tiles > 1 && ucsa_div_up(M, 256 * tiles) * n_run < 1024) tiles >>= 1;
  if (n_run > 0)
    hipLaunchKernelGGL(k_hashgrid_bwd<true>,
                       dim3(ucsa_div_up(M, 256 * tiles), n_run),
                       dim3(256), 0, coarse_stream, gd, rays_o, rays_d, z,
                       bb, T, M, 0u, tiles, (const float2*)d_feat, grad_table, mg);
  if (n_run < n_lo)
    hipLaunchKernelGGL(k_hashgrid_bwd<false>,
                       dim3(ucsa_div_up(M, 256), n_lo - n_run),
                       dim3(256), 0, coarse_stream, gd, rays_o, rays_d, z,
                       bb, T, M, n_run, 1u, (const float2*)d_feat, grad_table, mg);
  return join(ucsa_launch_status());
}

extern "C" int32_t ucsa_hashgrid_bwd_rays(const ucsa_grid* grid,
                                          const float* rays_o,
                                          const float* rays_d, const float* z,
                                          const float* aabb_host, uint32_t N,
                                          uint32_t T, const float* d_feat,
                                          float* grad_table, void* workspace,
                                          void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 1);
  UCSA_CHECK_ARG(aabb_host, 4);
  UCSA_CHECK_ARG(d_feat, 7);
  UCSA_CHECK_ARG(grad_table, 8);
  return hashgrid_bwd_launch(grid, rays_o, rays_d, z, aabb_host, N, T, d_feat,
                             grad_table, workspace, stream);
}

// Both density passes of a training step in ONE call, walking every ray's
// Tc + Tf samples in sorted depth order (src [N, Tc+Tf] as the forward
// composite wrote it: e < Tc coarse sample e, else fine sample e - Tc): see
// MergedSrc.  d_feat_c [L][N*Tc][2], d_feat_f [L][N*Tf][2]; the workspace is
// ucsa_hashgrid_bwd_workspace_bytes(N, Tc + Tf, L).  Same gradient as the two
// single-pass calls up to the order of fp32 additions.
extern "C" int32_t ucsa_hashgrid_bwd_rays_merged(
    const ucsa_grid* grid, const float* rays_o, const float* rays_d,
    const float* z_c, const float* z_f, const int32_t* src, const float* aabb_host,
    uint32_t N, uint32_t Tc, uint32_t Tf, const float* d_feat_c,
    const float* d_feat_f, float* grad_table, void* workspace, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(rays_o && rays_d && z_c, 1);
  UCSA_CHECK_ARG(z_f && src, 4);
  UCSA_CHECK_ARG(aabb_host, 6);
  UCSA_CHECK_ARG(Tc >= 1 && Tf >= 1 && (uint64_t)N * (Tc + Tf) < 0x80000000ull, 8);
  UCSA_CHECK_ARG(d_feat_c && d_feat_f, 10);
  UCSA_CHECK_ARG(grad_table, 12);
  UCSA_CHECK_ARG(workspace, 13);
  const MergedSrc mg{src, z_f, (const float2*)d_feat_f, Tc, Tf, N};
  return hashgrid_bwd_launch(grid, rays_o, rays_d, z_c, aabb_host, N, Tc + Tf, d_feat_c,
                             grad_table, workspace, stream, 0.0f, mg);
}

// ucsa_hashgrid_bwd_rays / _merged with PACKED bin records (p64_pack: every value
// rounded to its top (64 - L) / 2 bits, 2^-18 relative for the reference's grid;
// fp32 sums), since round 6 one 16-byte record per x-pair of corners.  For the training modes whose MLP backward is
// itself a two-term bf16 split (2^-16).  Require the workspace.
extern "C" int32_t ucsa_hashgrid_bwd_rays_p64(const ucsa_grid* grid,
                                              const float* rays_o,
                                              const float* rays_d,
                                              const float* z,
                                              const float* aabb_host, uint32_t N,
                                              uint32_t T, const float* d_feat,
                                              float* grad_table, void* workspace,
                                              void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 1);
  UCSA_CHECK_ARG(aabb_host, 4);
  UCSA_CHECK_ARG(d_feat, 7);
  UCSA_CHECK_ARG(grad_table, 8);
  UCSA_CHECK_ARG(workspace, 9);
  return hashgrid_bwd_launch(grid, rays_o, rays_d, z, aabb_host, N, T, d_feat,
                             grad_table, workspace, stream, -1.0f);
}

extern "C" int32_t ucsa_hashgrid_bwd_rays_merged_p64(
    const ucsa_grid* grid, const float* rays_o, const float* rays_d,
    const float* z_c, const float* z_f, const int32_t* src, const float* aabb_host,
    uint32_t N, uint32_t Tc, uint32_t Tf, const float* d_feat_c,
    const float* d_feat_f, float* grad_table, void* workspace, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(rays_o && rays_d && z_c, 1);
  UCSA_CHECK_ARG(z_f && src, 4);
  UCSA_CHECK_ARG(aabb_host, 6);
  UCSA_CHECK_ARG(Tc >= 1 && Tf >= 1 && (uint64_t)N * (Tc + Tf) < 0x80000000ull, 8);
  UCSA_CHECK_ARG(d_feat_c && d_feat_f, 10);
  UCSA_CHECK_ARG(grad_table, 12);
  UCSA_CHECK_ARG(workspace, 13);
  const MergedSrc mg{src, z_f, (const float2*)d_feat_f, Tc, Tf, N};
  return hashgrid_bwd_launch(grid, rays_o, rays_d, z_c, aabb_host, N, Tc + Tf, d_feat_c,
                             grad_table, workspace, stream, -1.0f, mg);
}

// ucsa_hashgrid_bwd_rays_merged with the half2 records of _h16 (the f16 / tcnn
// training modes): rec_scale > 0, a power of two.
extern "C" int32_t ucsa_hashgrid_bwd_rays_merged_h16(
    const ucsa_grid* grid, const float* rays_o, const float* rays_d,
    const float* z_c, const float* z_f, const int32_t* src, const float* aabb_host,
    uint32_t N, uint32_t Tc, uint32_t Tf, const float* d_feat_c,
    const float* d_feat_f, float* grad_table, void* workspace, float rec_scale,
    void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(rays_o && rays_d && z_c, 1);
  UCSA_CHECK_ARG(z_f && src, 4);
  UCSA_CHECK_ARG(aabb_host, 6);
  UCSA_CHECK_ARG(Tc >= 1 && Tf >= 1 && (uint64_t)N * (Tc + Tf) < 0x80000000ull, 8);
  UCSA_CHECK_ARG(d_feat_c && d_feat_f, 10);
  UCSA_CHECK_ARG(grad_table, 12);
  UCSA_CHECK_ARG(workspace, 13);
  UCSA_CHECK_ARG(rec_scale > 0.0f, 14);
  const MergedSrc mg{src, z_f, (const float2*)d_feat_f, Tc, Tf, N};
  return hashgrid_bwd_launch(grid, rays_o, rays_d, z_c, aabb_host, N, Tc + Tf, d_feat_c,
                             grad_table, workspace, stream, rec_scale, mg);
}

// The same with 8-byte bin records: value pairs stored as half2 x rec_scale
// (rec_scale > 0; a power of two).  Requires the workspace.
extern "C" int32_t ucsa_hashgrid_bwd_rays_h16(const ucsa_grid* grid,
                                              const float* rays_o,
                                              const float* rays_d,
                                              const float* z,
                                              const float* aabb_host, uint32_t N,
                                              uint32_t T, const float* d_feat,
                                              float* grad_table, void* workspace,
                                              float rec_scale, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 1);
  UCSA_CHECK_ARG(aabb_host, 4);
  UCSA_CHECK_ARG(d_feat, 7);
  UCSA_CHECK_ARG(grad_table, 8);
  UCSA_CHECK_ARG(workspace, 9);
  UCSA_CHECK_ARG(rec_scale > 0.0f, 10);
  return hashgrid_bwd_launch(grid, rays_o, rays_d, z, aabb_host, N, T, d_feat,
                             grad_table, workspace, stream, rec_scale);
}

extern "C" int32_t ucsa_hashgrid_bwd_points(const ucsa_grid* grid,
                                            const float* x, uint32_t M,
                                            const float* d_feat,
                                            float* grad_table, void* workspace,
                                            void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(x, 1);
  UCSA_CHECK_ARG(d_feat, 3);
  UCSA_CHECK_ARG(grad_table, 4);
  static const float no_aabb[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return hashgrid_bwd_launch(grid, x, nullptr, nullptr, no_aabb, M, 1u, d_feat,
                             grad_table, workspace, stream);
}


// ---------------------------------------------------------------------------
// Deterministic reduction (debug mode, SURVEY 5 "race detection" build note;
// VERDICT r4 "missing" item 4).  The kernels above add fp32 values in an order
// that depends on the hardware's scheduling (LDS float atomics on the coarse
// levels, bin records in reservation order on the fine ones): two runs of one
// step give gradients that differ in the last bits.  Here every contribution
// w * d_feat -- the same fp32 product the fast kernels form -- is converted to
// a 64-bit FIXED-POINT number (units of 2^-44) and added with integer atomics:
// integer addition is associative, so the sum does not depend on the order,
// and the result is the same bits on every run and every device.  A contribution
// must be finite and below 2^18 in magnitude (2^62 in fixed point); an entry's
// running sum must stay inside int64 (+-2^19): every atomic returns the value it
// added to, so the ONE addition that wraps sees it (signed overflow of old + add)
// -- exact whatever the order.  Either raises a flag that turns the whole gradient
// into NaN (found_inf semantics).  ~25 ms for a 4096 x 512
// step: for `UCSA_DETERMINISTIC=1` runs, not for production.
//   fix [total_entries * 2 + 1] int64, zeroed by the caller; the last word is the flag.
// ---------------------------------------------------------------------------
#define DET_ONE 17592186044416.0   // 2^44

// fix[i] += add; true when THIS addition wrapped int64 (old and add of one sign,
// the sum of the other)
__device__ __forceinline__ bool det_add(unsigned long long* p, long long add) {
  const long long old = (long long)atomicAdd(p, (unsigned long long)add);
  const long long sum = (long long)((unsigned long long)old + (unsigned long long)add);
  return ((old ^ sum) & (add ^ sum)) < 0;
}

__global__ void __launch_bounds__(256)
k_hashgrid_bwd_det(GridDev g, const float* __restrict__ rays_o,
                   const float* __restrict__ rays_d, const float* __restrict__ zs,
                   Aabb bb, uint32_t T, uint64_t M, const float2* __restrict__ d_feat,
                   unsigned long long* __restrict__ fix, uint64_t flag_at) {
  const uint32_t level = blockIdx.y;
  const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const uint32_t r = (uint32_t)(m / T);
  const float zz = zs[m];
  const float* o = rays_o + (size_t)r * 3;
  const float* d = rays_d + (size_t)r * 3;
  const float two_b = 2.0f * g.bound, inv = unit_inv_b(two_b);
  const float x = to_unit_b(clampf_b(o[0] + d[0] * zz, bb.lo[0], bb.hi[0]), g.bound, two_b, inv);
  const float y = to_unit_b(clampf_b(o[1] + d[1] * zz, bb.lo[1], bb.hi[1]), g.bound, two_b, inv);
  const float z = to_unit_b(clampf_b(o[2] + d[2] * zz, bb.lo[2], bb.hi[2]), g.bound, two_b, inv);
  const float scale = g.scale[level];
  const uint32_t res = g.res[level], entries = g.entries[level], hashed = g.hashed[level];
  const float px = x * scale + 0.5f, py = y * scale + 0.5f, pz = z * scale + 0.5f;
  const float fx0 = floorf(px), fy0 = floorf(py), fz0 = floorf(pz);
  const float wx = px - fx0, wy = py - fy0, wz = pz - fz0;
  const uint32_t gx = (uint32_t)(int32_t)fx0, gy = (uint32_t)(int32_t)fy0,
                 gz = (uint32_t)(int32_t)fz0;
  const float2 df = d_feat[(uint64_t)level * M + m];
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w = (c & 1) ? wx : 1.0f - wx;
    w = w * ((c & 2) ? wy : 1.0f - wy);
    w = w * ((c & 4) ? wz : 1.0f - wz);
    const uint32_t idx = grid_index_b(gx + (c & 1), gy + ((c >> 1) & 1), gz + ((c >> 2) & 1),
                                    res, entries, hashed);
    const float vx = w * df.x, vy = w * df.y;
    bad = bad || !(fabsf(vx) < 262144.0f) || !(fabsf(vy) < 262144.0f);   // NaN too
    const uint64_t e = ((uint64_t)g.offset[level] + idx) * 2u;
    if (bad) continue;          // (nothing defined to add; the flag below poisons the result)
    if (vx != 0.f) bad = det_add(&fix[e], __double2ll_rn((double)vx * DET_ONE)) || bad;
    if (vy != 0.f) bad = det_add(&fix[e + 1], __double2ll_rn((double)vy * DET_ONE)) || bad;
  }
  if (bad) atomicOr(&fix[flag_at], 1ull);
}

__global__ void __launch_bounds__(256)
k_hashgrid_bwd_det_finish(const long long* __restrict__ fix, uint64_t n,
                          float* __restrict__ grad_table) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool bad = fix[n] != 0;      // non-finite / out-of-range contribution or a wrapped sum
  const float v = (float)((double)fix[i] * (1.0 / DET_ONE));
  grad_table[i] = bad ? __builtin_nanf("") : grad_table[i] + v;
}

extern "C" uint64_t ucsa_hashgrid_bwd_det_workspace_bytes(const ucsa_grid* grid) {
  return grid ? ((uint64_t)grid->total_entries * 2ull + 1ull) * 8ull : 0ull;
}

extern "C" int32_t ucsa_hashgrid_bwd_rays_det(const ucsa_grid* grid, const float* rays_o,
                                              const float* rays_d, const float* z,
                                              const float* aabb_host, uint32_t N,
                                              uint32_t T, const float* d_feat,
                                              void* fix, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(rays_o && rays_d && z, 1);
  UCSA_CHECK_ARG(aabb_host, 4);
  UCSA_CHECK_ARG(d_feat, 7);
  UCSA_CHECK_ARG(fix && ((uintptr_t)fix & 7u) == 0, 8);
  const uint64_t M = (uint64_t)N * T;
  if (M == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_hashgrid_bwd_det, dim3(ucsa_div_up(M, 256), grid->n_levels),
                     dim3(256), 0, (hipStream_t)stream, ucsa_grid_dev(grid), rays_o, rays_d,
                     z, ucsa_aabb(aabb_host), T, M, (const float2*)d_feat,
                     (unsigned long long*)fix, (uint64_t)grid->total_entries * 2ull);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_hashgrid_bwd_det_finish(const ucsa_grid* grid, const void* fix,
                                                float* grad_table, void* stream) {
  UCSA_CHECK_ARG(grid && grid->total_entries > 0, 0);
  UCSA_CHECK_ARG(fix, 1);
  UCSA_CHECK_ARG(grad_table, 2);
  const uint64_t n = (uint64_t)grid->total_entries * 2ull;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_hashgrid_bwd_det_finish, dim3(ucsa_div_up(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const long long*)fix, n, grad_table);
  return ucsa_launch_status();
}
