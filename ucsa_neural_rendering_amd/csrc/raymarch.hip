// Occupancy-grid ray marching, early termination and alive-ray compaction
// (SURVEY 8f rank 1).  Restates the dormant CUDA kernels of
// reference nr4seg/nerf/raymarching/src/raymarching.cu:138-855 for gfx950:
//
//   * output spans / compacted slots are handed out by prefix sums over the
//     ray index (block sums -> per-block prefix -> wave64 scan), never by
//     atomics: results are run-to-run identical and in ray order, which is one
//     of the orders the reference's atomicAdd can produce;
//   * the training composite and its backward run one wave64 per ray (the
//     samples of a ray are contiguous): transmittance is a wave-wide
//     multiplicative scan, the per-ray sums are wave reductions, every global
//     access is coalesced;
//   * the marchers stay lane-per-ray (each step depends on the previous t).
//
// Arithmetic is fp32 in source order (-ffp-contract=off), identical to
// oracle/raymarch.c, so step counts and sample positions are bit-exact.
#include <cfloat>
#include <cmath>

#include "ucsa_common.h"
#include "wave_ops.h"

#define RM_MAX_STEPS 1024u                     // reference :23
#define RM_DENSITY_THRESH 0.01f                // reference :21
#define RM_SQRT3 1.73205080757f                // reference :22
#define RM_MIN_STEPSIZE (2 * RM_SQRT3 / 1024)  // reference :24
#define RM_BLOCK 256

// ---------------------------------------------------------------------------
// PCG32 (reference src/pcg32.h:44-117): only "seed, then one float" is used.
// ---------------------------------------------------------------------------
struct Pcg32 {
  uint64_t state, inc;
  __device__ __forceinline__ uint32_t next() {
    const uint64_t old = state;
    state = old * 0x5851f42d4c957f2dULL + inc;
    const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
  }
  __device__ __forceinline__ Pcg32(uint64_t initstate, uint64_t initseq) {
    state = 0u;
    inc = (initseq << 1u) | 1u;
    next();
    state += initstate;
    next();
  }
  __device__ __forceinline__ float next_float() {
    return __uint_as_float((next() >> 9) | 0x3f800000u) - 1.0f;
  }
};

__device__ __forceinline__ float rm_clamp(float x, float lo, float hi) {
  return fminf(hi, fmaxf(lo, x));
}

// ---------------------------------------------------------------------------
// One ray against the cascade grid [C,H,H,H].
// ---------------------------------------------------------------------------
struct Marcher {
  float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
  float bound, dt_gamma, dt_min, dt_max, thresh, far;
  float Hf, hm1, cmax;
  uint32_t H;
  const float* __restrict__ grid;

  __device__ __forceinline__ Marcher(const float* o, const float* d,
                                     const float* g, float mean_density,
                                     float bound_, float dt_gamma_, uint32_t C,
                                     uint32_t H_, float far_) {
    ox = o[0]; oy = o[1]; oz = o[2];
    dx = d[0]; dy = d[1]; dz = d[2];
    rdx = 1.0f / dx; rdy = 1.0f / dy; rdz = 1.0f / dz;
    bound = bound_;
    dt_gamma = dt_gamma_;
    dt_max = 2 * bound_ / H_;
    // fmin(fmax(x, dt_min), dt_max) is dt_max whenever dt_max < dt_min (tiny
    // bounds); with dt_min lowered to dt_max the median of three says the same
    dt_min = fminf(RM_MIN_STEPSIZE, dt_max);
    thresh = fminf(RM_DENSITY_THRESH, mean_density);
    far = far_;
    H = H_;
    Hf = (float)H_;
    hm1 = (float)(H_ - 1);
    cmax = (float)C - 1;
    grid = g;
  }

  __device__ __forceinline__ float step_size(float t) const {
    // median of three == clamp for dt_min <= dt_max and finite t (one
    // instruction instead of max + min in the wave marcher's serial chain)
    return __builtin_amdgcn_fmed3f(t * dt_gamma, dt_min, dt_max);
  }

  // One look at the cascade grid at ray parameter t (reference :188-220):
  // the clamped point, and either "occupied" or the parameter tt at which the
  // ray leaves the (empty) cell.
  __device__ __forceinline__ bool look(float t, float& x, float& y, float& z,
                                       float& tt) const {
    x = rm_clamp(ox + t * dx, -bound, bound);
    y = rm_clamp(oy + t * dy, -bound, bound);
    z = rm_clamp(oz + t * dz, -bound, bound);
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    const int level = (int)fminf(cmax, fmaxf(0.0f, (float)e));
    const float mip_bound = fminf(exp2f((float)level), bound);
    const float mip_rbound = 1.0f / mip_bound;
    const int nx = (int)rm_clamp(0.5f * (x * mip_rbound + 1) * Hf, 0.0f, hm1);
    const int ny = (int)rm_clamp(0.5f * (y * mip_rbound + 1) * Hf, 0.0f, hm1);
    const int nz = (int)rm_clamp(0.5f * (z * mip_rbound + 1) * Hf, 0.0f, hm1);
    const uint32_t idx = (uint32_t)level * H * H * H + (uint32_t)nx * H * H +
                         (uint32_t)ny * H + (uint32_t)nz;
    if (grid[idx] > thresh) return true;
    const float tx = (((nx + 0.5f + 0.5f * copysignf(1.0f, dx)) / hm1 * 2 - 1) * mip_bound - x) * rdx;
    const float ty = (((ny + 0.5f + 0.5f * copysignf(1.0f, dy)) / hm1 * 2 - 1) * mip_bound - y) * rdy;
    const float tz = (((nz + 0.5f + 0.5f * copysignf(1.0f, dz)) / hm1 * 2 - 1) * mip_bound - z) * rdz;
    tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    return false;
  }

  // reference :188-226.  Occupied cell: returns true with the point, t
  // untouched.  Empty cell: advances t past it (:221-225).  The inner loop
  // also stops once t >= far; the caller's loop ends there anyway and t is
  // dead then, so results are unchanged, but a degenerate ray cannot spin
  // forever.
  __device__ __forceinline__ bool probe(float& t, float& x, float& y,
                                        float& z) const {
    float tt;
    if (look(t, x, y, z, tt)) return true;
    do {
      t += step_size(t);
    } while (t < tt && t < far);
    return false;
  }
};

// block-wide exclusive scan of one uint per thread (RM_BLOCK threads);
// returns the exclusive prefix, *total = block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* sm,
                                                    uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan_add_u32(v, lane);
  if (lane == 63) sm[wid] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (uint32_t w = 0; w < RM_BLOCK / 64; ++w) {
    const uint32_t s = sm[w];
    if (w < wid) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

// exclusive prefix of block_sums[0 .. blockIdx.x) (+ optionally the grand
// total over all gridDim.x blocks), summed by the whole block.
__device__ __forceinline__ uint32_t prefix_of_blocks(const uint32_t* sums,
                                                     uint32_t upto,
                                                     uint32_t* sm) {
  uint32_t s = 0;
  for (uint32_t b = threadIdx.x; b < upto; b += blockDim.x) s += sums[b];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += (uint32_t)__shfl_xor((int)s, d, 64);
  __syncthreads();
  if (lane == 0) sm[8 + wid] = s;
  __syncthreads();
  uint32_t t = 0;
  for (uint32_t w = 0; w < blockDim.x / 64; ++w) t += sm[8 + w];
  return t;
}

// ===========================================================================
// march_rays_train.  reference :138-307.
// workspace (uint32): [0] old counter[0], [1] old counter[1], [2..3] pad,
//                     [4 .. 4+nb) block sums, [4+nb .. 4+nb+N) steps per ray
// ===========================================================================
__global__ void __launch_bounds__(RM_BLOCK)
k_march_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
              const float* __restrict__ grid, float mean_density, float bound,
              float dt_gamma, uint32_t N, uint32_t C, uint32_t H,
              const float* __restrict__ nears, const float* __restrict__ fars,
              const int32_t* __restrict__ counter, uint32_t perturb,
              uint32_t* __restrict__ ws) {
  __shared__ uint32_t sm[16];
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  uint32_t num_steps = 0;
  if (n < N) {
    const float far = fars[n];
    Marcher m(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid,
              mean_density, bound, dt_gamma, C, H, far);
    float t = nears[n];
    if (perturb) {
      Pcg32 rng((uint64_t)n, 1u);
      t += RM_MIN_STEPSIZE * rng.next_float();
    }
    float x, y, z;
    while (t < far && num_steps < RM_MAX_STEPS) {
      if (m.probe(t, x, y, z)) {
        ++num_steps;
        t += m.step_size(t);
      }
    }
    ws[4 + gridDim.x + n] = num_steps;
  }
  uint32_t total;
  block_excl_scan(num_steps, sm, &total);
  if (threadIdx.x == 0) {
    ws[4 + blockIdx.x] = total;
    if (blockIdx.x == 0) {
      ws[0] = (uint32_t)counter[0];
      ws[1] = (uint32_t)counter[1];
    }
  }
}

__global__ void __launch_bounds__(RM_BLOCK)
k_march_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
              const float* __restrict__ grid, float mean_density, float bound,
              float dt_gamma, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
              const float* __restrict__ nears, const float* __restrict__ fars,
              float* __restrict__ xyzs, float* __restrict__ dirs,
              float* __restrict__ deltas, int32_t* __restrict__ rays,
              int32_t* __restrict__ counter, uint32_t perturb,
              const uint32_t* __restrict__ ws) {
  __shared__ uint32_t sm[16];
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  const uint32_t num_steps = n < N ? ws[4 + gridDim.x + n] : 0u;
  const uint32_t before = prefix_of_blocks(ws + 4, blockIdx.x, sm);
  uint32_t total;
  const uint32_t in_block = block_excl_scan(num_steps, sm, &total);
  const uint32_t base0 = ws[0], base1 = ws[1];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    counter[0] = (int32_t)(base0 + before + total);
    counter[1] = (int32_t)(base1 + N);
  }
  if (n >= N) return;
  const uint32_t point_index = base0 + before + in_block;
  const uint32_t ray_index = base1 + n;
  if (ray_index < N) {  // (a caller that did not zero counter[1]: the
    rays[ray_index * 3] = (int32_t)n;            // reference writes past the
    rays[ray_index * 3 + 1] = (int32_t)point_index;  // buffer, we drop)
    rays[ray_index * 3 + 2] = (int32_t)num_steps;
  }
  if (num_steps == 0) return;
  if (point_index + num_steps >= M) return;

  const float far = fars[n];
  Marcher m(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, mean_density,
            bound, dt_gamma, C, H, far);
  float t = nears[n];
  if (perturb) {
    Pcg32 rng((uint64_t)n, 1u);
    t += RM_MIN_STEPSIZE * rng.next_float();
  }
  float* px = xyzs + (size_t)point_index * 3;
  float* pd = dirs + (size_t)point_index * 3;
  float* pl = deltas + (size_t)point_index * 2;
  float last_t = t, x, y, z;
  uint32_t step = 0;
  while (t < far && step < num_steps) {
    if (m.probe(t, x, y, z)) {
      px[0] = x; px[1] = y; px[2] = z;
      pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
      const float dt = m.step_size(t);
      t += dt;
      pl[0] = dt;
      pl[1] = t - last_t;
      last_t = t;
      px += 3; pd += 3; pl += 2;
      ++step;
    }
  }
}


// ===========================================================================
// Wave-cooperative marcher (one wave64 per ray) for small ray batches.
//
// Lane-per-ray marching is a chain of dependent grid look-ups: ~300 probes of
// ~1 us each per training ray, and a 4096-ray batch is only 64 waves on a
// 1024-SIMD chip (measured: 0.45 ms per pass, 36 % of a marched training
// step).  But every t the marcher can visit lies on one orbit
//     u_0 = t_start,  u_{k+1} = u_k + clamp(u_k * dt_gamma, dt_min, dt_max)
// whatever the grid holds: an occupied probe moves to the next orbit point, an
// empty one to the first orbit point at or beyond the cell exit tt.  So a wave
// generates 64 orbit points, looks all of them up AT ONCE (one memory round
// trip instead of up to 64), gives every lane its successor index (lane+1, or
// a binary search for tt in the orbit), and then walks the successor chain
// from the entry point with readlanes.  Same fp32 operations in the same
// order per point as the sequential loop -> bit-identical samples.
// ===========================================================================
#define WM_WAVES 4

__device__ __forceinline__ void wm_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// emit(rank, x, y, z, dt, t_after, last_t) is called by the lanes that hold a
// sample; rank counts the ray's samples from 0.  Returns the sample count
// (<= limit).  `last_t` = parameter the first depth delta is measured from.
template <typename Emit>
__device__ __forceinline__ uint32_t wave_march(const Marcher& m, float t_start,
                                               float last_t, uint32_t limit,
                                               float* u_lds, uint32_t lane,
                                               Emit emit) {
  float t_base = t_start;
  float pending = -INFINITY;  // next probe: first orbit point not below this
  uint32_t steps = 0;
  bool done = false;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  while (!done && t_base < m.far && steps < limit) {
    // ---- 64 orbit points (sequential by nature, uniform across the wave) --
    float t = t_base, u = 0.0f;
#pragma unroll 8
    for (uint32_t i = 0; i < 64; ++i) {
      if (lane == i) u = t;
      t += m.step_size(t);
    }
    const float u_end = t;
    const float dtv = m.step_size(u);  // same function of the same t
    u_lds[lane] = u;
    wm_sync();
    // ---- look all of them up ---------------------------------------------
    const bool valid = u < m.far;
    float x = 0.f, y = 0.f, z = 0.f, tt = 0.f;
    bool occ = false;
    if (valid) occ = m.look(u, x, y, z, tt);
    uint32_t nxt = lane + 1;
    if (valid && !occ) {  // first later orbit point not below tt (<= 64)
      uint32_t lo = lane + 1, hi = 64;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (u_lds[mid] < tt) lo = mid + 1; else hi = mid;
      }
      nxt = lo;
    }
    const uint64_t occ_mask = __ballot(occ), valid_mask = __ballot(valid);
    // ---- entry point of this chunk ----------------------------------------
    const uint64_t ge = __ballot(!(u < pending));
    uint32_t cur = ge ? (uint32_t)__ffsll((long long)ge) - 1u : 64u;
    uint64_t take = 0;
    const uint32_t room = limit - steps;
    if (cur < 64) {
      const uint64_t rest = ~0ull << cur;
      if ((occ_mask & rest) == (valid_mask & rest)) {
        // everything from the entry on is occupied (or past far): no walk
        uint64_t cand = valid_mask & rest;  // contiguous run starting at cur
        const uint32_t n = (uint32_t)__popcll(cand);
        if (n > room) cand &= ~(~0ull << (cur + room));  // room < n <= 64 - cur
        take = cand;
        pending = -INFINITY;
        done = (valid_mask != ~0ull) || n >= room;
      } else {
        uint32_t got = 0;
        while (cur < 64) {
          if (!((valid_mask >> cur) & 1ull)) {
            done = true;
            break;
          }
          const uint32_t nx = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)cur);
          if ((occ_mask >> cur) & 1ull) {
            take |= 1ull << cur;
            pending = -INFINITY;
            if (++got == room) {
              done = true;
              break;
            }
          } else {
            pending = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt), (int)cur));
          }
          cur = nx;
        }
      }
    }
    // ---- emit ---------------------------------------------------------------
    const float t_after = u + dtv;
    if (take) {
      const uint64_t below = take & lt_mask;
      const int prev = below ? 63 - __builtin_clzll(below) : 0;
      const float prev_after = __shfl(t_after, prev, 64);
      if ((take >> lane) & 1ull)
        emit(steps + (uint32_t)__popcll(below), x, y, z, dtv, t_after,
             below ? prev_after : last_t);
      const int top = 63 - __builtin_clzll(take);
      last_t = __builtin_bit_cast(
          float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_after), top));
      steps += (uint32_t)__popcll(take);
    }
    t_base = u_end;
    wm_sync();
  }
  return steps;
}

// workspace (uint32): [0] old counter[0], [1] old counter[1], [2..3] pad,
//   [4 + nb .. +N) steps per ray (same slot as the lane-per-ray kernels),
//   [4 + nb + N .. +N) first point per ray
__global__ void __launch_bounds__(64 * WM_WAVES)
k_march_count_w(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                const float* __restrict__ grid, float mean_density, float bound,
                float dt_gamma, uint32_t N, uint32_t C, uint32_t H,
                const float* __restrict__ nears, const float* __restrict__ fars,
                uint32_t perturb, uint32_t* __restrict__ steps_out,
                float* __restrict__ stage) {
  __shared__ float u_s[WM_WAVES][64];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t n = blockIdx.x * WM_WAVES + wid;
  if (n >= N) return;
  const float far = fars[n];
  Marcher m(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, mean_density,
            bound, dt_gamma, C, H, far);
  float t = nears[n];
  if (perturb) {
    Pcg32 rng((uint64_t)n, 1u);
    t += RM_MIN_STEPSIZE * rng.next_float();
  }
  // stage != NULL: keep the samples (x, y, z, dt, t_after - last_t) in the
  // ray's staging rows so that k_march_pack_w only has to copy them into
  // place once the offsets are known (one march per batch instead of two)
  float* st = stage ? stage + (size_t)n * RM_MAX_STEPS * 5 : nullptr;
  const uint32_t steps = wave_march(
      m, t, t, RM_MAX_STEPS, u_s[wid], lane,
      [&](uint32_t rank, float x, float y, float z, float dt, float t_after,
          float last_t) {
        if (st) {
          float* q = st + (size_t)rank * 5;
          q[0] = x; q[1] = y; q[2] = z; q[3] = dt; q[4] = t_after - last_t;
        }
      });
  if (lane == 0) steps_out[n] = steps;
}

// staging rows -> the ray's span (one wave per ray)
__global__ void __launch_bounds__(64 * WM_WAVES)
k_march_pack_w(const float* __restrict__ rays_d, uint32_t N, uint32_t M,
               const float* __restrict__ stage, float* __restrict__ xyzs,
               float* __restrict__ dirs, float* __restrict__ deltas,
               int32_t* __restrict__ rays, const uint32_t* __restrict__ steps_in,
               const uint32_t* __restrict__ first,
               const uint32_t* __restrict__ hdr) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n = blockIdx.x * WM_WAVES + (threadIdx.x >> 6);
  if (n >= N) return;
  const uint32_t num_steps = steps_in[n];
  const uint32_t point_index = first[n];
  const uint32_t ray_index = hdr[1] + n;
  if (lane == 0 && ray_index < N) {
    rays[ray_index * 3] = (int32_t)n;
    rays[ray_index * 3 + 1] = (int32_t)point_index;
    rays[ray_index * 3 + 2] = (int32_t)num_steps;
  }
  if (num_steps == 0 || point_index + num_steps >= M) return;
  const float dx = rays_d[(size_t)n * 3], dy = rays_d[(size_t)n * 3 + 1],
              dz = rays_d[(size_t)n * 3 + 2];
  const float* st = stage + (size_t)n * RM_MAX_STEPS * 5;
  for (uint32_t k = lane; k < num_steps; k += 64) {
    const float* q = st + (size_t)k * 5;
    const size_t p = (size_t)point_index + k;
    xyzs[p * 3] = q[0]; xyzs[p * 3 + 1] = q[1]; xyzs[p * 3 + 2] = q[2];
    dirs[p * 3] = dx; dirs[p * 3 + 1] = dy; dirs[p * 3 + 2] = dz;
    deltas[p * 2] = q[3]; deltas[p * 2 + 1] = q[4];
  }
}

// one workgroup: exclusive scan of steps[N] -> first[N], counters
__global__ void __launch_bounds__(1024)
k_march_scan(uint32_t N, const uint32_t* __restrict__ steps,
             uint32_t* __restrict__ first, int32_t* __restrict__ counter,
             uint32_t* __restrict__ hdr) {
  __shared__ uint32_t sm[16];
  __shared__ uint32_t carry_s;
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t base0 = (uint32_t)counter[0], base1 = (uint32_t)counter[1];
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t n0 = 0; n0 < N; n0 += 1024) {
    const uint32_t n = n0 + threadIdx.x;
    const uint32_t v = n < N ? steps[n] : 0u;
    const uint32_t incl = wave_incl_scan_add_u32(v, lane);
    if (lane == 63) sm[wid] = incl;
    __syncthreads();
    uint32_t before = carry_s, tot = 0;
    for (uint32_t w = 0; w < 16; ++w) {
      const uint32_t s = sm[w];
      if (w < wid) before += s;
      tot += s;
    }
    if (n < N) first[n] = base0 + before + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hdr[0] = base0;
    hdr[1] = base1;
    counter[0] = (int32_t)(base0 + carry_s);
    counter[1] = (int32_t)(base1 + N);
  }
}

__global__ void __launch_bounds__(64 * WM_WAVES)
k_march_write_w(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                const float* __restrict__ grid, float mean_density, float bound,
                float dt_gamma, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                const float* __restrict__ nears, const float* __restrict__ fars,
                float* __restrict__ xyzs, float* __restrict__ dirs,
                float* __restrict__ deltas, int32_t* __restrict__ rays,
                uint32_t perturb, const uint32_t* __restrict__ steps_in,
                const uint32_t* __restrict__ first,
                const uint32_t* __restrict__ hdr) {
  __shared__ float u_s[WM_WAVES][64];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t n = blockIdx.x * WM_WAVES + wid;
  if (n >= N) return;
  const uint32_t num_steps = steps_in[n];
  const uint32_t point_index = first[n];
  const uint32_t ray_index = hdr[1] + n;
  if (lane == 0 && ray_index < N) {
    rays[ray_index * 3] = (int32_t)n;
    rays[ray_index * 3 + 1] = (int32_t)point_index;
    rays[ray_index * 3 + 2] = (int32_t)num_steps;
  }
  if (num_steps == 0) return;
  if (point_index + num_steps >= M) return;
  const float far = fars[n];
  Marcher m(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, mean_density,
            bound, dt_gamma, C, H, far);
  float t = nears[n];
  if (perturb) {
    Pcg32 rng((uint64_t)n, 1u);
    t += RM_MIN_STEPSIZE * rng.next_float();
  }
  float* px = xyzs + (size_t)point_index * 3;
  float* pd = dirs + (size_t)point_index * 3;
  float* pl = deltas + (size_t)point_index * 2;
  const float ddx = m.dx, ddy = m.dy, ddz = m.dz;
  wave_march(m, t, t, num_steps, u_s[wid], lane,
             [&](uint32_t rank, float x, float y, float z, float dt,
                 float t_after, float last_t) {
               px[3 * rank] = x; px[3 * rank + 1] = y; px[3 * rank + 2] = z;
               pd[3 * rank] = ddx; pd[3 * rank + 1] = ddy; pd[3 * rank + 2] = ddz;
               pl[2 * rank] = dt;
               pl[2 * rank + 1] = t_after - last_t;
             });
}

// ray batches up to this size march one wave per ray
#define WM_MAX_RAYS 32768u
// ... and up to this size with staging rows (20 KB per ray: 168 MB)
#define WM_STAGE_RAYS 8192u

static uint64_t march_ws_head_bytes(uint32_t N) {
  const uint64_t b = 4ull * (4ull + ucsa_div_up(N ? N : 1, RM_BLOCK) + 2ull * N);
  return (b + 255ull) & ~255ull;
}

extern "C" uint64_t ucsa_march_workspace_bytes(uint32_t N) {
  uint64_t b = march_ws_head_bytes(N);
  if (N <= WM_STAGE_RAYS) b += (uint64_t)N * RM_MAX_STEPS * 5 * sizeof(float);
  return b;
}

extern "C" int32_t ucsa_march_rays_train(
    const float* rays_o, const float* rays_d, const float* density_grid,
    float mean_density, float bound, float dt_gamma, uint32_t N, uint32_t C,
    uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs,
    float* dirs, float* deltas, int32_t* rays, int32_t* counter,
    uint32_t perturb, void* workspace, void* stream) {
  UCSA_CHECK_ARG(rays_o, 0);
  UCSA_CHECK_ARG(rays_d, 1);
  UCSA_CHECK_ARG(density_grid, 2);
  UCSA_CHECK_ARG(bound > 0.f, 4);
  UCSA_CHECK_ARG(C >= 1 && C <= 32, 7);
  UCSA_CHECK_ARG(H >= 2 && (uint64_t)C * H * H * H < (1ull << 32), 8);
  UCSA_CHECK_ARG(nears && fars, 10);
  UCSA_CHECK_ARG(M == 0 || (xyzs && dirs && deltas), 12);
  UCSA_CHECK_ARG(rays, 15);
  UCSA_CHECK_ARG(counter, 16);
  UCSA_CHECK_ARG(workspace, 18);
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nb = ucsa_div_up(N, RM_BLOCK);
  uint32_t* ws = (uint32_t*)workspace;
  UCSA_CLEAR_ERR();
  if (N <= WM_MAX_RAYS) {
    uint32_t* steps = ws + 4 + nb;
    uint32_t* first = steps + N;
    const uint32_t nbw = ucsa_div_up(N, WM_WAVES);
    float* stage = N <= WM_STAGE_RAYS
        ? (float*)((char*)workspace + march_ws_head_bytes(N)) : nullptr;
    hipLaunchKernelGGL(k_march_count_w, dim3(nbw), dim3(64 * WM_WAVES), 0, s,
                       rays_o, rays_d, density_grid, mean_density, bound,
                       dt_gamma, N, C, H, nears, fars, perturb, steps, stage);
    hipLaunchKernelGGL(k_march_scan, dim3(1), dim3(1024), 0, s, N, steps, first,
                       counter, ws);
    if (stage) {
      hipLaunchKernelGGL(k_march_pack_w, dim3(nbw), dim3(64 * WM_WAVES), 0, s,
                         rays_d, N, M, stage, xyzs, dirs, deltas, rays, steps,
                         first, ws);
      return ucsa_launch_status();
    }
    hipLaunchKernelGGL(k_march_write_w, dim3(nbw), dim3(64 * WM_WAVES), 0, s,
                       rays_o, rays_d, density_grid, mean_density, bound,
                       dt_gamma, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                       rays, perturb, steps, first, ws);
    return ucsa_launch_status();
  }
  hipLaunchKernelGGL(k_march_count, dim3(nb), dim3(RM_BLOCK), 0, s, rays_o,
                     rays_d, density_grid, mean_density, bound, dt_gamma, N, C,
                     H, nears, fars, counter, perturb, ws);
  hipLaunchKernelGGL(k_march_write, dim3(nb), dim3(RM_BLOCK), 0, s, rays_o,
                     rays_d, density_grid, mean_density, bound, dt_gamma, N, C,
                     H, M, nears, fars, xyzs, dirs, deltas, rays, counter,
                     perturb, ws);
  return ucsa_launch_status();
}

// ===========================================================================
// composite_rays_train forward.  reference :318-394 (+ n_sem channels
// composited like rgb: the disabled variant raymarching.py:249-309).
// One wave per row of rays[]; 64 consecutive samples per trip.
// ===========================================================================
#define CT_WAVES 4

__global__ void __launch_bounds__(64 * CT_WAVES)
k_composite_train_fwd(const float* __restrict__ sigmas,
                      const float* __restrict__ rgbs,
                      const float* __restrict__ local_sem,
                      const float* __restrict__ deltas,
                      const int32_t* __restrict__ rays, uint32_t M, uint32_t N,
                      uint32_t n_sem, float* __restrict__ weights_sum,
                      float* __restrict__ depth, float* __restrict__ image,
                      float* __restrict__ semantics) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n = blockIdx.x * CT_WAVES + (threadIdx.x >> 6);
  if (n >= N) return;
  const uint32_t index = (uint32_t)rays[n * 3];
  const uint32_t offset = (uint32_t)rays[n * 3 + 1];
  uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
  if (num_steps == 0 || offset + num_steps >= M) num_steps = 0;

  float T_carry = 1.0f, t_carry = 0.0f;
  float r = 0, g = 0, b = 0, ws = 0, d = 0;
  // semantic classes: lane c owns class cb + c, summed in sample order
  const uint32_t n_cb = (n_sem + 63u) / 64u;
  float sem_acc[4] = {0, 0, 0, 0};  // up to 256 classes

  for (uint32_t s0 = 0; s0 < num_steps; s0 += 64) {
    const uint32_t s = s0 + lane;
    const bool live = s < num_steps;
    const size_t m = (size_t)offset + (live ? s : s0);
    const float sg = sigmas[m];
    const float2 dl = *reinterpret_cast<const float2*>(deltas + 2 * m);
    const float alpha = live ? 1.0f - __expf(-sg * dl.x) : 0.0f;
    const float Tin = wave_incl_scan_mul(1.0f - alpha, lane);
    float Tex = __shfl_up(Tin, 1, 64);
    if (lane == 0) Tex = 1.0f;
    const float w = alpha * (T_carry * Tex);
    const float t = t_carry + wave_incl_scan_add(live ? dl.y : 0.0f, lane);
    if (live) {
      r += w * rgbs[3 * m];
      g += w * rgbs[3 * m + 1];
      b += w * rgbs[3 * m + 2];
      d += w * t;
      ws += w;
    }
    if (n_sem) {
      const uint32_t cnt = min(64u, num_steps - s0);
      for (uint32_t cb = 0; cb < n_cb && cb < 4; ++cb) {
        const uint32_t c = cb * 64 + lane;
        const float* col = local_sem + ((size_t)offset + s0) * n_sem + c;
        float a = sem_acc[cb];
        for (uint32_t k = 0; k < cnt; ++k) {
          const float wk = __shfl(w, (int)k, 64);
          if (c < n_sem) a += wk * col[(size_t)k * n_sem];
        }
        sem_acc[cb] = a;
      }
    }
    T_carry *= __shfl(Tin, 63, 64);
    t_carry = __shfl(t, 63, 64);
  }
  r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
  ws = wave_sum(ws); d = wave_sum(d);
  if (lane == 0) {
    weights_sum[index] = ws;
    depth[index] = d;
    image[index * 3] = r;
    image[index * 3 + 1] = g;
    image[index * 3 + 2] = b;
  }
  for (uint32_t cb = 0; cb < n_cb && cb < 4; ++cb) {
    const uint32_t c = cb * 64 + lane;
    if (c < n_sem) semantics[(size_t)index * n_sem + c] = sem_acc[cb];
  }
}

extern "C" int32_t ucsa_composite_rays_train_fwd(
    const float* sigmas, const float* rgbs, const float* local_sem,
    const float* deltas, const int32_t* rays, uint32_t M, uint32_t N,
    uint32_t n_sem, float* weights_sum, float* depth, float* image,
    float* semantics, void* stream) {
  UCSA_CHECK_ARG(M == 0 || (sigmas && rgbs), 0);
  UCSA_CHECK_ARG(n_sem == 0 || M == 0 || local_sem, 2);
  UCSA_CHECK_ARG(M == 0 || deltas, 3);
  UCSA_CHECK_ARG(rays, 4);
  UCSA_CHECK_ARG(n_sem <= 256, 7);
  UCSA_CHECK_ARG(weights_sum && depth && image, 8);
  UCSA_CHECK_ARG(n_sem == 0 || semantics, 11);
  if (N == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_composite_train_fwd, dim3(ucsa_div_up(N, CT_WAVES)),
                     dim3(64 * CT_WAVES), 0, (hipStream_t)stream, sigmas, rgbs,
                     local_sem, deltas, rays, M, N, n_sem, weights_sum, depth,
                     image, semantics);
  return ucsa_launch_status();
}

// ===========================================================================
// composite_rays_train backward.  reference :408-487.  Semantic channels use
// DETACHED weights like the live path (reference renderer_semantics.py:268-271)
// : grad_local_sem = grad_semantics * w, nothing added to grad_sigmas.
// ===========================================================================
__global__ void __launch_bounds__(64 * CT_WAVES)
k_composite_train_bwd(const float* __restrict__ grad_ws,
                      const float* __restrict__ grad_image,
                      const float* __restrict__ grad_sem,
                      const float* __restrict__ sigmas,
                      const float* __restrict__ rgbs,
                      const float* __restrict__ deltas,
                      const int32_t* __restrict__ rays,
                      const float* __restrict__ weights_sum,
                      const float* __restrict__ image, uint32_t M, uint32_t N,
                      uint32_t n_sem, float* __restrict__ grad_sigmas,
                      float* __restrict__ grad_rgbs,
                      float* __restrict__ grad_local_sem) {
  __shared__ float w_tile[CT_WAVES][64];
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint32_t n = blockIdx.x * CT_WAVES + wid;
  if (n >= N) return;
  const uint32_t index = (uint32_t)rays[n * 3];
  const uint32_t offset = (uint32_t)rays[n * 3 + 1];
  const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
  if (num_steps == 0 || offset + num_steps >= M) return;

  const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1],
              gi2 = grad_image[index * 3 + 2], gw = grad_ws[index];
  const float rf = image[index * 3], gf = image[index * 3 + 1],
              bf = image[index * 3 + 2], wf = weights_sum[index];
  float T_carry = 1.0f, r_c = 0, g_c = 0, b_c = 0, ws_c = 0;

  for (uint32_t s0 = 0; s0 < num_steps; s0 += 64) {
    const uint32_t s = s0 + lane;
    const bool live = s < num_steps;
    const size_t m = (size_t)offset + (live ? s : s0);
    const float sg = sigmas[m];
    const float d0 = deltas[2 * m];
    const float c0 = rgbs[3 * m], c1 = rgbs[3 * m + 1], c2 = rgbs[3 * m + 2];
    const float alpha = live ? 1.0f - __expf(-sg * d0) : 0.0f;
    const float Tin = wave_incl_scan_mul(1.0f - alpha, lane);
    float Tex = __shfl_up(Tin, 1, 64);
    if (lane == 0) Tex = 1.0f;
    const float w = alpha * (T_carry * Tex);
    const float T = T_carry * Tin;  // transmittance after this sample
    const float r = r_c + wave_incl_scan_add(w * c0, lane);
    const float g = g_c + wave_incl_scan_add(w * c1, lane);
    const float b = b_c + wave_incl_scan_add(w * c2, lane);
    const float ws = ws_c + wave_incl_scan_add(w, lane);
    if (live) {
      grad_rgbs[3 * m] = gi0 * w;
      grad_rgbs[3 * m + 1] = gi1 * w;
      grad_rgbs[3 * m + 2] = gi2 * w;
      grad_sigmas[m] = d0 * (gi0 * (T * c0 - (rf - r)) + gi1 * (T * c1 - (gf - g)) +
                             gi2 * (T * c2 - (bf - b)) + gw * (T - (wf - ws)));
    }
    if (n_sem) {
      w_tile[wid][lane] = w;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t cnt = min(64u, num_steps - s0);
      const float* gs = grad_sem + (size_t)index * n_sem;
      float* out = grad_local_sem + ((size_t)offset + s0) * n_sem;
      for (uint32_t f = lane; f < cnt * n_sem; f += 64) {
        const uint32_t k = f / n_sem, c = f - k * n_sem;
        out[f] = gs[c] * w_tile[wid][k];
      }
      __builtin_amdgcn_wave_barrier();
    }
    T_carry = __shfl(T, 63, 64);
    r_c = __shfl(r, 63, 64);
    g_c = __shfl(g, 63, 64);
    b_c = __shfl(b, 63, 64);
    ws_c = __shfl(ws, 63, 64);
  }
}

extern "C" int32_t ucsa_composite_rays_train_bwd(
    const float* grad_weights_sum, const float* grad_image,
    const float* grad_semantics, const float* sigmas, const float* rgbs,
    const float* deltas, const int32_t* rays, const float* weights_sum,
    const float* image, uint32_t M, uint32_t N, uint32_t n_sem,
    float* grad_sigmas, float* grad_rgbs, float* grad_local_sem,
    void* stream) {
  UCSA_CHECK_ARG(grad_weights_sum && grad_image, 0);
  UCSA_CHECK_ARG(n_sem == 0 || grad_semantics, 2);
  UCSA_CHECK_ARG(M == 0 || (sigmas && rgbs && deltas), 3);
  UCSA_CHECK_ARG(rays, 6);
  UCSA_CHECK_ARG(weights_sum && image, 7);
  UCSA_CHECK_ARG(n_sem <= 256, 11);
  UCSA_CHECK_ARG(M == 0 || (grad_sigmas && grad_rgbs), 12);
  UCSA_CHECK_ARG(n_sem == 0 || M == 0 || grad_local_sem, 14);
  if (N == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_composite_train_bwd, dim3(ucsa_div_up(N, CT_WAVES)),
                     dim3(64 * CT_WAVES), 0, (hipStream_t)stream,
                     grad_weights_sum, grad_image, grad_semantics, sigmas, rgbs,
                     deltas, rays, weights_sum, image, M, N, n_sem, grad_sigmas,
                     grad_rgbs, grad_local_sem);
  return ucsa_launch_status();
}

// ===========================================================================
// march_rays (inference).  reference :528-634.  Rows past a ray's last step
// are left untouched (the caller zero-fills, raymarching.py:422-426).
// ===========================================================================
__global__ void __launch_bounds__(RM_BLOCK)
k_march_rays(uint32_t n_alive, uint32_t n_step,
             const int32_t* __restrict__ rays_alive,
             const float* __restrict__ rays_t, const float* __restrict__ rays_o,
             const float* __restrict__ rays_d, float bound, float dt_gamma,
             uint32_t C, uint32_t H, const float* __restrict__ grid,
             float mean_density, const float* __restrict__ fars,
             float* __restrict__ xyzs, float* __restrict__ dirs,
             float* __restrict__ deltas, uint32_t perturb) {
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  if (n >= n_alive) return;
  const uint32_t index = (uint32_t)rays_alive[n];
  const float far = fars[index];
  Marcher m(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid,
            mean_density, bound, dt_gamma, C, H, far);
  float t = rays_t[n];
  if (perturb) {
    Pcg32 rng((uint64_t)n, (uint64_t)perturb);
    t += RM_MIN_STEPSIZE * rng.next_float();
  }
  float* px = xyzs + (size_t)n * n_step * 3;
  float* pd = dirs + (size_t)n * n_step * 3;
  float* pl = deltas + (size_t)n * n_step * 2;
  float last_t = t, x, y, z;
  uint32_t step = 0;
  while (t < far && step < n_step) {
    if (m.probe(t, x, y, z)) {
      px[0] = x; px[1] = y; px[2] = z;
      pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
      const float dt = m.step_size(t);
      t += dt;
      pl[0] = dt;
      pl[1] = t - last_t;
      last_t = t;
      px += 3; pd += 3; pl += 2;
      ++step;
    }
  }
}

extern "C" int32_t ucsa_march_rays(
    uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive,
    const float* rays_t, const float* rays_o, const float* rays_d, float bound,
    float dt_gamma, uint32_t C, uint32_t H, const float* density_grid,
    float mean_density, const float* nears, const float* fars, float* xyzs,
    float* dirs, float* deltas, uint32_t perturb, void* stream) {
  UCSA_CHECK_ARG(n_step >= 1, 1);
  UCSA_CHECK_ARG(rays_alive, 2);
  UCSA_CHECK_ARG(rays_t, 3);
  UCSA_CHECK_ARG(rays_o && rays_d, 4);
  UCSA_CHECK_ARG(bound > 0.f, 6);
  UCSA_CHECK_ARG(C >= 1 && C <= 32, 8);
  UCSA_CHECK_ARG(H >= 2 && (uint64_t)C * H * H * H < (1ull << 32), 9);
  UCSA_CHECK_ARG(density_grid, 10);
  UCSA_CHECK_ARG(fars, 13);
  UCSA_CHECK_ARG(xyzs && dirs && deltas, 14);
  (void)nears;  // read by the reference (:563) but never used
  if (n_alive == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_march_rays, dim3(ucsa_div_up(n_alive, RM_BLOCK)),
                     dim3(RM_BLOCK), 0, (hipStream_t)stream, n_alive, n_step,
                     rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, C, H,
                     density_grid, mean_density, fars, xyzs, dirs, deltas,
                     perturb);
  return ucsa_launch_status();
}

// ===========================================================================
// composite_rays (inference, in place).  reference :647-729; semantic channels
// as in the disabled variant (:741-825, raymarching.py:507-558).
//   step(): one sample of the recurrence; returns false when the ray stops.
// `T < 1e-4` is a double comparison in the reference; for a float T that is
// T <= 1e-4f (1e-4f is the largest float below 1e-4).
// ===========================================================================
struct AliveAcc {
  float ws, t;
  __device__ __forceinline__ bool step(float sigma, float d0, float d1,
                                       float& w, bool& last) {
    if (d0 == 0.0f) return false;
    const float alpha = 1.0f - __expf(-sigma * d0);
    const float T = 1 - ws;
    w = alpha * T;
    ws += w;
    t += d1;
    last = T <= 1e-4f;
    return true;
  }
};

__global__ void __launch_bounds__(RM_BLOCK)
k_composite_rays(uint32_t n_alive, uint32_t n_step,
                 const int32_t* __restrict__ rays_alive,
                 float* __restrict__ rays_t, const float* __restrict__ sigmas,
                 const float* __restrict__ rgbs,
                 const float* __restrict__ deltas,
                 float* __restrict__ weights_sum, float* __restrict__ depth,
                 float* __restrict__ image) {
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  if (n >= n_alive) return;
  const uint32_t index = (uint32_t)rays_alive[n];
  AliveAcc a{weights_sum[index], rays_t[n]};
  float d = depth[index];
  float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
  uint32_t step = 0;
  while (step < n_step) {
    const size_t m = (size_t)n * n_step + step;
    float w;
    bool last;
    if (!a.step(sigmas[m], deltas[2 * m], deltas[2 * m + 1], w, last)) break;
    d += w * a.t;
    r += w * rgbs[3 * m];
    g += w * rgbs[3 * m + 1];
    b += w * rgbs[3 * m + 2];
    if (last) break;
    ++step;
  }
  rays_t[n] = step < n_step ? -1.0f : a.t;
  weights_sum[index] = a.ws;
  depth[index] = d;
  image[index * 3] = r;
  image[index * 3 + 1] = g;
  image[index * 3 + 2] = b;
}

// Semantic channels of the same update: one wave per alive ray, lane = class;
// every lane replays the (cheap, uniform) weight recurrence and adds its
// class in step order.  Runs BEFORE k_composite_rays (it needs the old
// weights_sum / rays_t and writes only `semantics`).
__global__ void __launch_bounds__(64 * CT_WAVES)
k_composite_rays_sem(uint32_t n_alive, uint32_t n_step,
                     const int32_t* __restrict__ rays_alive,
                     const float* __restrict__ rays_t,
                     const float* __restrict__ sigmas,
                     const float* __restrict__ local_sem,
                     const float* __restrict__ deltas, uint32_t n_sem,
                     const float* __restrict__ weights_sum,
                     float* __restrict__ semantics) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n = blockIdx.x * CT_WAVES + (threadIdx.x >> 6);
  if (n >= n_alive) return;
  const uint32_t index = (uint32_t)rays_alive[n];
  for (uint32_t cb = 0; cb < n_sem; cb += 64) {
    const uint32_t c = cb + lane;
    AliveAcc a{weights_sum[index], rays_t[n]};
    float acc = c < n_sem ? semantics[(size_t)index * n_sem + c] : 0.0f;
    for (uint32_t step = 0; step < n_step; ++step) {
      const size_t m = (size_t)n * n_step + step;
      float w;
      bool last;
      if (!a.step(sigmas[m], deltas[2 * m], deltas[2 * m + 1], w, last)) break;
      if (c < n_sem) acc += w * local_sem[m * n_sem + c];
      if (last) break;
    }
    if (c < n_sem) semantics[(size_t)index * n_sem + c] = acc;
  }
}

extern "C" int32_t ucsa_composite_rays(
    uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, float* rays_t,
    const float* sigmas, const float* rgbs, const float* local_sem,
    const float* deltas, uint32_t n_sem, float* weights_sum, float* depth,
    float* image, float* semantics, void* stream) {
  UCSA_CHECK_ARG(n_step >= 1, 1);
  UCSA_CHECK_ARG(rays_alive, 2);
  UCSA_CHECK_ARG(rays_t, 3);
  UCSA_CHECK_ARG(sigmas && rgbs, 4);
  UCSA_CHECK_ARG(n_sem == 0 || local_sem, 6);
  UCSA_CHECK_ARG(deltas, 7);
  UCSA_CHECK_ARG(weights_sum && depth && image, 9);
  UCSA_CHECK_ARG(n_sem == 0 || semantics, 12);
  if (n_alive == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  UCSA_CLEAR_ERR();
  if (n_sem)
    hipLaunchKernelGGL(k_composite_rays_sem, dim3(ucsa_div_up(n_alive, CT_WAVES)),
                       dim3(64 * CT_WAVES), 0, s, n_alive, n_step, rays_alive,
                       rays_t, sigmas, local_sem, deltas, n_sem, weights_sum,
                       semantics);
  hipLaunchKernelGGL(k_composite_rays, dim3(ucsa_div_up(n_alive, RM_BLOCK)),
                     dim3(RM_BLOCK), 0, s, n_alive, n_step, rays_alive, rays_t,
                     sigmas, rgbs, deltas, weights_sum, depth, image);
  return ucsa_launch_status();
}

// ===========================================================================
// compact_rays.  reference :838-855.  Stable (order-preserving) compaction:
// ballot + popcount within a wave, block sums, per-block prefix.
// workspace (uint32): [0] old alive_counter, [1..3] pad, [4 .. 4+nb) sums
// ===========================================================================
__device__ __forceinline__ uint32_t alive_rank(bool keep, uint32_t* sm,
                                               uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const uint64_t mask = __ballot(keep);
  const uint32_t below = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane == 0) sm[wid] = __popcll(mask);
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (uint32_t w = 0; w < RM_BLOCK / 64; ++w) {
    const uint32_t s = sm[w];
    if (w < wid) base += s;
    tot += s;
  }
  *total = tot;
  return base + below;
}

__global__ void __launch_bounds__(RM_BLOCK)
k_compact_count(uint32_t n_alive, const float* __restrict__ rays_t_old,
                const int32_t* __restrict__ alive_counter,
                uint32_t* __restrict__ ws) {
  __shared__ uint32_t sm[16];
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  const bool keep = n < n_alive && rays_t_old[n] >= 0.0f;
  uint32_t total;
  alive_rank(keep, sm, &total);
  if (threadIdx.x == 0) {
    ws[4 + blockIdx.x] = total;
    if (blockIdx.x == 0) ws[0] = (uint32_t)alive_counter[0];
  }
}

__global__ void __launch_bounds__(RM_BLOCK)
k_compact_write(uint32_t n_alive, int32_t* __restrict__ rays_alive,
                const int32_t* __restrict__ rays_alive_old,
                float* __restrict__ rays_t,
                const float* __restrict__ rays_t_old,
                int32_t* __restrict__ alive_counter,
                const uint32_t* __restrict__ ws) {
  __shared__ uint32_t sm[16];
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  const float t = n < n_alive ? rays_t_old[n] : -1.0f;
  const bool keep = n < n_alive && t >= 0.0f;
  const uint32_t before = prefix_of_blocks(ws + 4, blockIdx.x, sm);
  uint32_t total;
  const uint32_t rank = alive_rank(keep, sm, &total);
  const uint32_t base = ws[0];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    alive_counter[0] = (int32_t)(base + before + total);
  if (keep) {
    const uint32_t k = base + before + rank;
    rays_alive[k] = rays_alive_old[n];
    rays_t[k] = t;
  }
}

extern "C" uint64_t ucsa_compact_workspace_bytes(uint32_t n_alive) {
  return 4ull * (4ull + ucsa_div_up(n_alive ? n_alive : 1, RM_BLOCK));
}

extern "C" int32_t ucsa_compact_rays(uint32_t n_alive, int32_t* rays_alive,
                                     const int32_t* rays_alive_old,
                                     float* rays_t, const float* rays_t_old,
                                     int32_t* alive_counter, void* workspace,
                                     void* stream) {
  UCSA_CHECK_ARG(rays_alive && rays_alive_old, 1);
  UCSA_CHECK_ARG(rays_t && rays_t_old, 3);
  UCSA_CHECK_ARG(alive_counter, 5);
  UCSA_CHECK_ARG(workspace, 6);
  if (n_alive == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nb = ucsa_div_up(n_alive, RM_BLOCK);
  uint32_t* ws = (uint32_t*)workspace;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(RM_BLOCK), 0, s, n_alive,
                     rays_t_old, alive_counter, ws);
  hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(RM_BLOCK), 0, s, n_alive,
                     rays_alive, rays_alive_old, rays_t, rays_t_old,
                     alive_counter, ws);
  return ucsa_launch_status();
}

// ===========================================================================
// Density-grid maintenance.  The reference keeps the state (density_grid
// [cascade,128,128,128], mean_density, iter_density:
// renderer_semantics.py:91-103,111-121) but ships no updater -- its parent
// code base refreshed the grid from the field before each epoch.  These two
// kernels fill that gap for cuda_ray=True:
//   cell_points : one jittered point per cell of one cascade (the cell the
//                 marcher's lookup maps that point back to, see probe());
//   ema_update  : grid = max(grid * decay, fresh) where both are >= 0, and the
//                 mean of max(grid, 0) (deterministic block partials).
// ===========================================================================
__global__ void __launch_bounds__(RM_BLOCK)
k_grid_cell_points(uint32_t cas, uint32_t H, float bound, uint32_t seed,
                   float* __restrict__ xyz) {
  const uint32_t i = blockIdx.x * RM_BLOCK + threadIdx.x;
  const uint32_t n = H * H * H;
  if (i >= n) return;
  const uint32_t ix = i / (H * H), iy = (i / H) % H, iz = i % H;
  const float b = fminf(exp2f((float)cas), bound);
  const float half_cell = b / (float)H;
  float jx = 0.f, jy = 0.f, jz = 0.f;
  if (seed) {
    Pcg32 rng((uint64_t)cas * n + i, (uint64_t)seed);
    jx = rng.next_float() * 2 - 1;
    jy = rng.next_float() * 2 - 1;
    jz = rng.next_float() * 2 - 1;
  }
  xyz[(size_t)i * 3] = b * ((2 * ix + 1) / (float)H - 1) + jx * half_cell;
  xyz[(size_t)i * 3 + 1] = b * ((2 * iy + 1) / (float)H - 1) + jy * half_cell;
  xyz[(size_t)i * 3 + 2] = b * ((2 * iz + 1) / (float)H - 1) + jz * half_cell;
}

extern "C" int32_t ucsa_density_grid_points(uint32_t cascade, uint32_t H,
                                            float bound, uint32_t seed,
                                            float* xyz, void* stream) {
  UCSA_CHECK_ARG(cascade < 32, 0);
  UCSA_CHECK_ARG(H >= 2 && H <= 1024, 1);
  UCSA_CHECK_ARG(bound > 0.f, 2);
  UCSA_CHECK_ARG(xyz, 4);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_grid_cell_points, dim3(ucsa_div_up((uint64_t)H * H * H, RM_BLOCK)),
                     dim3(RM_BLOCK), 0, (hipStream_t)stream, cascade, H, bound,
                     seed, xyz);
  return ucsa_launch_status();
}

#define EMA_BLOCKS 1024

__global__ void __launch_bounds__(RM_BLOCK)
k_grid_ema(float* __restrict__ grid, const float* __restrict__ fresh,
           uint64_t n, float decay, float fresh_scale,
           float* __restrict__ partial) {
  __shared__ float sm[RM_BLOCK / 64];
  float acc = 0.f;
  for (uint64_t i = (uint64_t)blockIdx.x * RM_BLOCK + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * RM_BLOCK) {
    float g = grid[i];
    const float f = fresh[i] * fresh_scale;
    if (g >= 0.f && f >= 0.f) {
      g = fmaxf(g * decay, f);
      grid[i] = g;
    }
    acc += fmaxf(g, 0.f);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63u) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < RM_BLOCK / 64; ++w) s += sm[w];
    partial[blockIdx.x] = s;
  }
}

__global__ void k_grid_mean(const float* __restrict__ partial, uint32_t nb,
                            uint64_t n, float* __restrict__ mean) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0.0;
  for (uint32_t b = 0; b < nb; ++b) s += (double)partial[b];
  mean[0] = (float)(s / (double)n);
}

extern "C" uint64_t ucsa_density_grid_workspace_bytes(void) {
  return 4ull * EMA_BLOCKS;
}

extern "C" int32_t ucsa_density_grid_update(float* density_grid,
                                            const float* fresh, uint64_t n,
                                            float decay, float fresh_scale,
                                            float* mean_density,
                                            void* workspace, void* stream) {
  UCSA_CHECK_ARG(density_grid, 0);
  UCSA_CHECK_ARG(fresh, 1);
  UCSA_CHECK_ARG(n > 0, 2);
  UCSA_CHECK_ARG(mean_density, 5);
  UCSA_CHECK_ARG(workspace, 6);
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nb = (uint32_t)(n / RM_BLOCK + 1 < EMA_BLOCKS ? n / RM_BLOCK + 1
                                                              : EMA_BLOCKS);
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_grid_ema, dim3(nb), dim3(RM_BLOCK), 0, s, density_grid,
                     fresh, n, decay, fresh_scale, (float*)workspace);
  hipLaunchKernelGGL(k_grid_mean, dim3(1), dim3(64), 0, s,
                     (const float*)workspace, nb, n, mean_density);
  return ucsa_launch_status();
}

// ===========================================================================
// Segmented marching: the MI355X-native driver of the inference loop.
//
// The reference API above marches `n_step` samples for every alive ray into a
// zero-padded [n_alive*n_step] buffer, evaluates the field on all of it and
// asks the host for the survivor count after every iteration.  Here a round
// marches up to `cap` samples per alive ray into an EXACT-SIZE buffer (count
// -> prefix sums -> write, as in march_rays_train), the field runs once on
// those points, and one wave per ray composites its span with the reference's
// early-termination rule (stop after the first sample whose incoming
// transmittance is below 1e-4, raymarching.cu:693-706).  A few rounds with
// growing caps replace ~100 host-synchronised iterations; the samples a ray
// takes, and therefore the result, are the same up to fp32 re-association.
//
// The alive count of a round lives in device memory (`n_alive_dev`); launches
// are sized by the host's upper bound `n_cap` and surplus lanes exit.
// workspace (uint32): [0] total points, [1] n_alive seen, [2..3] pad,
//                     [4 .. 4+nb) block sums.   span[n] = (first point, count)
// ===========================================================================
__device__ __forceinline__ uint32_t seg_n_alive(const int32_t* n_alive_dev,
                                                uint32_t n_cap) {
  if (!n_alive_dev) return n_cap;
  const uint32_t n = (uint32_t)n_alive_dev[0];
  return n < n_cap ? n : n_cap;
}

__global__ void __launch_bounds__(RM_BLOCK)
k_seg_count(uint32_t n_cap, const int32_t* __restrict__ n_alive_dev,
            uint32_t cap, const int32_t* __restrict__ rays_alive,
            const float* __restrict__ rays_t, const float* __restrict__ rays_o,
            const float* __restrict__ rays_d, float bound, float dt_gamma,
            uint32_t C, uint32_t H, const float* __restrict__ grid,
            float mean_density, const float* __restrict__ fars,
            uint32_t perturb, int32_t* __restrict__ span,
            uint32_t* __restrict__ ws, float* __restrict__ stage) {
  __shared__ uint32_t sm[16];
  const uint32_t n_alive = seg_n_alive(n_alive_dev, n_cap);
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  uint32_t steps = 0;
  if (n < n_alive) {
    const uint32_t index = (uint32_t)rays_alive[n];
    const float far = fars[index];
    Marcher m(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid,
              mean_density, bound, dt_gamma, C, H, far);
    float t = rays_t[n];
    float last_t = t;  // as k_seg_write: the jitter is part of the first delta
    if (perturb) {
      Pcg32 rng((uint64_t)index, (uint64_t)perturb);
      t += RM_MIN_STEPSIZE * rng.next_float();
    }
    float x, y, z;
    // stage != NULL: the samples go to slot-major staging rows [n*cap + k]
    // = (x, y, z, dt, t - last_t) right away and k_seg_pack copies them to
    // their exact-size place once the offsets are known -- one march per
    // round instead of two
    float* st = stage ? stage + (size_t)n * cap * 5 : nullptr;
    while (t < far && steps < cap) {
      if (m.probe(t, x, y, z)) {
        const float dt = m.step_size(t);
        t += dt;
        if (st) {
          st[0] = x; st[1] = y; st[2] = z; st[3] = dt; st[4] = t - last_t;
          st += 5;
          last_t = t;
        }
        ++steps;
      }
    }
    span[2 * n + 1] = (int32_t)steps;
  }
  uint32_t total;
  block_excl_scan(steps, sm, &total);
  if (threadIdx.x == 0) ws[4 + blockIdx.x] = total;
}

__global__ void __launch_bounds__(RM_BLOCK)
k_seg_offsets(uint32_t n_cap, const int32_t* __restrict__ n_alive_dev,
              int32_t* __restrict__ span, uint32_t* __restrict__ ws) {
  __shared__ uint32_t sm[16];
  const uint32_t n_alive = seg_n_alive(n_alive_dev, n_cap);
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  const uint32_t steps = n < n_alive ? (uint32_t)span[2 * n + 1] : 0u;
  const uint32_t before = prefix_of_blocks(ws + 4, blockIdx.x, sm);
  uint32_t total;
  const uint32_t in_block = block_excl_scan(steps, sm, &total);
  if (n < n_alive) span[2 * n] = (int32_t)(before + in_block);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    ws[0] = before + total;
    ws[1] = n_alive;
  }
}

__global__ void __launch_bounds__(RM_BLOCK)
k_seg_write(uint32_t n_cap, const int32_t* __restrict__ n_alive_dev,
            const int32_t* __restrict__ rays_alive,
            const float* __restrict__ rays_t, const float* __restrict__ rays_o,
            const float* __restrict__ rays_d, float bound, float dt_gamma,
            uint32_t C, uint32_t H, const float* __restrict__ grid,
            float mean_density, const float* __restrict__ fars,
            uint32_t perturb, const int32_t* __restrict__ span,
            float* __restrict__ xyzs, float* __restrict__ dirs,
            float* __restrict__ deltas) {
  const uint32_t n_alive = seg_n_alive(n_alive_dev, n_cap);
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  if (n >= n_alive) return;
  const uint32_t count = (uint32_t)span[2 * n + 1];
  if (count == 0) return;
  const uint32_t index = (uint32_t)rays_alive[n];
  const float far = fars[index];
  Marcher m(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid,
            mean_density, bound, dt_gamma, C, H, far);
  float t = rays_t[n];
  // the depth deltas are measured from the slot's incoming t (like
  // kernel_march_rays :549,:577): the jitter is part of the first delta
  float last_t = t;
  if (perturb) {
    Pcg32 rng((uint64_t)index, (uint64_t)perturb);
    t += RM_MIN_STEPSIZE * rng.next_float();
  }
  const size_t p0 = (size_t)(uint32_t)span[2 * n];
  float* px = xyzs + p0 * 3;
  float* pd = dirs + p0 * 3;
  float* pl = deltas + p0 * 2;
  float x, y, z;
  uint32_t step = 0;
  while (t < far && step < count) {
    if (m.probe(t, x, y, z)) {
      px[0] = x; px[1] = y; px[2] = z;
      pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
      const float dt = m.step_size(t);
      t += dt;
      pl[0] = dt;
      pl[1] = t - last_t;
      last_t = t;
      px += 3; pd += 3; pl += 2;
      ++step;
    }
  }
}

// staging rows -> exact-size buffers: thread (slot n, sample k)
__global__ void __launch_bounds__(RM_BLOCK)
k_seg_pack(uint32_t n_cap, const int32_t* __restrict__ n_alive_dev,
           uint32_t cap, const int32_t* __restrict__ rays_alive,
           const float* __restrict__ rays_d, const int32_t* __restrict__ span,
           const float* __restrict__ stage, float* __restrict__ xyzs,
           float* __restrict__ dirs, float* __restrict__ deltas) {
  const uint32_t n_alive = seg_n_alive(n_alive_dev, n_cap);
  const uint64_t e = (uint64_t)blockIdx.x * RM_BLOCK + threadIdx.x;
  const uint32_t n = (uint32_t)(e / cap), k = (uint32_t)(e % cap);
  if (n >= n_alive || k >= (uint32_t)span[2 * n + 1]) return;
  const float* st = stage + ((size_t)n * cap + k) * 5;
  const size_t p = (size_t)(uint32_t)span[2 * n] + k;
  const float* d = rays_d + (size_t)(uint32_t)rays_alive[n] * 3;
  xyzs[p * 3] = st[0]; xyzs[p * 3 + 1] = st[1]; xyzs[p * 3 + 2] = st[2];
  dirs[p * 3] = d[0]; dirs[p * 3 + 1] = d[1]; dirs[p * 3 + 2] = d[2];
  deltas[p * 2] = st[3]; deltas[p * 2 + 1] = st[4];
}

extern "C" uint64_t ucsa_march_segment_stage_bytes(uint32_t n_cap,
                                                   uint32_t cap) {
  return (uint64_t)n_cap * cap * 5 * sizeof(float);
}

extern "C" uint64_t ucsa_march_segment_workspace_bytes(uint32_t n_cap) {
  return 4ull * (4ull + ucsa_div_up(n_cap ? n_cap : 1, RM_BLOCK));
}

extern "C" int32_t ucsa_march_segment_count(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, const float* rays_t, const float* rays_o,
    const float* rays_d, float bound, float dt_gamma, uint32_t C, uint32_t H,
    const float* density_grid, float mean_density, const float* fars,
    uint32_t perturb, int32_t* span, void* workspace, float* stage,
    void* stream) {
  UCSA_CHECK_ARG(cap >= 1 && cap <= RM_MAX_STEPS, 2);
  UCSA_CHECK_ARG(rays_alive, 3);
  UCSA_CHECK_ARG(rays_t, 4);
  UCSA_CHECK_ARG(rays_o && rays_d, 5);
  UCSA_CHECK_ARG(bound > 0.f, 7);
  UCSA_CHECK_ARG(C >= 1 && C <= 32, 9);
  UCSA_CHECK_ARG(H >= 2 && (uint64_t)C * H * H * H < (1ull << 32), 10);
  UCSA_CHECK_ARG(density_grid, 11);
  UCSA_CHECK_ARG(fars, 13);
  UCSA_CHECK_ARG(span, 15);
  UCSA_CHECK_ARG(workspace, 16);
  hipStream_t s = (hipStream_t)stream;
  uint32_t* ws = (uint32_t*)workspace;
  UCSA_CLEAR_ERR();
  if (n_cap == 0) {
    (void)hipMemsetAsync(ws, 0, 16, s);
    return ucsa_launch_status();
  }
  const uint32_t nb = ucsa_div_up(n_cap, RM_BLOCK);
  hipLaunchKernelGGL(k_seg_count, dim3(nb), dim3(RM_BLOCK), 0, s, n_cap,
                     n_alive_dev, cap, rays_alive, rays_t, rays_o, rays_d,
                     bound, dt_gamma, C, H, density_grid, mean_density, fars,
                     perturb, span, ws, stage);
  hipLaunchKernelGGL(k_seg_offsets, dim3(nb), dim3(RM_BLOCK), 0, s, n_cap,
                     n_alive_dev, span, ws);
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_march_segment_write(
    uint32_t n_cap, const int32_t* n_alive_dev, const int32_t* rays_alive,
    const float* rays_t, const float* rays_o, const float* rays_d, float bound,
    float dt_gamma, uint32_t C, uint32_t H, const float* density_grid,
    float mean_density, const float* fars, uint32_t perturb,
    const int32_t* span, float* xyzs, float* dirs, float* deltas,
    const float* stage, uint32_t cap, void* stream) {
  UCSA_CHECK_ARG(rays_alive, 2);
  UCSA_CHECK_ARG(rays_t, 3);
  UCSA_CHECK_ARG(rays_o && rays_d, 4);
  UCSA_CHECK_ARG(density_grid, 10);
  UCSA_CHECK_ARG(fars, 12);
  UCSA_CHECK_ARG(span, 14);
  UCSA_CHECK_ARG(xyzs && dirs && deltas, 15);
  if (n_cap == 0) return 0;
  UCSA_CLEAR_ERR();
  if (stage) {  // the samples were staged by ucsa_march_segment_count
    UCSA_CHECK_ARG(cap >= 1 && cap <= RM_MAX_STEPS, 19);
    hipLaunchKernelGGL(k_seg_pack, dim3(ucsa_div_up((uint64_t)n_cap * cap, RM_BLOCK)),
                       dim3(RM_BLOCK), 0, (hipStream_t)stream, n_cap,
                       n_alive_dev, cap, rays_alive, rays_d, span, stage, xyzs,
                       dirs, deltas);
    return ucsa_launch_status();
  }
  hipLaunchKernelGGL(k_seg_write, dim3(ucsa_div_up(n_cap, RM_BLOCK)),
                     dim3(RM_BLOCK), 0, (hipStream_t)stream, n_cap, n_alive_dev,
                     rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, C, H,
                     density_grid, mean_density, fars, perturb, span, xyzs,
                     dirs, deltas);
  return ucsa_launch_status();
}

// One wave per alive slot: composite the slot's span onto the ray's running
// sums with early termination; rays_t[n] <- -1 when the ray is finished (it
// stopped early, or its span came back shorter than `cap`, i.e. it reached
// far), else the ray parameter after its last sample.
__global__ void __launch_bounds__(64 * CT_WAVES)
k_seg_composite(uint32_t n_cap, const int32_t* __restrict__ n_alive_dev,
                uint32_t cap, const int32_t* __restrict__ rays_alive,
                float* __restrict__ rays_t, const int32_t* __restrict__ span,
                const float* __restrict__ sigmas, float sigma_scale,
                const float* __restrict__ rgbs,
                const float* __restrict__ local_sem,
                const float* __restrict__ deltas, uint32_t n_sem,
                float* __restrict__ weights_sum, float* __restrict__ depth,
                float* __restrict__ image, float* __restrict__ semantics) {
  const uint32_t n_alive = seg_n_alive(n_alive_dev, n_cap);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n = blockIdx.x * CT_WAVES + (threadIdx.x >> 6);
  if (n >= n_alive) return;
  const uint32_t index = (uint32_t)rays_alive[n];
  const uint32_t offset = (uint32_t)span[2 * n];
  const uint32_t count = (uint32_t)span[2 * n + 1];
  float T_carry = 1.0f - weights_sum[index];
  float t_carry = rays_t[n];
  bool stopped = false;  // a sample with incoming T <= 1e-4 has been taken
  float r = 0, g = 0, b = 0, ws = 0, d = 0;
  const uint32_t n_cb = (n_sem + 63u) / 64u;
  float sem_acc[4] = {0, 0, 0, 0};

  for (uint32_t s0 = 0; s0 < count && !stopped; s0 += 64) {
    const uint32_t s = s0 + lane;
    const bool live = s < count;
    const size_t m = (size_t)offset + (live ? s : s0);
    const float sg = sigmas[m] * sigma_scale;
    const float2 dl = *reinterpret_cast<const float2*>(deltas + 2 * m);
    const float alpha = live ? 1.0f - __expf(-sg * dl.x) : 0.0f;
    const float Tin = wave_incl_scan_mul(1.0f - alpha, lane);
    float Tex = __shfl_up(Tin, 1, 64);
    if (lane == 0) Tex = 1.0f;
    const float T = T_carry * Tex;  // transmittance reaching this sample
    // the reference takes the sample, then stops if T < 1e-4 (double compare
    // == T <= 1e-4f): a sample is used iff no earlier one saw T <= 1e-4
    const uint64_t stop_mask = __ballot(live && T <= 1e-4f);
    const uint32_t first_stop = stop_mask ? (uint32_t)__ffsll((long long)stop_mask) - 1u : 64u;
    const bool use = live && lane <= first_stop;
    const float w = use ? alpha * T : 0.0f;
    const float t = t_carry + wave_incl_scan_add(live ? dl.y : 0.0f, lane);
    if (use) {
      r += w * rgbs[3 * m];
      g += w * rgbs[3 * m + 1];
      b += w * rgbs[3 * m + 2];
      d += w * t;
      ws += w;
    }
    if (n_sem) {
      const uint32_t cnt = min(min(64u, count - s0), first_stop + 1u);
      for (uint32_t cb = 0; cb < n_cb && cb < 4; ++cb) {
        const uint32_t c = cb * 64 + lane;
        const float* col = local_sem + ((size_t)offset + s0) * n_sem + c;
        float a = sem_acc[cb];
        for (uint32_t k = 0; k < cnt; ++k) {
          const float wk = __shfl(w, (int)k, 64);
          if (c < n_sem) a += wk * col[(size_t)k * n_sem];
        }
        sem_acc[cb] = a;
      }
    }
    stopped = stop_mask != 0;
    T_carry *= __shfl(Tin, 63, 64);
    t_carry = __shfl(t, 63, 64);
  }
  r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
  ws = wave_sum(ws); d = wave_sum(d);
  if (lane == 0) {
    weights_sum[index] += ws;
    depth[index] += d;
    image[index * 3] += r;
    image[index * 3 + 1] += g;
    image[index * 3 + 2] += b;
    rays_t[n] = (stopped || count < cap) ? -1.0f : t_carry;
  }
  for (uint32_t cb = 0; cb < n_cb && cb < 4; ++cb) {
    const uint32_t c = cb * 64 + lane;
    if (c < n_sem) semantics[(size_t)index * n_sem + c] += sem_acc[cb];
  }
}

extern "C" int32_t ucsa_march_segment_composite(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, float* rays_t, const int32_t* span,
    const float* sigmas, float sigma_scale, const float* rgbs,
    const float* local_sem, const float* deltas, uint32_t n_sem,
    float* weights_sum, float* depth, float* image, float* semantics,
    void* stream) {
  UCSA_CHECK_ARG(rays_alive, 3);
  UCSA_CHECK_ARG(rays_t, 4);
  UCSA_CHECK_ARG(span, 5);
  UCSA_CHECK_ARG(sigmas && rgbs, 6);
  UCSA_CHECK_ARG(n_sem == 0 || local_sem, 9);
  UCSA_CHECK_ARG(deltas, 10);
  UCSA_CHECK_ARG(n_sem <= 256, 11);
  UCSA_CHECK_ARG(weights_sum && depth && image, 12);
  UCSA_CHECK_ARG(n_sem == 0 || semantics, 15);
  if (n_cap == 0) return 0;
  UCSA_CLEAR_ERR();
  hipLaunchKernelGGL(k_seg_composite, dim3(ucsa_div_up(n_cap, CT_WAVES)),
                     dim3(64 * CT_WAVES), 0, (hipStream_t)stream, n_cap,
                     n_alive_dev, cap, rays_alive, rays_t, span, sigmas,
                     sigma_scale, rgbs, local_sem, deltas, n_sem, weights_sum,
                     depth, image, semantics);
  return ucsa_launch_status();
}

// compact_rays with the alive count taken from device memory
// (count_in_dev may alias nothing written here; count_out_dev receives the
// number of survivors, starting from 0).
__global__ void __launch_bounds__(RM_BLOCK)
k_seg_compact_count(uint32_t n_cap, const int32_t* __restrict__ n_alive_dev,
                    const float* __restrict__ rays_t_old,
                    uint32_t* __restrict__ ws) {
  __shared__ uint32_t sm[16];
  const uint32_t n_alive = seg_n_alive(n_alive_dev, n_cap);
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  const bool keep = n < n_alive && rays_t_old[n] >= 0.0f;
  uint32_t total;
  alive_rank(keep, sm, &total);
  if (threadIdx.x == 0) ws[4 + blockIdx.x] = total;
}

__global__ void __launch_bounds__(RM_BLOCK)
k_seg_compact_write(uint32_t n_cap, const int32_t* __restrict__ n_alive_dev,
                    int32_t* __restrict__ rays_alive,
                    const int32_t* __restrict__ rays_alive_old,
                    float* __restrict__ rays_t,
                    const float* __restrict__ rays_t_old,
                    int32_t* __restrict__ n_alive_out,
                    const uint32_t* __restrict__ ws) {
  __shared__ uint32_t sm[16];
  const uint32_t n_alive = seg_n_alive(n_alive_dev, n_cap);
  const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
  const float t = n < n_alive ? rays_t_old[n] : -1.0f;
  const bool keep = n < n_alive && t >= 0.0f;
  const uint32_t before = prefix_of_blocks(ws + 4, blockIdx.x, sm);
  uint32_t total;
  const uint32_t rank = alive_rank(keep, sm, &total);
  if (keep) {
    rays_alive[before + rank] = rays_alive_old[n];
    rays_t[before + rank] = t;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    n_alive_out[0] = (int32_t)(before + total);
}

extern "C" int32_t ucsa_march_segment_compact(
    uint32_t n_cap, const int32_t* n_alive_dev, int32_t* rays_alive,
    const int32_t* rays_alive_old, float* rays_t, const float* rays_t_old,
    int32_t* n_alive_out, void* workspace, void* stream) {
  UCSA_CHECK_ARG(rays_alive && rays_alive_old, 2);
  UCSA_CHECK_ARG(rays_t && rays_t_old, 4);
  UCSA_CHECK_ARG(n_alive_out && n_alive_out != n_alive_dev, 6);
  UCSA_CHECK_ARG(workspace, 7);
  hipStream_t s = (hipStream_t)stream;
  UCSA_CLEAR_ERR();
  if (n_cap == 0) {
    (void)hipMemsetAsync(n_alive_out, 0, 4, s);
    return ucsa_launch_status();
  }
  const uint32_t nb = ucsa_div_up(n_cap, RM_BLOCK);
  uint32_t* ws = (uint32_t*)workspace;
  hipLaunchKernelGGL(k_seg_compact_count, dim3(nb), dim3(RM_BLOCK), 0, s, n_cap,
                     n_alive_dev, rays_t_old, ws);
  hipLaunchKernelGGL(k_seg_compact_write, dim3(nb), dim3(RM_BLOCK), 0, s, n_cap,
                     n_alive_dev, rays_alive, rays_alive_old, rays_t,
                     rays_t_old, n_alive_out, ws);
  return ucsa_launch_status();
}
