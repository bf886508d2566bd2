/*
 * CPU oracle for the occupancy-grid ray-marching functions
 * -- TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this.
 *
 * A scalar, single-thread fp32 restatement of the CUDA kernels of
 *   reference nr4seg/nerf/raymarching/src/raymarching.cu
 *     kernel_march_rays_train               :138-307
 *     kernel_composite_rays_train_forward   :318-394
 *     kernel_composite_rays_train_backward  :408-487
 *     kernel_march_rays                     :528-634
 *     kernel_composite_rays                 :647-729
 *     kernel_compact_rays                   :838-855
 *   and of the PCG32 generator it seeds per ray (src/pcg32.h:44-117).
 *
 * PARITY UNPINNED for the marching kernels: the reference is CUDA-only and
 * cannot run in this image, and ships no test vectors.  Pinned here by
 *   - the PCG32 known-answer sequence published with pcg-c-basic
 *     (seed 42, stream 54), and
 *   - hand-derived KATs (tests/test_oracle_raymarch.py): full / empty grids,
 *     closed-form step counts, composite against a float64 closed form, the
 *     backward against autograd of the forward.
 *
 * Where the CUDA kernels leave the result to the hardware scheduler (the
 * atomicAdd that hands out output spans / compacted slots in whatever order
 * threads arrive) this file takes rays in index order, which is one of the
 * orders the reference can produce.
 *
 * Arithmetic notes: every expression is evaluated in fp32 in source order
 * (build with -ffp-contract=off); nvcc may contract a*b+c into an FMA at its
 * own discretion, which is not restated.  __expf (reference :359,:451,:685)
 * is the fast-math exponential; expf is used here.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORC_MAX_STEPS 1024u                       /* reference :23 */
#define ORC_DENSITY_THRESH 0.01f                  /* reference :21 */
#define ORC_SQRT3 1.73205080757f                  /* reference :22 */
#define ORC_MIN_STEPSIZE (2 * ORC_SQRT3 / 1024)   /* reference :24 */

/* ---------------------------------------------------------------- PCG32 -- */
typedef struct { uint64_t state, inc; } orc_pcg32;

static uint32_t pcg_next(orc_pcg32* r) {
  const uint64_t old = r->state;
  r->state = old * 0x5851f42d4c957f2dULL + r->inc;
  const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  const uint32_t rot = (uint32_t)(old >> 59u);
  return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
}

static void pcg_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq) {
  r->state = 0u;
  r->inc = (initseq << 1u) | 1u;
  pcg_next(r);
  r->state += initstate;
  pcg_next(r);
}

static float pcg_float(orc_pcg32* r) {
  union { uint32_t u; float f; } x;
  x.u = (pcg_next(r) >> 9) | 0x3f800000u;
  return x.f - 1.0f;
}

/* first `n` raw outputs of pcg32(initstate, initseq) -- for the KAT */
void orc_pcg32_sequence(uint64_t initstate, uint64_t initseq, uint32_t n,
                        uint32_t* out) {
  orc_pcg32 r;
  pcg_seed(&r, initstate, initseq);
  for (uint32_t i = 0; i < n; ++i) out[i] = pcg_next(&r);
}

float orc_pcg32_first_float(uint64_t initstate, uint64_t initseq) {
  orc_pcg32 r;
  pcg_seed(&r, initstate, initseq);
  return pcg_float(&r);
}

/* -------------------------------------------------------------- marcher -- */
static float clampf(float x, float lo, float hi) {
  return fminf(hi, fmaxf(lo, x));
}

typedef struct {
  float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
  float bound, dt_gamma, dt_min, dt_max, thresh;
  uint32_t C, H;
  const float* grid;
} orc_ray;

static void ray_init(orc_ray* r, const float* o, const float* d,
                     const float* grid, float mean_density, float bound,
                     float dt_gamma, uint32_t C, uint32_t H) {
  r->ox = o[0]; r->oy = o[1]; r->oz = o[2];
  r->dx = d[0]; r->dy = d[1]; r->dz = d[2];
  r->rdx = 1 / r->dx; r->rdy = 1 / r->dy; r->rdz = 1 / r->dz;
  r->bound = bound;
  r->dt_gamma = dt_gamma;
  r->dt_min = ORC_MIN_STEPSIZE;
  r->dt_max = 2 * bound / H;
  r->thresh = fminf(ORC_DENSITY_THRESH, mean_density);
  r->C = C; r->H = H;
  r->grid = grid;
}

/* One probe of the cascade grid at ray parameter *t.  Returns 1 and the point
 * when the cell is occupied (the caller then takes one step), else jumps *t
 * past the cell and returns 0.  reference :188-226 (and the identical bodies
 * at :255-305, :580-632). */
static int probe(const orc_ray* r, float* t, float* px, float* py, float* pz) {
  const float x = clampf(r->ox + *t * r->dx, -r->bound, r->bound);
  const float y = clampf(r->oy + *t * r->dy, -r->bound, r->bound);
  const float z = clampf(r->oz + *t * r->dz, -r->bound, r->bound);
  const float H = (float)r->H;

  const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
  int e;
  frexpf(mx, &e);
  const int level = (int)fminf((float)r->C - 1, fmaxf(0.0f, (float)e));
  const float mip_bound = fminf(exp2f((float)level), r->bound);
  const float mip_rbound = 1 / mip_bound;

  const int nx = (int)clampf(0.5f * (x * mip_rbound + 1) * H, 0.0f, H - 1);
  const int ny = (int)clampf(0.5f * (y * mip_rbound + 1) * H, 0.0f, H - 1);
  const int nz = (int)clampf(0.5f * (z * mip_rbound + 1) * H, 0.0f, H - 1);
  const uint32_t h = r->H;
  const uint32_t idx = (uint32_t)level * h * h * h + (uint32_t)nx * h * h +
                       (uint32_t)ny * h + (uint32_t)nz;
  if (r->grid[idx] > r->thresh) {
    *px = x; *py = y; *pz = z;
    return 1;
  }
  const float hm1 = (float)(r->H - 1);
  const float tx = (((nx + 0.5f + 0.5f * copysignf(1.0f, r->dx)) / hm1 * 2 - 1) * mip_bound - x) * r->rdx;
  const float ty = (((ny + 0.5f + 0.5f * copysignf(1.0f, r->dy)) / hm1 * 2 - 1) * mip_bound - y) * r->rdy;
  const float tz = (((nz + 0.5f + 0.5f * copysignf(1.0f, r->dz)) / hm1 * 2 - 1) * mip_bound - z) * r->rdz;
  const float tt = *t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
  do {
    *t += clampf(*t * r->dt_gamma, r->dt_min, r->dt_max);
  } while (*t < tt);
  return 0;
}

/* reference :138-307.  counter[0] points, counter[1] rays are ADDED to (the
 * atomicAdd returns the old value as this ray's base).  Rays in index order. */
void orc_march_rays_train(const float* rays_o, const float* rays_d,
                          const float* grid, float mean_density, float bound,
                          float dt_gamma, uint32_t N, uint32_t C, uint32_t H,
                          uint32_t M, const float* nears, const float* fars,
                          float* xyzs, float* dirs, float* deltas,
                          int32_t* rays, int32_t* counter, uint32_t perturb) {
  for (uint32_t n = 0; n < N; ++n) {
    orc_ray r;
    ray_init(&r, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, grid,
             mean_density, bound, dt_gamma, C, H);
    const float far = fars[n];
    float t0 = nears[n];
    if (perturb) {
      orc_pcg32 rng;
      pcg_seed(&rng, (uint64_t)n, 1u);
      t0 += ORC_MIN_STEPSIZE * pcg_float(&rng);
    }
    float t = t0, x, y, z;
    uint32_t num_steps = 0;
    while (t < far && num_steps < ORC_MAX_STEPS) {
      if (probe(&r, &t, &x, &y, &z)) {
        ++num_steps;
        t += clampf(t * dt_gamma, r.dt_min, r.dt_max);
      }
    }
    const uint32_t point_index = (uint32_t)counter[0];
    const uint32_t ray_index = (uint32_t)counter[1];
    counter[0] += (int32_t)num_steps;
    counter[1] += 1;
    rays[ray_index * 3] = (int32_t)n;
    rays[ray_index * 3 + 1] = (int32_t)point_index;
    rays[ray_index * 3 + 2] = (int32_t)num_steps;
    if (num_steps == 0) continue;
    if (point_index + num_steps >= M) continue;

    float* px = xyzs + 3 * (size_t)point_index;
    float* pd = dirs + 3 * (size_t)point_index;
    float* pl = deltas + 2 * (size_t)point_index;
    t = t0;
    float last_t = t;
    uint32_t step = 0;
    while (t < far && step < num_steps) {
      if (probe(&r, &t, &x, &y, &z)) {
        px[0] = x; px[1] = y; px[2] = z;
        pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
        const float dt = clampf(t * dt_gamma, r.dt_min, r.dt_max);
        t += dt;
        pl[0] = dt;
        pl[1] = t - last_t;
        last_t = t;
        px += 3; pd += 3; pl += 2;
        ++step;
      }
    }
  }
}

/* reference :318-394, with n_sem extra channels composited like rgb (the
 * commented-out semantics variant, raymarching.py:249-309).  local_sem and
 * semantics may be NULL when n_sem == 0. */
void orc_composite_rays_train_fwd(const float* sigmas, const float* rgbs,
                                  const float* local_sem, const float* deltas,
                                  const int32_t* rays, uint32_t M, uint32_t N,
                                  uint32_t n_sem, float* weights_sum,
                                  float* depth, float* image,
                                  float* semantics) {
  for (uint32_t n = 0; n < N; ++n) {
    const uint32_t index = (uint32_t)rays[n * 3];
    const uint32_t offset = (uint32_t)rays[n * 3 + 1];
    const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
    float* sem = n_sem ? semantics + (size_t)index * n_sem : 0;
    for (uint32_t c = 0; c < n_sem; ++c) sem[c] = 0;
    weights_sum[index] = 0;
    depth[index] = 0;
    image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
    if (num_steps == 0 || offset + num_steps >= M) continue;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    for (uint32_t s = 0; s < num_steps; ++s) {
      const size_t m = (size_t)offset + s;
      const float alpha = 1.0f - expf(-sigmas[m] * deltas[2 * m]);
      const float w = alpha * T;
      r += w * rgbs[3 * m];
      g += w * rgbs[3 * m + 1];
      b += w * rgbs[3 * m + 2];
      for (uint32_t c = 0; c < n_sem; ++c) sem[c] += w * local_sem[m * n_sem + c];
      t += deltas[2 * m + 1];
      d += w * t;
      ws += w;
      T *= 1.0f - alpha;
    }
    weights_sum[index] = ws;
    depth[index] = d;
    image[index * 3] = r;
    image[index * 3 + 1] = g;
    image[index * 3 + 2] = b;
  }
}

/* reference :408-487.  The semantic channels are composited with DETACHED
 * weights, as on the live path (reference renderer_semantics.py:268-271): they
 * give grad_local_sem = grad_semantics * w and add nothing to grad_sigmas.
 * grad_depth is ignored (reference raymarching.py:209). */
void orc_composite_rays_train_bwd(const float* grad_ws, const float* grad_image,
                                  const float* grad_sem, const float* sigmas,
                                  const float* rgbs, const float* deltas,
                                  const int32_t* rays, const float* weights_sum,
                                  const float* image, uint32_t M, uint32_t N,
                                  uint32_t n_sem, float* grad_sigmas,
                                  float* grad_rgbs, float* grad_local_sem) {
  for (uint32_t n = 0; n < N; ++n) {
    const uint32_t index = (uint32_t)rays[n * 3];
    const uint32_t offset = (uint32_t)rays[n * 3 + 1];
    const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps >= M) continue;
    const float* gi = grad_image + (size_t)index * 3;
    const float gw = grad_ws[index];
    const float rf = image[index * 3], gf = image[index * 3 + 1],
                bf = image[index * 3 + 2], wf = weights_sum[index];
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
    for (uint32_t s = 0; s < num_steps; ++s) {
      const size_t m = (size_t)offset + s;
      const float alpha = 1.0f - expf(-sigmas[m] * deltas[2 * m]);
      const float w = alpha * T;
      r += w * rgbs[3 * m];
      g += w * rgbs[3 * m + 1];
      b += w * rgbs[3 * m + 2];
      ws += w;
      T *= 1.0f - alpha;
      grad_rgbs[3 * m] = gi[0] * w;
      grad_rgbs[3 * m + 1] = gi[1] * w;
      grad_rgbs[3 * m + 2] = gi[2] * w;
      for (uint32_t c = 0; c < n_sem; ++c)
        grad_local_sem[m * n_sem + c] = grad_sem[(size_t)index * n_sem + c] * w;
      grad_sigmas[m] = deltas[2 * m] * (gi[0] * (T * rgbs[3 * m] - (rf - r)) +
                                        gi[1] * (T * rgbs[3 * m + 1] - (gf - g)) +
                                        gi[2] * (T * rgbs[3 * m + 2] - (bf - b)) +
                                        gw * (T - (wf - ws)));
    }
  }
}

/* reference :528-634.  Output rows are left untouched past the last step
 * (the wrapper zero-fills, raymarching.py:422-426). */
void orc_march_rays(uint32_t n_alive, uint32_t n_step,
                    const int32_t* rays_alive, const float* rays_t,
                    const float* rays_o, const float* rays_d, float bound,
                    float dt_gamma, uint32_t C, uint32_t H, const float* grid,
                    float mean_density, const float* nears, const float* fars,
                    float* xyzs, float* dirs, float* deltas,
                    uint32_t perturb) {
  for (uint32_t n = 0; n < n_alive; ++n) {
    const int32_t index = rays_alive[n];
    orc_ray r;
    ray_init(&r, rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, grid,
             mean_density, bound, dt_gamma, C, H);
    const float far = fars[index];
    float t = rays_t[n];
    if (perturb) {
      orc_pcg32 rng;
      pcg_seed(&rng, (uint64_t)n, (uint64_t)perturb);
      t += ORC_MIN_STEPSIZE * pcg_float(&rng);
    }
    float* px = xyzs + 3 * (size_t)n * n_step;
    float* pd = dirs + 3 * (size_t)n * n_step;
    float* pl = deltas + 2 * (size_t)n * n_step;
    float last_t = t, x, y, z;
    uint32_t step = 0;
    while (t < far && step < n_step) {
      if (probe(&r, &t, &x, &y, &z)) {
        px[0] = x; px[1] = y; px[2] = z;
        pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
        const float dt = clampf(t * dt_gamma, r.dt_min, r.dt_max);
        t += dt;
        pl[0] = dt;
        pl[1] = t - last_t;
        last_t = t;
        px += 3; pd += 3; pl += 2;
        ++step;
      }
    }
  }
}

/* reference :647-729 (+ n_sem channels, the commented-out variant :741-825 /
 * raymarching.py:507-558).  In-place on weights_sum/depth/image/semantics and
 * rays_t. */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step,
                        const int32_t* rays_alive, float* rays_t,
                        const float* sigmas, const float* rgbs,
                        const float* local_sem, const float* deltas,
                        uint32_t n_sem, float* weights_sum, float* depth,
                        float* image, float* semantics) {
  for (uint32_t n = 0; n < n_alive; ++n) {
    const int32_t index = rays_alive[n];
    float t = rays_t[n];
    float ws = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1],
          b = image[index * 3 + 2];
    float* sem = n_sem ? semantics + (size_t)index * n_sem : 0;
    uint32_t step = 0;
    while (step < n_step) {
      const size_t m = (size_t)n * n_step + step;
      if (deltas[2 * m] == 0) break;
      const float alpha = 1.0f - expf(-sigmas[m] * deltas[2 * m]);
      const float T = 1 - ws;
      const float w = alpha * T;
      ws += w;
      t += deltas[2 * m + 1];
      d += w * t;
      r += w * rgbs[3 * m];
      g += w * rgbs[3 * m + 1];
      b += w * rgbs[3 * m + 2];
      for (uint32_t c = 0; c < n_sem; ++c) sem[c] += w * local_sem[m * n_sem + c];
      if ((double)T < 1e-4) break; /* the reference compares in double */
      ++step;
    }
    rays_t[n] = step < n_step ? -1.0f : t;
    weights_sum[index] = ws;
    depth[index] = d;
    image[index * 3] = r;
    image[index * 3 + 1] = g;
    image[index * 3 + 2] = b;
  }
}

/* reference :838-855; slots handed out in index order. */
void orc_compact_rays(uint32_t n_alive, int32_t* rays_alive,
                      const int32_t* rays_alive_old, float* rays_t,
                      const float* rays_t_old, int32_t* alive_counter) {
  for (uint32_t n = 0; n < n_alive; ++n) {
    if (rays_t_old[n] >= 0) {
      const int32_t k = alive_counter[0]++;
      rays_alive[k] = rays_alive_old[n];
      rays_t[k] = rays_t_old[n];
    }
  }
}
