// Fused BatchNorm2d (+ residual add) (+ ReLU), forward and backward, on
// channels-last activations: the memory-bound passes between the convolutions
// of DeepLabV3 (reference nr4seg/network/deeplabv3.py:6-19
// wraps torchvision's ResNet bottlenecks: conv -> BatchNorm2d -> ReLU, and
// conv -> BatchNorm2d -> (+ identity) -> ReLU; training mode during the joint
// step, joint_train_lightning_net.py:381,456-461).
//
// An NHWC activation is a row-major matrix X[M = N*H*W][C]; BatchNorm2d works
// per column.  PyTorch runs it as MIOpen's batch norm (two passes over X) plus
// separate elementwise kernels for the add and the ReLU and, in the backward,
// for their gradients: 20 / 32 bytes per element forward (without / with the
// residual) and 32 / 44 backward in fp32.  Here:
//
//   forward   k_bn_stats      X            -> per-column partial sums (read 4 B)
//             k_bn_finalize   partials     -> mean, invstd, running stats (one
//                                             workgroup per 64 channels)
//             k_bn_apply      X, (R)       -> Y = relu(a X + b (+ R))  (8 / 12 B)
//   backward  k_bn_bwd_stats  dY, Y, X     -> partial sum(g), sum(g xhat)
//             k_bn_bwd_final  partials     -> dgamma, dbeta
//             k_bn_bwd_apply  dY, Y, X     -> dX (, dR = g)
//   with g = dY * (Y > 0) when the ReLU is fused (Y is the layer's own output,
//   which autograd keeps alive for the next convolution anyway).
//
// Every kernel is a streaming pass: 16-byte accesses per lane (four fp32 or
// eight bf16 channels), rows of a workgroup contiguous in memory, HBM-bound.
// Sums: per-thread fp32 partials over at most a few hundred rows, an LDS tree
// per workgroup, and the per-workgroup partials combined in double by the
// finalize kernel; the variance is E[(x - s)^2] - (E[x - s])^2 around the
// shift s = running_mean (the best cheap guess of the mean: no cancellation
// once the statistics have settled).
#include <hip/hip_bf16.h>

#include "ucsa_common.h"

#define BN_THREADS 256

typedef float bn_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t bn_u32x2 __attribute__((ext_vector_type(2)));

// four consecutive channels of one row, widened to fp32
template <typename T>
struct Quad;
template <>
struct Quad<float> {
  static __device__ __forceinline__ bn_f32x4 load(const float* p) {
    return *reinterpret_cast<const bn_f32x4*>(p);
  }
  static __device__ __forceinline__ void store(float* p, bn_f32x4 v) {
    *reinterpret_cast<bn_f32x4*>(p) = v;
  }
};
template <>
struct Quad<__hip_bfloat16> {
  static __device__ __forceinline__ bn_f32x4 load(const __hip_bfloat16* p) {
    const bn_u32x2 u = *reinterpret_cast<const bn_u32x2*>(p);
    return bn_f32x4{__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xFFFF0000u),
                    __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xFFFF0000u)};
  }
  static __device__ __forceinline__ void store(__hip_bfloat16* p, bn_f32x4 v) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    bn_u32x2 u;
    u[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v[0], v[1]}, bf2));
    u[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v[2], v[3]}, bf2));
    *reinterpret_cast<bn_u32x2*>(p) = u;
  }
};

// Launch geometry shared by the two reduction kernels: a workgroup covers TQ
// channel quads (TQ = min(C/4, 16), a power of two) and 256/TQ rows per
// iteration of a contiguous range of rows.
struct BnGeom {
  uint32_t M, C;
  uint32_t tq;            // channel quads per workgroup
  uint32_t rows_per_it;   // 256 / tq
  uint32_t rows_per_wg;   // rows of one workgroup (multiple of rows_per_it)
  uint32_t n_row_wg;      // grid.x
  uint32_t n_col_wg;      // grid.y = (C/4) / tq
};

static BnGeom bn_geom(uint32_t M, uint32_t C) {
  BnGeom g;
  g.M = M;
  g.C = C;
  const uint32_t cq = C / 4;
  // 16 channel quads (64 channels: 256 contiguous bytes of a row in fp32) per
  // workgroup: wide layers get many column groups, and the workgroup has 16
  // row lanes for the final reduction of a column group (at most 512
  // partials, done by its last workgroup with four loads in flight per thread)
  uint32_t tq = 1;
  while (tq < 16 && tq * 2 <= cq && cq % (tq * 2) == 0) tq *= 2;
  g.tq = tq;
  g.rows_per_it = BN_THREADS / tq;
  g.n_col_wg = cq / tq;
  // ~2048 workgroups in all (8 per CU: the passes are latency-bound streams,
  // they need many loads in flight), at most 128 per column group (the finalize
  // launch sums them with 16 row lanes), at least 4 iterations each
  uint32_t want = 2048 / g.n_col_wg;
  if (want < 1) want = 1;
  if (want > 128) want = 128;
  uint32_t rows = (M + want - 1) / want;
  const uint32_t min_rows = 4 * g.rows_per_it;
  if (rows < min_rows) rows = min_rows;
  rows = (rows + g.rows_per_it - 1) / g.rows_per_it * g.rows_per_it;
  g.rows_per_wg = rows;
  g.n_row_wg = (M + rows - 1) / rows;
  return g;
}

extern "C" uint64_t ucsa_bn_workspace_bytes(uint32_t M, uint32_t C) {
  if (M == 0 || C == 0 || C % 4) return 0;
  const BnGeom g = bn_geom(M, C);
  // partials [n_row_wg][2][C] + per-channel coefficients [4][C]
  return ((uint64_t)g.n_row_wg * 2 * C + 4ull * C) * sizeof(float);
}

// sum over the workgroup's threads that share a channel quad (LDS tree over
// the row index), result valid in the threads of row 0
__device__ __forceinline__ void wg_reduce2(bn_f32x4& a, bn_f32x4& b, uint32_t tq,
                                           uint32_t rows_per_it) {
  __shared__ bn_f32x4 red[2][BN_THREADS];
  const uint32_t t = threadIdx.x;
  red[0][t] = a;
  red[1][t] = b;
  __syncthreads();
  for (uint32_t s = rows_per_it / 2; s >= 1; s >>= 1) {
    if (t < s * tq) {
      red[0][t] += red[0][t + s * tq];
      red[1][t] += red[1][t + s * tq];
    }
    __syncthreads();
  }
  a = red[0][t % tq];
  b = red[1][t % tq];
}


// Sum of a column group's per-workgroup partials, by ONE workgroup of the
// finalize launch (blockIdx.x = column group): thread (q, rr) adds partial rows
// rr, rr + rows_per_it, ... of its channel quad in double with eight loads in
// flight, an LDS tree over rr finishes.  Returns true in the threads of row 0,
// with the two sums of their quad in s / ss.
// (Tried first: no second launch, the LAST stats workgroup to arrive -- agent-
// scope ticket -- reduces.  Every workgroup then needs a release fence, and
// __threadfence() costs ~3.5 us per workgroup on this chip, x4 at four
// workgroups per CU (MI355X_MICROARCH.md, inter-workgroup visibility): the
// stats kernels took 45 - 130 us instead of ~8.  A dependent launch boundary
// is 1.5 - 1.9 us.)
struct D4 {
  double v[4];
};
__device__ __forceinline__ bool bn_group_reduce(const BnGeom& g, const float* partial,
                                                D4& s, D4& ss) {
  __shared__ double red[2][BN_THREADS][4];
  const uint32_t t = threadIdx.x;
  const uint32_t q = t % g.tq, rr = t / g.tq;
  const uint32_t c = 4 * (blockIdx.x * g.tq + q);
#pragma unroll
  for (int k = 0; k < 4; ++k) s.v[k] = ss.v[k] = 0.0;
  uint32_t k = rr;
  const uint32_t kstep = g.rows_per_it;
  for (; k + 3 * kstep < g.n_row_wg; k += 4 * kstep) {  // eight loads in flight
    bn_f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* p = partial + (size_t)(k + u * kstep) * 2 * g.C;
      a[u] = *reinterpret_cast<const bn_f32x4*>(p + c);
      b[u] = *reinterpret_cast<const bn_f32x4*>(p + g.C + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s.v[e] += (double)a[u][e];
        ss.v[e] += (double)b[u][e];
      }
  }
  for (; k < g.n_row_wg; k += kstep) {
    const float* p = partial + (size_t)k * 2 * g.C;
    const bn_f32x4 a = *reinterpret_cast<const bn_f32x4*>(p + c);
    const bn_f32x4 b = *reinterpret_cast<const bn_f32x4*>(p + g.C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s.v[e] += (double)a[e];
      ss.v[e] += (double)b[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][t][e] = s.v[e];
    red[1][t][e] = ss.v[e];
  }
  __syncthreads();
  for (uint32_t st = g.rows_per_it / 2; st >= 1; st >>= 1) {
    if (t < st * g.tq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[0][t][e] += red[0][t + st * g.tq][e];
        red[1][t][e] += red[1][t + st * g.tq][e];
      }
    }
    __syncthreads();
  }
  if (t >= g.tq) return false;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s.v[e] = red[0][t][e];
    ss.v[e] = red[1][t][e];
  }
  return true;
}

// ---- forward ---------------------------------------------------------------
struct BnFwdFin {   // what the finalizing workgroup needs
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float momentum, eps;
  float* save_mean;
  float* save_invstd;
  float* coef;
};

template <typename T>
__global__ void __launch_bounds__(BN_THREADS)
k_bn_stats(BnGeom g, const T* __restrict__ x, const float* __restrict__ shift,
           float* __restrict__ partial) {
  const uint32_t t = threadIdx.x;
  const uint32_t q = blockIdx.y * g.tq + t % g.tq;  // channel quad
  const uint32_t c = 4 * q;
  const uint32_t r0 = blockIdx.x * g.rows_per_wg + t / g.tq;
  uint32_t r1 = (blockIdx.x + 1) * g.rows_per_wg;
  if (r1 > g.M) r1 = g.M;
  const bn_f32x4 sh = shift ? *reinterpret_cast<const bn_f32x4*>(shift + c)
                            : bn_f32x4{0.f, 0.f, 0.f, 0.f};
  bn_f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
  uint32_t r = r0;
  const uint32_t step = g.rows_per_it;
  for (; r + 3 * step < r1; r += 4 * step) {  // four row loads in flight
    bn_f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = Quad<T>::load(x + (size_t)(r + k * step) * g.C + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bn_f32x4 d = v[k] - sh;
      s += d;
      ss += d * d;
    }
  }
  for (; r < r1; r += step) {
    const bn_f32x4 v = Quad<T>::load(x + (size_t)r * g.C + c) - sh;
    s += v;
    ss += v * v;
  }
  wg_reduce2(s, ss, g.tq, g.rows_per_it);
  if (t < g.tq) {
    float* p = partial + (size_t)blockIdx.x * 2 * g.C;
    *reinterpret_cast<bn_f32x4*>(p + c) = s;
    *reinterpret_cast<bn_f32x4*>(p + g.C + c) = ss;
  }
}

// mean / invstd, running statistics as torch.nn.BatchNorm2d updates them (the
// biased variance normalises, the unbiased one goes into running_var),
// coefficients a = gamma * invstd, b = beta - mean * a.  One workgroup per
// column group.
__global__ void __launch_bounds__(BN_THREADS)
k_bn_finalize(BnGeom g, const float* __restrict__ partial,
              const float* __restrict__ shift, BnFwdFin f) {
  D4 ds, dss;
  if (!bn_group_reduce(g, partial, ds, dss)) return;
  const uint32_t c = 4 * (blockIdx.x * g.tq + threadIdx.x);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t ch = c + e;
    const double shf = shift ? (double)shift[ch] : 0.0;
    const double m1 = ds.v[e] / g.M;                 // E[x - shift]
    double var = dss.v[e] / g.M - m1 * m1;
    if (var < 0.0) var = 0.0;
    const double mean = m1 + shf;
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float mf = (float)mean;
    f.save_mean[ch] = mf;
    f.save_invstd[ch] = invstd;
    if (f.running_mean) {
      const double unb = g.M > 1 ? var * ((double)g.M / (double)(g.M - 1)) : var;
      f.running_mean[ch] = (float)((1.0 - f.momentum) * shf + f.momentum * mean);
      f.running_var[ch] =
          (float)((1.0 - f.momentum) * (double)f.running_var[ch] + f.momentum * unb);
    }
    const float a = (f.gamma ? f.gamma[ch] : 1.0f) * invstd;
    f.coef[ch] = a;
    f.coef[g.C + ch] = (f.beta ? f.beta[ch] : 0.0f) - mf * a;
  }
}

// eval mode: coefficients from the running statistics
__global__ void k_bn_coef_eval(uint32_t C, const float* __restrict__ gamma,
                               const float* __restrict__ beta,
                               const float* __restrict__ running_mean,
                               const float* __restrict__ running_var, float eps,
                               float* __restrict__ coef) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float a = (gamma ? gamma[c] : 1.0f) * invstd;
  coef[c] = a;
  coef[C + c] = (beta ? beta[c] : 0.0f) - running_mean[c] * a;
}

template <typename T, bool RES, bool RELU>
__global__ void __launch_bounds__(BN_THREADS)
k_bn_apply(uint64_t n_quads, uint32_t cq, const T* __restrict__ x,
           const T* __restrict__ res, const float* __restrict__ coef,
           T* __restrict__ y) {
  const uint32_t C = cq * 4;
  for (uint64_t i = (uint64_t)blockIdx.x * BN_THREADS + threadIdx.x; i < n_quads;
       i += (uint64_t)gridDim.x * BN_THREADS) {
    const uint32_t c = (uint32_t)(i % cq) * 4;
    const bn_f32x4 a = *reinterpret_cast<const bn_f32x4*>(coef + c);
    const bn_f32x4 b = *reinterpret_cast<const bn_f32x4*>(coef + C + c);
    bn_f32x4 v = Quad<T>::load(x + i * 4) * a + b;
    if (RES) v += Quad<T>::load(res + i * 4);
    if (RELU) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
    }
    Quad<T>::store(y + i * 4, v);
  }
}

// ---- backward --------------------------------------------------------------
struct BnBwdFin {
  const float* gamma;
  float* dgamma;
  float* dbeta;
  float* coef;
};

template <typename T, bool RELU>
__global__ void __launch_bounds__(BN_THREADS)
k_bn_bwd_stats(BnGeom g, const T* __restrict__ dy, const T* __restrict__ y,
               const T* __restrict__ x, const float* __restrict__ mean,
               const float* __restrict__ invstd, float* __restrict__ partial) {
  const uint32_t t = threadIdx.x;
  const uint32_t q = blockIdx.y * g.tq + t % g.tq;
  const uint32_t c = 4 * q;
  const uint32_t r0 = blockIdx.x * g.rows_per_wg + t / g.tq;
  uint32_t r1 = (blockIdx.x + 1) * g.rows_per_wg;
  if (r1 > g.M) r1 = g.M;
  const bn_f32x4 mu = *reinterpret_cast<const bn_f32x4*>(mean + c);
  const bn_f32x4 is = *reinterpret_cast<const bn_f32x4*>(invstd + c);
  bn_f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
  auto add_row = [&](bn_f32x4 gv, bn_f32x4 yv, bn_f32x4 xv) {
    if (RELU) {
#pragma unroll
      for (int k = 0; k < 4; ++k) gv[k] = yv[k] > 0.0f ? gv[k] : 0.0f;
    }
    const bn_f32x4 xh = (xv - mu) * is;
    sg += gv;
    sgx += gv * xh;
  };
  uint32_t r = r0;
  const uint32_t step = g.rows_per_it;
  for (; r + step < r1; r += 2 * step) {  // two rows = up to six loads in flight
    const size_t o0 = (size_t)r * g.C + c, o1 = (size_t)(r + step) * g.C + c;
    const bn_f32x4 g0 = Quad<T>::load(dy + o0), g1 = Quad<T>::load(dy + o1);
    bn_f32x4 y0 = g0, y1 = g1;
    if (RELU) {
      y0 = Quad<T>::load(y + o0);
      y1 = Quad<T>::load(y + o1);
    }
    const bn_f32x4 x0 = Quad<T>::load(x + o0), x1 = Quad<T>::load(x + o1);
    add_row(g0, y0, x0);
    add_row(g1, y1, x1);
  }
  for (; r < r1; r += step) {
    const size_t o = (size_t)r * g.C + c;
    const bn_f32x4 gv = Quad<T>::load(dy + o);
    add_row(gv, RELU ? Quad<T>::load(y + o) : gv, Quad<T>::load(x + o));
  }
  wg_reduce2(sg, sgx, g.tq, g.rows_per_it);
  if (t < g.tq) {
    float* p = partial + (size_t)blockIdx.x * 2 * g.C;
    *reinterpret_cast<bn_f32x4*>(p + c) = sg;
    *reinterpret_cast<bn_f32x4*>(p + g.C + c) = sgx;
  }
}

// dbeta = sum g, dgamma = sum g xhat; coefficients of
// dx = c1 (g - c2 - xhat c3) = c1 g - (c1 c3 invstd) x + (c1 c3 invstd mean - c1 c2)
__global__ void __launch_bounds__(BN_THREADS)
k_bn_bwd_final(BnGeom g, const float* __restrict__ partial,
               const float* __restrict__ mean, const float* __restrict__ invstd,
               BnBwdFin f) {
  D4 dsg, dsgx;
  if (!bn_group_reduce(g, partial, dsg, dsgx)) return;
  const uint32_t c = 4 * (blockIdx.x * g.tq + threadIdx.x);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t ch = c + e;
    if (f.dbeta) f.dbeta[ch] = (float)dsg.v[e];
    if (f.dgamma) f.dgamma[ch] = (float)dsgx.v[e];
    const double c1 = (double)(f.gamma ? f.gamma[ch] : 1.0f) * (double)invstd[ch];
    const double kx = c1 * (dsgx.v[e] / g.M) * (double)invstd[ch];
    f.coef[ch] = (float)c1;                                            // * g
    f.coef[g.C + ch] = (float)(-kx);                                   // * x
    f.coef[2 * g.C + ch] = (float)(kx * (double)mean[ch] - c1 * (dsg.v[e] / g.M));
  }
}

template <typename T, bool RELU, bool DRES>
__global__ void __launch_bounds__(BN_THREADS)
k_bn_bwd_apply(uint64_t n_quads, uint32_t cq, const T* __restrict__ dy,
               const T* __restrict__ y, const T* __restrict__ x,
               const float* __restrict__ coef, T* __restrict__ dx,
               T* __restrict__ dres) {
  const uint32_t C = cq * 4;
  for (uint64_t i = (uint64_t)blockIdx.x * BN_THREADS + threadIdx.x; i < n_quads;
       i += (uint64_t)gridDim.x * BN_THREADS) {
    const uint32_t c = (uint32_t)(i % cq) * 4;
    const bn_f32x4 kg = *reinterpret_cast<const bn_f32x4*>(coef + c);
    const bn_f32x4 kx = *reinterpret_cast<const bn_f32x4*>(coef + C + c);
    const bn_f32x4 k0 = *reinterpret_cast<const bn_f32x4*>(coef + 2 * C + c);
    bn_f32x4 gv = Quad<T>::load(dy + i * 4);
    if (RELU) {
      const bn_f32x4 yv = Quad<T>::load(y + i * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) gv[k] = yv[k] > 0.0f ? gv[k] : 0.0f;
    }
    if (DRES) Quad<T>::store(dres + i * 4, gv);
    Quad<T>::store(dx + i * 4, kg * gv + kx * Quad<T>::load(x + i * 4) + k0);
  }
}

// ---- host ------------------------------------------------------------------
static uint32_t bn_apply_blocks(uint64_t n_quads) {
  const uint64_t b = (n_quads + BN_THREADS - 1) / BN_THREADS;
  return (uint32_t)(b < 8192 ? (b ? b : 1) : 8192);
}

template <typename T>
static int32_t bn_fwd(const void* x, const void* res, const float* gamma,
                      const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, uint32_t M, uint32_t C, int relu,
                      int training, void* y, float* save_mean, float* save_invstd,
                      void* workspace, hipStream_t s) {
  const BnGeom g = bn_geom(M, C);
  float* partial = (float*)workspace;
  float* coef = partial + (size_t)g.n_row_wg * 2 * C;
  UCSA_CLEAR_ERR();
  if (training) {
    const BnFwdFin f{gamma, beta, running_mean, running_var, momentum, eps,
                     save_mean, save_invstd, coef};
    // shift = running_mean (the best cheap guess of the mean) when there is one
    hipLaunchKernelGGL((k_bn_stats<T>), dim3(g.n_row_wg, g.n_col_wg),
                       dim3(BN_THREADS), 0, s, g, (const T*)x,
                       (const float*)running_mean, partial);
    hipLaunchKernelGGL(k_bn_finalize, dim3(g.n_col_wg), dim3(BN_THREADS), 0, s, g,
                       (const float*)partial, (const float*)running_mean, f);
  } else {
    hipLaunchKernelGGL(k_bn_coef_eval, dim3((C + 255) / 256), dim3(256), 0, s, C,
                       gamma, beta, running_mean, running_var, eps, coef);
  }
  const uint64_t nq = (uint64_t)M * (C / 4);
  const dim3 grid(bn_apply_blocks(nq));
#define BN_APPLY(RES, RELU)                                                     \
  hipLaunchKernelGGL((k_bn_apply<T, RES, RELU>), grid, dim3(BN_THREADS), 0, s, \
                     nq, C / 4, (const T*)x, (const T*)res, coef, (T*)y)
  if (res && relu) BN_APPLY(true, true);
  else if (res) BN_APPLY(true, false);
  else if (relu) BN_APPLY(false, true);
  else BN_APPLY(false, false);
#undef BN_APPLY
  return ucsa_launch_status();
}

template <typename T>
static int32_t bn_bwd(const void* dy, const void* x, const void* y,
                      const float* gamma, const float* save_mean,
                      const float* save_invstd, uint32_t M, uint32_t C, int relu,
                      void* dx, void* dres, float* dgamma, float* dbeta,
                      void* workspace, hipStream_t s) {
  const BnGeom g = bn_geom(M, C);
  float* partial = (float*)workspace;
  float* coef = partial + (size_t)g.n_row_wg * 2 * C;
  UCSA_CLEAR_ERR();
  const BnBwdFin f{gamma, dgamma, dbeta, coef};
  if (relu)
    hipLaunchKernelGGL((k_bn_bwd_stats<T, true>), dim3(g.n_row_wg, g.n_col_wg),
                       dim3(BN_THREADS), 0, s, g, (const T*)dy, (const T*)y,
                       (const T*)x, save_mean, save_invstd, partial);
  else
    hipLaunchKernelGGL((k_bn_bwd_stats<T, false>), dim3(g.n_row_wg, g.n_col_wg),
                       dim3(BN_THREADS), 0, s, g, (const T*)dy, (const T*)y,
                       (const T*)x, save_mean, save_invstd, partial);
  hipLaunchKernelGGL(k_bn_bwd_final, dim3(g.n_col_wg), dim3(BN_THREADS), 0, s, g,
                     (const float*)partial, save_mean, save_invstd, f);
  const uint64_t nq = (uint64_t)M * (C / 4);
  const dim3 grid(bn_apply_blocks(nq));
#define BN_BAPPLY(RELU, DRES)                                                      \
  hipLaunchKernelGGL((k_bn_bwd_apply<T, RELU, DRES>), grid, dim3(BN_THREADS), 0, s, \
                     nq, C / 4, (const T*)dy, (const T*)y, (const T*)x, coef,      \
                     (T*)dx, (T*)dres)
  if (relu && dres) BN_BAPPLY(true, true);
  else if (relu) BN_BAPPLY(true, false);
  else if (dres) BN_BAPPLY(false, true);
  else BN_BAPPLY(false, false);
#undef BN_BAPPLY
  return ucsa_launch_status();
}

extern "C" int32_t ucsa_bn_act_fwd(const void* x, const void* residual,
                                   const float* gamma, const float* beta,
                                   float* running_mean, float* running_var,
                                   float momentum, float eps, uint32_t M,
                                   uint32_t C, int32_t relu, int32_t training,
                                   int32_t dtype, void* y, float* save_mean,
                                   float* save_invstd, void* workspace,
                                   void* stream) {
  UCSA_CHECK_ARG(x && ((uintptr_t)x & 15u) == 0, 0);
  UCSA_CHECK_ARG(!residual || ((uintptr_t)residual & 15u) == 0, 1);
  UCSA_CHECK_ARG(training || (running_mean && running_var), 4);
  UCSA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), 5);
  UCSA_CHECK_ARG(C >= 4 && C % 4 == 0 && C <= 4096, 9);
  UCSA_CHECK_ARG(dtype == 0 || dtype == 1, 12);
  UCSA_CHECK_ARG(y && ((uintptr_t)y & 15u) == 0, 13);
  UCSA_CHECK_ARG(!training || (save_mean && save_invstd), 14);
  UCSA_CHECK_ARG(workspace, 16);
  if (M == 0) return 0;
  UCSA_CHECK_ARG(!training || M > 0, 8);
  if (dtype == 0)
    return bn_fwd<float>(x, residual, gamma, beta, running_mean, running_var,
                         momentum, eps, M, C, relu, training, y, save_mean,
                         save_invstd, workspace, (hipStream_t)stream);
  return bn_fwd<__hip_bfloat16>(x, residual, gamma, beta, running_mean,
                                running_var, momentum, eps, M, C, relu, training,
                                y, save_mean, save_invstd, workspace,
                                (hipStream_t)stream);
}

extern "C" int32_t ucsa_bn_act_bwd(const void* dy, const void* x, const void* y,
                                   const float* gamma, const float* save_mean,
                                   const float* save_invstd, uint32_t M,
                                   uint32_t C, int32_t relu, int32_t dtype,
                                   void* dx, void* dresidual, float* dgamma,
                                   float* dbeta, void* workspace, void* stream) {
  UCSA_CHECK_ARG(dy && ((uintptr_t)dy & 15u) == 0, 0);
  UCSA_CHECK_ARG(x && ((uintptr_t)x & 15u) == 0, 1);
  UCSA_CHECK_ARG(!relu || (y && ((uintptr_t)y & 15u) == 0), 2);
  UCSA_CHECK_ARG(save_mean && save_invstd, 4);
  UCSA_CHECK_ARG(C >= 4 && C % 4 == 0 && C <= 4096, 7);
  UCSA_CHECK_ARG(dtype == 0 || dtype == 1, 9);
  UCSA_CHECK_ARG(dx && ((uintptr_t)dx & 15u) == 0, 10);
  UCSA_CHECK_ARG(workspace, 14);
  if (M == 0) return 0;
  if (dtype == 0)
    return bn_bwd<float>(dy, x, y, gamma, save_mean, save_invstd, M, C, relu, dx,
                         dresidual, dgamma, dbeta, workspace, (hipStream_t)stream);
  return bn_bwd<__hip_bfloat16>(dy, x, y, gamma, save_mean, save_invstd, M, C,
                                relu, dx, dresidual, dgamma, dbeta, workspace,
                                (hipStream_t)stream);
}
