// Micro-benchmark: do VALU instructions hide under MFMAs on one SIMD?
// Per wave and iteration: 16 MFMAs (4 independent accumulators) and 16*R
// independent v_fma_f32, either as two blocks (MFMAs, then VALU) or
// interleaved in program order (M, R x V) x 16.  1 / 2 / 4 waves per SIMD.
//   TYPE 0: v_mfma_f32_16x16x4_f32 (32 cycles / SIMD)   R = 6
//   TYPE 1: v_mfma_f32_16x16x32_bf16 (16 cycles / SIMD)  R = 3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int TYPE>
__device__ __forceinline__ void mf(f32x4& acc, const f32x4& a, const f32x4& b) {
  if constexpr (TYPE == 0)
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a[0]), "v"(b[0]));
  else if constexpr (TYPE == 1)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  else if constexpr (TYPE == 3)  // K = 32, accumulator in AGPRs
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else if constexpr (TYPE == 4)  // K = 32, accumulator and A operand in AGPRs
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
  else {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 a2 = {a[0], a[1]}, b2 = {b[0], b[1]};
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a2), "v"(b2));
  }
}
__device__ __forceinline__ void va(float& x, float y, float z) {
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
}

// MODE 0: MFMA only, 1: VALU only, 2: blocked, 3: interleaved
template <int TYPE, int MODE, int R>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  f32x4 acc[4];
  f32x4 a, b;
  float v[8];
  for (int i = 0; i < 4; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    b[i] = 0.5f + i;
  }
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const float y = 0.999f, z = 1e-3f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int m = 0; m < 16; ++m) mf<TYPE>(acc[m & 3], a, b);
    }
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int q = 0; q < 16 * R; ++q) va(v[q & 7], y, z);
    }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        mf<TYPE>(acc[m & 3], a, b);
#pragma unroll
        for (int q = 0; q < R; ++q) va(v[(m * R + q) & 7], y, z);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// TYPE 5 (round 6): v_mfma_f32_32x32x16_bf16 -- 8 per iteration (the pipe time of 16
// 16x16x32), 2 independent 16-register accumulators, 2R VALU behind each in MODE 3.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int R>
__global__ void __launch_bounds__(1024) k32(float* out, int iters) {
  f32x16 acc[2];
  f32x4 a, b;
  float v[8];
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
  for (int i = 0; i < 4; ++i) {
    a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    b[i] = 0.5f + i;
  }
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const float y = 0.999f, z = 1e-3f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "v"(b));
    }
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int q = 0; q < 16 * R; ++q) va(v[q & 7], y, z);
    }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "v"(b));
#pragma unroll
        for (int q = 0; q < 2 * R; ++q) va(v[(m * 2 * R + q) & 7], y, z);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int R>
float run32(int waves_per_simd, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const dim3 grid(256), block(64 * 4 * waves_per_simd);
  hipLaunchKernelGGL((k32<MODE, R>), grid, block, 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k32<MODE, R>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f * 2.4e9f / iters;
}

template <int R>
void all32(float* out) {
  printf("bf16 32x32x16, per iteration and wave: 8 MFMA, %d v_fma_f32  (cycles at 2.4 GHz per iteration of ALL waves of a SIMD)\n", 16 * R);
  for (int w = 1; w <= 4; w *= 2) {
    const float m = run32<0, R>(w, out), v = run32<1, R>(w, out),
                bl = run32<2, R>(w, out), il = run32<3, R>(w, out);
    printf("  %d wave(s)/SIMD: MFMA only %7.1f | VALU only %7.1f | blocked %7.1f | interleaved %7.1f   (sum %7.1f, max %7.1f)\n",
           w, m, v, bl, il, m + v, m > v ? m : v);
  }
}

template <int TYPE, int MODE, int R>
float run(int waves_per_simd, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const dim3 grid(256), block(64 * 4 * waves_per_simd);
  hipLaunchKernelGGL((k<TYPE, MODE, R>), grid, block, 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<TYPE, MODE, R>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // cycles (at 2.4 GHz nominal) per iteration per SIMD
  return ms * 1e-3f * 2.4e9f / iters;
}

template <int TYPE, int R>
void all(const char* name, float* out) {
  printf("%s, per iteration and wave: 16 MFMA, %d v_fma_f32  (cycles at 2.4 GHz per iteration of ALL waves of a SIMD)\n", name, 16 * R);
  for (int w = 1; w <= 4; w *= 2) {
    const float m = run<TYPE, 0, R>(w, out), v = run<TYPE, 1, R>(w, out),
                bl = run<TYPE, 2, R>(w, out), il = run<TYPE, 3, R>(w, out);
    printf("  %d wave(s)/SIMD: MFMA only %7.1f | VALU only %7.1f | blocked %7.1f | interleaved %7.1f   (sum %7.1f, max %7.1f)\n",
           w, m, v, bl, il, m + v, m > v ? m : v);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  if (getenv("ONLY32") != nullptr) {   // round 6: 32x32x16 against 16x16x32 at equal work
    all<1, 3>("bf16 16x16x32", out);
    all32<3>(out);
    all<1, 6>("bf16 16x16x32", out);
    all32<6>(out);
    all<1, 8>("bf16 16x16x32", out);
    all32<8>(out);
    return 0;
  }
  all<0, 6>("f32 16x16x4", out);
  all<1, 3>("bf16 16x16x32", out);
  all<1, 6>("bf16 16x16x32", out);
  all<0, 3>("f32 16x16x4", out);
  all<2, 3>("bf16 16x16x16 (K = 16)", out);
  all<3, 3>("bf16 16x16x32, accumulators in AGPRs", out);
  all<4, 3>("bf16 16x16x32, accumulators and A in AGPRs", out);
  return 0;
}
