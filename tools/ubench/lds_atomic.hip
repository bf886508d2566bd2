// Micro-benchmark: LDS accumulate rate for the bin-accumulate pass of the
// hash-grid backward (2048-entry x 2-float slice per workgroup).
//   0: ds_add_f32 x2 at random entries      1: ds_add_u32 x2 (integer)
//   2: ds_add_f32 x2, conflict-free (lane-distinct consecutive entries)
//   3: one ds_add_u64 per entry (integer pair)
//   4: ds_add_f32, one float per record (half the ops)
//   5: ds_add_rtn_f32 x2 (returning)        6: non-atomic RMW x2 (wrong; bound)
//   7: ds_pk_add... not available for f32 -> packed 2xf32 via ds_add_f64? (skipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, uint32_t iters) {
  __shared__ float acc[4096];
  for (uint32_t e = threadIdx.x; e < 4096; e += 512) acc[e] = 0.f;
  __syncthreads();
  uint32_t* iacc = reinterpret_cast<uint32_t*>(acc);
  unsigned long long* lacc = reinterpret_cast<unsigned long long*>(acc);
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t h = hash32(it * 512u + threadIdx.x + blockIdx.x * 7919u);
    uint32_t il = h & 2047u;
    if (MODE == 2) il = (threadIdx.x + it * 64u) & 2047u;
    const float v = __uint_as_float(0x3f800000u | (h >> 9)) - 1.0f;
    if (MODE == 0) { atomicAdd(&acc[2 * il], v); atomicAdd(&acc[2 * il + 1], v); }
    else if (MODE == 1) { atomicAdd(&iacc[2 * il], h); atomicAdd(&iacc[2 * il + 1], h); }
    else if (MODE == 2) { atomicAdd(&acc[2 * il], v); atomicAdd(&acc[2 * il + 1], v); }
    else if (MODE == 3) { atomicAdd(&lacc[il], (unsigned long long)h); }
    else if (MODE == 4) { atomicAdd(&acc[2 * il], v); }
    else if (MODE == 5) { float a = atomicAdd(&acc[2 * il], v); float b = atomicAdd(&acc[2 * il + 1], v); if (a + b == 123.456f) out[0] = a; }
    else if (MODE == 6) { acc[2 * il] += v; acc[2 * il + 1] += v; }
    else if (MODE == 7) {  // 64-bit compare-and-swap loop on the (x, y) pair
      unsigned long long* p = &lacc[il];
      unsigned long long old = *(volatile unsigned long long*)p, assumed;
      do {
        assumed = old;
        float fx = __uint_as_float((uint32_t)assumed) + v;
        float fy = __uint_as_float((uint32_t)(assumed >> 32)) + v;
        old = atomicCAS(p, assumed, (unsigned long long)__float_as_uint(fx) | ((unsigned long long)__float_as_uint(fy) << 32));
      } while (old != assumed);
    }
    else if (MODE == 8) {  // two 32-bit CAS loops
      for (int c = 0; c < 2; ++c) {
        uint32_t* p = &iacc[2 * il + c];
        uint32_t old = *(volatile uint32_t*)p, assumed;
        do {
          assumed = old;
          old = atomicCAS(p, assumed, __float_as_uint(__uint_as_float(assumed) + v));
        } while (old != assumed);
      }
    }
  }
  __syncthreads();
  float s = 0.f;
  for (uint32_t e = threadIdx.x; e < 4096; e += 512) s += acc[e];
  if (s == 123.456f) out[blockIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const uint32_t blocks = 2560, iters = 64;  // 2560 x 512 x 64 = 84 M records
  const char* names[] = {"ds_add_f32 x2 random", "ds_add_u32 x2 random", "ds_add_f32 x2 conflict-free",
                         "ds_add_u64 x1 random", "ds_add_f32 x1 random", "ds_add_rtn_f32 x2 random",
                         "plain RMW x2 (wrong)", "64-bit CAS loop on the pair", "two 32-bit CAS loops"};
  for (int mode = 0; mode < 9; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    const double recs = (double)blocks * 512 * iters;
    printf("%-32s %8.3f ms  %7.1f G records/s\n", names[mode], best, recs / best / 1e6);
  }
  return 0;
}
