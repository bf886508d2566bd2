// Micro-benchmark: random fp32 atomic scatter-add into a slab, as in the
// hash-grid backward.  Variants: agent scope, workgroup scope, per-XCD private
// slab (+workgroup scope), packed 2 x f32 via one 64-bit CAS-free trick (none).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ void k(float* slab, uint32_t entries_mask, uint64_t n, uint32_t slab_stride) {
  uint32_t xcc = 0;
  if (MODE == 2) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
  }
  float* base = slab + (size_t)xcc * slab_stride;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t idx = hash32((uint32_t)i) & entries_mask;
    float* p = base + (size_t)idx * 2;
    if (MODE == 0) {
      atomicAdd(p, 1.0f); atomicAdd(p + 1, 2.0f);
    } else if (MODE == 1 || MODE == 2) {
      __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(p + 1, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 3) {  // non-atomic RMW (wrong, upper bound)
      p[0] += 1.0f; p[1] += 2.0f;
    } else if (MODE == 4) {  // one 64-bit integer atomic (fixed-point pair)
      atomicAdd((unsigned long long*)p, 0x0000000200000001ull);
    }
  }
}

int main() {
  const uint32_t entries = 1u << 19;           // one hashed level
  const uint64_t n = 84ull << 20;               // ~88 M corner updates
  float* slab;
  hipMalloc(&slab, (size_t)entries * 2 * 4 * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"agent atomics", "workgroup-scope atomics", "per-XCD slab + wg scope",
                         "non-atomic RMW", "one u64 atomic per entry"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(slab, 0, (size_t)entries * 2 * 4 * 8);
      hipEventRecord(a);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, slab, entries - 1, n, entries * 2); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, slab, entries - 1, n, entries * 2); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(4096), dim3(256), 0, 0, slab, entries - 1, n, entries * 2); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(4096), dim3(256), 0, 0, slab, entries - 1, n, entries * 2); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(4096), dim3(256), 0, 0, slab, entries - 1, n, entries * 2); break;
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep == 1) {
        std::vector<float> h(entries * 2 * 8);
        hipMemcpy(h.data(), slab, h.size() * 4, hipMemcpyDeviceToHost);
        double sx = 0; for (size_t i = 0; i < h.size(); i += 2) sx += h[i];
        printf("%-28s %8.3f ms  %7.1f G entry-updates/s  sum_x=%.0f (want %llu)\n", names[mode], ms,
               n / ms * 1e-6, sx, (unsigned long long)n);
      }
    }
  }
  return 0;
}
