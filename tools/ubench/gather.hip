// Micro-benchmark: random 8-byte (float2) / 4-byte (half2) gathers from one
// hash-grid level slab, as in k_hashgrid_encode on the fine levels.
// Variants: plain load, nontemporal, buffer loads with sc0 / sc1 / nt bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE, typename T>
__global__ void k(const T* __restrict__ slab, uint32_t mask, uint64_t n, float* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, (mask + 1) * sizeof(T), 0x00020000);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint32_t idx = hash32((uint32_t)i * 8u + c) & mask;
    if (sizeof(T) == 8) {
      float2 v;
      if (MODE == 0) v = *reinterpret_cast<const float2*>(&slab[idx]);
      else if (MODE == 1) { auto* p = reinterpret_cast<const float*>(&slab[idx]); v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1); }
      else {
        const int aux = MODE == 2 ? 1 : (MODE == 3 ? 16 : (MODE == 4 ? 17 : 2));  // sc0=1, sc1=16, nt=2
        auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, idx * 8, 0, aux);
        v.x = __builtin_bit_cast(float, r[0]); v.y = __builtin_bit_cast(float, r[1]);
      }
      acc += v.x + v.y;
    } else {
      uint32_t r;
      if (MODE == 0) r = *reinterpret_cast<const uint32_t*>(&slab[idx]);
      else if (MODE == 1) r = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(&slab[idx]));
      else {
        const int aux = MODE == 2 ? 1 : (MODE == 3 ? 16 : (MODE == 4 ? 17 : 2));
        r = __builtin_amdgcn_raw_buffer_load_b32(rsrc, idx * 4, 0, aux);
      }
      acc += (float)(r & 0xffff);
    }
  }
  out[i] = acc;
}

template <typename T>
void run(const char* tname) {
  const uint32_t entries = 1u << 19;
  const uint64_t n = 3145728;  // 32768 rays x 96 samples
  T* slab; float* out;
  hipMalloc(&slab, (size_t)entries * sizeof(T)); hipMemset(slab, 1, (size_t)entries * sizeof(T));
  hipMalloc(&out, n * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"plain", "nontemporal", "buffer sc0", "buffer sc1", "buffer sc0|sc1", "buffer nt"};
  for (int mode = 0; mode < 6; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(a);
      dim3 g((n + 255) / 256), bl(256);
      switch (mode) {
        case 0: hipLaunchKernelGGL((k<0, T>), g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 1: hipLaunchKernelGGL((k<1, T>), g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 2: hipLaunchKernelGGL((k<2, T>), g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 3: hipLaunchKernelGGL((k<3, T>), g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 4: hipLaunchKernelGGL((k<4, T>), g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 5: hipLaunchKernelGGL((k<5, T>), g, bl, 0, 0, slab, entries - 1, n, out); break;
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("%-8s %-16s %8.1f us  %7.1f G gathers/s\n", tname, names[mode], best * 1e3, n * 8 / best * 1e-6);
  }
  hipFree(slab); hipFree(out);
}

int main() {
  run<float2>("float2");
  run<uint32_t>("half2");
  return 0;
}
