// What does a divergent gather cost as a function of width and active lanes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(const float2* __restrict__ slab, uint32_t mask, uint64_t n, float* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  const bool odd = (hash32((uint32_t)i) >> 7) & 1;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t i0 = hash32((uint32_t)i * 8u + c) & mask;
    const uint32_t i1 = hash32((uint32_t)i * 8u + c + 4) & mask;
    if (MODE == 0) {            // 8 x float2, all lanes
      acc += slab[i0].x + slab[i1].y;
    } else if (MODE == 1) {     // 4 x float4 only
      const float4 p = *reinterpret_cast<const float4*>(slab + (i0 & ~1u)); acc += p.x + p.w;
    } else if (MODE == 2) {     // 4 x float4 + 4 x float2 on half of the lanes
      const float4 p = *reinterpret_cast<const float4*>(slab + (i0 & ~1u)); acc += p.x + p.w;
      if (odd) acc += slab[i1].y;
    } else if (MODE == 3) {     // 4 x float2 only
      acc += slab[i0].x;
    } else if (MODE == 4) {     // 4 x float2 all lanes + 4 x float2 half lanes
      acc += slab[i0].x; if (odd) acc += slab[i1].y;
    } else if (MODE == 5) {     // 4 x dword (4 B)
      acc += reinterpret_cast<const float*>(slab)[i0 * 2];
    }
  }
  out[i] = acc;
}
int main() {
  const uint32_t entries = 1u << 19; const uint64_t n = 3145728;
  float2* slab; float* out; hipMalloc(&slab, entries * 8); hipMemset(slab, 1, entries * 8); hipMalloc(&out, n * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"8 x float2", "4 x float4", "4 x float4 + 4 x float2(half lanes)", "4 x float2", "4 x float2 + 4 x float2(half lanes)", "4 x dword"};
  for (int mode = 0; mode < 6; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(a); dim3 g((n + 255) / 256), bl(256);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 1: hipLaunchKernelGGL(k<1>, g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 2: hipLaunchKernelGGL(k<2>, g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 3: hipLaunchKernelGGL(k<3>, g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 4: hipLaunchKernelGGL(k<4>, g, bl, 0, 0, slab, entries - 1, n, out); break;
        case 5: hipLaunchKernelGGL(k<5>, g, bl, 0, 0, slab, entries - 1, n, out); break;
      }
      hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-40s %8.1f us\n", names[mode], best * 1e3);
  }
  return 0;
}
