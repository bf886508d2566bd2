// Micro-benchmark: wave64 inclusive add-scan, __shfl_up (ds_bpermute_b32) steps
// vs DPP (row_shr 1/2/4/8 + row_bcast15 + row_bcast31).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float scan_shfl(float v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v = o + v;
  }
  return v;
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                            ROW_MASK, BANK_MASK, true);
  return v + __builtin_bit_cast(float, o);
}

__device__ __forceinline__ float scan_dpp(float v) {
  v = dpp_add<0x111, 0xf, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa, 0xf>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_add<0x143, 0xc, 0xf>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, uint32_t iters) {
  const uint32_t lane = threadIdx.x & 63u;
  float v = 1.0f + (float)(threadIdx.x & 7u) * 0.125f, acc = 0.f;
  for (uint32_t it = 0; it < iters; ++it) {
    const float s = MODE == 0 ? scan_shfl(v, lane) : scan_dpp(v);
    acc += s;
    v = v * 0.999f + 0.001f;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  float* h = (float*)malloc(256 * 4 * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const uint32_t blocks = 4096, iters = 512;
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    hipMemcpy(h + mode * 256, out, 256 * 4, hipMemcpyDeviceToHost);
    const double scans = (double)blocks * 4 * iters;
    printf("%-10s %8.3f ms  %7.1f G wave-scans/s   first sums %.4f %.4f %.4f\n",
           mode ? "dpp" : "shfl_up", best, scans / best / 1e6, h[mode * 256 + 1], h[mode * 256 + 17], h[mode * 256 + 63]);
  }
  return 0;
}
