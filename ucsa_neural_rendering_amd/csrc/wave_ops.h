// Wave64 scan / reduce helpers (deterministic: fixed tree, no atomics).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float wave_incl_scan_mul(float v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v = o * v;
  }
  return v;
}

__device__ __forceinline__ float wave_incl_scan_add(float v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v = o + v;
  }
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ float wave_bcast(float v, int src) {
  return __shfl(v, src, 64);
}

__device__ __forceinline__ uint32_t wave_incl_scan_add_u32(uint32_t v,
                                                           uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= (uint32_t)d) v = o + v;
  }
  return v;
}
