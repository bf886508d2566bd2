// Round 5: does the per-CU L1 (TCP) merge two LANES of one load instruction
// that fall into the same 128-byte line?  On a hashed level the x-neighbours
// of a corner are idx and idx ^ (2^k - 1): the same line for 15 of 16 x0.  The
// shipped encoder issues them as two instructions of ONE lane (two look-ups);
// a lane PAIR issuing them in one instruction would be one look-up if the TCP
// merges lanes.  3.1 M samples x 8 corner loads from a 4 MiB slab, 8-byte loads.
//   mode 0  lane = sample, 8 loads: 4 x (idx, idx ^ 1) back to back   (shipped pattern)
//   mode 1  lane = sample, 8 random loads                              (no sharing at all)
//   mode 2  lane pair (2i, 2i+1) = sample, 4 loads each: idx ^ side    (same 16-byte slot)
//   mode 3  lane pair, idx ^ (side * 7)                                (same line, other slot)
//   mode 4  lanes l and l + 32 = sample, idx ^ side                    (partners not adjacent)
//   mode 5  lane pair, partners in DIFFERENT lines (idx ^ side * 16)   (control: 2 x the work of mode 1 per sample)
//   mode 6  lane quad = sample pair ... idx ^ (l & 3): four lanes one line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ void __launch_bounds__(256)
k(const float2* __restrict__ slab, uint32_t mask, uint32_t n_threads, float* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_threads) return;
  float acc = 0.f;
  const uint32_t lane = threadIdx.x & 63u, wave = i >> 6;
  if (MODE <= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t i0 = hash32(i * 8u + c) & mask;
      const uint32_t i1 = MODE == 0 ? (i0 ^ 1u) : (hash32(i * 8u + c + 4) & mask);
      const float2 a = slab[i0], b = slab[i1];
      acc += a.x + b.y;
    }
  } else {
    uint32_t s, side;   // sample id, which partner
    if (MODE == 4) { s = wave * 32u + (lane & 31u); side = lane >> 5; }
    else if (MODE == 6) { s = i >> 2; side = lane & 3u; }
    else { s = i >> 1; side = lane & 1u; }
    const int loads = MODE == 6 ? 2 : 4;
#pragma unroll
    for (int c = 0; c < loads; ++c) {
      uint32_t idx = hash32(s * 8u + c) & mask;
      if (MODE == 2 || MODE == 4) idx ^= side;
      if (MODE == 3) idx ^= side * 7u;
      if (MODE == 5) idx ^= side * 16u;
      if (MODE == 6) idx ^= side;
      const float2 a = slab[idx];
      acc += a.x + a.y;
    }
  }
  out[i] = acc;
}

int main() {
  const uint32_t entries = 1u << 19;
  const uint32_t n = 3145728;  // samples
  float2* slab; float* out;
  hipMalloc(&slab, (size_t)entries * 8); hipMemset(slab, 1, (size_t)entries * 8);
  hipMalloc(&out, (size_t)n * 4 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"lane=sample 4x(idx,idx^1)", "lane=sample 8 random", "lane pair idx^side",
                         "lane pair idx^(7 side)", "lanes l,l+32 idx^side", "lane pair, 2 lines",
                         "lane quad idx^(l&3)"};
  for (int mode = 0; mode < 7; ++mode) {
    const uint32_t threads = mode <= 1 ? n : (mode == 6 ? n * 4u : n * 2u);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(a);
      dim3 g((threads + 255) / 256), bl(256);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, g, bl, 0, 0, slab, entries - 1, threads, out); break;
        case 1: hipLaunchKernelGGL(k<1>, g, bl, 0, 0, slab, entries - 1, threads, out); break;
        case 2: hipLaunchKernelGGL(k<2>, g, bl, 0, 0, slab, entries - 1, threads, out); break;
        case 3: hipLaunchKernelGGL(k<3>, g, bl, 0, 0, slab, entries - 1, threads, out); break;
        case 4: hipLaunchKernelGGL(k<4>, g, bl, 0, 0, slab, entries - 1, threads, out); break;
        case 5: hipLaunchKernelGGL(k<5>, g, bl, 0, 0, slab, entries - 1, threads, out); break;
        case 6: hipLaunchKernelGGL(k<6>, g, bl, 0, 0, slab, entries - 1, threads, out); break;
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    const double lane_loads = (double)n * 8.0;
    printf("mode %d %-28s %8.1f us  %6.2f G lane-loads/s  %5.2f lane-loads/clk/CU (2.4 GHz x 256)\n",
           mode, names[mode], best * 1e3, lane_loads / best * 1e-6,
           lane_loads / (best * 1e-3 * 2.4e9 * 256));
  }
  return 0;
}
