// Is  a - lo(p), b - hi(p)  (p = the two values rounded to bf16 in one dword,
// the residual of the bf16x3 operand split)  computable EXACTLY with one
// v_dot2_f32_bf16 each -- dot2(p, (-1, 0), a) and dot2(p, (0, -1), b) --
// instead of shift / and + subtract?  Prints the number of mismatches
// against the shift form over 2^22 random-ish values (0 = usable).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k(const float* x, uint32_t n, uint32_t mlo, uint32_t mhi, uint32_t* bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  const uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
  const float ra = a - __uint_as_float(p << 16), rb = b - __uint_as_float(p & 0xFFFF0000u);
  const bf16x2 pv = __builtin_bit_cast(bf16x2, p);
  const float da = __builtin_amdgcn_fdot2_f32_bf16(pv, __builtin_bit_cast(bf16x2, mlo), a, false);
  const float db = __builtin_amdgcn_fdot2_f32_bf16(pv, __builtin_bit_cast(bf16x2, mhi), b, false);
  if (__float_as_uint(da) != __float_as_uint(ra) && !(da == ra)) atomicAdd(bad, 1u);
  if (__float_as_uint(db) != __float_as_uint(rb) && !(db == rb)) atomicAdd(bad + 1, 1u);
}

int main() {
  const uint32_t n = 1u << 22;
  float* h = new float[n];
  uint32_t s = 12345u;
  for (uint32_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    const float m = (float)(s >> 8) / 16777216.0f * 2.0f - 1.0f;
    const int e = (int)((s >> 3) % 40) - 30;
    h[i] = ldexpf(m, e);
  }
  h[0] = 0.0f; h[1] = -0.0f; h[2] = 1e-38f; h[3] = 3e38f; h[4] = 1.0f; h[5] = -1.0f;
  float* d; uint32_t* bad;
  hipMalloc(&d, n * 4); hipMalloc(&bad, 8);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 8);
  // bf16 -1.0 = 0xBF80: (lo = -1, hi = 0) and (lo = 0, hi = -1)
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, d, n, 0x0000BF80u, 0xBF800000u, bad);
  uint32_t r[2];
  hipMemcpy(r, bad, 8, hipMemcpyDeviceToHost);
  printf("v_dot2_f32_bf16 residual vs shift/sub: %u mismatches (lo), %u (hi) of %u pairs\n", r[0], r[1], n / 2);
  return 0;
}
