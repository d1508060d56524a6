// GPU check (run with gpurun): the short correctly-rounded division / square-root sequences of cabs2_numpy against
// hipcc's correctly rounded operators -- sqrt exhaustively over [1, 2], division over 2^28 random pairs 0 <= a <= b.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float div_fast(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q;
  {
#pragma clang fp contract(off)
    q = a * r;
  }
  const float rem = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(rem, r, q);
}
__device__ __forceinline__ float sqrt_fast(float x) {   // x in [1, 2]
  const float s = __builtin_amdgcn_rsqf(x);
  float g, h;
  {
#pragma clang fp contract(off)
    g = x * s; h = 0.5f * s;
  }
  const float r = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, r, g);
  h = __builtin_fmaf(h, r, h);
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
__global__ void k_sqrt(unsigned long long* bad) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;   // 2^23 + 1 values
  if (i > (1u << 23)) return;
  const float x = __uint_as_float(0x3f800000u + i);
  if (sqrt_fast(x) != sqrtf(x)) atomicAdd(bad, 1ull);
}
__global__ void k_div(unsigned long long* bad, unsigned seed) {
  unsigned long long s = ((unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) + 1) * 6364136223846793005ull + seed;
  unsigned long long n = 0;
  for (int it = 0; it < 1024; ++it) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const unsigned ua = (unsigned)(s >> 32), ub = (unsigned)s;
    // b: random mantissa, exponent in [2^-40, 2^40]; a = b * u with u in [0, 1) through a random smaller float
    const float b = __uint_as_float(((87u + (ub >> 23) % 80u) << 23) | (ub & 0x7fffffu));
    float a = __uint_as_float(((87u + (ua >> 23) % 80u) << 23) | (ua & 0x7fffffu));
    if (a > b) { const float t = a; a = b; /* keep a <= b */ (void)t; a = __uint_as_float(__float_as_uint(b) - (ua & 0xfffffu)); }
    if (div_fast(a, b) != a / b) ++n;
  }
  if (n) atomicAdd(bad, n);
}
int main() {
  unsigned long long *d, h[2] = {0, 0};
  hipMalloc(&d, 16); hipMemset(d, 0, 16);
  hipLaunchKernelGGL(k_sqrt, dim3((1 << 23) / 256 + 1), dim3(256), 0, 0, d);
  hipLaunchKernelGGL(k_div, dim3(1024), dim3(256), 0, 0, d + 1, 12345u);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("sqrt mismatches over [1,2]: %llu of %u; div mismatches: %llu of %llu\n", h[0], (1u << 23) + 1, h[1], 1024ull * 256 * 1024);
  return (h[0] || h[1]) ? 1 : 0;
}
