// Micro-benchmark (diagnostic, not product): issue cost of the instructions the float64 parity kernel is made of, at
// 1 .. 8 waves per SIMD -- v_fma_f64 / v_add_f64 / v_mul_f64, the float64 <-> float32 conversions, v_permlane32_swap,
// ds_bpermute_b32, ds_read / ds_write_b64.  Same method as valu_rate.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_rate.hip -o /tmp/f64_rate && /tmp/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters, double seed) {
  double d[16];
  float s[16];
  for (int i = 0; i < 16; ++i) { d[i] = seed + i + threadIdx.x; s[i] = (float)seed * i + threadIdx.x; }
  extern __shared__ double lds[];
  lds[threadIdx.x] = seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(seed));
      REP16(X)
#undef X
    } else if (OP == 1) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(seed));
      REP16(X)
#undef X
    } else if (OP == 2) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(seed));
      REP16(X)
#undef X
    } else if (OP == 3) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s[i]) : "v"(d[i]));
      REP16(X)
#undef X
    } else if (OP == 4) {
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(s[i]));
      REP16(X)
#undef X
    } else if (OP == 5) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[i]), "+v"(s[(i + 1) & 15]));
      REP16(X)
#undef X
    } else if (OP == 6) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(s[i]) : "v"(threadIdx.x * 4 ^ 252));
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 7) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(threadIdx.x * 8), "i"(i * 8));
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 8) {
#define X(i) asm volatile("ds_write_b64 %1, %0 offset:%2" : : "v"(d[i]), "v"(threadIdx.x * 8), "i"(i * 8) : "memory");
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 12) {
#define X(i) asm volatile("ds_write_b32 %1, %0 offset:%2" : : "v"(s[i]), "v"(threadIdx.x * 4), "i"(i * 4) : "memory");
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 13) {   // 8 x ds_write2_b64 (two doubles each), counted as 16 values
#define X(i) if (i < 8) asm volatile("ds_write2_b64 %2, %0, %1 offset0:%3 offset1:%4" : : "v"(d[2 * i]), "v"(d[2 * i + 1]), "v"(threadIdx.x * 8), "i"(2 * i), "i"(2 * i + 1) : "memory");
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 14) {   // 8 x ds_write_b128, counted as 16 doubles
#define X(i) if (i < 8) asm volatile("ds_write_b128 %1, %0 offset:%2" : : "v"(*(const __attribute__((ext_vector_type(4))) float*)&d[2 * i]), "v"(threadIdx.x * 16), "i"(i * 16) : "memory");
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 15) {   // ds_write_b64 with only lanes 0..31 active
      if ((threadIdx.x & 63) < 32) {
#define X(i) asm volatile("ds_write_b64 %1, %0 offset:%2" : : "v"(d[i]), "v"(threadIdx.x * 8), "i"(i * 8) : "memory");
      REP16(X)
#undef X
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 16) {   // scratch-like: global loads that hit L2
#define X(i) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(s[i]) : "v"(sink + threadIdx.x), "i"(i * 256) : "memory");
      REP16(X)
#undef X
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (OP == 9) {   // dependent chain
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[0]) : "v"(seed));
      REP16(X)
#undef X
    } else if (OP == 10) {  // float64 fma and float32 packed fma 1:1
#define X(i) if (i & 1) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(seed)); } else { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"((float)seed)); }
      REP16(X)
#undef X
    } else if (OP == 11) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
      REP16(X)
#undef X
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double r = 0;
  for (int i = 0; i < 16; ++i) r += d[i] + s[i];
  if (r == 12345.678) sink[0] = (float)r;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, unsigned long long* d_out, float* d_sink) {
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 1000;   // (long runs take the clock ramp out of the wall time)
  for (int wg_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * wg_per_cu;
    const size_t lds = (size_t)(163840 / wg_per_cu - 1024) & ~(size_t)255;   // pins the residency: wg_per_cu workgroups of 4 waves per CU
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 256, lds>>>(d_out, d_sink, 10, 1.0);
    hipEventRecord(e0);
    k<OP><<<grid, 256, lds>>>(d_out, d_sink, iters, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= grid;
    const double n = 16.0 * iters;
    printf("%-28s %d wave/SIMD: %6.2f cycles/instr per wave, %5.2f cycles/instr per SIMD, wall %.3f ms = %.2f ns per instruction and SIMD\n",
           name, wg_per_cu, cyc / n, cyc / n / wg_per_cu, ms, ms * 1e6 / (n * wg_per_cu));
  }
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 4096 * 8); hipMalloc(&d_sink, 1 << 20);
  run<0>("v_fma_f64", d_out, d_sink);
  run<1>("v_add_f64", d_out, d_sink);
  run<2>("v_mul_f64", d_out, d_sink);
  run<9>("v_fma_f64 dependent chain", d_out, d_sink);
  run<10>("v_fma_f64 + v_fma_f32 1:1", d_out, d_sink);
  run<3>("v_cvt_f32_f64", d_out, d_sink);
  run<4>("v_cvt_f64_i32", d_out, d_sink);
  run<11>("v_rcp_f32", d_out, d_sink);
  run<5>("v_permlane32_swap_b32", d_out, d_sink);
  run<6>("ds_bpermute_b32 x16 + wait", d_out, d_sink);
  run<7>("ds_read_b64 x16 + wait", d_out, d_sink);
  run<8>("ds_write_b64 x16 + wait", d_out, d_sink);
  run<12>("ds_write_b32 x16 + wait", d_out, d_sink);
  run<13>("ds_write2_b64 x8 (per double)", d_out, d_sink);
  run<14>("ds_write_b128 x8 (per double)", d_out, d_sink);
  run<15>("ds_write_b64 x16 half exec", d_out, d_sink);
  run<16>("global_load_dword x16 + wait", d_out, d_sink);
  return 0;
}
