// Micro-benchmark (diagnostic, not product): issue cost of the vector instructions the STFT kernels are made of, at 1, 2 and 4
// waves per SIMD.  Each kernel runs N iterations of 16 independent instructions of one kind per wave and reports
// s_memtime cycles per instruction as seen by one wave, and the SIMD's instruction throughput.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float cf __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters, float seed) {
  cf a[16];
  float s[16];
  f32x4 acc[4];
  for (int i = 0; i < 16; ++i) { a[i] = cf{seed + i + threadIdx.x, seed - i}; s[i] = seed * i + threadIdx.x; }
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  cf c = {seed, 1.0f};
  cf sc = {seed * 0.5f, 2.0f};   // wave-uniform: lives in an SGPR pair
  cf scs[8], b[8];
  for (int i = 0; i < 8; ++i) { scs[i] = cf{seed * (0.5f + i), 2.0f - i}; b[i] = cf{seed + threadIdx.x * 0.001f * i, 1.0f / (1 + i)}; }
  extern __shared__ float lds[];
  lds[threadIdx.x] = seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == 1) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == 2) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"(seed));
      REP16(X)
#undef X
    } else if (OP == 3) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(s[i]));
      REP16(X)
#undef X
    } else if (OP == 4) {
#define X(i) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(s[i]));
      REP16(X)
#undef X
    } else if (OP == 5) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == 6) {   // dependent chain of packed adds
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == 7) {   // dependent chain of plain fma
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[0]) : "v"(seed));
      REP16(X)
#undef X
    } else if (OP == 8) {   // 4x4x1 MFMA, 4 independent accumulators
#define X(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(s[i], seed, acc[i & 3], 0, 0, 0);
      REP16(X)
#undef X
    } else if (OP == 9) {   // packed add interleaved 1:1 with 4x4x1 MFMA (8 + 8)
#define X(i) if (i & 1) { acc[(i >> 1) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(s[i], seed, acc[(i >> 1) & 3], 0, 0, 0); } else { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
      REP16(X)
#undef X
    } else if (OP == 10) {  // v_log_f32
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(s[i]));
      REP16(X)
#undef X
    } else if (OP == 11) {  // ds_read_b32 (16 in flight) + wait
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s[i]) : "v"(threadIdx.x * 4), "i"(i * 4));
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 12) {  // v_cndmask
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(seed));
      REP16(X)
#undef X
    } else if (OP == 13) {  // 16x16x4 f32 MFMA
#define X(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(s[i], seed, acc[i & 3], 0, 0, 0);
      REP16(X)
#undef X
    } else if (OP == 15) {
#define X(i) asm volatile("s_nop 0");
      REP16(X)
#undef X
    } else if (OP == 16) {
#define X(i) if (i & 1) { asm volatile("s_nop 0"); } else { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
      REP16(X)
#undef X
    } else if (OP == 17) {
#define X(i) if ((i & 3) == 3) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s[i]) : "v"(threadIdx.x * 4), "i"(i * 4)); } else { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 18) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(acc[i & 3]) : "v"(threadIdx.x * 16), "i"(i * 16));
      REP16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 19) {  // packed fma with a scalar (SGPR pair) multiplicand and a broadcast half, as the CNN's conv1 issues it
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(c), "s"(sc));
      REP16(X)
#undef X
    } else if (OP == 21) {  // conv1's stream: eight different SGPR pairs, both broadcast halves, varying vector sources
#define X(i) if (i & 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(a[i]) : "v"(b[(i + 5) & 7]), "s"(scs[i & 7])); } else { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b[(i + 3) & 7]), "s"(scs[i & 7])); }
      REP16(X)
#undef X
    } else if (OP == 14) {  // 8 packed adds + 8 plain fma interleaved
#define X(i) if (i & 1) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[i]) : "v"(seed)); } else { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
      REP16(X)
#undef X
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a[i][0] + a[i][1] + s[i];
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (r == 12345.678f) sink[0] = r;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, unsigned long long* d_out, float* d_sink) {
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 2000;   // (long runs take the clock ramp out of the wall time)
  for (int wg_per_cu : {1, 2, 4, 5, 6, 8}) {
    const int grid = 256 * wg_per_cu;
    const size_t lds = (size_t)(163840 / wg_per_cu - 1024) & ~(size_t)255;   // pins the residency: wg_per_cu workgroups of 4 waves per CU
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 256, lds>>>(d_out, d_sink, 10, 1.0f);
    hipEventRecord(e0);
    k<OP><<<grid, 256, lds>>>(d_out, d_sink, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= grid;
    const double n = 16.0 * iters;
    printf("%-34s %d wave/SIMD: %6.2f cycles/instr per wave, %5.2f cycles/instr per SIMD, wall %.3f ms = %.2f ns per instruction and SIMD (counter at %.2f GHz)\n",
           name, wg_per_cu, cyc / n, cyc / n / wg_per_cu, ms, ms * 1e6 / (n * wg_per_cu), cyc / (ms * 1e6));
  }
}

int main(int argc, char** argv) {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 4096 * 8); hipMalloc(&d_sink, 64);
  if (argc > 1) {   // only the packed-fma operand forms
    run<0>("v_pk_fma_f32", d_out, d_sink);
    run<19>("v_pk_fma_f32 v, v.lo, s[pair], v", d_out, d_sink);
    run<21>("v_pk_fma_f32, 8 SGPR pairs, lo/hi", d_out, d_sink);
    return 0;
  }
  run<0>("v_pk_fma_f32", d_out, d_sink);
  run<1>("v_pk_add_f32 op_sel/neg", d_out, d_sink);
  run<5>("v_pk_mul_f32", d_out, d_sink);
  run<2>("v_fma_f32", d_out, d_sink);
  run<15>("s_nop 0", d_out, d_sink);
  run<16>("v_pk_add + s_nop 0 1:1", d_out, d_sink);
  run<17>("v_pk_add + ds_read_b32 3:1 (+wait)", d_out, d_sink);
  run<18>("ds_read_b128 x16 + wait", d_out, d_sink);
  run<8>("v_mfma_f32_4x4x1 (4 acc)", d_out, d_sink);
  run<9>("v_pk_add + mfma_4x4x1 1:1", d_out, d_sink);
  run<11>("ds_read_b32 x16 + wait", d_out, d_sink);
  return 0;
}
