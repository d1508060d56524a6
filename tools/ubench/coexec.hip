// Diagnostic (not product): does a kernel's result change when a float16-MFMA kernel of another stream shares its CU?
// Each victim kernel computes a deterministic function of its thread index with one kind of instruction; it is run
// alone, then while an aggressor (a chain of v_mfma_f32_16x16x32_f16, or of v_mfma_f32_16x16x4_f32 as a control) runs on
// a second stream, and the outputs are compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float cf __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 4) void aggressor(int kind, int iters, float* sink) {
  const int tid = threadIdx.x;
  float acc;
  if (kind == 0) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (tid + j)); b[j] = (_Float16)(0.02f * j); }
    f4 c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d, 0, 0, 0);
    }
    acc = c[0] + d[1];
  } else if (kind == 2) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    b8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (tid + j)); b[j] = (__bf16)(0.02f * j); }
    f4 c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, d, 0, 0, 0);
    }
    acc = c[0] + d[1];
  } else {
    f4 c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(0.01f * tid, 0.5f, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(0.5f, 0.01f * tid, d, 0, 0, 0);
    }
    acc = c[0] + d[1];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int V>
__global__ __launch_bounds__(256) void victim(float* out, int iters) {
  __shared__ float sl[4096];
  const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
  float r = 0.f;
  const float x0 = 1.0f + 1e-3f * (gid & 1023);
  if (V == 0) {        // packed fma chain (plain operands)
    cf a = {x0, 0.5f * x0}; const cf m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    for (int i = 0; i < iters; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
    r = a[0] + a[1];
  } else if (V == 1) { // packed add with op_sel / neg modifiers + packed mul
    cf a = {x0, 0.5f * x0}; const cf m = {0.9995f, 0.9995f}, c = {1e-3f, 2e-3f};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a) : "v"(c));
      asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(m));
    }
    r = a[0] + a[1];
  } else if (V == 2) { // plain fma chain
    float a = x0;
    for (int i = 0; i < iters; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(0.999f), "v"(1e-3f));
    r = a;
  } else if (V == 3) { // float32 MFMA 16x16x4 dependent chain
    f4 c = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, 1e-3f * (1 + (i & 7)), c, 0, 0, 0);
    r = c[0] + c[1] + c[2] + c[3];
  } else if (V == 4) { // float32 MFMA 4x4x1 dependent chain
    f4 c = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, 1e-3f * (1 + (i & 7)), c, 0, 0, 0);
    r = c[0] + c[1] + c[2] + c[3];
  } else if (V == 5) { // float64 MFMA chain
    d4 c = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x0, 1e-3 * (1 + (i & 7)), c, 0, 0, 0);
    r = (float)(c[0] + c[1] + c[2] + c[3]);
  } else if (V == 6) { // DPP row_mirror + add
    float a = x0;
    for (int i = 0; i < iters; ++i) {
      const float m = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x140, 0xf, 0xf, true));
      a = a * 0.5f + m * 0.4999f + 1e-3f;
    }
    r = a;
  } else if (V == 7) { // SDWA int16 -> float conversions
    int w = (gid * 2654435761u) >> 3; float a = 0.f;
    for (int i = 0; i < iters; ++i) { a += (float)(short)(w & 0xffff) * 1e-6f + (float)(w >> 16) * 1e-7f; w = w * 1664525 + 1013904223; }
    r = a;
  } else if (V == 8) { // LDS transpose traffic
    float a = x0;
    for (int i = 0; i < iters; ++i) {
      sl[(tid * 17 + i) & 4095] = a;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      a = a * 0.5f + sl[((tid ^ 37) * 17 + i) & 4095] * 0.25f + 1e-3f;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    r = a;
  } else if (V == 9) { // v_log_f32 / v_exp_f32
    float a = x0;
    for (int i = 0; i < iters; ++i) a = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(a) * 0.999f) + 1e-3f;
    r = a;
  } else if (V == 11) { // packed add, op_sel only (swapped halves of src1)
    cf a = {x0, 0.5f * x0}; const cf c = {1e-3f, 2e-3f}, m = {0.9995f, 0.9995f};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(c));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
    }
    r = a[0] + a[1];
  } else if (V == 12) { // packed add, neg modifier only
    cf a = {x0, 0.5f * x0}; const cf c = {1e-3f, -2e-3f}, m = {0.9995f, 0.9995f};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %1 neg_hi:[0,1]" : "+v"(a) : "v"(c));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
    }
    r = a[0] + a[1];
  } else if (V == 13) { // packed mul with a broadcast operand (op_sel_hi:[1,0])
    cf a = {x0, 0.5f * x0}; const cf c = {1e-3f, 2e-3f}, m = {0.9995f, 123.0f};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
      asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(m));
    }
    r = a[0] + a[1];
  } else if (V == 14) { // packed fma with op_sel on two operands (the complex multiply's second half)
    cf a = {x0, 0.5f * x0}; const cf w = {0.9995f, 1e-4f}, t = {1e-3f, 2e-3f};
    for (int i = 0; i < iters; ++i)
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(a) : "v"(w), "v"(t));
    r = a[0] + a[1];
  } else if (V == 15) { // compiler-generated packed math with a scalar broadcast
    cf a = {x0, 0.5f * x0};
    const float sc = 0.9995f + 1e-9f * blockIdx.x;
    for (int i = 0; i < iters; ++i) { a = a * cf{sc, sc} + cf{1e-3f, 2e-3f}; asm volatile("" : "+v"(a)); }
    r = a[0] + a[1];
  } else if (V == 16) { // v_swap_b32 + plain packed add (the op_sel-free way to swap halves)
    cf a = {x0, 0.5f * x0}; const cf m = {0.9995f, 0.9995f}; cf c = {1e-3f, 2e-3f};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_swap_b32 %0, %1" : "+v"(c[0]), "+v"(c[1]));
      asm volatile("v_pk_add_f32 %0, %0, %1 neg_hi:[0,1]" : "+v"(a) : "v"(c));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
    }
    r = a[0] + a[1];
  } else if (V == 17) { // packed mul with swapped halves of src1
    cf a = {x0, 0.5f * x0}; const cf c = {1e-3f, 2e-3f}, m = {0.9995f, 0.9996f};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
      asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(m));
    }
    r = a[0] + a[1];
  } else if (V == 18) { // packed add, both results from the low halves (op_sel_hi:[0,0]) and from the high halves
    cf a = {x0, 0.5f * x0}; const cf c = {1e-3f, 2e-3f}, m = {0.9995f, 0.9995f};
    for (int i = 0; i < iters; ++i) {
      cf u, v;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(u) : "v"(a), "v"(c));
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(v) : "v"(a), "v"(c));
      a = (u + v) * m;
      asm volatile("" : "+v"(a));
    }
    r = a[0] + a[1];
  } else if (V == 19) { // the fma form of a + swap(b) with an inline constant 1.0
    cf a = {x0, 0.5f * x0}; const cf c = {1e-3f, 2e-3f}, m = {0.9995f, 0.9995f};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_fma_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "+v"(a) : "v"(c));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
    }
    r = a[0] + a[1];
  } else if (V == 20) { // the same sum through v_pk_add_f32 op_sel (reference for V == 19's value when run alone)
    cf a = {x0, 0.5f * x0}; const cf c = {1e-3f, 2e-3f}, m = {0.9995f, 0.9995f};
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a) : "v"(c));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
    }
    r = a[0] + a[1];
  } else if (V == 10) { // packed fma fed by float32 MFMA results (the mix of the STFT kernels)
    f4 c = {0, 0, 0, 0}; cf a = {x0, 0.5f * x0};
    for (int i = 0; i < iters; ++i) {
      c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], 1e-3f, c, 0, 0, 0);
      const cf m = {0.999f, 1.001f}, k = {c[0] * 1e-6f, 1e-3f};
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
    }
    r = a[0] + a[1] + c[1];
  }
  out[gid] = r;
}

template <int V>
void test(const char* name, float* d_out, float* d_sink, hipStream_t s1, hipStream_t s2) {
  const int blocks = 2048, n = blocks * 256, iters = 2000;
  std::vector<float> ref(n), got(n);
  victim<V><<<blocks, 256, 0, s1>>>(d_out, iters);
  hipStreamSynchronize(s1);
  hipMemcpy(ref.data(), d_out, n * 4, hipMemcpyDeviceToHost);
  for (int kind = 0; kind < 3; ++kind) {
    int bad_runs = 0; long bad_vals = 0;
    for (int rep = 0; rep < 6; ++rep) {
      aggressor<<<4096, 512, 0, s2>>>(kind, 4000, d_sink);
      victim<V><<<blocks, 256, 0, s1>>>(d_out, iters);
      hipDeviceSynchronize();
      hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost);
      long b = 0;
      for (int i = 0; i < n; ++i) b += std::memcmp(&ref[i], &got[i], 4) != 0;
      bad_runs += b != 0; bad_vals += b;
    }
    printf("%-44s beside %-22s: %d of 6 runs differ from the solo run (%ld values)\n", name,
           kind == 0 ? "float16 MFMA chains" : (kind == 1 ? "float32 MFMA chains" : "bfloat16 MFMA chains"), bad_runs, bad_vals);
  }
}

int main() {
  float *d_out, *d_sink;
  hipMalloc(&d_out, 2048 * 256 * 4); hipMalloc(&d_sink, 64);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  test<0>("v_pk_fma_f32 chain", d_out, d_sink, s1, s2);
  test<1>("v_pk_add_f32 op_sel/neg + v_pk_mul_f32", d_out, d_sink, s1, s2);
  test<2>("v_fma_f32 chain", d_out, d_sink, s1, s2);
  test<3>("v_mfma_f32_16x16x4_f32 chain", d_out, d_sink, s1, s2);
  test<4>("v_mfma_f32_4x4x1_16B_f32 chain", d_out, d_sink, s1, s2);
  test<5>("v_mfma_f64_16x16x4_f64 chain", d_out, d_sink, s1, s2);
  test<6>("v_mov_b32_dpp row_mirror", d_out, d_sink, s1, s2);
  test<7>("v_cvt_f32_i32_sdwa", d_out, d_sink, s1, s2);
  test<8>("LDS write / barrier / read", d_out, d_sink, s1, s2);
  test<9>("v_log_f32 / v_exp_f32", d_out, d_sink, s1, s2);
  test<10>("4x4x1 MFMA feeding v_pk_fma_f32", d_out, d_sink, s1, s2);
  test<11>("v_pk_add_f32 op_sel only", d_out, d_sink, s1, s2);
  test<12>("v_pk_add_f32 neg_hi only", d_out, d_sink, s1, s2);
  test<13>("v_pk_mul_f32 op_sel_hi:[1,0] (broadcast)", d_out, d_sink, s1, s2);
  test<14>("v_pk_fma_f32 op_sel + neg (cmul half)", d_out, d_sink, s1, s2);
  test<15>("compiler packed math, scalar broadcast", d_out, d_sink, s1, s2);
  test<16>("v_swap_b32 + v_pk_add_f32 neg_hi", d_out, d_sink, s1, s2);
  test<17>("v_pk_mul_f32 op_sel (swapped src1)", d_out, d_sink, s1, s2);
  test<18>("v_pk_add_f32 op_sel_hi:[0,0] / op_sel:[1,1]", d_out, d_sink, s1, s2);
  test<19>("v_pk_fma_f32 x, 1.0, a op_sel (= a + swap x)", d_out, d_sink, s1, s2);
  {  // the fma form must give the add form's value (run alone)
    const int n = 2048 * 256; std::vector<float> r19(n), r20(n);
    victim<19><<<2048, 256, 0, s1>>>(d_out, 2000); hipStreamSynchronize(s1); hipMemcpy(r19.data(), d_out, n * 4, hipMemcpyDeviceToHost);
    victim<20><<<2048, 256, 0, s1>>>(d_out, 2000); hipStreamSynchronize(s1); hipMemcpy(r20.data(), d_out, n * 4, hipMemcpyDeviceToHost);
    long b = 0; for (int i = 0; i < n; ++i) b += std::memcmp(&r19[i], &r20[i], 4) != 0;
    printf("fma form vs add form, both alone: %ld of %d values differ\n", b, n);
  }
  return 0;
}
