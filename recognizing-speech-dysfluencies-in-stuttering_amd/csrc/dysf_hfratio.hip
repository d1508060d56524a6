// dysf_hfratio.hip -- high_freq_energy_ratio (pipeline1.py:177-186) on the GPU:
//     fft = np.fft.rfft(y);  ratio = sum(|fft[freqs > 4000]|**2) / (sum(|fft|**2) + 1e-10)
// The transform spans the WHOLE clip, whose length N is arbitrary (the reference's clips run from 7 140 to ~160 000
// samples, most of them with large prime factors), so it is evaluated as a chirp-z (Bluestein) convolution on
// power-of-two FFTs of one common size M >= 2 * N_max - 1 for the whole batch:
//     X[k] = w[k] * sum_n (y[n] w[n]) conj(w[k - n]),   w[n] = exp(-i pi n^2 / N)
// Kernels (all on the caller's stream, batched over clips; a reporting metric, not a throughput path):
//   hf_prep      a[m] = y[m] w[m] (zero padded), b = conj(w) wrapped around M; n^2 mod 2N in 64-bit integers, sincospi in
//                float64, data in float32 (NumPy >= 2 transforms float32 input in single precision too)
//   hf_fft_pass  one Stockham autosort radix-4 (or radix-2) pass over a and b, ping-pong buffers in global memory
//   hf_multiply  conj(A * B): the inverse transform is the conjugate of the forward transform of the conjugate
//   hf_finish    X[k] = w[k] conj(c[k]) / M for k <= N/2, |X[k]|**2 as np.abs(complex64)**2, the two sums in float64
#include <hip/hip_runtime.h>

#include "dysf_kernels.h"

namespace dysf {

typedef float c32 __attribute__((ext_vector_type(2)));

// a * b as one v_pk_mul_f32 + one v_pk_fma_f32 whose operand selections are safe beside float16 MFMAs of other waves (the
// plain expression compiles to a v_pk_mul_f32 with mixed source halves: see dysf_fft.hpp add_mi and tools/check_isa.py)
__device__ __forceinline__ c32 cmulf(c32 a, c32 b) {
  c32 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));  // (a0 b0, a1 b0)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));  // (-a1 b1 + t0, a0 b1 + t1)
  return r;
}

// w[n] = exp(-i pi n^2 / N): the exponent is reduced exactly (n^2 mod 2N) before it meets floating point
__device__ __forceinline__ c32 chirp(long long n, long long N) {
  const unsigned long long r = ((unsigned long long)n * (unsigned long long)n) % (unsigned long long)(2 * N);
  double sn, cs;
  sincospi((double)r / (double)N, &sn, &cs);
  return c32{(float)cs, (float)(-sn)};
}

struct HfArgs {
  const void* pcm;
  const long long* offsets;
  int pcm_type;
  int sr;
  long long M;        // transform size (power of two, >= 2 * longest clip - 1)
  c32* a0; c32* a1;   // [clip][M] ping-pong buffers of the data sequence
  c32* b0; c32* b1;   // [clip][M] ping-pong buffers of the chirp filter
  const int* flags;   // [clip] non-finite flag of the feature kernels
  float* metrics;     // [clip][3]; slot 2 is written here
};

__global__ __launch_bounds__(256) void hf_prep_kernel(HfArgs h) {
  const int clip = blockIdx.y;
  const long long off = h.offsets[clip];
  long long N = h.offsets[clip + 1] - off;
  if (2 * N - 1 > h.M) N = 0;   // a clip beyond the caller's max_samples: the chirp convolution would alias -- rejected like an empty one
  c32* a = h.a0 + (size_t)clip * h.M;
  c32* b = h.b0 + (size_t)clip * h.M;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < h.M; m += (long long)gridDim.x * 256) {
    c32 av = {0.f, 0.f}, bv = {0.f, 0.f};
    if (N > 0) {
      if (m < N) {
        const float y = (h.pcm_type == DYSF_PCM_I16) ? (float)((const short*)h.pcm)[off + m] * (1.0f / 32768.0f)
                                                     : ((const float*)h.pcm)[off + m];
        const c32 w = chirp(m, N);
        av = c32{y * w[0], y * w[1]};
        bv = c32{w[0], -w[1]};
      } else if (h.M - m < N) {   // negative lags m - M = -(N-1) .. -1
        const c32 w = chirp(h.M - m, N);
        bv = c32{w[0], -w[1]};
      }
    }
    a[m] = av;
    b[m] = bv;
  }
}

// One Stockham pass of radix R over `batches` sequences of length M laid out back to back: thread j of a sequence takes
// the inputs j + r * M / R, applies the twiddles exp(-2 pi i r k / (Ns R)), k = j mod Ns, and a radix-R butterfly, and
// stores to (j / Ns) * Ns * R + k + r * Ns (the same index maps as the wave FFT in dysf_fft.hpp).
template <int R>
__global__ __launch_bounds__(256) void hf_fft_pass_kernel(const c32* in, c32* out, long long M, long long Ns) {
  const c32* src = in + (size_t)blockIdx.y * M;
  c32* dst = out + (size_t)blockIdx.y * M;
  const long long q = M / R;
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < q; j += (long long)gridDim.x * 256) {
    const long long k = j % Ns;
    c32 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = src[j + r * q];
      if (r > 0) {
        double sn, cs;
        sincospi(-2.0 * (double)(r * k) / (double)(Ns * R), &sn, &cs);
        v[r] = cmulf(v[r], c32{(float)cs, (float)sn});
      }
    }
    if (R == 2) {
      const c32 x = v[0], y = v[1];
      v[0] = x + y; v[1] = x - y;
    } else {
      const c32 t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = v[1] - v[3];
      const c32 mi = c32{t3[1], -t3[0]};  // -i t3
      v[0] = t0 + t2; v[1] = t1 + mi; v[2] = t0 - t2; v[3] = t1 - mi;
    }
    const long long base = (j / Ns) * Ns * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[base + r * Ns] = v[r];
  }
}

__global__ __launch_bounds__(256) void hf_multiply_kernel(const c32* A, const c32* B, c32* out, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const c32 p = cmulf(A[i], B[i]);
    out[i] = c32{p[0], -p[1]};
  }
}

// np.abs(complex64)**2 as in dysf_kernels.hip (cabs2_numpy): larger * sqrt(fma(r, r, 1)), squared, float32 steps
__device__ __forceinline__ float cabs2_np(float re, float im) {
  const float ar = fabsf(re), ai = fabsf(im);
  const float lg = fmaxf(ar, ai), sm = fminf(ai, ar);
  const float ratio = (lg > 0.f && lg < INFINITY) ? sm / lg : 0.f;
  float m, s;
  {
#pragma clang fp contract(off)
    m = sqrtf(__builtin_fmaf(ratio, ratio, 1.0f)) * lg;
    s = m * m;
  }
  return s;
}

__global__ __launch_bounds__(256) void hf_finish_kernel(HfArgs h, const c32* conv) {
  __shared__ double red[8];
  const int clip = blockIdx.x, tid = threadIdx.x;
  const long long off = h.offsets[clip], N = h.offsets[clip + 1] - off;
  float* out = h.metrics + (size_t)clip * 3 + 2;
  if (N <= 0 || 2 * N - 1 > h.M || h.flags[clip]) { if (tid == 0) *out = 0.f; return; }   // (too long for M: see hf_prep_kernel)
  const c32* c = conv + (size_t)clip * h.M;
  // np.fft.rfftfreq(N, 1 / sr): k * (1 / (N * d)) with d = 1 / sr, all float64
  const double val = 1.0 / ((double)N * (1.0 / (double)h.sr));
  const float inv_m = 1.0f / (float)h.M;
  double tot = 0.0, high = 0.0;
  for (long long k = tid; k <= N / 2; k += 256) {
    const c32 w = chirp(k, N);
    const c32 z = c32{c[k][0] * inv_m, -c[k][1] * inv_m};   // conj(forward(conj(.))) / M = the inverse transform
    const c32 x = cmulf(z, w);
    const double pw = (double)cabs2_np(x[0], x[1]);
    tot += pw;
    if ((double)k * val > 4000.0) high += pw;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { tot += __shfl_xor(tot, o); high += __shfl_xor(high, o); }
  if ((tid & 63) == 0) { red[tid >> 6] = tot; red[4 + (tid >> 6)] = high; }
  __syncthreads();
  if (tid == 0) {
    const float t = (float)(red[0] + red[1] + red[2] + red[3]);
    const float hg = (float)(red[4] + red[5] + red[6] + red[7]);
    *out = hg / (t + 1e-10f);
  }
}

// Workspace bytes per clip for a transform of size M.
size_t hf_ratio_bytes_per_clip(long long M) { return (size_t)M * sizeof(c32) * 4; }

long long hf_ratio_fft_size(long long max_samples) {
  long long M = 16;
  while (M < 2 * max_samples - 1) M <<= 1;
  return M;
}

hipError_t launch_hf_ratio(const KArgs& a, int n_clips, long long M, void* workspace, hipStream_t s) {
  HfArgs h;
  h.pcm = a.pcm; h.offsets = a.offsets; h.pcm_type = a.pcm_type; h.sr = a.sr; h.M = M;
  h.a0 = (c32*)workspace; h.a1 = h.a0 + (size_t)n_clips * M;
  h.b0 = h.a1 + (size_t)n_clips * M; h.b1 = h.b0 + (size_t)n_clips * M;
  h.flags = a.ws_flags; h.metrics = a.metrics;
  const int bx = (int)std::min<long long>(4096, (M + 255) / 256);
  hipLaunchKernelGGL(hf_prep_kernel, dim3(bx, n_clips), dim3(256), 0, s, h);
  // a and b are adjacent ([a0 | a1 | b0 | b1]): transform both with one launch per pass by treating a0/b0 and a1/b1 as
  // batches of 2 * n_clips sequences with a stride of 2 * n_clips * M between the ping and the pong halves
  auto forward = [&](c32* x0, c32* x1, int batches, c32** result) {
    c32 *in = x0, *out = x1;
    long long Ns = 1;
    int lg = 0;
    while ((1LL << lg) < M) ++lg;
    int passes4 = lg / 2, passes2 = lg % 2;
    const int gx = (int)std::min<long long>(2048, (M / 4 + 255) / 256);
    for (int p = 0; p < passes4; ++p) {
      hipLaunchKernelGGL(hf_fft_pass_kernel<4>, dim3(gx, batches), dim3(256), 0, s, in, out, M, Ns);
      Ns *= 4;
      std::swap(in, out);
    }
    if (passes2) {
      hipLaunchKernelGGL(hf_fft_pass_kernel<2>, dim3(gx, batches), dim3(256), 0, s, in, out, M, Ns);
      std::swap(in, out);
    }
    *result = in;
  };
  c32 *fa = nullptr, *fb = nullptr, *conv = nullptr;
  forward(h.a0, h.a1, n_clips, &fa);
  forward(h.b0, h.b1, n_clips, &fb);
  c32* prod = (fa == h.a0) ? h.a1 : h.a0;   // the free half of the a pair
  const long long total = (long long)n_clips * M;
  hipLaunchKernelGGL(hf_multiply_kernel, dim3((int)std::min<long long>(65535, (total + 255) / 256)), dim3(256), 0, s, fa, fb, prod, total);
  forward(prod, (prod == h.a0) ? h.a1 : h.a0, n_clips, &conv);
  hipLaunchKernelGGL(hf_finish_kernel, dim3(n_clips), dim3(256), 0, s, h, conv);
  return hipGetLastError();
}

}  // namespace dysf
