// dysf_resample.hip -- rational sample-rate conversion: the step librosa.load(path, sr=16000)
// performs on raw-rate audio before the feature path (pipeline1.py:100-106).  Polyphase FIR exactly as
// scipy.signal.resample_poly / upfirdn define it:
//   y[m] = sum_k h[m*down - k*up] * x[k],   out[n] = y[n + n_pre_remove],   len(out) = ceil(len(x) * up / down)
// with zero extension of x; h (front-padded, float64) comes from the host (rsd_amd/resample.py: design_taps).
// ~len(h)/up taps per output (28 for 22.05 -> 16 kHz), float64 accumulation, float32 result.  Two kernels: a general
// one-thread-per-output form that reads the taps from L2, and (up >= 64, <= 32 taps per phase) a phase-resident form
// whose taps live in registers and whose input span lives in LDS.
#include <hip/hip_runtime.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/dysf_hip.h"

namespace {

struct ResDev {
  int up, down, n_taps, n_pre_remove;
  const double* taps;
  int ntap;              // taps per phase = ceil(n_taps / up)
  const double* phase;   // [ntap][up] taps in thread order: phase[t][c] = taps[ph(c) + t*up], ph(c) = ((c + n_pre_remove)*down) % up
};

constexpr int kRepeats = 16;     // outputs per thread of the phase-resident kernel
constexpr int kMaxPhaseTaps = 32;

template <int PCM>
__global__ __launch_bounds__(256) void dysf_resample_kernel(ResDev r, const void* pcm, const long long* in_off,
                                                            float* out, const long long* out_off) {
  const int clip = blockIdx.y;
  const long long i0 = in_off[clip], n_in = in_off[clip + 1] - i0;
  const long long o0 = out_off[clip], n_out = out_off[clip + 1] - o0;
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_out) return;
  const long long pos = (n + r.n_pre_remove) * (long long)r.down;  // position of this output on the up-sampled grid
  long long k = pos / r.up;
  if (k > n_in - 1) k = n_in - 1;
  long long idx = pos - k * r.up;                                  // tap that meets x[k]
  double acc = 0.0;
  for (; k >= 0 && idx < r.n_taps; --k, idx += r.up) {
    const double x = (PCM == DYSF_PCM_I16) ? (double)((const short*)pcm)[i0 + k] * (1.0 / 32768.0)
                                           : (double)((const float*)pcm)[i0 + k];
    acc = fma(r.taps[idx], x, acc);
  }
  out[o0 + n] = (float)acc;
}

// Phase-resident form for up >= 64: thread c of a workgroup owns the outputs n = base + c + up*r, r < kRepeats, which
// all have the same filter phase, so its ntap (<= 32) float64 taps sit in registers; the input span of the chunk is
// converted once into LDS as float64 (zero extension included, so the inner loop has no bounds checks and no
// conversions): one 8-byte LDS read + one float64 FMA per tap.
template <int PCM>
__global__ __launch_bounds__(512) void dysf_resample_phase_kernel(ResDev r, const void* pcm, const long long* in_off,
                                                                  float* out, const long long* out_off) {
  extern __shared__ double xs[];  // the chunk's input span as float64 (converted once, not once per tap)
  const int clip = blockIdx.y;
  const long long i0 = in_off[clip], n_in = in_off[clip + 1] - i0;
  const long long o0 = out_off[clip], n_out = out_off[clip + 1] - o0;
  const long long base = (long long)blockIdx.x * r.up * kRepeats;
  if (base >= n_out) return;
  const long long k_min = ((base + r.n_pre_remove) * (long long)r.down) / r.up - (r.ntap - 1);
  const int span = r.down * kRepeats + r.ntap + 1;
  if (PCM == DYSF_PCM_I16) {
    // 16-byte groups of eight samples, aligned in global memory; the group that holds x[k_min] starts the span
    const short* src = (const short*)pcm + i0;
    const long long g0 = (i0 + k_min) & ~7LL;           // absolute sample index of the first group (floor, also if negative)
    const int lead = (int)(i0 + k_min - g0);            // samples of that group ahead of the span
    const int ngroups = (span + lead + 7) / 8;
    for (int g = threadIdx.x; g < ngroups; g += blockDim.x) {
      const long long s0 = g0 + 8LL * g - i0;           // clip-relative index of the group's first sample
      short v[8];
      if (s0 >= 0 && s0 + 8 <= n_in) {
        *(int4*)v = *(const int4*)(src + s0);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (s0 + j >= 0 && s0 + j < n_in) ? src[s0 + j] : (short)0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = 8 * g + j - lead;
        if (i >= 0 && i < span) xs[i] = (double)v[j] * (1.0 / 32768.0);
      }
    }
  } else {
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
      const long long s = k_min + i;
      xs[i] = (s >= 0 && s < n_in) ? (double)((const float*)pcm)[i0 + s] : 0.0;
    }
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= r.up) return;
  const long long pos = (base + c + r.n_pre_remove) * (long long)r.down;
  const long long k0 = pos / r.up;
  double tap[kMaxPhaseTaps];
#pragma unroll
  for (int t = 0; t < kMaxPhaseTaps; ++t) tap[t] = (t < r.ntap) ? r.phase[(size_t)t * r.up + c] : 0.0;  // coalesced
  int x0 = (int)(k0 - k_min);  // LDS index of x[k0]; advances by `down` per repeat (up*down/up)
  for (int rep = 0; rep < kRepeats; ++rep, x0 += r.down) {
    const long long n = base + c + (long long)r.up * rep;
    if (n >= n_out) break;
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < kMaxPhaseTaps; ++t)
      if (t < r.ntap) acc = fma(tap[t], xs[x0 - t], acc);
    out[o0 + n] = (float)acc;
  }
}

}  // namespace

struct DysfResampler {
  int device = 0;
  ResDev d{};
  void* buf = nullptr;
  void* buf_phase = nullptr;
  bool phase_form = false;  // use dysf_resample_phase_kernel
};

static thread_local std::string g_rerr;

extern "C" {

const char* dysf_resampler_last_error(void) { return g_rerr.c_str(); }

int32_t dysf_resampler_create(int32_t device, int32_t up, int32_t down, const double* taps, int32_t n_taps,
                              int32_t n_pre_remove, DysfResampler** out) {
  if (!out || !taps) { g_rerr = "null argument"; return DYSF_E_INVALID; }
  *out = nullptr;
  if (up < 1 || down < 1 || n_taps < 1 || n_pre_remove < 0) { g_rerr = "bad resampler shape"; return DYSF_E_INVALID; }
  DysfResampler* r = new (std::nothrow) DysfResampler();
  if (!r) { g_rerr = "out of host memory"; return DYSF_E_NOMEM; }
  r->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete r; g_rerr = "hipSetDevice failed"; return DYSF_E_HIP; }
  if (hipMalloc(&r->buf, (size_t)n_taps * sizeof(double)) != hipSuccess ||
      hipMemcpy(r->buf, taps, (size_t)n_taps * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    if (r->buf) (void)hipFree(r->buf);
    delete r; g_rerr = "device upload failed"; return DYSF_E_HIP;
  }
  r->d.up = up; r->d.down = down; r->d.n_taps = n_taps; r->d.n_pre_remove = n_pre_remove; r->d.taps = (const double*)r->buf;
  r->d.ntap = (n_taps + up - 1) / up;
  const size_t lds = ((size_t)down * kRepeats + r->d.ntap + 1) * sizeof(double);
  if (up >= 64 && up <= 512 && r->d.ntap <= kMaxPhaseTaps && lds <= 64 * 1024) {
    std::vector<double> ph((size_t)up * r->d.ntap, 0.0);
    for (int c = 0; c < up; ++c) {  // a chunk starts at a multiple of up, so thread c always has this phase
      const int p = (int)(((long long)(c + n_pre_remove) * down) % up);
      for (int t = 0; t < r->d.ntap; ++t)
        if (p + (long long)t * up < n_taps) ph[(size_t)t * up + c] = taps[p + (size_t)t * up];
    }
    if (hipMalloc(&r->buf_phase, ph.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(r->buf_phase, ph.data(), ph.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      if (r->buf_phase) (void)hipFree(r->buf_phase);
      (void)hipFree(r->buf); delete r; g_rerr = "device upload failed"; return DYSF_E_HIP;
    }
    r->d.phase = (const double*)r->buf_phase;
    r->phase_form = true;
  }
  *out = r;
  return 0;
}

void dysf_resampler_destroy(DysfResampler* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  if (r->buf) (void)hipFree(r->buf);
  if (r->buf_phase) (void)hipFree(r->buf_phase);
  delete r;
}

int64_t dysf_resampler_out_len(const DysfResampler* r, int64_t n_in) {
  if (!r || n_in < 0) return -1;
  const int64_t t = n_in * r->d.up;
  return t / r->d.down + (t % r->d.down ? 1 : 0);
}

int32_t dysf_resample_batch_dev(DysfResampler* r, const void* d_pcm, int32_t pcm_type, const int64_t* d_in_offsets,
                                int32_t n_clips, float* d_out, const int64_t* d_out_offsets, int64_t max_out_len,
                                void* stream) {
  if (!r || !d_pcm || !d_in_offsets || !d_out || !d_out_offsets || n_clips < 0 || max_out_len < 0) { g_rerr = "bad argument"; return DYSF_E_INVALID; }
  if (pcm_type != DYSF_PCM_I16 && pcm_type != DYSF_PCM_F32) { g_rerr = "bad pcm_type"; return DYSF_E_INVALID; }
  if (n_clips == 0 || max_out_len == 0) return 0;
  if (n_clips > 65535) { g_rerr = "at most 65535 clips per call"; return DYSF_E_INVALID; }
  if (hipSetDevice(r->device) != hipSuccess) { g_rerr = "hipSetDevice failed"; return DYSF_E_HIP; }
  if (r->phase_form) {
    const long long chunk = (long long)r->d.up * kRepeats;
    const dim3 pgrid((unsigned)((max_out_len + chunk - 1) / chunk), (unsigned)n_clips);
    const dim3 pblock((unsigned)((r->d.up + 63) / 64 * 64));
    const size_t lds = ((size_t)r->d.down * kRepeats + r->d.ntap + 1) * sizeof(double);
    if (pcm_type == DYSF_PCM_I16)
      hipLaunchKernelGGL(dysf_resample_phase_kernel<DYSF_PCM_I16>, pgrid, pblock, lds, (hipStream_t)stream, r->d, d_pcm,
                         (const long long*)d_in_offsets, d_out, (const long long*)d_out_offsets);
    else
      hipLaunchKernelGGL(dysf_resample_phase_kernel<DYSF_PCM_F32>, pgrid, pblock, lds, (hipStream_t)stream, r->d, d_pcm,
                         (const long long*)d_in_offsets, d_out, (const long long*)d_out_offsets);
    const hipError_t pe = hipGetLastError();
    if (pe != hipSuccess) { g_rerr = hipGetErrorString(pe); return DYSF_E_HIP; }
    return 0;
  }
  const dim3 grid((unsigned)((max_out_len + 255) / 256), (unsigned)n_clips);
  if (pcm_type == DYSF_PCM_I16)
    hipLaunchKernelGGL(dysf_resample_kernel<DYSF_PCM_I16>, grid, dim3(256), 0, (hipStream_t)stream, r->d, d_pcm,
                       (const long long*)d_in_offsets, d_out, (const long long*)d_out_offsets);
  else
    hipLaunchKernelGGL(dysf_resample_kernel<DYSF_PCM_F32>, grid, dim3(256), 0, (hipStream_t)stream, r->d, d_pcm,
                       (const long long*)d_in_offsets, d_out, (const long long*)d_out_offsets);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_rerr = hipGetErrorString(e); return DYSF_E_HIP; }
  return 0;
}

}  // extern "C"
