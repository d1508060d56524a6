// dysf_resample.hip -- rational sample-rate conversion: the step librosa.load(path, sr=16000)
// performs on raw-rate audio before the feature path (pipeline1.py:100-106).  Polyphase FIR exactly as
// scipy.signal.resample_poly / upfirdn define it:
//   y[m] = sum_k h[m*down - k*up] * x[k],   out[n] = y[n + n_pre_remove],   len(out) = ceil(len(x) * up / down)
// with zero extension of x; h (front-padded, float64) comes from the host (rsd_amd/resample.py: design_taps).
// One thread per output sample: ~len(h)/up taps (28 for 22.05 -> 16 kHz), float64 accumulation, float32 result.
// HBM-bound by nature (2 or 4 bytes in, 4 bytes out per sample); the taps (70 KB) stay in L2.
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "../../include/dysf_hip.h"

namespace {

struct ResDev {
  int up, down, n_taps, n_pre_remove;
  const double* taps;
};

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

}  // namespace

struct DysfResampler {
  int device = 0;
  ResDev d{};
  void* buf = nullptr;
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
  *out = r;
  return 0;
}

void dysf_resampler_destroy(DysfResampler* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  if (r->buf) (void)hipFree(r->buf);
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
