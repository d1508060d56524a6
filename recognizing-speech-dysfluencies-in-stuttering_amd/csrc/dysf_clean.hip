// dysf_clean.hip -- the audio cleaning stage ahead of the feature path:
//     y_clean = nr.reduce_noise(y=y, sr=sr)        pipeline1.py:139  (prop_decrease=0.8 at main.py:655 / main1.py:604)
//     y_clean = librosa.util.normalize(y_clean)    pipeline1.py:140
//     sf.write(out_path, y_clean, sr)              pipeline1.py:141  (PCM_16)
// noisereduce's default gate (stationary=False, SpectralGateNonStationary) is glue over SciPy; every step below follows
// the SciPy call it makes, in float64 like the float64 chunk noisereduce builds (oracle/dysf_clean_oracle.py):
//   chunk   clips up to chunk_size (600 000 samples) are one chunk: the clip between 30 000 zeros (SpectralGate._read_chunk);
//           longer clips (the reference's 126.5 s sidebar uploads, main1.py:598-611) are cut into full 600 000-sample spans,
//           each padded with its 30 000 real neighbouring samples either side (zeros outside the clip) and gated on its own
//           (SpectralGate.get_traces / filter_chunk); then scipy.signal.stft's own 512 zeros
//   stft    hann(1024, periodic), hop 256, rfft / sum(window)                               clean_stft_kernel
//   smooth  |X| through filtfilt([b], [1, b - 1], padtype=None) along time: a one-pole low-pass run forward and
//           backward from steady-state initial conditions; mask = sigmoid((|X| - s) / s, -2, 10)   clean_gate_kernel
//   blur    fftconvolve(mask, outer(triangle_f, triangle_t) / sum, "same"): a zero-boundary separable FIR, X *= mask
//                                                                  clean_blur_freq_kernel, clean_blur_time_kernel
//   istft   irfft * sum(window), overlap-add of the windowed segments in ascending frame order, / sum(window^2),
//           crop the two paddings, cast to float32, clip peak                     clean_istft_kernel, clean_ola_kernel
//   write   / peak in float64 stored as float32 (librosa.util.normalize), * 32768 clipped to [-32768, 32767] and
//           rounded half-to-even (libsndfile f2s_clip_array as python-soundfile sets it up)  clean_quant_kernel
// One workgroup per frame for the two transforms (1024-point Stockham radix-4 in LDS, twiddles from a host table),
// one thread per bin for the recursive filter, one thread per cell / sample elsewhere.  This is a preprocessing stage
// (the reference runs it once per file and memoises the WAV), not the throughput path: correctness first.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "../../include/dysf_hip.h"

// SciPy's loops (lfilter, pocketfft) are plain multiply / add code on the reference's host: no fused contractions here
#pragma clang fp contract(off)

namespace {

constexpr int kN = 1024;          // n_fft = win_length (noisereduce default)
constexpr int kHop = 256;         // win_length // 4
constexpr int kBins = kN / 2 + 1;
constexpr int kMaxTaps = 129;     // per axis of the mask blur

struct CleanDev {
  const void* pcm; const long long* offsets; int pcm_type;
  int clip0;                 // first clip of this sub-batch
  int padding;               // samples either side of a chunk (30 000): zeros outside the clip
  long long chunk;           // noisereduce's chunk_size (600 000)
  int nch;                   // chunks of the longest clip: the workspaces hold [clip][nch] jobs
  int t_max;                 // frame stride of the workspaces
  long long max_samples;     // the caller's bound on a clip's length: longer clips are rejected (DYSF_TOO_LONG), not processed
  double b;                  // one-pole coefficient of the time smoothing
  double shift, slope;       // sigmoid(x, shift, slope): shift = -thresh_n_mult_nonstationary
  double prop;               // prop_decrease
  int nf, nt;                // blur taps along frequency / time (odd)
  double inv_tot;            // 1 / sum of the outer-product filter
  const double2* tw;         // [1024] exp(-2 pi i j / 1024)
  const double* win;         // [1024] hann, periodic
  const double* wf; const double* wt;   // triangles
  double2* spec;             // [clip][t_max][513]
  double* smooth;            // [clip][t_max][513] forward pass of the smoother, then the frequency pass of the blur
  double* mask;              // [clip][t_max][513]
  double* seg;               // [clip][t_max][1024] irfft segments * sum(window)
  float* y;                  // [total samples] denoised float32 (noisereduce's return value), indexed like the input
  unsigned* peak;            // [clip] bits of max |y| (non-negative floats order like their bits)
  int* status;               // [clip] DYSF_OK / DYSF_EMPTY / DYSF_NONFINITE
  float* out_norm; short* out_pcm16;
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return double2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

__device__ __forceinline__ int frames_of(long long n, int padding) { return (int)((n + 2LL * padding) / kHop) + 1; }

// One gating job = one chunk of one clip: the span [c0, c0 + L) of the clip inside `padding` samples either side.
struct Job { long long off, n, c0, L; int T; bool ok; };
__device__ __forceinline__ Job job_of(const CleanDev& c, int lc, int k) {
  Job j;
  const int clip = c.clip0 + lc;
  j.off = c.offsets[clip]; j.n = c.offsets[clip + 1] - j.off;
  j.ok = j.n > 0 && j.n <= c.max_samples;
  if (j.n <= c.chunk) { j.c0 = 0; j.L = j.n; j.ok = j.ok && k == 0; }
  else { j.c0 = (long long)k * c.chunk; j.L = c.chunk; j.ok = j.ok && j.c0 < j.n; }   // every chunk spans chunk_size, the last one too
  j.T = j.ok ? frames_of(j.L, c.padding) : 0;
  return j;
}

// 1024-point forward FFT of buf[0] (Stockham autosort, five radix-4 passes, thread j owns butterfly j of a pass);
// returns the buffer index that holds the result.  The index maps are those of dysf_hfratio.hip's global passes.
__device__ __forceinline__ int fft1024(double2 (*buf)[kN], const double2* tw, int tid) {
  int cur = 0;
#pragma unroll 1
  for (int Ns = 1; Ns < kN; Ns *= 4) {
    __syncthreads();
    const double2* in = buf[cur];
    double2* out = buf[cur ^ 1];
    const int k = tid % Ns;
    double2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = in[tid + r * (kN / 4)];
      if (r > 0) v[r] = cmul(v[r], tw[r * k * (kN / 4 / Ns)]);
    }
    const double2 t0 = {v[0].x + v[2].x, v[0].y + v[2].y}, t1 = {v[0].x - v[2].x, v[0].y - v[2].y};
    const double2 t2 = {v[1].x + v[3].x, v[1].y + v[3].y}, t3 = {v[1].x - v[3].x, v[1].y - v[3].y};
    const int base = (tid / Ns) * Ns * 4 + k;
    out[base] = double2{t0.x + t2.x, t0.y + t2.y};
    out[base + Ns] = double2{t1.x + t3.y, t1.y - t3.x};      // t1 - i t3
    out[base + 2 * Ns] = double2{t0.x - t2.x, t0.y - t2.y};
    out[base + 3 * Ns] = double2{t1.x - t3.y, t1.y + t3.x};  // t1 + i t3
    cur ^= 1;
  }
  __syncthreads();
  return cur;
}

__global__ __launch_bounds__(256) void clean_stft_kernel(CleanDev c) {
  __shared__ double2 buf[2][kN];
  const int job = blockIdx.y, lc = job / c.nch, t = blockIdx.x, tid = threadIdx.x;
  const Job jb = job_of(c, lc, job % c.nch);
  if (!jb.ok || t >= jb.T) return;
  const long long off = jb.off, n = jb.n;
  for (int j = tid; j < kN; j += 256) {
    // tap j sits at index u of the padded chunk; scipy.signal.stft's own boundary zeros lie outside [0, L + 2 padding) --
    // where a chunk of a long clip has real neighbours, so the test is on u, not only on the clip's ends
    const long long u = (long long)t * kHop + j - kN / 2;
    const long long s = jb.c0 - c.padding + u;   // sample of the clip under tap j
    double v = 0.0;
    if (u >= 0 && u < jb.L + 2LL * c.padding && s >= 0 && s < n)
      v = (c.pcm_type == DYSF_PCM_I16) ? (double)((const short*)c.pcm)[off + s] * (1.0 / 32768.0)
                                       : (double)((const float*)c.pcm)[off + s];
    buf[0][j] = double2{v * c.win[j], 0.0};
  }
  const int cur = fft1024(buf, c.tw, tid);
  double2* dst = c.spec + ((size_t)job * c.t_max + t) * kBins;
  const double scale = 1.0 / (kN / 2);   // 1 / sum(hann(1024, periodic)) = 1 / 512
  for (int f = tid; f < kBins; f += 256) dst[f] = double2{buf[cur][f].x * scale, buf[cur][f].y * scale};
}

// scipy.signal.filtfilt([b], [1, b - 1], x, padtype=None): y[n] = b x[n] + (1 - b) y[n-1] started from the steady state
// of x[0] (lfilter_zi * x[0], i.e. y[0] = x[0]), then the same filter over the reversed result.
__global__ __launch_bounds__(64) void clean_gate_kernel(CleanDev c) {
  const int job = blockIdx.y, f = blockIdx.x * 64 + threadIdx.x;
  const Job jb = job_of(c, job / c.nch, job % c.nch);
  if (!jb.ok || f >= kBins) return;
  const int T = jb.T;
  const double2* sp = c.spec + (size_t)job * c.t_max * kBins + f;
  double* sm = c.smooth + (size_t)job * c.t_max * kBins + f;
  double* mk = c.mask + (size_t)job * c.t_max * kBins + f;
  const double b = c.b, a = 1.0 - b;
  double y = 0.0;
  for (int t = 0; t < T; ++t) {
    const double2 z = sp[(size_t)t * kBins];
    const double mag = hypot(z.x, z.y);
    y = (t == 0) ? mag : b * mag + a * y;
    sm[(size_t)t * kBins] = y;
  }
  for (int t = T - 1; t >= 0; --t) {
    const double yf = sm[(size_t)t * kBins];
    y = (t == T - 1) ? yf : b * yf + a * y;
    const double2 z = sp[(size_t)t * kBins];
    const double mag = hypot(z.x, z.y);
    const double above = (mag - y) / y;
    mk[(size_t)t * kBins] = 1.0 / (1.0 + exp(-(above + c.shift) * c.slope));
  }
}

// Zero-boundary "same" convolution of the mask with the triangle filter, which is an outer product: a pass along
// frequency into the (now free) smoother buffer, a pass along time, then X *= mask * prop + (1 - prop).
__global__ __launch_bounds__(256) void clean_blur_freq_kernel(CleanDev c) {
  const int job = blockIdx.z, t = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  const Job jb = job_of(c, job / c.nch, job % c.nch);
  if (!jb.ok || f >= kBins || t >= jb.T) return;
  const double* mk = c.mask + ((size_t)job * c.t_max + t) * kBins;
  const int hf = c.nf / 2;
  double row = 0.0;
  for (int df = -hf; df <= hf; ++df) {
    const int ff = f + df;
    if (ff >= 0 && ff < kBins) row += mk[ff] * c.wf[hf - df];
  }
  c.smooth[((size_t)job * c.t_max + t) * kBins + f] = row;
}

__global__ __launch_bounds__(256) void clean_blur_time_kernel(CleanDev c) {
  const int job = blockIdx.z, t = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
  const Job jb = job_of(c, job / c.nch, job % c.nch);
  if (!jb.ok || f >= kBins) return;
  const int T = jb.T;
  if (t >= T) return;
  const double* rows = c.smooth + (size_t)job * c.t_max * kBins + f;
  const int ht = c.nt / 2;
  double acc = 0.0;
  for (int dt = -ht; dt <= ht; ++dt) {
    const int tt = t + dt;
    if (tt >= 0 && tt < T) acc += rows[(size_t)tt * kBins] * c.wt[ht - dt];
  }
  const double m = (acc * c.inv_tot) * c.prop + (1.0 - c.prop);
  double2* z = c.spec + ((size_t)job * c.t_max + t) * kBins + f;
  *z = double2{z->x * m, z->y * m};
}

// irfft of one masked frame: the Hermitian extension (imaginary parts of bins 0 and 512 ignored, as pocketfft's c2r
// does), inverse = conj(forward(conj(.))) / 1024, times sum(window) = 512.
__global__ __launch_bounds__(256) void clean_istft_kernel(CleanDev c) {
  __shared__ double2 buf[2][kN];
  const int job = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const Job jb = job_of(c, job / c.nch, job % c.nch);
  if (!jb.ok || t >= jb.T) return;
  const double2* src = c.spec + ((size_t)job * c.t_max + t) * kBins;
  for (int f = tid; f < kBins; f += 256) {
    double2 z = src[f];
    if (f == 0 || f == kN / 2) z.y = 0.0;
    buf[0][f] = double2{z.x, -z.y};                       // conj(Z[f])
    if (f > 0 && f < kN / 2) buf[0][kN - f] = double2{z.x, z.y};   // conj(Z[1024 - f]) = Z[f]
  }
  const int cur = fft1024(buf, c.tw, tid);
  double* dst = c.seg + ((size_t)job * c.t_max + t) * kN;
  for (int j = tid; j < kN; j += 256) dst[j] = buf[cur][j].x * 0.5;   // / 1024 * 512, both exact
}

// scipy.signal.istft's overlap-add for one output sample: segments in ascending frame order, normalised by the summed
// squared window where that exceeds 1e-10; positions past the last whole hop of the padded chunk stay zero.
__global__ __launch_bounds__(256) void clean_ola_kernel(CleanDev c) {
  const int lc = blockIdx.y, clip = c.clip0 + lc;
  const long long off = c.offsets[clip], n = c.offsets[clip + 1] - off;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || n > c.max_samples) return;
  const int k = n <= c.chunk ? 0 : (int)(i / c.chunk);   // the chunk this sample is cropped from
  const Job jb = job_of(c, lc, k);
  const int T = jb.T;
  const long long q = i - jb.c0 + c.padding;  // index in the padded chunk
  float out = 0.f;
  if (q < (long long)(T - 1) * kHop) {
    const long long p = q + kN / 2;           // index in stft's zero-extended signal
    const int t_hi = (int)std::min<long long>(p / kHop, T - 1);
    const int t_lo = (int)std::max<long long>(0, (p - (kN - 1) + kHop - 1) / kHop);
    const double* seg = c.seg + ((size_t)lc * c.nch + k) * c.t_max * kN;
    double x = 0.0, norm = 0.0;
    for (int t = t_lo; t <= t_hi; ++t) {
      const int j = (int)(p - (long long)t * kHop);
      const double w = c.win[j];
      x += seg[(size_t)t * kN + j] * w;
      norm += w * w;
    }
    out = (float)(x / (norm > 1e-10 ? norm : 1.0));
  }
  c.y[off + i] = out;
  if (!isfinite(out)) c.status[clip] = DYSF_NONFINITE;
  else atomicMax(c.peak + clip, __float_as_uint(fabsf(out)));
}

// librosa.util.normalize (float64 peak, float64 quotient stored as float32; a peak below float32's tiny divides by 1)
// and libsndfile's clipping float -> PCM_16 conversion.
__global__ __launch_bounds__(256) void clean_quant_kernel(CleanDev c) {
  const int clip = c.clip0 + blockIdx.y;
  const long long off = c.offsets[clip], n = c.offsets[clip + 1] - off;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || n > c.max_samples) return;
  const bool bad = c.status[clip] != DYSF_OK;   // librosa raises on non-finite input: the reference writes no file
  double peak = (double)__uint_as_float(c.peak[clip]);
  if (peak < 1.1754943508222875e-38) peak = 1.0;
  const float v = bad ? 0.f : (float)((double)c.y[off + i] / peak);
  if (c.out_norm) c.out_norm[off + i] = v;
  if (c.out_pcm16) {
    const float s = v * 32768.0f;
    c.out_pcm16[off + i] = (s >= 32767.0f) ? (short)32767 : (s <= -32768.0f) ? (short)-32768 : (short)__float2int_rn(s);
  }
}

__global__ void clean_init_kernel(CleanDev c, int n_sub) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_sub) return;
  const int clip = c.clip0 + i;
  c.peak[clip] = 0u;
  const long long n = c.offsets[clip + 1] - c.offsets[clip];
  c.status[clip] = n <= 0 ? DYSF_EMPTY : (n > c.max_samples ? DYSF_TOO_LONG : DYSF_OK);
}

struct Buf {
  void* p = nullptr; size_t cap = 0;
  bool reserve(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return false; }
    cap = bytes;
    return true;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct DysfCleaner {
  int device = 0, sr = 16000, padding = 30000;
  long long chunk = 600000;   // noisereduce's chunk_size default (the reference passes none)
  double b = 0, prop = 1, shift = -2, slope = 10, inv_tot = 1;
  int nf = 1, nt = 1;
  Buf tables, spec, smooth, mask, seg, y, small;
  const double2* d_tw = nullptr; const double *d_win = nullptr, *d_wf = nullptr, *d_wt = nullptr;
  size_t ws_budget = (size_t)6 << 30;   // bytes of spectra kept in flight; larger batches run as sub-batches
};

static thread_local std::string g_clerr;

extern "C" {

const char* dysf_clean_last_error(void) { return g_clerr.c_str(); }

int32_t dysf_clean_create(int32_t device, int32_t sr, double prop_decrease, double time_constant_s,
                          double freq_mask_smooth_hz, double time_mask_smooth_ms, double thresh_n_mult,
                          double sigmoid_slope, int32_t padding, DysfCleaner** out) {
  if (!out) { g_clerr = "null argument"; return DYSF_E_INVALID; }
  *out = nullptr;
  if (sr < 1 || padding < 0 || time_constant_s <= 0 || !(prop_decrease >= 0.0 && prop_decrease <= 1.0)) {
    g_clerr = "bad cleaner parameters"; return DYSF_E_INVALID;
  }
  // SpectralGate._generate_mask_smoothing_filter: int() truncation, at least one bin / frame
  const int ngf = (int)(freq_mask_smooth_hz / ((double)sr / (kN / 2)));
  const int ngt = (int)(time_mask_smooth_ms / (((double)kHop / sr) * 1000));
  if (ngf < 1 || ngt < 1 || 2 * ngf + 1 > kMaxTaps || 2 * ngt + 1 > kMaxTaps) {
    g_clerr = "mask smoothing width outside 1.." + std::to_string(kMaxTaps / 2) + " bins / frames"; return DYSF_E_INVALID;
  }
  DysfCleaner* c = new (std::nothrow) DysfCleaner();
  if (!c) { g_clerr = "out of host memory"; return DYSF_E_NOMEM; }
  c->device = device; c->sr = sr; c->padding = padding; c->prop = prop_decrease;
  c->shift = -thresh_n_mult; c->slope = sigmoid_slope;
  const double t_frames = time_constant_s * sr / (double)kHop;   // get_time_smoothed_representation
  c->b = (std::sqrt(1 + 4 * t_frames * t_frames) - 1) / (2 * t_frames * t_frames);
  // triangles: concatenate(linspace(0, 1, n + 1, endpoint=False), linspace(1, 0, n + 2))[1:-1]  (2n + 1 taps)
  auto triangle = [](int n) {
    std::vector<double> v;
    for (int i = 1; i <= n; ++i) v.push_back(i * (1.0 / (n + 1)));               // 0 + i * step, step = 1 / (n + 1)
    for (int i = 0; i <= n; ++i) v.push_back(i == n + 1 ? 0.0 : 1.0 + i * (-1.0 / (n + 1)));   // 1 + i * step, step = -1 / (n + 1)
    return v;
  };
  // (n_grad_freq == 1 and n_grad_time == 1 switches the smoothing off in noisereduce: a single unit tap here)
  const bool smooth = !(ngf == 1 && ngt == 1);
  const std::vector<double> wf = smooth ? triangle(ngf) : std::vector<double>{1.0}, wt = smooth ? triangle(ngt) : std::vector<double>{1.0};
  c->nf = (int)wf.size(); c->nt = (int)wt.size();
  double tot = 0.0;
  for (double a : wf) for (double b2 : wt) tot += a * b2;
  c->inv_tot = 1.0 / tot;
  std::vector<double> host(2 * kN + kN + wf.size() + wt.size());
  for (int j = 0; j < kN; ++j) {
    const long double ang = -2.0L * 3.14159265358979323846264338327950288L * j / kN;
    host[2 * j] = (double)cosl(ang); host[2 * j + 1] = (double)sinl(ang);
  }
  {  // scipy.signal.get_window("hann", 1024): general_cosine over linspace(-pi, pi, 1025)[:-1]
    const double step = (M_PI - (-M_PI)) / kN;
    for (int j = 0; j < kN; ++j) host[2 * kN + j] = 0.5 + 0.5 * std::cos(j * step + (-M_PI));
  }
  std::copy(wf.begin(), wf.end(), host.begin() + 3 * kN);
  std::copy(wt.begin(), wt.end(), host.begin() + 3 * kN + wf.size());
  if (hipSetDevice(device) != hipSuccess) { delete c; g_clerr = "hipSetDevice failed"; return DYSF_E_HIP; }
  if (!c->tables.reserve(host.size() * sizeof(double)) ||
      hipMemcpy(c->tables.p, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    c->tables.release(); delete c; g_clerr = "device upload failed"; return DYSF_E_HIP;
  }
  const double* base = (const double*)c->tables.p;
  c->d_tw = (const double2*)base; c->d_win = base + 2 * kN; c->d_wf = base + 3 * kN; c->d_wt = base + 3 * kN + wf.size();
  *out = c;
  return 0;
}

void dysf_clean_destroy(DysfCleaner* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  c->tables.release(); c->spec.release(); c->smooth.release(); c->mask.release(); c->seg.release(); c->y.release(); c->small.release();
  delete c;
}

/* pipeline1.py:139-141 for a batch of clips resident in device memory.  Any of the three outputs may be null:
   d_denoised (float32, nr.reduce_noise's return value), d_normalized (float32, after librosa.util.normalize),
   d_pcm16 (int16, the samples sf.write stores); all are indexed like the input (d_offsets).  d_status[n_clips]:
   DYSF_OK, DYSF_EMPTY (no samples: outputs untouched), DYSF_TOO_LONG (longer than max_samples: outputs untouched) or DYSF_NONFINITE (the gate produced a non-finite sample --
   librosa.util.normalize raises there and the reference writes no file; the normalised / PCM outputs are zero). */
int32_t dysf_clean_batch_dev(DysfCleaner* c, const void* d_pcm, int32_t pcm_type, const int64_t* d_offsets, int32_t n_clips,
                             int64_t max_samples, int64_t total_samples, float* d_denoised, float* d_normalized,
                             int16_t* d_pcm16, int32_t* d_status, void* stream) {
  if (!c || !d_pcm || !d_offsets || !d_status || n_clips < 0 || max_samples < 0 || total_samples < 0) { g_clerr = "bad argument"; return DYSF_E_INVALID; }
  if (pcm_type != DYSF_PCM_I16 && pcm_type != DYSF_PCM_F32) { g_clerr = "bad pcm_type"; return DYSF_E_INVALID; }
  if (n_clips == 0) return 0;
  if (max_samples > 0x7ff00000LL) { g_clerr = "clip too long"; return DYSF_E_INVALID; }
  if (hipSetDevice(c->device) != hipSuccess) { g_clerr = "hipSetDevice failed"; return DYSF_E_HIP; }
  hipStream_t s = (hipStream_t)stream;
  const long long chunk = c->chunk;
  const int nch = max_samples <= chunk ? 1 : (int)((max_samples - 1) / chunk) + 1;   // gating jobs per clip
  const int t_max = (int)((std::min<long long>(max_samples, chunk) + 2LL * c->padding) / kHop) + 1;
  const size_t per_clip = (size_t)nch * t_max * (kBins * (sizeof(double2) + 2 * sizeof(double)) + kN * sizeof(double));
  const int sub = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_clips, std::min<size_t>(c->ws_budget / per_clip, (size_t)(65535 / nch))));
  const size_t jobs = (size_t)sub * nch;
  if (nch > 65535) { g_clerr = "clip too long"; return DYSF_E_INVALID; }
  if (!c->spec.reserve(jobs * t_max * kBins * sizeof(double2)) || !c->smooth.reserve(jobs * t_max * kBins * sizeof(double)) ||
      !c->mask.reserve(jobs * t_max * kBins * sizeof(double)) || !c->seg.reserve(jobs * t_max * kN * sizeof(double)) ||
      !c->small.reserve((size_t)n_clips * sizeof(unsigned)) || (!d_denoised && !c->y.reserve((size_t)std::max<int64_t>(total_samples, 1) * sizeof(float)))) {
    g_clerr = "out of device memory for the cleaning workspace"; return DYSF_E_NOMEM;
  }
  CleanDev d{};
  d.pcm = d_pcm; d.offsets = (const long long*)d_offsets; d.pcm_type = pcm_type; d.padding = c->padding; d.chunk = chunk; d.nch = nch; d.t_max = t_max; d.max_samples = max_samples;
  d.b = c->b; d.shift = c->shift; d.slope = c->slope; d.prop = c->prop; d.nf = c->nf; d.nt = c->nt; d.inv_tot = c->inv_tot;
  d.tw = c->d_tw; d.win = c->d_win; d.wf = c->d_wf; d.wt = c->d_wt;
  d.spec = (double2*)c->spec.p; d.smooth = (double*)c->smooth.p; d.mask = (double*)c->mask.p; d.seg = (double*)c->seg.p;
  d.y = d_denoised ? d_denoised : (float*)c->y.p; d.peak = (unsigned*)c->small.p; d.status = d_status;
  d.out_norm = d_normalized; d.out_pcm16 = d_pcm16;
  const unsigned sample_blocks = (unsigned)((max_samples + 255) / 256);
  for (int c0 = 0; c0 < n_clips; c0 += sub) {
    const int m = std::min(sub, n_clips - c0);
    d.clip0 = c0;
    hipLaunchKernelGGL(clean_init_kernel, dim3((m + 255) / 256), dim3(256), 0, s, d, m);
    hipLaunchKernelGGL(clean_stft_kernel, dim3(t_max, m * nch), dim3(256), 0, s, d);
    hipLaunchKernelGGL(clean_gate_kernel, dim3((kBins + 63) / 64, m * nch), dim3(64), 0, s, d);
    hipLaunchKernelGGL(clean_blur_freq_kernel, dim3((kBins + 255) / 256, t_max, m * nch), dim3(256), 0, s, d);
    hipLaunchKernelGGL(clean_blur_time_kernel, dim3((kBins + 255) / 256, t_max, m * nch), dim3(256), 0, s, d);
    hipLaunchKernelGGL(clean_istft_kernel, dim3(t_max, m * nch), dim3(256), 0, s, d);
    if (sample_blocks) {
      hipLaunchKernelGGL(clean_ola_kernel, dim3(sample_blocks, m), dim3(256), 0, s, d);
      if (d_normalized || d_pcm16) hipLaunchKernelGGL(clean_quant_kernel, dim3(sample_blocks, m), dim3(256), 0, s, d);
    }
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_clerr = hipGetErrorString(e); return DYSF_E_HIP; }
  return 0;
}

}  // extern "C"
