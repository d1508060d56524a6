// dysf_capi.hip -- the extern "C" boundary declared in include/dysf_hip.h.
// Owns tables (host + device copies), the per-workgroup global workspace and the LDS carve-up; the
// hot entry point dysf_extract_batch_dev only enqueues one kernel on the caller's stream.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "dysf_kernels.h"
#include "dysf_tables.hpp"

using namespace dysf;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
static int fail_hip(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return DYSF_E_HIP;
}
#define HIPCHK(expr)                                        \
  do {                                                      \
    hipError_t e_ = (expr);                                 \
    if (e_ != hipSuccess) return fail_hip(e_, #expr);       \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { p = nullptr; return fail_hip(e, "hipMalloc"); }
    cap = bytes;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct DysfHandle {
  HostTables t;
  int device = -1;
  int num_cu = 256;
  size_t lds_limit = 163840;
  // device tables
  float* d_window_i16 = nullptr;
  float *d_window = nullptr, *d_stage_tw = nullptr, *d_post_tw = nullptr,
        *d_chroma_pack = nullptr;
  double *d_window64 = nullptr, *d_window64_i16 = nullptr, *d_stage_tw64 = nullptr, *d_post_tw64 = nullptr;  // parity presets
  double* d_dct_pack = nullptr;
  float* d_dct_pack32 = nullptr;
  double* d_dct_rowsum = nullptr;
  uint16_t* d_dct_pack16 = nullptr;     // float16 hi / lo split DCT operands (kernel B on the float16 MFMA)
  double* d_dct_rowsum16 = nullptr;
  bool dct16 = true;        // DYSF_DCT16=0: keep the DCT of the throughput presets on the float32 MFMA
  int32_t* d_rich_tab = nullptr;
  bool dct64 = false;       // DYSF_DCT64=1: run the DCT on the float64 MFMA instead of the mean-free float32 form
  // workspace / staging
  DevBuf ws_rich, ws_S, ws_stat, ws_flat, ws_energy, ws_hf, ws_rowmax;
  DevBuf ws_logmel, ws_peak_mag, ws_peak_bin, ws_small;
  DevBuf ws_cand, ws_framecnt;   // kernel A-exact -> kernel C: per-frame lists of piptrack's local maxima
  size_t ws_budget = (size_t)4 << 30;  // bytes of per-clip workspace before a batch is split (DYSF_WS_MB overrides)
  bool sched_built = false;
  int64_t last_sub = 0, last_peak_stride = 0;  // geometry of the most recent launch (dysf_debug_peaks)
  bool timing = false;           // dysf_debug_timing: HIP events around each kernel of the last sub-batch
  hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long* dbg_stamps = nullptr;  // DYSF_STAMP diagnostic builds
  DevBuf st_pcm, st_off, st_pooled, st_frames, st_status, st_tuning;
  bool ws_clean = false;         // ws_small is known to be all zero once the work queued on ws_stream has run
  hipStream_t ws_stream = nullptr;
  DevBuf d_mel_sched, d_mel_spack;  // kernel A's per-wave mel schedule and the taps laid out in its order
  int mel_slots = 0;
  // kernel A2 (wave-private STFT + mel): block schedule of the 4x4x1 mel product
  DevBuf d_mel4_pack, d_mel4_kbeg;
  int mel4_steps = 0, mel4_rounds = 0;
  int exact_variant = 0; // DYSF_EXACT: 1 keeps kernel A for the parity presets at n_fft 2048 (A/B checks), else kernel A-exact
  int stft_variant = 0;  // DYSF_STFT: 0 choose (A2 where it applies), 1 always kernel A, 2 same as 0
};

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
  if (src.empty()) { *dst = nullptr; return 0; }
  HIPCHK(hipMalloc((void**)dst, src.size() * sizeof(T)));
  HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

extern "C" {

int32_t dysf_abi_version(void) { return DYSF_ABI_VERSION; }
const char* dysf_last_error(void) { return g_err.c_str(); }

int32_t dysf_params_preset(int32_t preset, DysfParams* p) {
  if (!p) return fail(DYSF_E_INVALID, "null params");
  std::memset(p, 0, sizeof(*p));
  p->sr = 16000; p->fmin = 0.f; p->fmax = 0.f; p->top_db = 80.f; p->amin = 1e-10f; p->delta_width = 9;
  if (preset == DYSF_PRESET_REFERENCE_149) {
    p->n_fft = 2048; p->hop = 512; p->n_mels = 128; p->n_mfcc = 20; p->chroma = 1; p->parity = 1;
  } else if (preset == DYSF_PRESET_BENCH_512) {
    p->n_fft = 512; p->hop = 160; p->n_mels = 64; p->n_mfcc = 40; p->chroma = 0;
  } else if (preset == DYSF_PRESET_MAIN_286) {
    p->n_fft = 512; p->hop = 256; p->n_mels = 128; p->n_mfcc = 40; p->chroma = 1; p->rich = 1; p->parity = 1;
  } else {
    return fail(DYSF_E_INVALID, "unknown preset");
  }
  return 0;
}

int32_t dysf_create(const DysfParams* p, int32_t device, DysfHandle** out) {
  if (!p || !out) return fail(DYSF_E_INVALID, "null argument");
  *out = nullptr;
  DysfHandle* h = new (std::nothrow) DysfHandle();
  if (!h) return fail(DYSF_E_NOMEM, "out of host memory");
  const char* err = "";
  if (!build_tables(*p, h->t, &err)) { delete h; return fail(DYSF_E_INVALID, err); }
  h->device = device;
  if (const char* e = std::getenv("DYSF_DCT64")) h->dct64 = std::atoi(e) != 0;
  if (const char* e = std::getenv("DYSF_DCT16")) h->dct16 = std::atoi(e) != 0;
  if (const char* e = std::getenv("DYSF_WS_MB")) { if (std::atoll(e) > 0) h->ws_budget = (size_t)std::atoll(e) << 20; }
  if (const char* e = std::getenv("DYSF_STFT")) h->stft_variant = std::atoi(e);
  if (const char* e = std::getenv("DYSF_EXACT")) h->exact_variant = std::atoi(e);
  if (device >= 0) {
    int rc = 0;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete h; return fail_hip(e, "hipSetDevice"); }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) h->num_cu = v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0)
      h->lds_limit = (size_t)v;
    rc |= upload(&h->d_window, h->t.window_half);
    {
      std::vector<float> w16(h->t.window_half);
      for (float& w : w16) w *= 1.0f / 32768.0f;  // exact
      rc |= upload(&h->d_window_i16, w16);
    }
    if (p->parity) {
      rc |= upload(&h->d_window64, h->t.window64_half);
      std::vector<double> w16(h->t.window64_half);
      for (double& w : w16) w *= 1.0 / 32768.0;  // exact
      rc |= upload(&h->d_window64_i16, w16);
      rc |= upload(&h->d_stage_tw64, h->t.stage_tw64);
      rc |= upload(&h->d_post_tw64, h->t.post_tw64);
    }
    rc |= upload(&h->d_stage_tw, h->t.stage_tw);
    rc |= upload(&h->d_post_tw, h->t.post_tw);
    rc |= upload(&h->d_dct_pack, h->t.dct_pack64);
    rc |= upload(&h->d_dct_pack32, h->t.dct_pack32);
    rc |= upload(&h->d_dct_rowsum, h->t.dct_rowsum);
    if (p->n_mels % 32 == 0 && p->n_mels <= 64) {
      // D = hi + lo * 2^-11 with hi = float16(D), lo = float16((D - hi) * 2^11): three float16 products (hi x, hi x_lo, lo x)
      // carry the float32 coefficient to ~2^-22; lo is scaled so that it stays a normal float16
      const HostTables& t = h->t;
      const int rbs = t.dct_pack.row_blocks, blks = p->n_mels / 32;
      std::vector<uint16_t> pk((size_t)rbs * blks * 2 * 64 * 8, 0);
      std::vector<double> rs((size_t)rbs * 16, 0.0);
      auto bits = [](_Float16 v) { uint16_t u; std::memcpy(&u, &v, 2); return u; };
      for (int rb = 0; rb < rbs; ++rb)
        for (int b = 0; b < blks; ++b)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const int c = rb * 16 + (l & 15), m = 32 * b + 8 * (l >> 4) + j;
              if (c >= p->n_mfcc || m >= p->n_mels) continue;
              const float d = t.dct[(size_t)c * p->n_mels + m];
              const _Float16 hi = (_Float16)d;
              const _Float16 lo = (_Float16)((d - (float)hi) * 2048.0f);
              pk[((((size_t)rb * blks + b) * 2 + 0) * 64 + l) * 8 + j] = bits(hi);
              pk[((((size_t)rb * blks + b) * 2 + 1) * 64 + l) * 8 + j] = bits(lo);
              rs[c] += (double)(float)hi + (double)(float)lo * (1.0 / 2048.0);
            }
      rc |= upload(&h->d_dct_pack16, pk);
      rc |= upload(&h->d_dct_rowsum16, rs);
    }
    rc |= upload(&h->d_rich_tab, h->t.rich_tab);
    rc |= upload(&h->d_chroma_pack, h->t.chroma_pack);
    if (rc) { dysf_destroy(h); return DYSF_E_HIP; }
  }
  *out = h;
  return 0;
}

void dysf_destroy(DysfHandle* h) {
  if (!h) return;
  if (h->device >= 0) {
    (void)hipSetDevice(h->device);
    (void)hipFree(h->d_window); (void)hipFree(h->d_stage_tw); (void)hipFree(h->d_post_tw);
    (void)hipFree(h->d_window_i16); (void)hipFree(h->d_dct_pack); (void)hipFree(h->d_dct_pack32); (void)hipFree(h->d_dct_rowsum); (void)hipFree(h->d_dct_pack16); (void)hipFree(h->d_dct_rowsum16); (void)hipFree(h->d_rich_tab);
    (void)hipFree(h->d_chroma_pack);
    (void)hipFree(h->d_window64); (void)hipFree(h->d_window64_i16); (void)hipFree(h->d_stage_tw64); (void)hipFree(h->d_post_tw64);
    h->ws_S.release(); h->ws_stat.release(); h->ws_flat.release(); h->ws_energy.release(); h->ws_hf.release(); h->ws_rowmax.release();
    h->ws_rich.release(); h->ws_logmel.release(); h->ws_peak_mag.release(); h->ws_peak_bin.release(); h->ws_small.release();
    h->ws_cand.release(); h->ws_framecnt.release();
    h->st_pcm.release(); h->st_off.release(); h->st_pooled.release(); h->st_frames.release();
    h->st_status.release(); h->st_tuning.release(); h->d_mel_sched.release(); h->d_mel_spack.release();
    h->d_mel4_pack.release(); h->d_mel4_kbeg.release();
    for (int i = 0; i < 4; ++i) if (h->tev[i]) (void)hipEventDestroy(h->tev[i]);
  }
  delete h;
}

int32_t dysf_pooled_dim(const DysfHandle* h) {
  return h ? 6 * h->t.p.n_mfcc + (h->t.p.chroma ? 2 * kChroma : 0) + (h->t.p.rich ? kRichDim : 0) : DYSF_E_INVALID;
}
int32_t dysf_frame_rows(const DysfHandle* h) { return h ? 3 * h->t.p.n_mfcc : DYSF_E_INVALID; }
int64_t dysf_num_frames(const DysfHandle* h, int64_t n_samples) {
  if (!h || n_samples < 0) return DYSF_E_INVALID;
  return 1 + n_samples / h->t.p.hop;
}

int64_t dysf_debug_table(const DysfHandle* h, int32_t which, int32_t arg, float* dst, int64_t cap) {
  if (!h || !dst) return fail(DYSF_E_INVALID, "null argument");
  const float* src = nullptr;
  int64_t n = 0;
  switch (which) {
    case 0: src = h->t.window.data(); n = (int64_t)h->t.window.size(); break;
    case 1: src = h->t.mel.data(); n = (int64_t)h->t.mel.size(); break;
    case 2: src = h->t.dct.data(); n = (int64_t)h->t.dct.size(); break;
    case 3:
      if (!h->t.p.chroma || arg < 0 || arg >= kTunings) return fail(DYSF_E_INVALID, "no such chroma bank");
      n = (int64_t)kChroma * h->t.n_bins;
      src = h->t.chroma.data() + (size_t)arg * n;
      break;
    default: return fail(DYSF_E_INVALID, "unknown table");
  }
  if (n > cap) return fail(DYSF_E_INVALID, "destination too small");
  std::memcpy(dst, src, (size_t)n * sizeof(float));
  return n;
}


#ifdef DYSF_STAMP
// diagnostic build only: a kernel that leaves its whole dynamic LDS allocation filled with NaN bit patterns (what a kernel of
// another stream may leave behind on a CU), to expose reads of LDS words a kernel never wrote
__global__ void dysf_poison_lds_kernel(int words, unsigned* sink) {
  extern __shared__ unsigned pl[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) pl[i] = 0x7fc00000u | (unsigned)i;
  __syncthreads();
  if (threadIdx.x == 0 && pl[words - 1] == 1u) sink[0] = pl[0];
}
// ... and one that leaves NaN bit patterns in all 256 vector registers of its waves (what the waves of another kernel may
// leave in the physical registers a new wave is given): exposes reads of registers a kernel never wrote
__global__ __launch_bounds__(256) void dysf_poison_vgpr_kernel(unsigned* sink) {
  unsigned keep = 0x7fc00000u + threadIdx.x;
  asm volatile(".irp r,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255\n\tv_mov_b32 v\\r, %0\n\t.endr\n\ts_nop 4" : : "v"(keep) : "v255", "memory");
  if (keep == 1u) sink[0] = keep;
}
int32_t dysf_debug_poison_vgpr(void* stream, int32_t blocks, void* d_sink) {
  hipLaunchKernelGGL(dysf_poison_vgpr_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)d_sink);
  return hipGetLastError() == hipSuccess ? 0 : DYSF_E_HIP;
}
// ... co-residency stress kernels: which ingredient of the float16 conv2 kernel disturbs a feature kernel on the same CU?
//   kind 0: barriers + LDS traffic (512 threads), 1: float16 MFMA chain, 2: float32 MFMA chain, 3: ds_read_b128 storm
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 4) void dysf_stress_kernel(int kind, int iters, float* sink) {
  extern __shared__ float sl[];
  const int tid = threadIdx.x;
  float acc = 0.f;
  if (kind == 0) {
    for (int i = 0; i < iters; ++i) {
      sl[(tid * 17 + i) & 4095] = (float)i;
      __syncthreads();
      acc += sl[(tid * 5 + i) & 4095];
      __syncthreads();
    }
  } else if (kind == 1) {
    h8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (tid + j)); b[j] = (_Float16)(0.02f * j); }
    f4_t c = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d, 0, 0, 0);
    }
    acc = c[0] + d[1];
  } else if (kind == 2) {
    f4_t c = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(0.01f * tid, 0.5f, c, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(0.5f, 0.01f * tid, d, 0, 0, 0);
    }
    acc = c[0] + d[1];
  } else {
    for (int i = tid; i < 8192; i += 512) sl[i] = (float)i;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
      const float4 v = *(const float4*)(sl + ((tid * 4 + 64 * i) & 8188));
      acc += v.x + v.w;
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}
int32_t dysf_debug_stress(void* stream, int32_t kind, int32_t blocks, int32_t iters, int32_t lds_bytes, void* d_sink) {
  (void)hipFuncSetAttribute((const void*)dysf_stress_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(dysf_stress_kernel, dim3(blocks), dim3(512), (size_t)lds_bytes, (hipStream_t)stream, kind, iters, (float*)d_sink);
  return hipGetLastError() == hipSuccess ? 0 : DYSF_E_HIP;
}
int32_t dysf_debug_poison_lds(void* stream, int32_t blocks, int32_t bytes, void* d_sink) {
  (void)hipFuncSetAttribute((const void*)dysf_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  hipLaunchKernelGGL(dysf_poison_lds_kernel, dim3(blocks), dim3(256), (size_t)bytes, (hipStream_t)stream, bytes / 4, (unsigned*)d_sink);
  return hipGetLastError() == hipSuccess ? 0 : DYSF_E_HIP;
}
// diagnostic build only (not declared in dysf_hip.h): per-phase cycle sums of thread 0 of every workgroup
int32_t dysf_debug_stamps(DysfHandle* h, uint64_t* out16, int32_t reset) {
  if (!h || h->device < 0) return DYSF_E_INVALID;
  const size_t bytes = (size_t)8192 * 16 * 8;  // one private row of 16 slots per workgroup (index mod 8192)
  if (!h->dbg_stamps) { if (hipMalloc((void**)&h->dbg_stamps, bytes) != hipSuccess) return DYSF_E_HIP; (void)hipMemset(h->dbg_stamps, 0, bytes); }
  (void)hipDeviceSynchronize();
  if (out16) {
    std::vector<uint64_t> all((size_t)8192 * 16);
    (void)hipMemcpy(all.data(), h->dbg_stamps, bytes, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) { out16[i] = 0; for (int w = 0; w < 8192; ++w) out16[i] += all[(size_t)w * 16 + i]; }
  }
  if (reset) (void)hipMemset(h->dbg_stamps, 0, bytes);
  return 0;
}
#endif

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Mel schedule: every (column block, row block) band is a run of K-steps; runs are dealt to the waves in order with
// a per-wave budget of ceil(total / waves), splitting a band across waves where needed (partial sums meet in LDS).
// The taps of every segment are re-laid into `spack`, zero-padded to whole batches of eight K-steps, so the kernel
// needs no tap mask: the steps past a segment's end multiply by zero.
static void build_mel_schedule(const HostTables& t, int nw, int cb_count, std::vector<int32_t>& sched, int* nslots,
                               std::vector<float>& spack) {
  const int stride = kMelSchedStride;
  sched.assign((size_t)nw * stride, 0);
  const MfmaPack& mp = t.mel_pack;
  struct Part { int w, idx; };
  int slots = 0;
  const bool per_cb = (cb_count > 0) && (nw % cb_count == 0);
  const int groups = per_cb ? cb_count : 1;
  const int wpg = nw / groups;
  for (int g = 0; g < groups; ++g) {
    int total = 0;
    for (int cb = (per_cb ? g : 0); cb < (per_cb ? g + 1 : cb_count); ++cb)
      for (int rb = 0; rb < mp.row_blocks; ++rb) total += mp.meta[rb * 3 + 1];
    const int target = (total + wpg - 1) / wpg;
    int w = g * wpg, wend = (g + 1) * wpg - 1, cap = target;
    for (int cb = (per_cb ? g : 0); cb < (per_cb ? g + 1 : cb_count); ++cb)
      for (int rb = 0; rb < mp.row_blocks; ++rb) {
        const int kbeg = mp.meta[rb * 3], nst = mp.meta[rb * 3 + 1], poff = mp.meta[rb * 3 + 2];
        int s0 = 0, rem = nst;
        std::vector<Part> parts;
        while (rem > 0) {
          int& nseg = sched[(size_t)w * stride + kMelMaxSeg * 8];
          if (w < wend && (cap < std::min(rem, 4) || nseg >= kMelMaxSeg)) { ++w; cap = target; continue; }
          int take = (w == wend) ? rem : std::min(rem, cap);
          if (w < wend && rem - take > 0 && rem - take < 4) take = rem;  // no slivers
          int* q = &sched[(size_t)w * stride + (size_t)std::min(nseg, kMelMaxSeg - 1) * 8];
          const int soff = (int)spack.size();
          spack.resize(soff + (size_t)((take + 7) / 8 * 8) * 64, 0.f);
          std::copy(mp.data.begin() + poff + (size_t)s0 * 64, mp.data.begin() + poff + (size_t)(s0 + take) * 64,
                    spack.begin() + soff);
          q[0] = rb; q[1] = cb; q[2] = kbeg + 4 * s0; q[3] = take; q[4] = soff; q[5] = 0; q[6] = 0; q[7] = 0;
          parts.push_back({w, std::min(nseg, kMelMaxSeg - 1)});
          if (nseg < kMelMaxSeg) ++nseg;
          s0 += take; rem -= take; cap -= take;
          if (cap <= 0 && w < wend) { ++w; cap = target; }
        }
        if (parts.size() > 1) {
          const int first_slot = slots;
          for (size_t i = 0; i + 1 < parts.size(); ++i) {
            int* q = &sched[(size_t)parts[i].w * stride + (size_t)parts[i].idx * 8];
            q[5] = 1; q[6] = slots++;
          }
          int* q = &sched[(size_t)parts.back().w * stride + (size_t)parts.back().idx * 8];
          q[5] = 2; q[6] = first_slot; q[7] = (int)parts.size() - 1;
        }
      }
  }
  *nslots = slots;
  // the kernel fetches the taps of four consecutive K-steps with one 16-byte load per lane: [step / 4][lane][step % 4]
  // (every segment starts at a multiple of eight steps)
  std::vector<float> quad(spack.size());
  for (size_t s = 0; s < spack.size() / 64; ++s)
    for (int l = 0; l < 64; ++l) quad[(s / 4) * 256 + (size_t)l * 4 + (s % 4)] = spack[s * 64 + l];
  spack.swap(quad);
}

// ------------------------------------------------------------------------------------------------
// Kernel A2's mel product on v_mfma_f32_4x4x1_16B_f32: sixteen independent 4 x 4 blocks per instruction (four mels against
// the wave's four frames), one K-step (bin) each.  A group of four consecutive mels has a band of 7 (low) to 54 (high) bins
// at 512 / 64 mels; walking every group in its own block would cost the longest band's steps.  The schedule spreads the
// long bands over idle blocks instead -- two accumulator sets of sixteen blocks, S steps each:
//   set 0: row r (= DPP row: blocks 4r .. 4r+3) holds the r-th longest group, its band cut into four pieces;
//   set 1: blocks 4r, 4r+1 hold two short groups, blocks 4r+2, 4r+3 the two halves of a middle one.
// After the products the kernel adds set 0's pieces within each row (two DPP row shifts), the halves of set 1 (one), and
// parks the row's set-0 result in block 4r+3 of set 1: every block of set 1 then holds one finished group, out[] says
// which.  Taps outside a slot's piece are zero; piece starts are multiples of four bins (16-byte LDS reads).
static void build_mel4(const HostTables& t, std::vector<float>& pack, std::vector<int32_t>& meta, int* steps_out, int* rounds_out) {
  const int n_mels = t.p.n_mels, n_bins = t.n_bins;
  const int rounds = (n_mels + 63) / 64;
  struct Slot { int group = -1, k0 = 0, k1 = 0; };  // bins [k0, k1) of the group's band
  std::vector<Slot> slots((size_t)rounds * 32);    // [round][set][block]
  std::vector<int> out((size_t)rounds * 16, -1);
  int S = 4;
  for (int rd = 0; rd < rounds; ++rd) {
    int lo[16], hi[16], order[16];
    for (int g = 0; g < 16; ++g) {
      int l = n_bins, h = 0;
      for (int i = 0; i < 4; ++i) {
        const int m = 64 * rd + 4 * g + i;
        if (m >= n_mels) continue;
        for (int k = 0; k < n_bins; ++k)
          if (t.mel[(size_t)m * n_bins + k] != 0.f) { l = std::min(l, k); h = std::max(h, k + 1); }
      }
      if (l >= h) { l = 0; h = 0; }
      lo[g] = l & ~3; hi[g] = h; order[g] = g;
    }
    std::stable_sort(order, order + 16, [&](int x, int y) { return hi[x] - lo[x] > hi[y] - lo[y]; });
    auto cut = [&](int g, int parts, int j, Slot& sl) {   // piece j of the band of group g cut into `parts` pieces
      const int len = hi[g] - lo[g];
      const int pl = ((len + parts - 1) / parts + 3) & ~3;
      sl.group = g; sl.k0 = std::min(hi[g], lo[g] + j * pl); sl.k1 = std::min(hi[g], lo[g] + (j + 1) * pl);
      S = std::max(S, (sl.k1 - sl.k0 + 3) & ~3);
    };
    for (int r = 0; r < 4; ++r) {
      Slot* s0 = &slots[((size_t)rd * 2 + 0) * 16 + 4 * r];
      Slot* s1 = &slots[((size_t)rd * 2 + 1) * 16 + 4 * r];
      for (int j = 0; j < 4; ++j) cut(order[r], 4, j, s0[j]);          // the four longest: quarters in set 0
      cut(order[8 + 2 * r], 1, 0, s1[0]);                              // the eight shortest: whole
      cut(order[9 + 2 * r], 1, 0, s1[1]);
      cut(order[4 + r], 2, 0, s1[2]); cut(order[4 + r], 2, 1, s1[3]);  // the next four: halves in set 1
      out[rd * 16 + 4 * r] = order[8 + 2 * r]; out[rd * 16 + 4 * r + 1] = order[9 + 2 * r];
      out[rd * 16 + 4 * r + 2] = order[4 + r]; out[rd * 16 + 4 * r + 3] = order[r];
    }
  }
  // meta: [round][0..15] first bin of set 0's blocks, [16..31] of set 1's, [32..47] mel offset of the finished block (-1: none)
  meta.assign((size_t)rounds * 48, 0);
  pack.assign((size_t)rounds * (S / 4) * 2 * 64 * 4, 0.f);
  for (int rd = 0; rd < rounds; ++rd) {
    for (int set = 0; set < 2; ++set)
      for (int b = 0; b < 16; ++b) {
        const Slot& sl = slots[((size_t)rd * 2 + set) * 16 + b];
        const int kb = std::min(sl.k0 & ~3, (kA2Kpad - S) & ~3);   // the S steps stay inside the frame's row slot
        meta[rd * 48 + set * 16 + b] = kb;
        for (int st = 0; st < S; ++st)
          for (int i = 0; i < 4; ++i) {
            const int k = kb + st, m = 64 * rd + 4 * sl.group + i, l = 4 * b + i;
            if (sl.group >= 0 && m < n_mels && k >= sl.k0 && k < sl.k1)
              pack[((((size_t)rd * (S / 4) + st / 4) * 2 + set) * 64 + l) * 4 + (st & 3)] = t.mel[(size_t)m * n_bins + k];
          }
      }
    for (int b = 0; b < 16; ++b) {
      const int g = out[rd * 16 + b];
      meta[rd * 48 + 32 + b] = (g >= 0 && 64 * rd + 4 * g < t.mpad) ? 64 * rd + 4 * g : -1;
    }
  }
  *steps_out = S; *rounds_out = rounds;
}

// ------------------------------------------------------------------------------------------------
static int run_dev(DysfHandle* h, const void* d_pcm, int pcm_type, const int64_t* d_offsets, int n_clips,
                   int64_t max_samples, float* d_pooled, float* d_frames, int64_t frame_stride, int32_t* d_status,
                   int32_t* d_tuning, float* dbg_S, float* dbg_logmel, hipStream_t stream, float* d_metrics = nullptr) {
  if (!h) return fail(DYSF_E_INVALID, "null handle");
  if (h->device < 0) return fail(DYSF_E_INVALID, "handle was created without a device (device < 0)");
  if (n_clips < 0 || max_samples < 0) return fail(DYSF_E_INVALID, "negative size");
  if (n_clips == 0) return 0;
  if (!d_pcm || !d_offsets) return fail(DYSF_E_INVALID, "null input pointer");
  if (pcm_type != DYSF_PCM_I16 && pcm_type != DYSF_PCM_F32) return fail(DYSF_E_INVALID, "bad pcm_type");
  if (((uintptr_t)d_pcm & 15u) != 0) return fail(DYSF_E_INVALID, "d_pcm must be 16-byte aligned");
  const HostTables& t = h->t;
  const DysfParams& p = t.p;
  const int64_t t_max = 1 + max_samples / p.hop;
  if (d_frames && frame_stride < t_max) return fail(DYSF_E_INVALID, "frame_stride smaller than the longest clip's frame count");
  if (t_max > (1 << 24) || max_samples > 0x7ff00000LL) return fail(DYSF_E_INVALID, "clip too long");
  HIPCHK(hipSetDevice(h->device));

  KArgs a;
  std::memset(&a, 0, sizeof(a));
  a.pcm = d_pcm; a.pcm_type = pcm_type; a.hop = p.hop; a.sr = p.sr; a.kpad = t.kpad; a.n_mels = p.n_mels; a.n_mfcc = p.n_mfcc;
  a.pooled_dim = dysf_pooled_dim(h); a.frame_stride = frame_stride;
  a.dbg_S = dbg_S; a.dbg_logmel = dbg_logmel; a.dbg_stamps = h->dbg_stamps;
  // power_to_db of a floor cell: float32(10) * float32(log10(amin)), the reference's two roundings (-100 for 1e-10)
  a.top_db = p.top_db; a.amin = p.amin; a.db_amin = 10.0f * (float)std::log10((double)p.amin); a.chroma = p.chroma ? 1 : 0;
  a.parity = p.parity ? 1 : 0; a.t_max = (int)t_max;
  a.window64 = pcm_type == DYSF_PCM_I16 ? h->d_window64_i16 : h->d_window64; a.stage_tw64 = h->d_stage_tw64; a.post_tw64 = h->d_post_tw64;
  a.window = h->d_window; a.window_i16 = h->d_window_i16; a.stage_tw = h->d_stage_tw; a.post_tw = h->d_post_tw;
  a.dct_pack = h->d_dct_pack; a.dct_rb = t.dct_pack.row_blocks; a.dct_groups = t.dct_groups;
  a.dct_pack32 = h->dct64 ? nullptr : h->d_dct_pack32; a.dct_rowsum = h->d_dct_rowsum; a.inv_n_mels = 1.0f / (float)p.n_mels;
  // throughput presets: the DCT on the float16 MFMA (hi / lo split); the parity presets keep their float64 / float32 forms
  a.dct_pack16 = (h->dct16 && !h->dct64 && !p.parity) ? (const void*)h->d_dct_pack16 : nullptr; a.dct_rowsum16 = h->d_dct_rowsum16;
  a.chroma_pack = h->d_chroma_pack; a.chroma_groups = t.chroma_groups;
  a.pip_kmin = t.pip_kmin; a.pip_kmax = t.pip_kmax;
  a.cand_cap = std::max(4096, 16 * ((t.pip_kmax - t.pip_kmin) / 2 + 2));   // 64 KB of LDS, or one round (16 frames) if that is more
  if (p.chroma && (size_t)a.cand_cap * 16 + 4096 > h->lds_limit)   // (+ the chroma kernel's static LDS)
    return fail(DYSF_E_INVALID, "pitch band too wide for the chroma kernel's LDS candidate list (sr too low for this n_fft)");

  // ---- kernel A geometry: small workgroups, several per CU
  const int nt_a = stft_threads(t.NC), nw_a = nt_a / 64;
  const int fpi_a = t.FPW * nw_a, cb_a = (fpi_a + 15) / 16;
  // per-wave LDS region: FFT exchange buffers (doubles in the parity presets), later the power-spectrum rows
  int rstride = std::max(t.FPW * (t.NC + t.NC / 16) * (p.parity ? 2 : 1), t.FPW * t.kpad);
  while (rstride % 32 != 8) ++rstride;
  a.rstride = rstride;
  if (!h->sched_built) {
    std::vector<int32_t> sched;
    std::vector<float> spack;
    build_mel_schedule(t, nw_a, cb_a, sched, &h->mel_slots, spack);
    if (int rc = h->d_mel_sched.reserve(sched.size() * 4)) return rc;
    HIPCHK(hipMemcpy(h->d_mel_sched.p, sched.data(), sched.size() * 4, hipMemcpyHostToDevice));
    if (int rc = h->d_mel_spack.reserve(spack.size() * 4)) return rc;
    HIPCHK(hipMemcpy(h->d_mel_spack.p, spack.data(), spack.size() * 4, hipMemcpyHostToDevice));
    h->sched_built = true;
  }
  a.mel_sched = (const int*)h->d_mel_sched.p;
  a.mel_pack = (const float*)h->d_mel_spack.p;
  a.mel_slots = h->mel_slots;
  a.tiles_per_wg = p.parity ? 12 : 4;   // the parity kernels copy 40-60 KB of float64 tables per workgroup: amortise them
  if (const char* e = std::getenv("DYSF_TILES")) { if (std::atoi(e) > 0) a.tiles_per_wg = std::atoi(e); }
  a.lm_stride = t.mpad;
  a.ws_logmel_stride = t_max * (int64_t)t.mpad;
  a.ws_peak_stride = p.chroma ? t_max * ((t.pip_kmax - t.pip_kmin) / 2 + 2) : 0;
  const int tables = (2 * t.NC + ((t.NC + 2 + 3) & ~3) + 2 * 16 * std::min(t.L, 16) + 3) & ~3;   // (stage-1 twiddles: one column per ls % 16)
  const int tables_a = p.parity ? ((stft_exact_table_floats(p.n_fft) + 3) & ~3) : tables;
  const int tile_a = ((fpi_a - 1) * p.hop + p.n_fft + 8 + 7) & ~7;  // + alignment slack of the staged tile
  const int misc_a = (kStftMiscBase + h->mel_slots * 256 + 3) & ~3;
  const int lds_a_region = 0, lds_a_pcm = nw_a * rstride, lds_a_misc = lds_a_pcm + tile_a, lds_a_tables = lds_a_misc + misc_a;
  size_t lds_a = (size_t)(lds_a_tables + tables_a) * 4;
  if (const char* e = std::getenv("DYSF_LDS_PAD")) lds_a += (size_t)std::atoi(e);  // occupancy experiments only
  if (lds_a > h->lds_limit) return fail(DYSF_E_INVALID, "hop too large for the LDS tile");

  // ---- kernel A-exact: the parity presets at n_fft 2048 (the reference's parameterisation), two workgroups per CU
  const bool use_ex = p.parity && p.n_fft == 2048 && h->exact_variant != 1;
  const int lds_ex_misc = nw_a * kExRegion, lds_ex_tables = lds_ex_misc + misc_a;
  const size_t lds_ex = (size_t)(lds_ex_tables + stft_exact2048_table_floats()) * 4;
  if (use_ex) {
    if (lds_ex > h->lds_limit) return fail(DYSF_E_INVALID, "mel schedule too large for the LDS of the 2048-point parity kernel");
    if (const char* e = std::getenv("DYSF_TILES")) { if (std::atoi(e) > 0) a.tiles_per_wg = std::atoi(e); } else a.tiles_per_wg = 4;
  }

  // ---- kernel A2 (wave-private STFT + mel): the throughput presets at n_fft 512
  const bool use_a2 = h->stft_variant != 1 && p.n_fft == 512 && !p.parity && !p.chroma && !dbg_S && !d_metrics &&
                      p.hop <= 4096;
  int a2_qpw = 0, a2_groups = 0, a2_nw = 8, a2_occ = 6;
  size_t lds_a2 = 0;
  if (use_a2) {
    if (!h->mel4_steps) {
      std::vector<float> pack; std::vector<int32_t> meta;
      build_mel4(t, pack, meta, &h->mel4_steps, &h->mel4_rounds);
      if (int rc = h->d_mel4_pack.reserve(pack.size() * 4)) return rc;
      HIPCHK(hipMemcpy(h->d_mel4_pack.p, pack.data(), pack.size() * 4, hipMemcpyHostToDevice));
      if (int rc = h->d_mel4_kbeg.reserve(meta.size() * 4)) return rc;
      HIPCHK(hipMemcpy(h->d_mel4_kbeg.p, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
    }
    a.mel4_pack = (const float*)h->d_mel4_pack.p; a.mel4_kbeg = (const int*)h->d_mel4_kbeg.p;
    a.mel4_steps = h->mel4_steps; a.mel4_rounds = h->mel4_rounds;
    if (const char* e = std::getenv("DYSF_A2_NW")) { const int v = std::atoi(e); if (v == 4 || v == 5 || v == 6 || v == 8) a2_nw = v; }
    if (a2_nw == 5) a2_occ = 5;
    if (a2_nw == 6) a2_occ = 6;
    if (const char* e = std::getenv("DYSF_A2_OCC")) { if (a2_nw == 4 || a2_nw == 8) a2_occ = std::max(4, std::min(6, std::atoi(e))); }
    if (a2_nw == 8 && a2_occ == 5) a2_occ = 4;
    const int tab_floats = 16 * 84 + h->mel4_rounds * h->mel4_steps * 2 * 64;
    lds_a2 = (size_t)(a2_nw * kA2Region + tab_floats) * 4;
    // the occupancy a 4-wave build is compiled for must fit the CU's LDS; step down otherwise (wider mel banks)
    while (a2_nw == 4 && a2_occ > 4 && lds_a2 * (size_t)a2_occ > h->lds_limit) --a2_occ;
    // Quads per wave: the longest clip is cut into G workgroups of NW waves with about six quads per wave -- more, smaller
    // workgroups than the CUs hold at once, so that they start at different times (waves that run this program in phase
    // -- all in their VALU passes, then all in their LDS / MFMA passes -- are measurably slower: 120 vs 127-134 us same-box
    // between geometries that differ only in this), and no wave of the last workgroup sits idle (76 quads of a 3 s clip at
    // hop 160: 2 workgroups x 8 waves x 5 quads).
    const int64_t nq_max = (t_max + 3) / 4;
    {
      const int64_t g = std::max<int64_t>(1, (nq_max + 3 * a2_nw) / (6 * a2_nw));
      a2_qpw = (int)std::max<int64_t>(1, (nq_max + g * a2_nw - 1) / (g * a2_nw));
    }
    if (const char* e = std::getenv("DYSF_QUADS")) { if (std::atoi(e) > 0) a2_qpw = std::atoi(e); }
    a2_groups = (int)((nq_max + a2_nw * a2_qpw - 1) / (a2_nw * a2_qpw));
    a.quads_per_wave = a2_qpw;
    a.stagger_mode = 0; a.stagger_units = 6;
    if (const char* e = std::getenv("DYSF_LDS_PAD")) lds_a2 += (size_t)std::atoi(e);  // occupancy experiments only
    if (lds_a2 > h->lds_limit) return fail(DYSF_E_INVALID, "mel bands too wide for kernel A2's LDS table");
  }

  // ---- kernel D geometry (rich presets: its own float32 STFT pass): 512 threads, one workgroup per clip
  const int nw_c = 8, fpi_c = t.FPW * nw_c, cb_c = (fpi_c + 15) / 16;
  const int tile_c = ((fpi_c - 1) * p.hop + p.n_fft + 8 + 7) & ~7;
  const int misc_c = (128 + nw_c * cb_c * 256 + 16 * cb_c * 16 + 3) & ~3;
  int rstride_c = std::max(t.FPW * (t.NC + t.NC / 16), t.FPW * t.kpad);  // kernel D keeps the float32 FFT
  while (rstride_c % 32 != 8) ++rstride_c;
  const int lds_c_pcm = nw_c * rstride_c, lds_c_misc = lds_c_pcm + tile_c, lds_c_tables = lds_c_misc + misc_c;
  const size_t lds_c = (size_t)(lds_c_tables + tables) * 4;
  if (p.rich && lds_c > h->lds_limit) return fail(DYSF_E_INVALID, "hop too large for the LDS tile");
  if (p.rich && (size_t)nw_c * rstride_c * 4 < (size_t)kRichSegBytes) return fail(DYSF_E_INVALID, "LDS regions too small for the rich preset");

  // ---- kernel B LDS: frame-0 features, float64 DCT operands, MFCC chunk (the final reduction reuses the last two)
  const int nd = t.dct_pack.row_blocks * t.dct_groups * 256;
  const size_t lds_b = (size_t)(kMfccHead + std::max(2 * nd + t.cpad * kMfccCstr, 8 * (18 * 65 + 3) + 2 * 8 * 18)) * 4;

  // ---- per-clip workspace; very large batches run as several sub-batches that reuse it
  a.ws_rich_stride = p.rich ? t_max : 0;
  a.rich_offset = 6 * p.n_mfcc + (p.chroma ? 2 * kChroma : 0); a.rich_fl = kRichFrameLength; a.rich_tab = h->d_rich_tab;
  a.s_stride = p.chroma ? 16 * t.chroma_groups : 0;
  a.ws_S_stride = p.chroma ? t_max * (int64_t)a.s_stride : 0;
  const bool need_stat = p.parity && d_pooled && !d_frames;  // kernel B re-reads the frame rows for NumPy-order statistics
  a.stat_stride = need_stat ? ((t_max + 3) & ~(int64_t)3) : 0;
  // metrics launches (pipeline1.py:151-174): per-frame spectral flatness and the 25 ms / 10 ms frame energies of snr_db
  const int snr_fl = (int)(0.025 * p.sr), snr_hop = (int)(0.010 * p.sr);
  const long long hf_M = d_metrics ? hf_ratio_fft_size(max_samples) : 0;
  a.energy_stride = d_metrics ? std::max<int64_t>(1, max_samples >= snr_fl ? 1 + (max_samples - snr_fl) / std::max(1, snr_hop) : 1) : 0;
  // kernel A-exact picks piptrack's local maxima from the rows in LDS (kernel C then keeps one prefix sum per frame in LDS)
  const bool ex_picks = use_ex && p.chroma && (size_t)(t_max + 2) * 4 + 4096 <= h->lds_limit;
  a.cand_per_frame = (t.pip_kmax - t.pip_kmin) / 2 + 2;
  const size_t per_clip = (size_t)a.ws_logmel_stride * 4 + 16 + (size_t)a.ws_peak_stride * 8 + (ex_picks ? (size_t)a.ws_peak_stride * 16 + (size_t)t_max * 4 : 0) + (size_t)a.ws_rich_stride * 14 * 4 +
                          (size_t)a.ws_S_stride * 4 + (p.chroma ? (size_t)t_max * 4 : 0) + (size_t)a.stat_stride * 3 * p.n_mfcc * 4 +
                          (d_metrics ? (size_t)(t_max + a.energy_stride) * 4 + hf_ratio_bytes_per_clip(hf_M) : 0);
  // a sub-batch is gridDim.y of kernel A: at most 65535 clips
  const int64_t sub = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_clips, 65535), (int64_t)(h->ws_budget / per_clip)));
  if (int rc = h->ws_logmel.reserve((size_t)sub * a.ws_logmel_stride * 4)) return rc;
  if ((size_t)sub * 12 > h->ws_small.cap) h->ws_clean = false;
  if (int rc = h->ws_small.reserve((size_t)sub * 12)) return rc;
  if (p.chroma) {
    if (int rc = h->ws_peak_mag.reserve((size_t)sub * a.ws_peak_stride * 4)) return rc;
    if (int rc = h->ws_peak_bin.reserve((size_t)sub * a.ws_peak_stride * 4)) return rc;
  }
  if (ex_picks) {
    if (int rc = h->ws_cand.reserve((size_t)sub * a.ws_peak_stride * 16)) return rc;
    if (int rc = h->ws_framecnt.reserve((size_t)sub * t_max * 4)) return rc;
  }
  a.ws_cand = ex_picks ? (float4*)h->ws_cand.p : nullptr;
  a.ws_framecnt = (int*)h->ws_framecnt.p;
  if (p.rich) { if (int rc = h->ws_rich.reserve((size_t)sub * a.ws_rich_stride * 14 * 4)) return rc; }
  if (p.chroma) {
    if (int rc = h->ws_S.reserve((size_t)sub * a.ws_S_stride * 4)) return rc;
    if (int rc = h->ws_rowmax.reserve((size_t)sub * t_max * 4)) return rc;
  }
  a.ws_rowmax = p.chroma ? (float*)h->ws_rowmax.p : nullptr;
  if (need_stat) { if (int rc = h->ws_stat.reserve((size_t)sub * a.stat_stride * 3 * p.n_mfcc * 4)) return rc; }
  if (d_metrics) {
    if (int rc = h->ws_flat.reserve((size_t)sub * t_max * 4)) return rc;
    if (int rc = h->ws_energy.reserve((size_t)sub * a.energy_stride * 4)) return rc;
    if (int rc = h->ws_hf.reserve((size_t)sub * hf_ratio_bytes_per_clip(hf_M))) return rc;
  }
  a.ws_S = (float*)h->ws_S.p;
  a.ws_stat = (float*)h->ws_stat.p;
  a.ws_flat = d_metrics ? (float*)h->ws_flat.p : nullptr;
  a.ws_energy = (float*)h->ws_energy.p;
  a.ws_rich = (float*)h->ws_rich.p;
  a.ws_logmel = (float*)h->ws_logmel.p;
  a.ws_peak_mag = (float*)h->ws_peak_mag.p;
  a.ws_peak_bin = (int*)h->ws_peak_bin.p;
  const int tiles_max = (int)((t_max + fpi_a - 1) / fpi_a);
  const int groups = (tiles_max + a.tiles_per_wg - 1) / a.tiles_per_wg;
  if (std::getenv("DYSF_VERBOSE"))
    std::fprintf(stderr, "[dysf] stft: %d thr, grid %dx%lld, lds %zu B | mfcc: lds %zu B | rich: lds %zu B | sub-batch %lld clips, slots %d\n",
                 nt_a, groups, (long long)sub, use_ex ? lds_ex : lds_a, lds_b, p.rich ? lds_c : (size_t)0, (long long)sub, h->mel_slots);

  h->last_sub = sub; h->last_peak_stride = a.ws_peak_stride;
  for (int64_t c0 = 0; c0 < n_clips; c0 += sub) {
    const int nc = (int)std::min<int64_t>(sub, n_clips - c0);
    a.offsets = (const long long*)d_offsets + c0;
    a.n_clips = nc;
    a.pooled = d_pooled ? d_pooled + (size_t)c0 * a.pooled_dim : nullptr;
    a.frames = d_frames ? d_frames + (size_t)c0 * 3 * p.n_mfcc * frame_stride : nullptr;
    a.status = d_status ? d_status + c0 : nullptr;
    a.tuning = d_tuning ? d_tuning + c0 : nullptr;
    a.ws_clipmax = (unsigned*)h->ws_small.p;
    a.ws_flags = (int*)h->ws_small.p + sub;
    a.ws_peak_cnt = (int*)h->ws_small.p + 2 * sub;
    // presets without chroma leave these words zero themselves (kernel B, after reading them): the memset node is
    // only needed the first time, after a reallocation, a failed launch or a change of stream
    a.self_clean = (p.chroma || p.rich) ? 0 : 1;  // (kernels C and D read the flag after kernel B)
    if (!(a.self_clean && h->ws_clean && h->ws_stream == stream)) HIPCHK(hipMemsetAsync(h->ws_small.p, 0, h->ws_small.cap, stream));
    h->ws_clean = false;
    a.rstride = rstride;
    a.lds_region = lds_a_region; a.lds_pcm = lds_a_pcm; a.lds_misc = lds_a_misc; a.lds_tables = lds_a_tables;
    if (h->timing) HIPCHK(hipEventRecord(h->tev[0], stream));
    if (use_ex) {
      a.rstride = kExRegion; a.lds_region = 0; a.lds_pcm = 0; a.lds_misc = lds_ex_misc; a.lds_tables = lds_ex_tables;
      HIPCHK(launch_stft_mel_exact2048(a, pcm_type, dim3(groups, nc), lds_ex, stream));
    } else if (use_a2) {
      a.lds_region = 0; a.lds_tables = a2_nw * kA2Region; a.lds_misc = a2_nw * kA2Region + 16 * 84; a.lds_pcm = 0;
      HIPCHK(launch_stft_mel_wave(a, pcm_type, a2_nw, a2_occ, dim3(a2_groups, nc), lds_a2, stream));
    } else {
      HIPCHK(launch_stft_mel(a, p.n_fft, pcm_type, dim3(groups, nc), lds_a, stream));
    }
    if (h->timing) HIPCHK(hipEventRecord(h->tev[1], stream));
    HIPCHK(launch_mfcc(a, dim3(nc), lds_b, stream));
    h->ws_clean = a.self_clean != 0; h->ws_stream = stream;
    if (h->timing) HIPCHK(hipEventRecord(h->tev[2], stream));
    if (d_metrics) {   // (reads the non-finite flag kernel B leaves in place for chroma presets)
      a.metrics = d_metrics + (size_t)c0 * 3;
      HIPCHK(launch_metrics(a, dim3(nc), stream));
      HIPCHK(launch_hf_ratio(a, nc, hf_M, h->ws_hf.p, stream));
    }
    if (p.chroma) HIPCHK(launch_chroma(a, p.n_fft, dim3(nc), stream));
    if (p.rich) {
      a.rstride = rstride_c;
      a.lds_region = 0; a.lds_pcm = lds_c_pcm; a.lds_misc = lds_c_misc; a.lds_tables = lds_c_tables;
      HIPCHK(launch_rich(a, p.n_fft, pcm_type, dim3(nc), lds_c, stream));
    }
    if (h->timing) HIPCHK(hipEventRecord(h->tev[3], stream));
  }
  return 0;
}

extern "C" {

int32_t dysf_extract_batch_dev(DysfHandle* h, const void* d_pcm, int32_t pcm_type, const int64_t* d_offsets,
                               int32_t n_clips, int64_t max_samples, float* d_pooled, float* d_frames,
                               int64_t frame_stride, int32_t* d_status, int32_t* d_tuning, void* stream) {
  return run_dev(h, d_pcm, pcm_type, d_offsets, n_clips, max_samples, d_pooled, d_frames, frame_stride, d_status,
                 d_tuning, nullptr, nullptr, (hipStream_t)stream);
}

int32_t dysf_extract_batch(DysfHandle* h, const void* pcm, int32_t pcm_type, const int64_t* offsets, int32_t n_clips,
                           float* pooled, float* frames, int64_t frame_stride, int32_t* status, int32_t* tuning) {
  if (!h) return fail(DYSF_E_INVALID, "null handle");
  if (h->device < 0) return fail(DYSF_E_INVALID, "handle was created without a device (device < 0)");
  if (n_clips < 0) return fail(DYSF_E_INVALID, "negative n_clips");
  if (n_clips == 0) return 0;
  if (!pcm || !offsets) return fail(DYSF_E_INVALID, "null input pointer");
  if (pcm_type != DYSF_PCM_I16 && pcm_type != DYSF_PCM_F32) return fail(DYSF_E_INVALID, "bad pcm_type");
  const size_t esz = pcm_type == DYSF_PCM_I16 ? 2 : 4;
  int64_t max_samples = 0;
  for (int i = 0; i < n_clips; ++i) {
    if (offsets[i + 1] < offsets[i] || offsets[i] < 0) return fail(DYSF_E_INVALID, "offsets must be non-decreasing");
    max_samples = std::max<int64_t>(max_samples, offsets[i + 1] - offsets[i]);
  }
  const int64_t first = offsets[0], total = offsets[n_clips] - first;
  HIPCHK(hipSetDevice(h->device));
  const int pd = dysf_pooled_dim(h), fr = dysf_frame_rows(h);
  std::vector<int64_t> rel(n_clips + 1);
  for (int i = 0; i <= n_clips; ++i) rel[i] = offsets[i] - first;
  if (int rc = h->st_pcm.reserve(std::max<size_t>((size_t)total * esz, 4))) return rc;
  if (int rc = h->st_off.reserve(rel.size() * 8)) return rc;
  if (pooled) if (int rc = h->st_pooled.reserve((size_t)n_clips * pd * 4)) return rc;
  if (frames) if (int rc = h->st_frames.reserve((size_t)n_clips * fr * frame_stride * 4)) return rc;
  if (status) if (int rc = h->st_status.reserve((size_t)n_clips * 4)) return rc;
  if (tuning) if (int rc = h->st_tuning.reserve((size_t)n_clips * 4)) return rc;
  if (total > 0) HIPCHK(hipMemcpy(h->st_pcm.p, (const char*)pcm + (size_t)first * esz, (size_t)total * esz, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->st_off.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice));
  int rc = run_dev(h, h->st_pcm.p, pcm_type, (const int64_t*)h->st_off.p, n_clips, max_samples,
                   pooled ? (float*)h->st_pooled.p : nullptr, frames ? (float*)h->st_frames.p : nullptr, frame_stride,
                   status ? (int32_t*)h->st_status.p : nullptr, tuning ? (int32_t*)h->st_tuning.p : nullptr, nullptr,
                   nullptr, nullptr);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  if (pooled) HIPCHK(hipMemcpy(pooled, h->st_pooled.p, (size_t)n_clips * pd * 4, hipMemcpyDeviceToHost));
  if (frames) HIPCHK(hipMemcpy(frames, h->st_frames.p, (size_t)n_clips * fr * frame_stride * 4, hipMemcpyDeviceToHost));
  if (status) HIPCHK(hipMemcpy(status, h->st_status.p, (size_t)n_clips * 4, hipMemcpyDeviceToHost));
  if (tuning) HIPCHK(hipMemcpy(tuning, h->st_tuning.p, (size_t)n_clips * 4, hipMemcpyDeviceToHost));
  return 0;
}

int32_t dysf_metrics_batch_dev(DysfHandle* h, const void* d_pcm, int32_t pcm_type, const int64_t* d_offsets, int32_t n_clips,
                               int64_t max_samples, float* d_metrics, void* stream) {
  if (!h || !d_metrics) return fail(DYSF_E_INVALID, "null argument");
  const DysfParams& p = h->t.p;
  // librosa.stft defaults are what spectral_flatness_mean (pipeline1.py:170) transforms with; the kernels that zero the
  // per-clip flag words themselves (presets without chroma) would do so before the metrics kernel reads them
  if (p.n_fft != 2048 || p.hop != 512 || !p.chroma)
    return fail(DYSF_E_INVALID, "metrics need a handle with librosa's STFT defaults (n_fft 2048, hop 512) and chroma on, e.g. REFERENCE_149");
  return run_dev(h, d_pcm, pcm_type, d_offsets, n_clips, max_samples, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                 (hipStream_t)stream, d_metrics);
}

int64_t dysf_debug_peaks(DysfHandle* h, float* mag, int32_t* bin, int64_t cap) {
  if (!h || h->device < 0 || !h->ws_peak_mag.p || !h->ws_small.p) return fail(DYSF_E_INVALID, "no chroma launch to inspect");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipDeviceSynchronize());
  int32_t cnt = 0;
  HIPCHK(hipMemcpy(&cnt, (const int32_t*)h->ws_small.p + 2 * h->last_sub, 4, hipMemcpyDeviceToHost));
  const int64_t n = std::min<int64_t>(std::min<int64_t>(cnt, cap), h->last_peak_stride);
  if (n > 0 && mag) HIPCHK(hipMemcpy(mag, h->ws_peak_mag.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (n > 0 && bin) HIPCHK(hipMemcpy(bin, h->ws_peak_bin.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return cnt;
}

int32_t dysf_debug_timing(DysfHandle* h, int32_t enable, float* ms3) {
  if (!h || h->device < 0) return fail(DYSF_E_INVALID, "null handle or no device");
  HIPCHK(hipSetDevice(h->device));
  if (enable && !h->tev[0])
    for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&h->tev[i]));
  if (ms3 && h->timing && h->tev[0]) {
    HIPCHK(hipEventSynchronize(h->tev[3]));
    for (int i = 0; i < 3; ++i) HIPCHK(hipEventElapsedTime(&ms3[i], h->tev[i], h->tev[i + 1]));
  }
  h->timing = enable != 0;
  return 0;
}

int32_t dysf_debug_stft_power_dev(DysfHandle* h, const void* d_pcm, int32_t pcm_type, int64_t n_samples, float* d_S,
                                  void* stream) {
  if (!h || !d_S) return fail(DYSF_E_INVALID, "null argument");
  if (h->device < 0) return fail(DYSF_E_INVALID, "handle was created without a device (device < 0)");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = h->st_off.reserve(16)) return rc;
  const int64_t off[2] = {0, n_samples};
  HIPCHK(hipMemcpy(h->st_off.p, off, sizeof(off), hipMemcpyHostToDevice));
  return run_dev(h, d_pcm, pcm_type, (const int64_t*)h->st_off.p, 1, n_samples, nullptr, nullptr, 0, nullptr, nullptr,
                 d_S, nullptr, (hipStream_t)stream);
}

int32_t dysf_debug_logmel_dev(DysfHandle* h, const void* d_pcm, int32_t pcm_type, int64_t n_samples, float* d_logmel,
                              void* stream) {
  if (!h || !d_logmel) return fail(DYSF_E_INVALID, "null argument");
  if (h->device < 0) return fail(DYSF_E_INVALID, "handle was created without a device (device < 0)");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = h->st_off.reserve(16)) return rc;
  const int64_t off[2] = {0, n_samples};
  HIPCHK(hipMemcpy(h->st_off.p, off, sizeof(off), hipMemcpyHostToDevice));
  return run_dev(h, d_pcm, pcm_type, (const int64_t*)h->st_off.p, 1, n_samples, nullptr, nullptr, 0, nullptr, nullptr,
                 nullptr, d_logmel, (hipStream_t)stream);
}

}  // extern "C"
