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
  int threads = 512;      // workgroup shape (DYSF_THREADS=256|512 overrides; see run_dev)
  int force_logmel = 0;   // DYSF_LOGMEL=global forces the workspace variant
  unsigned long long* dbg_stamps = nullptr;  // DYSF_STAMP diagnostic builds
  size_t lds_limit = 163840;
  // device tables
  float *d_window = nullptr, *d_stage_tw = nullptr, *d_post_tw = nullptr, *d_mel_pack = nullptr,
        *d_chroma_pack = nullptr;
  double* d_dct_pack = nullptr;
  int* d_mel_meta = nullptr;
  // workspace / staging
  DevBuf ws_logmel, ws_peak_mag, ws_peak_bin;
  DevBuf st_pcm, st_off, st_pooled, st_frames, st_status, st_tuning;
  DevBuf d_mel_sched;
  int sched_threads = 0, mel_slots = 0;
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
    p->n_fft = 2048; p->hop = 512; p->n_mels = 128; p->n_mfcc = 20; p->chroma = 1;
  } else if (preset == DYSF_PRESET_BENCH_512) {
    p->n_fft = 512; p->hop = 160; p->n_mels = 64; p->n_mfcc = 40; p->chroma = 0;
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
  if (const char* e = std::getenv("DYSF_THREADS")) { if (std::atoi(e) == 512) h->threads = 512; else if (std::atoi(e) == 256) h->threads = 256; }
  if (const char* e = std::getenv("DYSF_LOGMEL")) { if (std::string(e) == "global") h->force_logmel = 1; }
  if (device >= 0) {
    int rc = 0;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete h; return fail_hip(e, "hipSetDevice"); }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) h->num_cu = v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0)
      h->lds_limit = (size_t)v;
    rc |= upload(&h->d_window, h->t.window_half);
    rc |= upload(&h->d_stage_tw, h->t.stage_tw);
    rc |= upload(&h->d_post_tw, h->t.post_tw);
    rc |= upload(&h->d_mel_pack, h->t.mel_pack.data);
    rc |= upload(&h->d_mel_meta, h->t.mel_pack.meta);
    rc |= upload(&h->d_dct_pack, h->t.dct_pack64);
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
    (void)hipFree(h->d_mel_pack); (void)hipFree(h->d_mel_meta); (void)hipFree(h->d_dct_pack);
    (void)hipFree(h->d_chroma_pack);
    h->ws_logmel.release(); h->ws_peak_mag.release(); h->ws_peak_bin.release();
    h->st_pcm.release(); h->st_off.release(); h->st_pooled.release(); h->st_frames.release();
    h->st_status.release(); h->st_tuning.release(); h->d_mel_sched.release();
  }
  delete h;
}

int32_t dysf_pooled_dim(const DysfHandle* h) { return h ? 6 * h->t.p.n_mfcc + (h->t.p.chroma ? 2 * kChroma : 0) : DYSF_E_INVALID; }
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
// diagnostic build only (not declared in dysf_hip.h): per-phase cycle sums of wave 0 [0..15] and the last wave [16..31]
int32_t dysf_debug_stamps(DysfHandle* h, uint64_t* out32, int32_t reset) {
  if (!h || h->device < 0) return DYSF_E_INVALID;
  if (!h->dbg_stamps) { if (hipMalloc((void**)&h->dbg_stamps, 32 * 8) != hipSuccess) return DYSF_E_HIP; (void)hipMemset(h->dbg_stamps, 0, 32 * 8); }
  (void)hipDeviceSynchronize();
  if (out32) (void)hipMemcpy(out32, h->dbg_stamps, 32 * 8, hipMemcpyDeviceToHost);
  if (reset) (void)hipMemset(h->dbg_stamps, 0, 32 * 8);
  return 0;
}
#endif

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Mel schedule: every (column block, row block) band is a run of K-steps; runs are dealt to the waves in order with
// a per-wave budget of ceil(total / waves), splitting a band across waves where needed (partial sums meet in LDS).
static void build_mel_schedule(const HostTables& t, int nw, int cb_count, std::vector<int32_t>& sched, int* nslots) {
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
          q[0] = rb; q[1] = cb; q[2] = kbeg + 4 * s0; q[3] = take; q[4] = poff + s0 * 64; q[5] = 0; q[6] = 0; q[7] = 0;
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
}

// ------------------------------------------------------------------------------------------------
static int run_dev(DysfHandle* h, const void* d_pcm, int pcm_type, const int64_t* d_offsets, int n_clips,
                   int64_t max_samples, float* d_pooled, float* d_frames, int64_t frame_stride, int32_t* d_status,
                   int32_t* d_tuning, float* dbg_S, float* dbg_logmel, hipStream_t stream) {
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
  if (t_max > (1 << 24)) return fail(DYSF_E_INVALID, "clip too long");
  HIPCHK(hipSetDevice(h->device));

  KArgs a;
  std::memset(&a, 0, sizeof(a));
  a.pcm = d_pcm; a.offsets = (const long long*)d_offsets; a.n_clips = n_clips;
  a.pooled = d_pooled; a.pooled_dim = dysf_pooled_dim(h); a.frames = d_frames; a.frame_stride = frame_stride;
  a.status = d_status; a.tuning = d_tuning; a.dbg_S = dbg_S; a.dbg_logmel = dbg_logmel;
  a.dbg_stamps = h->dbg_stamps;
  a.sr = p.sr; a.hop = p.hop; a.kpad = t.kpad; a.n_mels = p.n_mels; a.n_mfcc = p.n_mfcc;
  a.top_db = p.top_db; a.amin = p.amin; a.db_amin = (float)(10.0 * std::log10((double)p.amin)); a.chroma = p.chroma ? 1 : 0;
  int mp = p.n_mels;
  while (mp % 32 != 2) ++mp;
  a.mp = mp;
  a.window = h->d_window; a.stage_tw = h->d_stage_tw; a.post_tw = h->d_post_tw;
  a.mel_pack = h->d_mel_pack; a.mel_meta = h->d_mel_meta; a.mel_rb = t.mel_pack.row_blocks; a.mel_max_steps = t.mel_pack.max_steps;
  a.dct_pack = h->d_dct_pack; a.dct_rb = t.dct_pack.row_blocks; a.dct_steps = t.dct_pack.max_steps;
  a.chroma_pack = h->d_chroma_pack; a.chroma_steps = t.chroma_steps;
  a.pip_kmin = t.pip_kmin; a.pip_kmax = t.pip_kmax;

  // LDS carve-up (floats).  Two shapes: 256-thread workgroups with the clip's log-mel in a per-workgroup
  // global workspace (small LDS footprint -> several workgroups per CU hide each other's barriers and LDS
  // latency), or 512-thread workgroups that keep the log-mel in LDS (one workgroup per CU).
  int threads = h->threads;
  const int nw = threads / 64;
  const int fpi = t.FPW * nw;
  int rstride = std::max(t.FPW * (t.NC + t.NC / 16), t.FPW * t.kpad);
  while (rstride % 32 != 8) ++rstride;
  a.rstride = rstride;
  const int tile_samples = ((fpi - 1) * p.hop + p.n_fft + 8 + 7) & ~7;  // + alignment slack of the staged tile
  const int cb = (fpi + 15) / 16;
  if (h->sched_threads != threads) {
    std::vector<int32_t> sched;
    build_mel_schedule(t, nw, cb, sched, &h->mel_slots);
    if (int rc = h->d_mel_sched.reserve(sched.size() * 4)) return rc;
    HIPCHK(hipMemcpy(h->d_mel_sched.p, sched.data(), sched.size() * 4, hipMemcpyHostToDevice));
    h->sched_threads = threads;
  }
  a.mel_sched = (const int*)h->d_mel_sched.p;
  a.mel_slots = h->mel_slots;
  const int misc = kMiscBaseFloats + h->mel_slots * 256 + (p.chroma ? nw * cb * 256 + 16 * cb * 16 : 0);
  const int tables = (2 * t.NC + t.NC + 2 + 3) & ~3;
  a.lds_region = 0;
  a.lds_pcm = nw * rstride;
  a.lds_misc = a.lds_pcm + tile_samples;
  a.lds_tables = a.lds_misc + misc;
  a.lds_logmel = a.lds_tables + tables;
  {  // phase B reuses [0, lds_misc): MFCC chunk first, then the float64 DCT operands if they fit
    const int chunk_floats = ((t.cpad * (16 * nw + 1) + 3) & ~3);
    const int dct_floats = t.dct_pack.row_blocks * t.dct_pack.max_steps * 64 * 2;
    if (chunk_floats > a.lds_misc) return fail(DYSF_E_INVALID, "internal: MFCC chunk does not fit the LDS scratch");
    a.lds_dct = (chunk_floats + dct_floats <= a.lds_misc) ? chunk_floats : -1;
  }
  const size_t fixed_bytes = (size_t)a.lds_logmel * 4;
  if (fixed_bytes > h->lds_limit) return fail(DYSF_E_INVALID, "hop too large for the LDS tile");
  bool lm_lds = (threads == 512) && fixed_bytes + (size_t)t_max * mp * 4 <= h->lds_limit;
  if (h->force_logmel == 1) lm_lds = false;
  size_t lds_bytes = lm_lds ? fixed_bytes + (size_t)t_max * mp * 4 : fixed_bytes;
  // the packed mel taps live in LDS for the whole kernel when that does not cost a resident workgroup
  a.mel_pack_floats = (int)t.mel_pack.data.size();
  a.lds_melpack = -1;
  {
    const size_t with_pack = (((lds_bytes / 4 + 3) & ~(size_t)3) + a.mel_pack_floats) * 4;
    const size_t budget = lm_lds ? h->lds_limit : h->lds_limit / (threads == 256 ? 3 : 2);
    if (with_pack <= budget) {
      a.lds_melpack = (int)((lds_bytes / 4 + 3) & ~(size_t)3);
      lds_bytes = with_pack;
    }
  }
  int wg_per_cu = 1;
  if (!lm_lds) wg_per_cu = std::max<int>(1, std::min<int>(threads == 256 ? 3 : 1, (int)(h->lds_limit / lds_bytes)));
  const int grid = std::min(n_clips, h->num_cu * wg_per_cu);
  if (!lm_lds) {
    if (int rc = h->ws_logmel.reserve((size_t)grid * t_max * mp * 4)) return rc;
    a.ws_logmel = (float*)h->ws_logmel.p;
    a.ws_logmel_stride = t_max * mp;
  }
  if (p.chroma) {
    const int64_t cap = t_max * ((t.pip_kmax - t.pip_kmin) / 2 + 2);
    if (int rc = h->ws_peak_mag.reserve((size_t)grid * cap * 4)) return rc;
    if (int rc = h->ws_peak_bin.reserve((size_t)grid * cap * 4)) return rc;
    a.ws_peak_mag = (float*)h->ws_peak_mag.p;
    a.ws_peak_bin = (int*)h->ws_peak_bin.p;
    a.ws_peak_stride = cap;
  }
  if (std::getenv("DYSF_VERBOSE"))
    std::fprintf(stderr, "[dysf] threads %d grid %d lds %zu B (limit %zu) logmel %s melpack %s dct %s slots %d tile %d floats\n", threads, grid, lds_bytes, h->lds_limit, lm_lds ? "lds" : "global", a.lds_melpack >= 0 ? "lds" : "global", a.lds_dct >= 0 ? "lds" : "global", a.mel_slots, tile_samples);
  HIPCHK(launch_clip_kernel(a, p.n_fft, threads, pcm_type, lm_lds, grid, lds_bytes, stream));
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
