// dysf_kernels.h -- argument block shared by the C ABI (dysf_capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dysf_hip.h"

namespace dysf {

struct KArgs {
  // input
  const void* pcm;
  const long long* offsets;
  int n_clips;
  // outputs (any may be null)
  float* pooled;
  int pooled_dim;
  float* frames;
  long long frame_stride;
  int* status;
  int* tuning;
  float* dbg_S;       // [T][n_bins] of clip 0 (single-clip debug launches only)
  float* dbg_logmel;  // [T][n_mels]
  unsigned long long* dbg_stamps;  // [32] cycle sums, DYSF_STAMP builds only
  // parameters
  int sr, hop, kpad, n_mels, mp, n_mfcc;
  float top_db, amin, db_amin;
  int chroma;
  int rstride;  // floats per wave region (FFT exchange + power-spectrum rows)
  // tables
  const float* window;
  const float* stage_tw;
  const float* post_tw;
  const float* mel_pack;
  const int* mel_meta;
  int mel_rb, mel_max_steps;
  const int* mel_sched;  // per wave kMelSchedStride ints (see below)
  int mel_slots;         // split-K partial tiles kept in LDS
  const double* dct_pack;  // float64 A operands
  int dct_rb, dct_steps;
  const float* chroma_pack;
  int chroma_steps;
  int pip_kmin, pip_kmax;
  // per-workgroup global workspace
  float* ws_logmel;
  long long ws_logmel_stride;
  float* ws_peak_mag;
  int* ws_peak_bin;
  long long ws_peak_stride;
  // LDS carve-up (float offsets)
  int lds_region, lds_pcm, lds_logmel, lds_misc, lds_tables;
  int lds_melpack, mel_pack_floats;  // float offset of the kernel-lifetime LDS copy of mel_pack, or -1
  int lds_dct;  // float offset of the phase-B copy of dct_pack, or -1 to read it from global memory
};

constexpr int kMelMaxSeg = 4;
constexpr int kMelSchedStride = 40;  // ints per wave: kMelMaxSeg x {rb, cb, kbeg, nsteps, poff, kind, slot, nparts}, nseg at [32]
constexpr int kMiscBaseFloats = 288 + 8 * kMelSchedStride;  // fixed part of the misc LDS area (see dysf_kernels.hip)

// threads: 256 (4 waves, several workgroups per CU, log-mel in the global workspace) or 512 (8 waves)
hipError_t launch_clip_kernel(const KArgs& a, int n_fft, int threads, int pcm_type, bool lm_lds, int grid,
                              size_t lds_bytes, hipStream_t s);

}  // namespace dysf
