// dysf_kernels.h -- argument block shared by the C ABI (dysf_capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/dysf_hip.h"

namespace dysf {

struct KArgs {
  // input
  const void* pcm;
  const long long* offsets;
  int n_clips;
  int pcm_type;      // DYSF_PCM_I16 / DYSF_PCM_F32 (the STFT kernels are templated on it; the metrics kernel reads it)
  // outputs (any may be null)
  float* pooled;
  int pooled_dim;
  float* frames;
  long long frame_stride;
  int* status;
  int* tuning;
  float* dbg_S;       // [T][n_bins] of clip 0 (single-clip debug launches only)
  float* dbg_logmel;  // [T][n_mels]
  unsigned long long* dbg_stamps;  // [16] cycle sums, DYSF_STAMP diagnostic builds only
  // parameters
  int sr, hop, kpad, n_mels, n_mfcc;
  float top_db, amin, db_amin;
  int chroma;
  int parity;        // 1: float64 FFT + NumPy's complex64 abs (kernel A), NumPy-order float32 statistics (kernel B)
  int t_max;         // frames the workspace rows are sized for; a clip with more is rejected (DYSF_TOO_LONG)
  int rstride;       // floats per wave region (FFT exchange + power-spectrum rows)
  int tiles_per_wg;  // kernel A: consecutive frame tiles one workgroup processes
  // tables
  const float* window;    // pre-scaled by 1/2
  const float* window_i16;  // the same times 2^-15: int16 PCM is staged as unscaled integers
  const float* stage_tw;
  const float* post_tw;
  const double* window64;   // parity presets: [n_fft] window / 2 (times 2^-15 for int16 PCM), float64
  const double* stage_tw64; // [(nstage-1)][L][16][2] float64
  const double* post_tw64;  // [NC/2+1][2] float64
  const float* mel_pack;
  const int* mel_sched;   // per wave kMelSchedStride ints (see below)
  int mel_slots;          // split-K partial tiles kept in LDS
  const double* dct_pack;  // float64 A operands
  const float* dct_pack32;   // float32 A operands of the same layout (null: run the DCT on the float64 MFMA)
  const double* dct_rowsum;  // [16*dct_rb] float64 sums of the float32 coefficients of every row
  const void* dct_pack16;    // float16 hi / lo split of the same coefficients for v_mfma_f32_16x16x32_f16 (n_mels % 32 == 0):
                             // [row block][K block of 32 mels][hi, lo * 2^11][64 lanes][8 halfs]; null: float32 MFMA
  const double* dct_rowsum16;  // [16*dct_rb] float64 row sums of hi + lo * 2^-11 (the coefficients the split form multiplies by)
  float inv_n_mels;
  int dct_rb, dct_groups;  // row blocks of 16 coefficients, K groups of 16 mels
  const float* chroma_pack;
  int chroma_groups;        // K groups of 16 bins in a bank = s_stride / 16
  // rich presets (kernel D)
  int rich_offset;          // first float of the rich descriptors in a pooled row
  int rich_fl;              // frame length of zero_crossing_rate / rms (2048)
  const int* rich_tab;      // [7][3] contrast bands: first bin, end bin, bins averaged at each end
  float* ws_rich;           // [clip][14][ws_rich_stride] per-frame peak (7 rows) and valley (7 rows) magnitudes
  long long ws_rich_stride;
  int pip_kmin, pip_kmax;
  int cand_cap;             // kernel C: local maxima its LDS list holds (>= one round's worst case)
  // per-clip global workspace
  float* ws_logmel;            // [clip][T_max][lm_stride] log-mel (dB, un-clamped), frame-major
  long long ws_logmel_stride;  // floats per clip
  int lm_stride;               // floats per log-mel row = n_mels rounded up to 16
  unsigned* ws_clipmax;        // [clip] bits of the clip's largest mel power (non-negative floats order like their bits)
  float* ws_S;                 // chroma presets: [clip][T_max][s_stride] power spectrogram (kernel A -> kernel C)
  long long ws_S_stride;       // floats per clip
  int s_stride;                // floats per spectrum row = n_bins rounded up to 16 (padding columns are zero)
  float* ws_rowmax;            // chroma presets: [clip][t_max] largest power-spectrum bin of every frame (piptrack's threshold)
  float* ws_flat;              // metrics launches: [clip][t_max] spectral flatness of every frame (kernel A -> metrics kernel)
  float* ws_energy;            // metrics launches: [clip][energy_stride] 25 ms frame energies of snr_db
  long long energy_stride;
  float* metrics;              // [clip][3] snr_db, spectral_flatness_mean, high_freq_energy_ratio (pipeline1.py:151-186)
  float* ws_stat;              // parity presets without a frames output: [clip][3*n_mfcc][stat_stride] frame rows
  long long stat_stride;
  int self_clean;           // kernel B zeroes ws_clipmax / ws_flags after reading them (presets without chroma)
  int* ws_flags;               // [clip] 1: non-finite sample seen
  int* ws_peak_cnt;            // [clip] piptrack peaks appended so far
  float* ws_peak_mag;
  int* ws_peak_bin;
  long long ws_peak_stride;    // peak capacity per clip
  float4* ws_cand;             // kernel A-exact -> kernel C: [clip][t_max][cand_per_frame] piptrack's local maxima (bin, power, left,
                               // right neighbour) of every frame, picked from the spectrum row while it is in LDS; null: kernel C
                               // picks them from the stored spectrogram itself (a second pass over it)
  int* ws_framecnt;            // [clip][t_max] local maxima of every frame
  int cand_per_frame;          // slots per frame = (pip_kmax - pip_kmin) / 2 + 2 (local maxima cannot be adjacent)
  // LDS carve-up (float offsets) of the kernel being launched
  int lds_region, lds_pcm, lds_misc, lds_tables;
  // kernel A2 (wave-private STFT + mel, throughput presets at n_fft 512): mel product on v_mfma_f32_4x4x1_16B_f32
  const float* mel4_pack;  // [rounds][steps / 4][2 sets][64 lanes][4] A operands: lane (block b = l >> 2, row i = l & 3) holds
                           // the taps of its slot's mel i at the bins kbeg + 4*s .. + 3 (dysf_capi.hip: build_mel4)
  const int* mel4_kbeg;    // [rounds][48]: first bin of set 0's blocks, of set 1's blocks, mel offset of the finished blocks
  int mel4_steps;          // K-steps per set (a multiple of 4)
  int mel4_rounds;         // rounds of 16 blocks x 4 mels = ceil(n_mels / 64)
  int quads_per_wave;      // consecutive groups of four frames one wave transforms
  int stagger_mode, stagger_units;  // start delay per wave (see the kernel): key source, units of 512 cycles per key step
};

constexpr int kA2Kpad = 272;      // kernel A2: floats per power-spectrum row in LDS (= the FFT exchange stride of a frame)
constexpr int kA2Region = 4 * kA2Kpad;  // floats of LDS per wave: exchange buffer of its four frames, then their spectrum rows

constexpr int kMelMaxSeg = 4;
constexpr int kMelSchedStride = 40;  // ints per wave: kMelMaxSeg x {rb, cb, kbeg, nsteps, poff, kind, slot, nparts}, nseg at [32]
constexpr int kStftMiscBase = 32 + 8 * kMelSchedStride;  // kernel A misc area before the split-K slots
constexpr int kRichDim = 17;                             // rich presets: 7 + 7 contrast statistics, zcr, rms, centroid
constexpr int kRichFrameLength = 2048;                   // librosa's default frame_length of zero_crossing_rate / rms
constexpr int kRichSegBytes = 1024 * 16;                 // kernel D: LDS taken from the regions by its segment sums
constexpr int kMfccCstr = 132;                           // kernel B: LDS stride (float2) of a row pair of the MFCC chunk (128 columns)
constexpr int kPwMaxFrames = 16384;                      // kernel B: longest clip whose statistics follow NumPy's pairwise order
constexpr int kPwProg = 2 * (kPwMaxFrames / 64 + 2);     // entries of the pairwise-summation program (leaves + adds)
constexpr int kMfccHead = 144;                           // kernel B: frame-0 features ahead of the DCT operands

// kernel A workgroup size: 16 frames per tile up to n_fft 1024, 8 frames at n_fft 2048
constexpr int stft_threads(int nc) { return nc <= 256 ? 256 : 512; }

hipError_t launch_stft_mel(const KArgs& a, int n_fft, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s);
hipError_t launch_stft_mel_wave(const KArgs& a, int pcm_type, int waves_per_wg, int waves_per_simd, dim3 grid, size_t lds_bytes, hipStream_t s);  // kernel A2
// kernel A-exact (parity presets at n_fft 2048): 512 threads, kExRegion floats of LDS per wave
constexpr int kExRegion = 1096;
int stft_exact2048_table_floats();
hipError_t launch_stft_mel_exact2048(const KArgs& a, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s);
hipError_t launch_mfcc(const KArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s);
hipError_t launch_chroma(const KArgs& a, int n_fft, dim3 grid, hipStream_t s);
hipError_t launch_metrics(const KArgs& a, dim3 grid, hipStream_t s);
int stft_exact_table_floats(int n_fft);  // LDS floats of the parity presets' float64 FFT tables
// high_freq_energy_ratio (dysf_hfratio.hip): chirp-z transform of every whole clip on FFTs of size M
long long hf_ratio_fft_size(long long max_samples);
size_t hf_ratio_bytes_per_clip(long long M);
hipError_t launch_hf_ratio(const KArgs& a, int n_clips, long long M, void* workspace, hipStream_t s);
hipError_t launch_rich(const KArgs& a, int n_fft, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s);

}  // namespace dysf
