// dysf_tables.hpp -- host-side (float64) construction of every constant table the kernels read:
// Hann window, FFT twiddles, Slaney mel bank (librosa.filters.mel), orthonormal DCT-II rows
// (scipy.fft.dct norm='ortho'), the 100 chroma banks librosa.filters.chroma can be asked for by
// estimate_tuning, and their MFMA-ready packed forms.  The reference reaches these through
// librosa.feature.mfcc / chroma_stft (pipeline1.py:216, :227); nothing here is copied from librosa --
// the formulas are its published definitions.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/dysf_hip.h"
#include "dysf_fft.hpp"

namespace dysf {

// A[rows x cols] (row-major) packed for v_mfma_f32_16x16x4_f32 A-operands: per 16-row block a run of
// K-steps [kbeg, kbeg + 4*nsteps); step s, lane l holds A[rb*16 + (l & 15)][kbeg + 4*s + (l >> 4)].
struct MfmaPack {
  std::vector<float> data;
  std::vector<int32_t> meta;  // per row block: kbeg, nsteps, offset (floats)
  int row_blocks = 0;
  int max_steps = 0;
};

inline MfmaPack pack_rows(const std::vector<float>& A, int rows, int cols, bool banded) {
  MfmaPack p;
  p.row_blocks = (rows + 15) / 16;
  for (int rb = 0; rb < p.row_blocks; ++rb) {
    int lo = cols, hi = 0;
    if (banded) {
      for (int r = rb * 16; r < std::min(rows, rb * 16 + 16); ++r)
        for (int k = 0; k < cols; ++k)
          if (A[(size_t)r * cols + k] != 0.f) { lo = std::min(lo, k); hi = std::max(hi, k + 1); }
      if (lo >= hi) { lo = 0; hi = 4; }
    } else {
      lo = 0; hi = cols;
    }
    const int kbeg = lo & ~3;
    const int nsteps = (hi - kbeg + 3) / 4;
    const int off = (int)p.data.size();
    p.meta.push_back(kbeg); p.meta.push_back(nsteps); p.meta.push_back(off);
    p.max_steps = std::max(p.max_steps, nsteps);
    p.data.resize(off + (size_t)nsteps * 64, 0.f);
    for (int s = 0; s < nsteps; ++s)
      for (int l = 0; l < 64; ++l) {
        const int r = rb * 16 + (l & 15), k = kbeg + 4 * s + (l >> 4);
        if (r < rows && k < cols) p.data[off + s * 64 + l] = A[(size_t)r * cols + k];
      }
  }
  return p;
}

struct HostTables {
  DysfParams p{};
  int NC = 0, L = 0, FPW = 0, nstage = 0;
  int n_bins = 0, kpad = 0, mpad = 0, cpad = 0;
  std::vector<float> window;     // [n_fft]
  std::vector<float> window_half;  // window / 2 (exact): what the kernel multiplies by
  std::vector<float> stage_tw;   // [(nstage-1)][L][16][2]
  std::vector<float> post_tw;    // [NC/2+1][2]
  std::vector<float> mel;        // [n_mels][n_bins]
  std::vector<float> dct;        // [n_mfcc][n_mels]
  std::vector<float> chroma;     // [100][12][n_bins]   (only when p.chroma)
  MfmaPack mel_pack, dct_pack;
  std::vector<double> dct_pack64;  // exact float64 coefficients, K-grouped layout (see build_tables)
  std::vector<float> dct_pack32;   // the float32 coefficients in the same layout
  std::vector<double> dct_rowsum;  // [16 * row blocks] float64 sum over mels of every row's float32 coefficients
  int dct_groups = 0;               // groups of 16 mels = mpad / 16
  std::vector<int32_t> rich_tab;    // rich presets: [7][3] contrast bands (first bin, end bin, bins averaged per end)
  std::vector<float> chroma_pack;  // [100][chroma_groups][64 lanes][4]: lane (row = l & 15, q = l >> 4) holds bins 16g + 4q + s
  int chroma_groups = 0;           // ceil(n_bins / 16); spectrum rows in the workspace are 16 * chroma_groups floats
  // parity presets: the FFT tables in float64
  std::vector<double> window64_half;  // [n_fft] window / 2
  std::vector<double> stage_tw64;     // [(nstage-1)][L][16][2]
  std::vector<double> post_tw64;      // [NC/2+1][2]
  int pip_kmin = 0, pip_kmax = 0;  // piptrack bins with 150 <= f < min(4000, sr/2)
};

inline double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
inline double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}
// numpy.linspace(start, stop, num)
inline std::vector<double> linspace(double start, double stop, int num, bool endpoint = true) {
  std::vector<double> y(num);
  const int div = endpoint ? num - 1 : num;
  const double step = (stop - start) / div;
  for (int i = 0; i < num; ++i) y[i] = i * step + start;
  if (endpoint && num > 1) y[num - 1] = stop;
  return y;
}
inline double tuning_value(int idx) { return (double)idx * 0.01 + (-0.5); }

template <int NC, int S>
inline void fill_stage_tw(float* dst, double* dst64) {
  constexpr int L = FftGeom<NC>::L;
  for (int ls = 0; ls < L; ++ls)
    for (int p = 0; p < kPts; ++p) {
      int num, den;
      fft_twiddle_frac<NC, S>(ls, p, &num, &den);
      num %= den;
      const double a = -2.0 * M_PI * (double)num / (double)den;
      dst[(ls * kPts + p) * 2 + 0] = (float)std::cos(a);
      dst[(ls * kPts + p) * 2 + 1] = (float)std::sin(a);
      dst64[(ls * kPts + p) * 2 + 0] = std::cos(a);
      dst64[(ls * kPts + p) * 2 + 1] = std::sin(a);
    }
}
template <int NC>
inline void fill_fft_tables(HostTables& t) {
  using G = FftGeom<NC>;
  t.NC = NC; t.L = G::L; t.FPW = G::FPW; t.nstage = G::NSTAGE;
  t.stage_tw.assign((size_t)(G::NSTAGE - 1) * G::L * kPts * 2, 0.f);
  t.stage_tw64.assign(t.stage_tw.size(), 0.0);
  fill_stage_tw<NC, 1>(t.stage_tw.data(), t.stage_tw64.data());
  if (G::NSTAGE == 3) fill_stage_tw<NC, 2>(t.stage_tw.data() + (size_t)G::L * kPts * 2, t.stage_tw64.data() + (size_t)G::L * kPts * 2);
  t.post_tw.resize((NC / 2 + 1) * 2);
  t.post_tw64.resize((NC / 2 + 1) * 2);
  for (int k = 0; k <= NC / 2; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)(2 * NC);
    // V[k] = -i exp(-2 pi i k / N) = (sin a, -cos a) with a = -2 pi k / N   (see rfft_pair_power)
    t.post_tw[2 * k] = (float)std::sin(a);
    t.post_tw[2 * k + 1] = (float)(-std::cos(a));
    t.post_tw64[2 * k] = std::sin(a);
    t.post_tw64[2 * k + 1] = -std::cos(a);
  }
}

// Row stride of the power-spectrum rows in LDS.  The MFMA B reads (16 frames x 2 K-values per 32-lane group) are
// conflict-free for kpad % 32 == 2 and, with 4 rows per wave region (region stride % 32 == 8), also for 18; 18
// additionally puts the rows of the two frames that share a 32-lane group 18 banks apart for the row stores.
inline int round_kpad(int n_bins, int frames_per_wave) {
  const int want = frames_per_wave == 4 ? 18 : 2;
  int k = n_bins;
  while (k % 32 != want) ++k;
  return k;
}

inline bool build_tables(const DysfParams& p, HostTables& t, const char** err) {
  t.p = p;
  if (p.n_fft != 256 && p.n_fft != 512 && p.n_fft != 1024 && p.n_fft != 2048) { *err = "n_fft must be 256, 512, 1024 or 2048"; return false; }
  if (p.hop < 1 || p.hop > p.n_fft) { *err = "hop must be in [1, n_fft]"; return false; }
  if (p.n_mels < 1 || p.n_mels > 128) { *err = "n_mels must be in [1, 128]"; return false; }
  if (p.n_mfcc < 1 || p.n_mfcc > 48 || p.n_mfcc > p.n_mels) { *err = "n_mfcc must be in [1, min(48, n_mels)]"; return false; }
  if (p.delta_width != 9) { *err = "delta_width must be 9"; return false; }
  if (p.sr < 1) { *err = "sr must be positive"; return false; }
  if (!(p.amin > 0.f)) { *err = "amin must be positive"; return false; }
  if (p.parity != 0 && p.parity != 1) { *err = "parity must be 0 or 1"; return false; }
  if (p.rich && (2048 % p.hop != 0)) { *err = "rich presets need a hop that divides 2048 (frame_length of zcr / rms)"; return false; }
  if (p.rich && 6400.0 >= 0.5 * p.sr) { *err = "rich presets need sr > 12800 Hz (spectral_contrast bands)"; return false; }
  const int n_fft = p.n_fft, sr = p.sr;
  t.n_bins = n_fft / 2 + 1;
  t.mpad = (p.n_mels + 15) / 16 * 16;
  t.cpad = (p.n_mfcc + 15) / 16 * 16;
  switch (n_fft) {
    case 256: fill_fft_tables<128>(t); break;
    case 512: fill_fft_tables<256>(t); break;
    case 1024: fill_fft_tables<512>(t); break;
    default: fill_fft_tables<1024>(t); break;
  }
  t.kpad = round_kpad(t.n_bins, t.FPW);
  // periodic Hann (scipy.signal.get_window('hann', n, fftbins=True))
  t.window.resize(n_fft);
  for (int n = 0; n < n_fft; ++n) t.window[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)n_fft));
  t.window_half.resize(n_fft);
  for (int n = 0; n < n_fft; ++n) t.window_half[n] = 0.5f * t.window[n];
  t.window64_half.resize(n_fft);  // the float64 window librosa.stft multiplies the frames by (scipy get_window), halved
  for (int n = 0; n < n_fft; ++n) t.window64_half[n] = 0.5 * (0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)n_fft));

  // np.fft.rfftfreq(n_fft, 1/sr)
  const double val = 1.0 / ((double)n_fft * (1.0 / (double)sr));
  std::vector<double> fftfreqs(t.n_bins);
  for (int k = 0; k < t.n_bins; ++k) fftfreqs[k] = (double)k * val;

  // Slaney mel bank: triangles in float64, stored float32, then scaled by the float64 area norm
  {
    const double fmax = p.fmax > 0.f ? (double)p.fmax : (double)sr / 2.0;
    std::vector<double> mels = linspace(hz_to_mel((double)p.fmin), hz_to_mel(fmax), p.n_mels + 2);
    std::vector<double> mel_f(mels.size());
    for (size_t i = 0; i < mels.size(); ++i) mel_f[i] = mel_to_hz(mels[i]);
    t.mel.assign((size_t)p.n_mels * t.n_bins, 0.f);
    for (int i = 0; i < p.n_mels; ++i) {
      const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
      const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
      for (int k = 0; k < t.n_bins; ++k) {
        const double lower = -(mel_f[i] - fftfreqs[k]) / fd0;
        const double upper = (mel_f[i + 2] - fftfreqs[k]) / fd1;
        const float w = (float)std::max(0.0, std::min(lower, upper));
        t.mel[(size_t)i * t.n_bins + k] = (float)((double)w * enorm);
      }
    }
    t.mel_pack = pack_rows(t.mel, p.n_mels, t.n_bins, true);
    t.mel_pack.data.resize(t.mel_pack.data.size() + 8 * 64, 0.f);  // slack for the kernel's batched tap reads
  }
  // DCT-II ortho rows
  {
    t.dct.assign((size_t)p.n_mfcc * p.n_mels, 0.f);
    const double M = (double)p.n_mels;
    for (int c = 0; c < p.n_mfcc; ++c)
      for (int m = 0; m < p.n_mels; ++m) {
        double d = std::cos(M_PI * (double)c * (2.0 * m + 1.0) / (2.0 * M)) * std::sqrt(2.0 / M);
        if (c == 0) d *= std::sqrt(0.5);
        t.dct[(size_t)c * p.n_mels + m] = (float)d;
      }
    t.dct_pack = pack_rows(t.dct, p.n_mfcc, p.n_mels, false);
    // the kernel runs the DCT on the float64 MFMA: identical silent / clamped frames would otherwise repeat
    // one float32 rounding error in every frame and bias the pooled means by ~1e-4
    // K is walked in groups of 16 mels; within a group lane (i = l & 15, k4 = l >> 4) owns the four consecutive
    // mels 16g + 4*k4 + s (s = MFMA step), so its B operand is ONE 16-byte load of the frame-major log-mel row
    // and its A operands are four consecutive doubles of this pack: [row block][group][lane][s]
    const int groups = t.mpad / 16;
    t.dct_groups = groups;
    t.dct_pack64.assign((size_t)t.dct_pack.row_blocks * groups * 64 * 4, 0.0);
    t.dct_pack32.assign(t.dct_pack64.size(), 0.f);
    t.dct_rowsum.assign((size_t)t.dct_pack.row_blocks * 16, 0.0);
    for (int c = 0; c < p.n_mfcc; ++c)
      for (int m = 0; m < p.n_mels; ++m) t.dct_rowsum[c] += (double)t.dct[(size_t)c * p.n_mels + m];
    for (int rb = 0; rb < t.dct_pack.row_blocks; ++rb)
      for (int g = 0; g < groups; ++g)
        for (int l = 0; l < 64; ++l)
          for (int s = 0; s < 4; ++s) {
            const int c = rb * 16 + (l & 15), m = 16 * g + 4 * (l >> 4) + s;
            if (c < p.n_mfcc && m < p.n_mels) {
              double d = std::cos(M_PI * (double)c * (2.0 * m + 1.0) / (2.0 * M)) * std::sqrt(2.0 / M);
              if (c == 0) d *= std::sqrt(0.5);
              t.dct_pack64[(((size_t)rb * groups + g) * 64 + l) * 4 + s] = d;
              t.dct_pack32[(((size_t)rb * groups + g) * 64 + l) * 4 + s] = t.dct[(size_t)c * p.n_mels + m];
            }
          }
  }
  // spectral_contrast bands (librosa defaults fmin 200 Hz, 6 octave bands + the remainder, quantile 0.02): the bins
  // that get sorted and how many are averaged at each end -- the index arithmetic of librosa.feature.spectral_contrast
  if (p.rich) {
    const int n_bands = 6;
    std::vector<double> octa(n_bands + 2, 0.0);
    for (int i = 0; i <= n_bands; ++i) octa[i + 1] = 200.0 * std::pow(2.0, (double)i);
    for (int k = 0; k <= n_bands; ++k) {
      int a0 = -1, b0 = -1;  // first / last bin with f_low <= freq <= f_high
      for (int b = 0; b < t.n_bins; ++b)
        if (fftfreqs[b] >= octa[k] && fftfreqs[b] <= octa[k + 1]) { if (a0 < 0) a0 = b; b0 = b; }
      if (a0 < 0) { t.rich_tab.clear(); break; }  // band above Nyquist: the caller rejects the preset
      int first = a0 - (k > 0 ? 1 : 0);
      int last = (k == n_bands) ? t.n_bins - 1 : b0;  // inclusive end of `current_band`
      const int n_cur = last - first + 1;
      const int end = (k < n_bands) ? last : last + 1;  // bands below the top drop their last bin before sorting
      const int n_avg = std::max((int)std::nearbyint(0.02 * (double)n_cur), 1);
      t.rich_tab.push_back(first); t.rich_tab.push_back(end); t.rich_tab.push_back(n_avg);
    }
  }
  // piptrack frequency mask
  {
    const double fmin = 150.0, fmax = std::min(4000.0, (double)sr / 2.0);
    t.pip_kmin = t.n_bins; t.pip_kmax = t.n_bins;
    for (int k = 0; k < t.n_bins; ++k) if (fftfreqs[k] >= fmin) { t.pip_kmin = k; break; }
    for (int k = 0; k < t.n_bins; ++k) if (!(fftfreqs[k] < fmax)) { t.pip_kmax = k; break; }
    t.pip_kmin = std::max(t.pip_kmin, 1);
    t.pip_kmax = std::min(t.pip_kmax, t.n_bins - 1);
  }
  // 100 chroma banks
  if (p.chroma) {
    const int nc = kChroma;
    t.chroma.assign((size_t)kTunings * nc * t.n_bins, 0.f);
    t.chroma_groups = (t.n_bins + 15) / 16;
    t.chroma_pack.assign((size_t)kTunings * t.chroma_groups * 256, 0.f);
    std::vector<double> freqs = linspace(0.0, (double)sr, n_fft, false);
    std::vector<double> frq(n_fft), bw(n_fft), col(nc);
    std::vector<float> bank((size_t)nc * t.n_bins);
    for (int ti = 0; ti < kTunings; ++ti) {
      const double a440 = 440.0 * std::pow(2.0, tuning_value(ti) / nc);
      for (int j = 1; j < n_fft; ++j) frq[j] = nc * std::log2(freqs[j] / (a440 / 16));
      frq[0] = frq[1] - 1.5 * nc;
      for (int j = 0; j + 1 < n_fft; ++j) bw[j] = std::max(frq[j + 1] - frq[j], 1.0);
      bw[n_fft - 1] = 1.0;
      for (int j = 0; j < t.n_bins; ++j) {
        double ss = 0.0;
        for (int c = 0; c < nc; ++c) {
          double D = frq[j] - (double)c;
          D = std::fmod(D + 6.0 + 120.0, 12.0);
          if (D < 0) D += 12.0;
          D -= 6.0;
          const double q = 2.0 * D / bw[j];
          col[c] = std::exp(-0.5 * q * q);
          ss += col[c] * col[c];
        }
        double len = std::sqrt(ss);
        if (len < 2.2250738585072014e-308) len = 1.0;
        const double oq = (frq[j] / nc - 5.0) / 2.0;
        const double scale = std::exp(-0.5 * oq * oq);
        for (int c = 0; c < nc; ++c) {
          const int dst = ((c - 3) % nc + nc) % nc;  // np.roll(wts, -3, axis=0)
          bank[(size_t)dst * t.n_bins + j] = (float)(col[c] / len * scale);
        }
      }
      std::copy(bank.begin(), bank.end(), t.chroma.begin() + (size_t)ti * nc * t.n_bins);
      float* cp = t.chroma_pack.data() + (size_t)ti * t.chroma_groups * 256;
      for (int g = 0; g < t.chroma_groups; ++g)
        for (int l = 0; l < 64; ++l)
          for (int q = 0; q < 4; ++q) {
            const int row = l & 15, k = 16 * g + 4 * (l >> 4) + q;
            if (row < nc && k < t.n_bins) cp[((size_t)g * 64 + l) * 4 + q] = bank[(size_t)row * t.n_bins + k];
          }
    }
  }
  return true;
}

}  // namespace dysf
