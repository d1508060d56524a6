// dysf_fft.hpp -- register-resident Stockham FFT used by the STFT kernel (gfx950), plus the
// index helpers the host emulation test (csrc/emu_fft_test.cpp) shares with the device code.
//
// A real frame of NFFT samples is packed as NC = NFFT/2 complex points z[n] = x[2n] + i x[2n+1],
// transformed with a complex FFT, and un-packed into the NFFT/2+1 bins of the real spectrum.
// Every lane owns 16 complex points in registers (L = NC/16 lanes per frame, 64/L frames per wave);
// stages are Stockham autosort passes (radix 16 / 4 / 2 in registers) and the only data movement
// between stages is a transpose through a wave-private LDS buffer.  Replaces the pocketfft rfft call
// inside librosa.stft (reached from pipeline1.py:216 and :227).
#pragma once

#if defined(__HIPCC__)
#define DYSF_HD __host__ __device__ __forceinline__
#else
#define DYSF_HD inline
#endif

namespace dysf {

constexpr int kPts = 16;  // complex points per lane
constexpr int kTunings = 100;  // histogram bins of librosa.pitch_tuning at resolution 0.01
constexpr int kChroma = 12;

// LDS exchange-buffer padding: one float of padding every 16 keeps the stride-16 writes of the first
// radix-16 pass on distinct banks.
DYSF_HD int pad_idx(int i) { return i + (i >> 4); }

template <int NC>
struct FftGeom {
  static_assert(NC == 128 || NC == 256 || NC == 512 || NC == 1024, "n_fft must be 256..2048");
  static constexpr int L = NC / kPts;            // lanes per frame
  static constexpr int FPW = 64 / L;             // frames per wave
  static constexpr int NSTAGE = (NC == 256) ? 2 : 3;
  static constexpr int XSTRIDE = NC + NC / 16;   // floats per frame in the exchange buffer
  DYSF_HD static constexpr int radix(int s) {
    return s == 0 ? 16 : (NC == 128 ? (s == 1 ? 4 : 2) : (s == 1 ? 16 : (NC == 512 ? 2 : 4)));
  }
  // product of the radices before stage s
  DYSF_HD static constexpr int ns(int s) { return s == 0 ? 1 : (s == 1 ? 16 : 16 * radix(1)); }
};

// ---- small DFTs on register arrays, elements at stride ST -------------------------------------
template <int ST>
DYSF_HD void dft2(float* re, float* im) {
  float ar = re[0], ai = im[0], br = re[ST], bi = im[ST];
  re[0] = ar + br; im[0] = ai + bi;
  re[ST] = ar - br; im[ST] = ai - bi;
}

// forward radix-4: X[u] = sum_q x[q] (-i)^(q u)
template <int ST>
DYSF_HD void dft4(float* re, float* im) {
  float t0r = re[0] + re[2 * ST], t0i = im[0] + im[2 * ST];
  float t1r = re[0] - re[2 * ST], t1i = im[0] - im[2 * ST];
  float t2r = re[ST] + re[3 * ST], t2i = im[ST] + im[3 * ST];
  float t3r = re[ST] - re[3 * ST], t3i = im[ST] - im[3 * ST];
  re[0] = t0r + t2r;      im[0] = t0i + t2i;
  re[ST] = t1r + t3i;     im[ST] = t1i - t3r;       // t1 - i t3
  re[2 * ST] = t0r - t2r; im[2 * ST] = t0i - t2i;
  re[3 * ST] = t1r - t3i; im[3 * ST] = t1i + t3r;   // t1 + i t3
}

DYSF_HD void cmul(float& r, float& i, float wr, float wi) {
  float tr = r * wr - i * wi;
  i = r * wi + i * wr;
  r = tr;
}

// forward radix-16 on 16 contiguous registers, output in natural order.
// n = q + 4p, k = s + 4u:  a[q][s] = sum_p x[q+4p] W4^(ps);  X[s+4u] = sum_q a[q][s] W16^(qs) W4^(qu).
DYSF_HD void dft16(float* re, float* im) {
  constexpr float C1 = 0.92387953251128675613f;  // cos(pi/8)
  constexpr float S1 = 0.38268343236508977173f;  // sin(pi/8)
  constexpr float H = 0.70710678118654752440f;   // sqrt(1/2)
  // inner radix-4 over p for each q: elements q, q+4, q+8, q+12 (stride 4); result index s at q+4s
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4<4>(re + q, im + q);
  // twiddles W16^(q s) on element q + 4s
  cmul(re[1 + 4], im[1 + 4], C1, -S1);   // q1 s1: W^1
  cmul(re[1 + 8], im[1 + 8], H, -H);     // q1 s2: W^2
  cmul(re[1 + 12], im[1 + 12], S1, -C1); // q1 s3: W^3
  cmul(re[2 + 4], im[2 + 4], H, -H);     // q2 s1: W^2
  { float t = re[2 + 8]; re[2 + 8] = im[2 + 8]; im[2 + 8] = -t; }  // q2 s2: W^4 = -i
  cmul(re[2 + 12], im[2 + 12], -H, -H);  // q2 s3: W^6
  cmul(re[3 + 4], im[3 + 4], S1, -C1);   // q3 s1: W^3
  cmul(re[3 + 8], im[3 + 8], -H, -H);    // q3 s2: W^6
  cmul(re[3 + 12], im[3 + 12], -C1, S1); // q3 s3: W^9
  // outer radix-4 over q for each s: elements 4s + {0,1,2,3} (stride 1); result index u at 4s + u
#pragma unroll
  for (int s = 0; s < 4; ++s) dft4<1>(re + 4 * s, im + 4 * s);
  // element 4s + u now holds X[s + 4u]; transpose the 4x4 register tile to natural order
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int u = s + 1; u < 4; ++u) {
      float t = re[4 * s + u]; re[4 * s + u] = re[4 * u + s]; re[4 * u + s] = t;
      t = im[4 * s + u]; im[4 * s + u] = im[4 * u + s]; im[4 * u + s] = t;
    }
}

// One Stockham stage on a lane's 16 points.  Point p = b + NB*r belongs to butterfly b (NB = 16/R
// butterflies per lane), input slot r.  tw{r,i}[p] is the stage twiddle (ignored for stage 0).
template <int NC, int S>
DYSF_HD void fft_stage(float* re, float* im, const float* twr, const float* twi) {
  constexpr int R = FftGeom<NC>::radix(S);
  constexpr int NB = kPts / R;
  if (S > 0) {
#pragma unroll
    for (int p = NB; p < kPts; ++p) cmul(re[p], im[p], twr[p], twi[p]);
  }
  if (R == 16) {
    dft16(re, im);
  } else if (R == 4) {
#pragma unroll
    for (int b = 0; b < NB; ++b) dft4<NB>(re + b, im + b);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) dft2<NB>(re + b, im + b);
  }
}

// where stage S stores point p of lane ls (natural order after the last stage)
template <int NC, int S>
DYSF_HD int fft_out_index(int ls, int p) {
  constexpr int R = FftGeom<NC>::radix(S);
  constexpr int NB = kPts / R;
  constexpr int NS = FftGeom<NC>::ns(S);
  constexpr int L = FftGeom<NC>::L;
  const int b = p % NB, r = p / NB;
  const int j = ls + L * b;
  const int k = j % NS;
  return (j / NS) * NS * R + k + r * NS;
}

// which element every stage loads into point p of lane ls
template <int NC>
DYSF_HD int fft_in_index(int ls, int p) { return ls + FftGeom<NC>::L * p; }

// twiddle exponent numerator / denominator for stage S: exp(-2 pi i num / den)
template <int NC, int S>
DYSF_HD void fft_twiddle_frac(int ls, int p, int* num, int* den) {
  constexpr int R = FftGeom<NC>::radix(S);
  constexpr int NB = kPts / R;
  constexpr int NS = FftGeom<NC>::ns(S);
  constexpr int L = FftGeom<NC>::L;
  const int b = p % NB, r = p / NB;
  const int j = ls + L * b;
  *num = r * (j % NS);
  *den = NS * R;
}

// Real-spectrum un-packing for the bin pair (k, NC-k): Zk = (a,b), Zm = Z[(NC-k)%NC] = (c,d),
// W = exp(-2 pi i k / (2 NC)) = (wr, wi).  Returns |X[k]|^2 and |X[NC-k]|^2.
DYSF_HD void rfft_pair_power(float a, float b, float c, float d, float wr, float wi, float* pk, float* pm) {
  const float er = 0.5f * (a + c), ei = 0.5f * (b - d);
  const float orr = 0.5f * (a - c), oi = 0.5f * (b + d);
  const float tr = wr * oi + wi * orr;
  const float ti = -(wr * orr - wi * oi);
  const float xr = er + tr, xi = ei + ti;
  const float yr = er - tr, yi = ei - ti;
  *pk = xr * xr + xi * xi;
  *pm = yr * yr + yi * yi;
}

}  // namespace dysf
