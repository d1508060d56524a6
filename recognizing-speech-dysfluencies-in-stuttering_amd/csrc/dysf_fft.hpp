// dysf_fft.hpp -- register-resident Stockham FFT used by the STFT kernel (gfx950), plus the
// index helpers the host emulation test (csrc/emu_fft_test.cpp) shares with the device code.
//
// A real frame of NFFT samples is packed as NC = NFFT/2 complex points z[n] = x[2n] + i x[2n+1],
// transformed with a complex FFT, and un-packed into the NFFT/2+1 bins of the real spectrum.
// Every lane owns 16 complex points in registers (L = NC/16 lanes per frame, 64/L frames per wave);
// stages are Stockham autosort passes (radix 16 / 4 / 2 in registers) and the only data movement
// between stages is a transpose through a wave-private LDS buffer.  Replaces the pocketfft rfft call
// inside librosa.stft (reached from pipeline1.py:216 and :227).
//
// Complex values are two-float vectors so that on gfx950 every complex add is one v_pk_add_f32 (the
// multiplications by -i fold into its op_sel / neg modifiers) and every complex multiply is one
// v_pk_mul_f32 + one v_pk_fma_f32.
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

#if defined(__clang__)
typedef float cf __attribute__((ext_vector_type(2)));
typedef double cd __attribute__((ext_vector_type(2)));   // the parity presets run the same FFT in float64
#define DYSF_SWAP(z) __builtin_shufflevector((z), (z), 1, 0)
#else
typedef float cf __attribute__((vector_size(8)));
typedef double cd __attribute__((vector_size(16)));
typedef int cf_mask_ __attribute__((vector_size(8)));
typedef long long cd_mask_ __attribute__((vector_size(16)));
#define DYSF_SWAP(z) dysf_swap_(z)
static inline cf dysf_swap_(cf z) { return __builtin_shuffle(z, (cf_mask_){1, 0}); }
static inline cd dysf_swap_(cd z) { return __builtin_shuffle(z, (cd_mask_){1, 0}); }
#endif

template <class C> struct scalar_of;
template <> struct scalar_of<cf> { typedef float type; };
template <> struct scalar_of<cd> { typedef double type; };

DYSF_HD cf mk(float re, float im) { cf r = {re, im}; return r; }
// z * (-i) = (im, -re)
template <class C>
DYSF_HD C mul_mi(C z) { C s = DYSF_SWAP(z); C r = {s[0], -s[1]}; return r; }

// z * (wr + i wi) with compile-time-constant twiddles: the (-wi, wi) vector is a literal
template <class C>
DYSF_HD C cmul_const(C z, double wr, double wi) {
  typedef typename scalar_of<C>::type T;
  const C a = {(T)wr, (T)wr}, b = {(T)-wi, (T)wi};
  return z * a + DYSF_SWAP(z) * b;
}
DYSF_HD cf cmul(cf z, cf w);
// float32 form on the device: the run-time complex multiply below with a literal pair (its operand selections are the ones
// the co-execution check of add_mi clears; the generic form above compiles to a v_pk_mul_f32 with swapped halves)
DYSF_HD cf cmul_const(cf z, double wr, double wi) {
#if defined(__HIP_DEVICE_COMPILE__)
  const cf w = {(float)wr, (float)wi};
  return cmul(z, w);
#else
  const cf a = {(float)wr, (float)wr}, b = {(float)-wi, (float)wi};
  return z * a + DYSF_SWAP(z) * b;
#endif
}
// z * w with a run-time twiddle held in a register pair
DYSF_HD cf cmul(cf z, cf w) {
#if defined(__HIP_DEVICE_COMPILE__)
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(z), "v"(w));  // (zr*wr, zi*wr)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
      : "=v"(r) : "v"(z), "v"(w), "v"(t));                                     // (-zi*wi + t.lo, zr*wi + t.hi)
  return r;
#else
  cf r = {z[0] * w[0] - z[1] * w[1], z[0] * w[1] + z[1] * w[0]};
  return r;
#endif
}
DYSF_HD cd cmul(cd z, cd w) {
  cd r = {z[0] * w[0] - z[1] * w[1], z[0] * w[1] + z[1] * w[0]};
  return r;
}

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

// a + (-i) b and a - (-i) b without materialising (-i) b: the swap of b's halves and the sign ride on the operand
// modifiers of ONE packed instruction.  It is a v_pk_fma_f32 with the constant 1.0 (b * 1 + a is exactly a + b), not a
// v_pk_add_f32: on gfx950 v_pk_add_f32 / v_pk_mul_f32 whose op_sel picks DIFFERENT halves of their two sources
// (op_sel:[0,1] op_sel_hi:[1,0]) return wrong values while a float16 / bfloat16 MFMA of another wave -- any other kernel
// sharing the CU, e.g. a GEMM of another stream -- is executing (tools/ubench/coexec.hip: 160 000 of 524 288 results differ
// beside v_mfma_f32_16x16x32_f16 chains, none beside float32 MFMAs; v_pk_fma_f32 with the same operand selection is not
// affected, nor are equal-half selections, broadcasts and neg modifiers).  `make check-isa` keeps the form out of the
// library.
DYSF_HD cf add_mi(cf a, cf b) {
#if defined(__HIP_DEVICE_COMPILE__)
  cf r;
  asm("v_pk_fma_f32 %0, %2, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b));  // (ar + bi, ai - br)
  return r;
#else
  return a + mul_mi(b);
#endif
}
DYSF_HD cf sub_mi(cf a, cf b) {
#if defined(__HIP_DEVICE_COMPILE__)
  cf r;
  asm("v_pk_fma_f32 %0, %2, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b));  // (ar - bi, ai + br)
  return r;
#else
  return a - mul_mi(b);
#endif
}
DYSF_HD cd add_mi(cd a, cd b) { return a + mul_mi(b); }
DYSF_HD cd sub_mi(cd a, cd b) { return a - mul_mi(b); }

// ---- small DFTs on register arrays, elements at stride ST -------------------------------------
template <int ST, class C>
DYSF_HD void dft2(C* z) {
  const C a = z[0], b = z[ST];
  z[0] = a + b;
  z[ST] = a - b;
}

// forward radix-4: X[u] = sum_q x[q] (-i)^(q u)
template <int ST, class C>
DYSF_HD void dft4(C* z) {
  const C t0 = z[0] + z[2 * ST], t1 = z[0] - z[2 * ST];
  const C t2 = z[ST] + z[3 * ST], t3 = z[ST] - z[3 * ST];
  z[0] = t0 + t2;
  z[ST] = add_mi(t1, t3);      // t1 - i t3
  z[2 * ST] = t0 - t2;
  z[3 * ST] = sub_mi(t1, t3);  // t1 + i t3
}

// forward radix-16 on 16 contiguous registers, output in natural order.
// n = q + 4p, k = s + 4u:  a[q][s] = sum_p x[q+4p] W4^(ps);  X[s+4u] = sum_q a[q][s] W16^(qs) W4^(qu).
template <class C>
DYSF_HD void dft16(C* z) {
  constexpr double C1 = 0.92387953251128675613;  // cos(pi/8)
  constexpr double S1 = 0.38268343236508977173;  // sin(pi/8)
  constexpr double H = 0.70710678118654752440;   // sqrt(1/2)
  // inner radix-4 over p for each q: elements q, q+4, q+8, q+12 (stride 4); result index s at q+4s
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4<4>(z + q);
  // twiddles W16^(q s) on element q + 4s
  z[1 + 4] = cmul_const(z[1 + 4], C1, -S1);    // q1 s1: W^1
  z[1 + 8] = cmul_const(z[1 + 8], H, -H);      // q1 s2: W^2
  z[1 + 12] = cmul_const(z[1 + 12], S1, -C1);  // q1 s3: W^3
  z[2 + 4] = cmul_const(z[2 + 4], H, -H);      // q2 s1: W^2
  z[2 + 8] = mul_mi(z[2 + 8]);                 // q2 s2: W^4 = -i
  z[2 + 12] = cmul_const(z[2 + 12], -H, -H);   // q2 s3: W^6
  z[3 + 4] = cmul_const(z[3 + 4], S1, -C1);    // q3 s1: W^3
  z[3 + 8] = cmul_const(z[3 + 8], -H, -H);     // q3 s2: W^6
  z[3 + 12] = cmul_const(z[3 + 12], -C1, S1);  // q3 s3: W^9
  // outer radix-4 over q for each s: elements 4s + {0,1,2,3} (stride 1); result index u at 4s + u
#pragma unroll
  for (int s = 0; s < 4; ++s) dft4<1>(z + 4 * s);
  // element 4s + u now holds X[s + 4u]; transpose the 4x4 register tile to natural order
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int u = s + 1; u < 4; ++u) {
      const C t = z[4 * s + u]; z[4 * s + u] = z[4 * u + s]; z[4 * u + s] = t;
    }
}

// The butterflies of one Stockham stage on a lane's 16 points.  Point p = b + NB*r belongs to butterfly b (NB = 16/R
// butterflies per lane), input slot r.
template <int NC, int S, class C>
DYSF_HD void fft_butterflies(C* z) {
  constexpr int R = FftGeom<NC>::radix(S);
  constexpr int NB = kPts / R;
  if (R == 16) {
    dft16(z);
  } else if (R == 4) {
#pragma unroll
    for (int b = 0; b < NB; ++b) dft4<NB>(z + b);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) dft2<NB>(z + b);
  }
}

// One Stockham stage: twiddles tw[p] on the points p >= NB (ignored for stage 0), then the butterflies.
template <int NC, int S, class C>
DYSF_HD void fft_stage(C* z, const C* tw) {
  constexpr int R = FftGeom<NC>::radix(S);
  constexpr int NB = kPts / R;
  if (S > 0) {
#pragma unroll
    for (int p = NB; p < kPts; ++p) z[p] = cmul(z[p], tw[p]);
  }
  fft_butterflies<NC, S>(z);
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

// Real-spectrum un-packing for the bin pair (k, NC-k).  The FFT input is pre-scaled by 1/2 (folded into the
// window table), so with Zk = Z[k], Zm = Z[(NC-k)%NC] and V = -i exp(-2 pi i k / (2 NC)) = (wi, -wr):
//   X[k] = (Zk + conj Zm) + V (Zk - conj Zm),   X[NC-k] = conj((Zk + conj Zm) - V (Zk - conj Zm)).
// Returns |X[k]|^2 and |X[NC-k]|^2.
// float64 form used by the parity presets: the two spectrum values themselves (the caller rounds them to complex64)
DYSF_HD void rfft_pair(cd zk, cd zm, cd v, cd* xk, cd* xm) {
  const cd cm = {zm[0], -zm[1]};
  const cd e = zk + cm, o = zk - cm;
  const cd t = cmul(o, v);
  const cd x = e + t, y = e - t;
  *xk = x;
  cd yc = {y[0], -y[1]};
  *xm = yc;
}
DYSF_HD void rfft_pair_power(cf zk, cf zm, cf v, float* pk, float* pm) {
  const cf cm = {zm[0], -zm[1]};
  const cf e = zk + cm, o = zk - cm;
  const cf t = cmul(o, v);
  const cf x = e + t, y = e - t;
  const cf x2 = x * x, y2 = y * y;
  *pk = x2[0] + x2[1];
  *pm = y2[0] + y2[1];
}

}  // namespace dysf
