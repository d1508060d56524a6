// dysf_kernels.hip -- gfx950 kernels for the dysfluency feature path.
//
// librosa.feature.mfcc -> delta -> delta(order=2) -> stat_pair (+ chroma_stft), pipeline1.py:216-228, as two kernels on
// one stream (three with chroma).  The HBM intermediates are the log-mel spectrogram (clip-global top_db needs the whole
// clip before the DCT) and, for chroma presets, the power spectrogram (librosa's second STFT is not recomputed).
//   dysf_stft_mel_kernel  grid (tile groups, clips), small LDS footprint -> 4 workgroups per CU:
//            coalesced PCM -> LDS tile; per-wave register FFT (dysf_fft.hpp; float64 in the parity presets) ->
//            power-spectrum tile in LDS; banded Slaney-mel GEMM on v_mfma_f32_16x16x4_f32 (split-K schedule);
//            10*log10 -> log-mel [T][n_mels] in the workspace; clip max of the mel power by atomicMax; chroma presets:
//            spectrogram rows and row maxima to the workspace; metrics launches: per-frame spectral flatness
//   dysf_mfcc_kernel      one workgroup per clip: clip-global top_db clamp, DCT-II on the MFMA, Savitzky-Golay deltas,
//            frame-level rows [3*n_mfcc][T], pooled mean/std (NumPy's float32 summation order in the parity presets)
//   dysf_chroma_kernel    (chroma only) one workgroup per clip: piptrack peak picking on the stored spectrogram,
//            median / histogram tuning decision, chroma GEMM on MFMA against the selected one of 100 banks, max-norm,
//            mean/std
//   dysf_rich_kernel      (rich presets, main.py:738-754) spectral contrast, zero-crossing rate, RMS, centroid
//   dysf_metrics_kernel   (metrics launches, pipeline1.py:151-174) snr_db and the mean spectral flatness
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "dysf_fft.hpp"
#include "dysf_kernels.h"

namespace dysf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in order; this only stops the compiler from reordering them.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Single IEEE float32 operations that the compiler may not contract into an fma (HIP compiles with
// -ffp-contract=fast, and clang's __fadd_rn / __fmul_rn are plain operators; its __fsqrt_rn is the approximate
// v_sqrt_f32).  Used wherever a NumPy float32 rounding sequence is reproduced step by step.
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double dadd_rn(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ double dmul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }   // correctly rounded (hipcc default)
__device__ __forceinline__ float sqrt_rn(float a) { return sqrtf(a); }        // correctly rounded (hipcc default)

// ---------------------------------------------------------------------------------------------
// PCM tile staging in two halves so the HBM latency hides under the previous tile's FFT:
//   pcm_prefetch  issues 16-byte coalesced loads of the samples [s0, s0 + count) of a clip into registers
//   pcm_commit    converts and writes them to the LDS tile as float32 (int16 stays unscaled: see pcm_store_group)
// The tile is stored from the 16-byte-aligned element a0 <= off + s0 onwards, so element (off + s) sits at
// tile[(off + s) - a0] and every group lands on an aligned ds_write_b128; samples outside [0, n) are zero.
template <int PCM>
struct PcmStage {
  static constexpr int EPG = (PCM == DYSF_PCM_I16) ? 8 : 4;            // elements per 16-byte group
  static constexpr int MAXG = (PCM == DYSF_PCM_I16) ? 2 : 3;           // pipelined groups per thread; any tile
  int4 g[MAXG];                                                        // beyond MAXG*NT groups is staged in place
};

// All positions are 32-bit and relative to the clip's first sample (clips are shorter than 2^31 samples); `base`
// points at that sample.  off_lo = off & (EPG - 1) carries the clip's alignment inside its 16-byte group.
template <int PCM>
__device__ __forceinline__ int pcm_tile_origin(int off_lo, int s0) {
  return ((off_lo + s0) & ~(PcmStage<PCM>::EPG - 1)) - off_lo;  // floor to a 16-byte group, also for negative values
}

template <int PCM>
__device__ __forceinline__ int4 pcm_load_group(const char* base, int n, int r) {
  constexpr int EPG = PcmStage<PCM>::EPG;
  constexpr int ES = (PCM == DYSF_PCM_I16) ? 2 : 4;
  int4 v = make_int4(0, 0, 0, 0);
  if (r >= 0 && r + EPG <= n) {
    v = *(const int4*)(base + (long long)r * ES);
  } else if (r + EPG > 0 && r < n) {  // straddles a clip edge: element-wise, zero outside
    if (PCM == DYSF_PCM_I16) {
      const unsigned short* b = (const unsigned short*)base;
      unsigned e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = (r + i >= 0 && r + i < n) ? (unsigned)b[r + i] : 0u;
      v = make_int4((int)(e[0] | (e[1] << 16)), (int)(e[2] | (e[3] << 16)), (int)(e[4] | (e[5] << 16)),
                    (int)(e[6] | (e[7] << 16)));
    } else {
      const int* b = (const int*)base;
      int e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = (r + i >= 0 && r + i < n) ? b[r + i] : 0;
      v = make_int4(e[0], e[1], e[2], e[3]);
    }
  }
  return v;
}

// Returns true if a non-finite sample was seen (float input only).
template <int PCM>
__device__ __forceinline__ bool pcm_store_group(float* tile, int gi, int4 v) {
  if (PCM == DYSF_PCM_I16) {
    // the 2^-15 of int16 -> [-1, 1) is folded into the window table (KArgs::window_i16; exact, a power of two)
    float4 lo, hi;
    lo.x = (float)(short)(v.x & 0xffff); lo.y = (float)(v.x >> 16);
    lo.z = (float)(short)(v.y & 0xffff); lo.w = (float)(v.y >> 16);
    hi.x = (float)(short)(v.z & 0xffff); hi.y = (float)(v.z >> 16);
    hi.z = (float)(short)(v.w & 0xffff); hi.w = (float)(v.w >> 16);
    *(float4*)(tile + gi * 8) = lo;
    *(float4*)(tile + gi * 8 + 4) = hi;
    return false;
  }
  float4 f;
  f.x = __int_as_float(v.x); f.y = __int_as_float(v.y); f.z = __int_as_float(v.z); f.w = __int_as_float(v.w);
  *(float4*)(tile + gi * 4) = f;
  return !(isfinite(f.x) && isfinite(f.y) && isfinite(f.z) && isfinite(f.w));
}

template <int PCM, int NT>
__device__ __forceinline__ void pcm_prefetch(PcmStage<PCM>& st, const char* base, int n, int r0, int ngroups) {
  constexpr int EPG = PcmStage<PCM>::EPG;
  constexpr int ES = (PCM == DYSF_PCM_I16) ? 2 : 4;
  // A tile that lies inside the clip (all but the first and last one or two of a clip; the test is wave-uniform) takes
  // plain 16-byte loads: the group index is clamped instead of guarded -- the surplus lanes reload the last group and
  // pcm_commit does not store them -- so the path has no exec-mask branches at all.
  if (r0 >= 0 && r0 + ngroups * EPG <= n) {
#pragma unroll
    for (int j = 0; j < PcmStage<PCM>::MAXG; ++j) {
      const int gi = min((int)threadIdx.x + j * NT, ngroups - 1);
      st.g[j] = *(const int4*)(base + (long long)(r0 + gi * EPG) * ES);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < PcmStage<PCM>::MAXG; ++j) {
    const int gi = threadIdx.x + j * NT;
    st.g[j] = (gi < ngroups) ? pcm_load_group<PCM>(base, n, r0 + gi * PcmStage<PCM>::EPG) : make_int4(0, 0, 0, 0);
  }
}

template <int PCM, int NT>
__device__ __forceinline__ bool pcm_commit(const PcmStage<PCM>& st, float* tile, const char* base, int n, int r0,
                                           int ngroups) {
  bool bad = false;
#pragma unroll
  for (int j = 0; j < PcmStage<PCM>::MAXG; ++j) {
    const int gi = threadIdx.x + j * NT;
    if (gi < ngroups) bad |= pcm_store_group<PCM>(tile, gi, st.g[j]);
  }
  for (int gi = threadIdx.x + PcmStage<PCM>::MAXG * NT; gi < ngroups; gi += NT)  // oversized tiles
    bad |= pcm_store_group<PCM>(tile, gi, pcm_load_group<PCM>(base, n, r0 + gi * PcmStage<PCM>::EPG));
  return bad;
}

// Two-pass (re, then im) transpose of a lane's 16 points through the frame's exchange buffer.
template <int NC, int SPREV>
__device__ __forceinline__ void fft_exchange(cf* z, float* xb, int ls) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int p = 0; p < kPts; ++p) xb[pad_idx(fft_out_index<NC, SPREV>(ls, p))] = z[p][h];
    wave_lds_sync();
#pragma unroll
    for (int p = 0; p < kPts; ++p) z[p][h] = xb[pad_idx(fft_in_index<NC>(ls, p))];
    wave_lds_sync();
  }
}

// Windowed frame (in the LDS PCM tile at float offset frame_off) -> power spectrum row srow[0..NC].
// xb and srow both live in the calling wave's private LDS region.  win holds the window pre-scaled by 1/2
// (see rfft_pair_power), ptw the un-packing twiddles V[k].
struct FftTabs {     // LDS tables shared by the workgroup (consecutive lanes -> consecutive words)
  const float* win;  // [2*NC] window, pre-scaled by 1/2
  const float* ptw;  // [NC/2+1][2] un-packing twiddles V[k]
  const cf* t1;      // [16][L] stage-1 twiddles, point-major
};

// copy the tables from global memory (once per workgroup)
template <int NC, int NT, int PCM>
__device__ __forceinline__ void load_fft_tabs(float* dst, const KArgs& a, FftTabs& t) {
  constexpr int L = FftGeom<NC>::L;
  float* win = dst;
  float* ptw = win + 2 * NC;
  cf* t1 = (cf*)(ptw + ((NC + 2 + 3) & ~3));
  const float* wsrc = (PCM == DYSF_PCM_I16) ? a.window_i16 : a.window;
  for (int i = threadIdx.x; i < 2 * NC; i += NT) win[i] = wsrc[i];
  for (int i = threadIdx.x; i < NC + 2; i += NT) ptw[i] = a.post_tw[i];
  // stage-1 twiddles turn point p by W^(p (ls % 16)): one column per ls % 16 (lanes 16 apart read the same word: a broadcast)
  constexpr int T1L = L < 16 ? L : 16;
  for (int i = threadIdx.x; i < kPts * T1L; i += NT) t1[i] = ((const cf*)a.stage_tw)[(i % T1L) * kPts + (i / T1L)];
  t.win = win; t.ptw = ptw; t.t1 = t1;
}
template <int NC>
constexpr int fft_tabs_floats() { return 2 * NC + ((NC + 2 + 3) & ~3) + 2 * kPts * (FftGeom<NC>::L < 16 ? FftGeom<NC>::L : 16); }

template <int NC>
__device__ __forceinline__ void wave_frame_power(const float* tile, int frame_off, float* xb, float* srow,
                                                 int ls, const FftTabs& tb, const KArgs& a) {
  const float* win = tb.win;
  const float* ptw = tb.ptw;
  using G = FftGeom<NC>;
  constexpr int L = G::L;
  constexpr int LAST = G::NSTAGE - 1;
  cf z[kPts];
#pragma unroll
  for (int p = 0; p < kPts; ++p) {
    const int n = ls + L * p;
    const cf x = {tile[frame_off + 2 * n], tile[frame_off + 2 * n + 1]};
    z[p] = x * *(const cf*)(win + 2 * n);
  }
  fft_stage<NC, 0>(z, (const cf*)nullptr);
  fft_exchange<NC, 0>(z, xb, ls);
  {
    cf t1[kPts];
    constexpr int T1L = L < 16 ? L : 16;
#pragma unroll
    for (int p = 0; p < kPts; ++p) t1[p] = tb.t1[p * T1L + (ls % T1L)];
    fft_stage<NC, 1>(z, t1);
  }
  if (G::NSTAGE == 3) {
    fft_exchange<NC, 1>(z, xb, ls);
    cf t2[kPts];
    const cf* tw2 = (const cf*)a.stage_tw + (size_t)L * kPts + ls * kPts;
#pragma unroll
    for (int p = 0; p < kPts; ++p) t2[p] = tw2[p];
    fft_stage<NC, 2>(z, t2);
  }
  // gather the (k, NC-k) bin pairs this lane un-packs: k = ls + L*i, i < 8 (+ the self-paired bin NC/2 in lane 0)
  cf zk[9], zm[9];
  if (NC == 256) {
    // two-stage layout: register p of lane ls already holds Z[ls + 16 p], and Z[NC - k] sits in register 15 - i
    // of lane 16 - ls of the same frame (lane 0 pairs with itself, shifted by one register) -> two DPP moves
    // per value instead of a round trip through the exchange buffer
    const unsigned long long lane0 = __ballot(ls == 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      zk[i] = z[i];
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // lane ls <- lane (16 - ls) & 15 of its row: row_mirror, then row_ror:1 (two DPP moves)
        // the select is inline asm on purpose: written as ?: the compiler folds it into a dynamically indexed
        // register array (a 32-way compare chain per value)
        float offer;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(offer) : "v"(z[15 - i][h]), "v"(z[(16 - i) & 15][h]), "s"(lane0));
        // every lane of a row has a source lane under row_mirror / row_ror, so no "old" value is needed (mov_dpp
        // leaves it undefined: no zero-initialising v_mov in front of each DPP move)
        const int m = __builtin_amdgcn_mov_dpp(__float_as_int(offer), 0x140, 0xf, 0xf, true);
        zm[i][h] = __int_as_float(__builtin_amdgcn_mov_dpp(m, 0x121, 0xf, 0xf, true));
      }
    }
    zk[8] = z[8];
    zm[8] = z[8];
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int p = 0; p < kPts; ++p) xb[pad_idx(fft_out_index<NC, LAST>(ls, p))] = z[p][h];
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int k = (i == 8) ? NC / 2 : ls + L * i;
        zk[i][h] = xb[pad_idx(k)];
        zm[i][h] = xb[pad_idx((NC - k) % NC)];
      }
      wave_lds_sync();
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int k = (i == 8) ? NC / 2 : ls + L * i;
    float pk, pm;
    rfft_pair_power(zk[i], zm[i], *(const cf*)(ptw + 2 * k), &pk, &pm);
    if (i < 8 || ls == 0) {
      srow[k] = pk;
      srow[NC - k] = pm;
    }
  }
  wave_lds_sync();
}

// ---------------------------------------------------------------------------------------------
// Parity presets (KArgs::parity): the same wave FFT in float64.  librosa.stft multiplies the float32 frame by the
// float64 window, transforms in float64 and stores complex64; _spectrogram then takes np.abs(D)**2.  A float64 FFT
// of a different factorisation agrees with pocketfft's to ~1e-16 of the frame's largest value, so the complex64
// roundings agree except for ~1e-8 of the strong bins -- and only strong bins take part in piptrack's decisions.
template <int NC, int SPREV>
__device__ __forceinline__ void fft_exchange64(cd* z, double* xb, int ls) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int p = 0; p < kPts; ++p) xb[pad_idx(fft_out_index<NC, SPREV>(ls, p))] = z[p][h];
    wave_lds_sync();
#pragma unroll
    for (int p = 0; p < kPts; ++p) z[p][h] = xb[pad_idx(fft_in_index<NC>(ls, p))];
    wave_lds_sync();
  }
}

// Correctly rounded a / b for 0 <= a <= b, b in [2^-60, 2^60], and correctly rounded sqrt(x) for x in [1, 2]: the
// reciprocal / reciprocal-square-root instruction plus fma corrections, without the scaling and fix-up steps of the
// general library sequences (half their instructions).  tools/checks/rn_check.hip compares them on the GPU with the
// correctly rounded operators: sqrt over all 2^23 + 1 floats of [1, 2], division over 2^28 random pairs, no mismatch.
__device__ __forceinline__ float div_rn_unit(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  const float q = mul_rn(a, r);
  const float rem = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(rem, r, q);
}
__device__ __forceinline__ float sqrt_rn_12(float x) {
  const float s = __builtin_amdgcn_rsqf(x);
  float g = mul_rn(x, s), h = mul_rn(0.5f, s);
  const float r = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, r, g);
  h = __builtin_fmaf(h, r, h);
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}

// np.abs of one complex64 value as NumPy's SIMD loop computes it (loops_unary_complex.dispatch.c.src, FMA form):
// larger * sqrt(fma(r, r, 1)) with r = smaller / larger, every step rounded to float32; then squared.
__device__ __forceinline__ float cabs2_numpy(float re, float im) {
  const float ar = fabsf(re), ai = fabsf(im);
  const float lg = fmaxf(ar, ai), sm = fminf(ai, ar);
  float ratio;
  if (lg > 8.7e-19f && lg < 1.1e18f) ratio = div_rn_unit(sm, lg);            // every bin that can matter
  else ratio = (lg > 0.f && lg < INFINITY) ? div_rn(sm, lg) : 0.f;
  const float m = mul_rn(sqrt_rn_12(__builtin_fmaf(ratio, ratio, 1.0f)), lg);
  return mul_rn(m, m);
}

// float64 tables of the parity presets, copied into LDS once per workgroup (read from L2 for every frame they cost
// 57 KB per 2048-point frame -- 5.5 GB per 1024-clip launch, which bound the first version of this path).
// Twiddles of a stage are stored as [q][lane] with q counting the points that carry one; a radix-4 stage keeps only its
// NB base twiddles per lane (exponent r = 1) and squares / cubes them in registers (r = 2, 3).
template <int NC, int S>
constexpr int tw64_entries() {   // complex doubles per lane the LDS copy of stage S holds
  return FftGeom<NC>::radix(S) == 4 ? kPts / 4 : kPts - kPts / FftGeom<NC>::radix(S);
}
template <int NC>
constexpr int fft_tabs64_floats() {
  return 4 * NC + 2 * (NC + 2) + 4 * FftGeom<NC>::L * (tw64_entries<NC, 1>() + (FftGeom<NC>::NSTAGE == 3 ? tw64_entries<NC, 2>() : 0));
}
struct FftTabs64 {
  const cd* win;   // [NC] (w[2n], w[2n+1]), pre-scaled by 1/2 (and 2^-15 for int16 PCM)
  const cd* ptw;   // [NC/2+1] un-packing twiddles
  const cd* t1;    // [q][L] stage-1 twiddles
  const cd* t2;    // [q][L] stage-2 twiddles (three-stage sizes)
};
template <int NC, int NT>
__device__ __forceinline__ void load_fft_tabs64(float* dst, const KArgs& a, FftTabs64& t) {
  using G = FftGeom<NC>;
  constexpr int L = G::L;
  cd* win = (cd*)dst;
  cd* ptw = win + NC;
  cd* t1 = ptw + (NC / 2 + 1);
  cd* t2 = t1 + L * tw64_entries<NC, 1>();
  for (int i = threadIdx.x; i < NC; i += NT) win[i] = ((const cd*)a.window64)[i];
  for (int i = threadIdx.x; i < NC / 2 + 1; i += NT) ptw[i] = ((const cd*)a.post_tw64)[i];
  constexpr int NB1 = kPts / G::radix(1);
  for (int i = threadIdx.x; i < L * tw64_entries<NC, 1>(); i += NT)
    t1[i] = ((const cd*)a.stage_tw64)[(i % L) * kPts + NB1 + i / L];
  if (G::NSTAGE == 3) {
    constexpr int NB2 = kPts / G::radix(2);
    for (int i = threadIdx.x; i < L * tw64_entries<NC, 2>(); i += NT)
      t2[i] = ((const cd*)a.stage_tw64)[(size_t)L * kPts + (i % L) * kPts + NB2 + i / L];
  }
  t.win = win; t.ptw = ptw; t.t1 = t1; t.t2 = t2;
}

// twiddles of stage S from the LDS copy, then the butterflies
template <int NC, int S>
__device__ __forceinline__ void fft_stage64(cd* z, const cd* tw, int ls) {
  using G = FftGeom<NC>;
  constexpr int L = G::L, R = G::radix(S), NB = kPts / R;
  if (R == 4) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const cd w1 = tw[b * L + ls];
      const cd w2 = cmul(w1, w1), w3 = cmul(w2, w1);
      z[b + NB] = cmul(z[b + NB], w1);
      z[b + 2 * NB] = cmul(z[b + 2 * NB], w2);
      z[b + 3 * NB] = cmul(z[b + 3 * NB], w3);
    }
  } else {
#pragma unroll
    for (int p = NB; p < kPts; ++p) z[p] = cmul(z[p], tw[(p - NB) * L + ls]);
  }
  fft_butterflies<NC, S>(z);
}

// grow: this frame's row in the spectrogram workspace (null: not kept); returns the frame's largest bin in every lane of
// the frame's lane group (piptrack's threshold) -- both come straight from the registers that hold the new bins.
template <int NC>
__device__ __forceinline__ float wave_frame_power_exact(const float* tile, int frame_off, double* xb, float* srow,
                                                        int ls, const FftTabs64& tb, const KArgs& a, float* grow) {
  using G = FftGeom<NC>;
  constexpr int L = G::L;
  constexpr int LAST = G::NSTAGE - 1;
  cd z[kPts];
#pragma unroll
  for (int p = 0; p < kPts; ++p) {
    const int n = ls + L * p;
    const cd x = {(double)tile[frame_off + 2 * n], (double)tile[frame_off + 2 * n + 1]};
    z[p] = x * tb.win[n];  // one float64 rounding per sample, as numpy's float64(frame) * window
  }
  fft_butterflies<NC, 0>(z);
  fft_exchange64<NC, 0>(z, xb, ls);
  fft_stage64<NC, 1>(z, tb.t1, ls);
  if (G::NSTAGE == 3) {
    fft_exchange64<NC, 1>(z, xb, ls);
    fft_stage64<NC, 2>(z, tb.t2, ls);
  }
  cd zk[9], zm[9];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int p = 0; p < kPts; ++p) xb[pad_idx(fft_out_index<NC, LAST>(ls, p))] = z[p][h];
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int k = (i == 8) ? NC / 2 : ls + L * i;
      zk[i][h] = xb[pad_idx(k)];
      zm[i][h] = xb[pad_idx((NC - k) % NC)];
    }
    wave_lds_sync();
  }
  // the mel product reads up to 32 words past a row's last bin (mel_segment): the words between the rows and after the
  // wave's last row still hold halves of doubles, which need not be finite floats
  for (int k = NC + 1 + ls; k < a.kpad; k += L) srow[k] = 0.f;
  for (int k = ls; k < 40; k += L) srow[a.kpad + k] = 0.f;  // (the next row's head is rewritten below by its own lanes)
  wave_lds_sync();
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int k = (i == 8) ? NC / 2 : ls + L * i;
    cd xk, xm;
    rfft_pair(zk[i], zm[i], tb.ptw[k], &xk, &xm);
    if (i < 8 || ls == 0) {
      const float pk = cabs2_numpy((float)xk[0], (float)xk[1]);       // complex64 rounding, then |.|**2 in float32
      const float pm = cabs2_numpy((float)xm[0], (float)xm[1]);
      srow[k] = pk;
      srow[NC - k] = pm;
      mx = fmaxf(mx, fmaxf(pk, pm));
      if (grow) { grow[k] = pk; grow[NC - k] = pm; }   // consecutive lanes, consecutive (or descending) bins: 256-byte segments
    }
  }
  if (grow) for (int k = NC + 1 + ls; k < a.s_stride; k += L) grow[k] = 0.f;   // padding columns (the chroma GEMM reads them)
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  wave_lds_sync();
  return mx;
}

// ---------------------------------------------------------------------------------------------
// piptrack, second half: one local maximum (bin k, power x0, neighbours xm / x1) -> interpolated magnitude and the
// histogram bin of its pitch residual.  Follows librosa.piptrack / pitch_tuning step by step (numba promotes the integer
// literals of the parabolic interpolation to float64; the residual is float32; np.histogram's edges are i * 0.01 - 0.5 in
// float64 with a closed last bin).
template <int NC>
__device__ __forceinline__ void peak_to_mag_bin(int k, float x0, float xm, float x1, const KArgs& a, float* mag_out, int* bin_out) {
  const float dsum = add_rn(x1, xm), ddif = sub_rn(x1, xm);
  const double pa = (double)dsum - 2.0 * (double)x0;
  const double pb = (double)ddif / 2.0;
  const float shift = (fabs(pb) >= fabs(pa)) ? 0.f : (float)(-pb / pa);
  const float avg = mul_rn(ddif, 0.5f);
  *mag_out = add_rn(x0, mul_rn(mul_rn(0.5f, avg), shift));
  const float pitch = (float)((((double)k + (double)shift) * (double)a.sr) / (double)(2 * NC));
  // pitch_tuning residual, float32 like numpy; log2 is correctly rounded via float64
  const float octs = (float)log2((double)div_rn(pitch, 27.5f));
  const float x12 = mul_rn(12.0f, octs);
  float r = x12 - floorf(x12);
  if (r >= 0.5f) r = sub_rn(r, 1.0f);
  const double rd = (double)r;
  int b = (int)floor((rd + 0.5) * 100.0);
  b = b < 0 ? 0 : (b > 99 ? 99 : b);
  while (b > 0 && rd < dadd_rn(dmul_rn((double)b, 0.01), -0.5)) --b;
  while (b < 99 && rd >= dadd_rn(dmul_rn((double)(b + 1), 0.01), -0.5)) ++b;
  *bin_out = b;
}

// r-th smallest (0-based) of the n positive floats in v (bit order = numeric order), by most-significant-byte-first
// radix selection: four passes over v, each with a 256-bin LDS histogram of the values that still match the prefix.
// All threads call; hist = 256 ints of LDS scratch, ctl = 2 ints.
// reg[j] caches v[threadIdx.x + j * NT] for j < KR (the four passes then read registers, not memory).
template <int NT, int KR>
__device__ float block_select(const float* v, const float (&reg)[KR], int n, int r, int* hist, int* ctl) {
  unsigned prefix = 0u, mask = 0u;
  for (int pass = 3; pass >= 0; --pass) {
    for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KR; ++j) {
      const unsigned u = __float_as_uint(reg[j]);
      if ((int)threadIdx.x + j * NT < n && (u & mask) == prefix) atomicAdd(&hist[(u >> (8 * pass)) & 255u], 1);
    }
    for (int i = threadIdx.x + KR * NT; i < n; i += NT) {
      const unsigned u = __float_as_uint(v[i]);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> (8 * pass)) & 255u], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // wave 0: lane l owns bins 4l .. 4l+3; exclusive prefix over the lanes by shuffles
      const int l = threadIdx.x;
      const int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
      const int tot = h0 + h1 + h2 + h3;
      int inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (l >= o) inc += t; }
      const int before = inc - tot;
      if (r >= before && r < inc) {  // exactly one lane: the r-th value falls into one of its four bins
        int d = 4 * l, c = before;
        if (r >= c + h0) { c += h0; ++d; if (r >= c + h1) { c += h1; ++d; if (r >= c + h2) { c += h2; ++d; } } }
        ctl[0] = d; ctl[1] = r - c;
      }
    }
    __syncthreads();
    prefix |= (unsigned)ctl[0] << (8 * pass);
    mask |= 0xffu << (8 * pass);
    r = ctl[1];
    __syncthreads();
  }
  return __uint_as_float(prefix);
}

// The selection again with ONE workgroup barrier per radix pass instead of four (the chroma kernel runs four workgroups per
// CU in lockstep: barriers are what its tuning phase waits for).  hist3 = three 256-bin histograms used in rotation -- all
// zero on entry, with a barrier since they were zeroed: the pass that counts into buffer b also zeroes the buffer two passes
// ahead, which nobody reads or updates between the two barriers around that pass.  Every wave scans the histogram itself
// (four bins per lane, shuffle prefix), so there is no broadcast through LDS.  The top byte of positive floats in a narrow
// range takes two or three values: that pass adds wave-aggregated counts (one atomic per distinct digit and wave) instead of
// 64 colliding ones.  Leaves hist3 dirty.
template <int NT, int KR>
__device__ float block_select3(const float* v, const float (&reg)[KR], int n, int r, int* hist3) {
  const int lane = threadIdx.x & 63;
  unsigned prefix = 0u, mask = 0u;
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    const int pass = 3 - it;
    int* h = hist3 + (it % 3) * 256;
    auto count = [&](bool active, unsigned u) {
      const unsigned digit = (u >> (8 * pass)) & 255u;
      active = active && (u & mask) == prefix;
      if (it == 0) {
        unsigned long long todo = __ballot(active);
        while (todo) {   // wave-uniform
          const int src = __builtin_ctzll(todo);
          const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)digit, src);
          const unsigned long long m = __ballot(active && digit == d0);
          if (lane == src) atomicAdd(&h[d0], __popcll(m));
          todo &= ~m;
        }
      } else if (active) {
        atomicAdd(&h[digit], 1);
      }
    };
#pragma unroll
    for (int j = 0; j < KR; ++j) count((int)threadIdx.x + j * NT < n, __float_as_uint(reg[j]));
    for (int i0 = KR * NT; i0 < n; i0 += NT) {   // (uniform trip count: count() holds wave-wide operations)
      const int i = i0 + (int)threadIdx.x;
      count(i < n, __float_as_uint(v[i < n ? i : n - 1]));
    }
    __syncthreads();
    if (it < 2) { int* hz = hist3 + ((it + 2) % 3) * 256; for (int i = threadIdx.x; i < 256; i += NT) hz[i] = 0; }
    const int h0 = h[4 * lane], h1 = h[4 * lane + 1], h2 = h[4 * lane + 2], h3 = h[4 * lane + 3];
    const int tot = h0 + h1 + h2 + h3;
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    const int before = inc - tot;
    int d = 4 * lane, c = before;
    if (r >= c + h0) { c += h0; ++d; if (r >= c + h1) { c += h1; ++d; if (r >= c + h2) { c += h2; ++d; } } }
    const int owner = __builtin_ctzll(__ballot(r >= before && r < inc));   // exactly one lane holds the r-th value's bin
    prefix |= (unsigned)__builtin_amdgcn_readlane(d, owner) << (8 * pass);
    mask |= 0xffu << (8 * pass);
    r = __builtin_amdgcn_readlane(r - c, owner);
  }
  return __uint_as_float(prefix);
}

// The same selection with nothing cached (metrics kernel).
template <int NT>
__device__ float block_select(const float* v, int n, int r, int* hist, int* ctl) {
  unsigned prefix = 0u, mask = 0u;
  for (int pass = 3; pass >= 0; --pass) {
    for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) {
      const unsigned u = __float_as_uint(v[i]);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> (8 * pass)) & 255u], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
      const int tot = h0 + h1 + h2 + h3;
      int inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (l >= o) inc += t; }
      const int before = inc - tot;
      if (r >= before && r < inc) {
        int d = 4 * l, c = before;
        if (r >= c + h0) { c += h0; ++d; if (r >= c + h1) { c += h1; ++d; if (r >= c + h2) { c += h2; ++d; } } }
        ctl[0] = d; ctl[1] = r - c;
      }
    }
    __syncthreads();
    prefix |= (unsigned)ctl[0] << (8 * pass);
    mask |= 0xffu << (8 * pass);
    r = ctl[1];
    __syncthreads();
  }
  return __uint_as_float(prefix);
}

// ---------------------------------------------------------------------------------------------
// One segment of the mel schedule: K-steps [kbeg, kbeg + 4*nsteps) of row block rb against column block cb.
struct MelSeg { int rb, cb, kbeg, nsteps, poff, kind, slot, nparts; };  // kind 0 whole, 1 partial writer, 2 finisher

template <int NC, int FPW, int FPI>
__device__ __forceinline__ f32x4 mel_segment(const float* regions, const KArgs& a, const float* mel_src,
                                             const MelSeg& g, int lane) {
  int f = g.cb * 16 + (lane & 15);
  f = f < FPI ? f : FPI - 1;
  const float* srow = regions + (f / FPW) * a.rstride + (f % FPW) * a.kpad + g.kbeg + (lane >> 4);
  const float4* wp = (const float4*)(mel_src + g.poff) + lane;  // the L2-resident pack: four K-steps per 16-byte load
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains hide the MFMA latency
  // batches of 8 K-steps: the 16 LDS / L2 reads of the NEXT batch are in flight while the 8 MFMAs of this one run.  No
  // masks: the host pads every segment's taps with zeros to a whole batch, and the spectrum reads that run past bin NC stay
  // inside the workgroup's LDS, which holds finite values only (regions are zeroed at start-up; a clip with non-finite
  // samples is rejected anyway).
  // (two register sets used alternately: the loads of batch s + 8 are issued before the MFMAs of batch s)
  float a0[8], b0[8], a1[8], b1[8];
  auto fetch = [&](float (&av)[8], float (&bv)[8], int s) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 v = wp[((s >> 2) + q) * 64];
      av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) bv[u] = srow[4 * (s + u)];
  };
  auto multiply = [&](const float (&av)[8], const float (&bv)[8]) {
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], acc1, 0, 0, 0);
    }
  };
  fetch(a0, b0, 0);
  for (int s = 0; s < g.nsteps; s += 16) {
    if (s + 8 < g.nsteps) fetch(a1, b1, s + 8);
    multiply(a0, b0);
    if (s + 8 >= g.nsteps) break;
    if (s + 16 < g.nsteps) fetch(a0, b0, s + 16);
    multiply(a1, b1);
  }
  return acc0 + acc1;
}

// librosa.power_to_db computes float32(10) * np.log10(x): two float32 roundings.  log10 = v_log_f32 (log2) times
// log10(2) as a two-term constant (about 1 ulp); cells at the amin floor take the host-computed constant
// (-100 exactly for 1e-10) so that silent cells are exact.
__device__ __forceinline__ float power_to_db(float x, const KArgs& a) {
  const float l2 = __builtin_amdgcn_logf(x);
  const float lg = fmaf(l2, -1.432098883924482e-08f, l2 * 0.3010300099849701f);
  return (x <= a.amin) ? a.db_amin : mul_rn(10.0f, lg);
}

// finished mel tile -> four consecutive mels of one frame per lane -> one 16-byte store into the frame-major
// log-mel workspace [T][lm_stride]; running max of the mel POWER (log10 is monotonic: kernel B turns the clip's largest
// power into the exact float32 dB value the reference's clamp starts from)
template <int FPI>
__device__ __forceinline__ void mel_epilogue(f32x4 acc, const MelSeg& g, const KArgs& a, float* lm, int f0, int T,
                                             int lane, float& lmax) {
  const int f = g.cb * 16 + (lane & 15);
  const int t = f0 + f;
  if (f < FPI && t < T) {
    const int m0 = g.rb * 16 + 4 * (lane >> 4);
    float4 v;
    v.x = power_to_db(acc[0], a); v.y = power_to_db(acc[1], a); v.z = power_to_db(acc[2], a); v.w = power_to_db(acc[3], a);
    *(float4*)(lm + (size_t)t * a.lm_stride + m0) = v;
    // rows past n_mels have all-zero taps (acc = 0), so the clip's largest mel power needs no row mask
    lmax = fmaxf(fmaxf(lmax, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
  }
}

// ---------------------------------------------------------------------------------------------
#ifdef DYSF_STAMP
// diagnostic build only (make stamp): lane 0 of wave 0 adds the cycles since its previous stamp to slot i
#define STAMP_INIT unsigned long long stamp_last = __builtin_readcyclecounter()
#define STAMP(i)                                                              \
  do {                                                                        \
    if (threadIdx.x == 0 && a.dbg_stamps) {                                   \
      const unsigned long long now_ = __builtin_readcyclecounter();           \
      a.dbg_stamps[(size_t)((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + (i)] += now_ - stamp_last; \
      stamp_last = now_;                                                      \
    }                                                                         \
  } while (0)
struct Stamper {   // the same for code in device functions (kernel A-exact)
  unsigned long long* slot;
  unsigned long long last;
  __device__ __forceinline__ void init(const KArgs& a) {
    slot = (threadIdx.x == 0 && a.dbg_stamps) ? a.dbg_stamps + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 : nullptr;
    last = __builtin_readcyclecounter();
  }
  __device__ __forceinline__ void mark(int i) {
    if (slot) { const unsigned long long now = __builtin_readcyclecounter(); slot[i] += now - last; last = now; }
  }
};
#else
#define STAMP_INIT do {} while (0)
#define STAMP(i) do {} while (0)
struct Stamper {
  __device__ __forceinline__ void init(const KArgs&) {}
  __device__ __forceinline__ void mark(int) {}
};
#endif

// =============================================================================================
// Kernel A: PCM -> STFT power -> mel -> dB.  grid = (tile groups, clips); NT threads, NT/64 waves.
// EX: float64 FFT (parity presets).  CH: chroma presets (piptrack peak picking, spectrogram rows kept for kernel C) and
// the single-clip debug dumps; the throughput instantiation (CH = false) carries none of that code.
template <int NC, int PCM, int NT, bool EX, bool CH>
__global__ __launch_bounds__(NT, (EX ? 2 : 4)) void dysf_stft_mel_kernel(const KArgs a) {
  using G = FftGeom<NC>;
  constexpr int L = G::L, FPW = G::FPW;
  constexpr int NW = NT / 64;
  constexpr int FPI = FPW * NW;  // frames per tile
  constexpr int EPG = PcmStage<PCM>::EPG;
  extern __shared__ __align__(16) float lds[];
  float* regions = lds + a.lds_region;
  float* tile = lds + a.lds_pcm;
  float* misc = lds + a.lds_misc;   // [0..15] wave max, [16] peak count (int), then schedule, slots, peak stage
  int* imisc = (int*)misc;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: per-wave bases stay in SGPRs
  const int q = lane / L;
  // odd frames of a wave take their FFT lane roles rotated by L/2: the two frames that share a 32-lane LDS access
  // group then read the PCM tile 16 banks apart (frames are a multiple of 32 banks apart whenever hop % 32 == 0)
  const int ls = (FPW == 4) ? ((lane + ((q & 1) ? L / 2 : 0)) % L) : lane % L;
  STAMP_INIT;
  const int clip = blockIdx.y;
  const long long off = a.offsets[clip];
  const long long n = a.offsets[clip + 1] - off;
  const int T = (int)(1 + n / a.hop);
  const int ti0 = blockIdx.x * a.tiles_per_wg;
  // rejected clips and empty tile groups (uniform exit); a metrics launch still wants the spectral flatness of short clips
  if (n <= 0 || (T < 9 && !(CH && a.ws_flat)) || T > a.t_max || ti0 * FPI >= T) return;
  const int ntiles = min(a.tiles_per_wg, (T - ti0 * FPI + FPI - 1) / FPI);

  float* wregion = regions + wave * a.rstride;
  float* xb = wregion + q * G::XSTRIDE;
  double* xb64 = (double*)wregion + q * G::XSTRIDE;  // EX: the exchange buffer holds doubles (rstride is sized for it)
  float* srow_own = wregion + q * a.kpad;
  FftTabs tb;
  FftTabs64 tb64;
  if (EX) load_fft_tabs64<NC, NT>(lds + a.lds_tables, a, tb64);
  else load_fft_tabs<NC, NT, PCM>(lds + a.lds_tables, a, tb);
  // the mel product reads a few words past a spectrum row's last bin (see mel_segment): no stale LDS there
  for (int i = tid * 4; i < NW * a.rstride; i += NT * 4) *(float4*)(regions + i) = float4{0.f, 0.f, 0.f, 0.f};
  int* sched = imisc + 32;
  for (int i = tid; i < NW * kMelSchedStride; i += NT) sched[i] = a.mel_sched[i];
  const int* wsched = sched + wave * kMelSchedStride;
  float* slots = misc + 32 + 8 * kMelSchedStride;
  f32x4 pend = {0.f, 0.f, 0.f, 0.f};
  float* lm = a.ws_logmel + (size_t)clip * a.ws_logmel_stride;

  float lmax = 0.f;  // largest mel power seen by this thread
  bool bad = false;
  const int tile_samples = (FPI - 1) * a.hop + 2 * NC;
  PcmStage<PCM> stage;
  const char* pbase = (const char*)a.pcm + off * (PCM == DYSF_PCM_I16 ? 2 : 4);
  const int n32 = (int)n, off_lo = (int)(off & (EPG - 1));
  float* sdst = a.ws_S ? a.ws_S + (size_t)clip * a.ws_S_stride : nullptr;
  auto tile_geom = [&](int ti, int* r0, int* shift, int* ngroups) {
    const int s0 = ti * FPI * a.hop - NC;
    *r0 = pcm_tile_origin<PCM>(off_lo, s0);
    *shift = s0 - *r0;
    *ngroups = (s0 + tile_samples - *r0 + EPG - 1) / EPG;
  };
  int r0 = 0, shift = 0, ngroups = 0;
  tile_geom(ti0, &r0, &shift, &ngroups);
  pcm_prefetch<PCM, NT>(stage, pbase, n32, r0, ngroups);
  bad |= pcm_commit<PCM, NT>(stage, tile, pbase, n32, r0, ngroups);
  __syncthreads();
  STAMP(0);  // setup: tables, schedule, first tile staged
  for (int j = 0; j < ntiles; ++j) {
    const int f0 = (ti0 + j) * FPI;
    const int cur_shift = shift;
    if (j + 1 < ntiles) {  // loads for the next tile fly while this one is transformed
      tile_geom(ti0 + j + 1, &r0, &shift, &ngroups);
      pcm_prefetch<PCM, NT>(stage, pbase, n32, r0, ngroups);
    }
    {
      const int fl = wave * FPW + q;  // frame within tile
      const int t_own = f0 + fl;
      float* grow = (CH && sdst && t_own < T) ? sdst + (size_t)t_own * a.s_stride : nullptr;
      float mx_own = 0.f;
      if (EX) mx_own = wave_frame_power_exact<NC>(tile + cur_shift, fl * a.hop, xb64, srow_own, ls, tb64, a, grow);
      else wave_frame_power<NC>(tile + cur_shift, fl * a.hop, xb, srow_own, ls, tb, a);
      STAMP(1);  // prefetch issue + FFT
      if (CH && a.ws_rowmax) {  // piptrack's threshold is 0.1 * the frame's largest bin: kernel C picks the peaks from the stored rows
        if (!EX) {
          for (int k = lane % L; k <= NC; k += L) mx_own = fmaxf(mx_own, srow_own[k]);
#pragma unroll
          for (int o = L / 2; o > 0; o >>= 1) mx_own = fmaxf(mx_own, __shfl_xor(mx_own, o));
        }
        if (lane % L == 0 && t_own < T) a.ws_rowmax[(size_t)clip * a.t_max + t_own] = mx_own;
      }
      STAMP(7);  // row maximum
      if (CH && a.ws_flat) {
        // librosa.feature.spectral_flatness of this frame (pipeline1.py:168-174): geometric over arithmetic mean of
        // max(1e-10, |D|**2) across the bins
        float slog = 0.f, ssum = 0.f;
        for (int k = lane % L; k <= NC; k += L) {
          const float v = fmaxf(1e-10f, srow_own[k]);
          slog += logf(v); ssum += v;
        }
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) { slog += __shfl_xor(slog, o); ssum += __shfl_xor(ssum, o); }
        if (lane % L == 0 && t_own < T)
          a.ws_flat[(size_t)clip * a.t_max + t_own] = div_rn(expf(slog / (float)(NC + 1)), ssum / (float)(NC + 1));
      }
      if (CH && sdst && !EX) {  // float32-FFT chroma presets: the wave's own FPW rows from LDS, 256-byte segments
#pragma unroll
        for (int qq = 0; qq < FPW; ++qq) {
          const int t = f0 + wave * FPW + qq;
          if (t < T) {
            const float* src = wregion + qq * a.kpad;
            float* dst = sdst + (size_t)t * a.s_stride;
            for (int k = lane; k < a.s_stride; k += 64) dst[k] = (k <= NC) ? src[k] : 0.f;
          }
        }
      }
    }
    STAMP(14);  // spectrogram rows to the workspace
    __syncthreads();  // S tile complete; every wave is done reading the PCM tile
    STAMP(2);  // barrier
    if (j + 1 < ntiles) bad |= pcm_commit<PCM, NT>(stage, tile, pbase, n32, r0, ngroups);
    if (CH && a.dbg_S) {
      for (int i = tid; i < FPI * (NC + 1); i += NT) {
        const int f = i / (NC + 1), k = i % (NC + 1);
        if (f0 + f < T) a.dbg_S[(size_t)(f0 + f) * (NC + 1) + k] = regions[(f / FPW) * a.rstride + (f % FPW) * a.kpad + k];
      }
    }
    // banded mel GEMM on MFMA: D[16 mels x 16 frames] += Wpack[16 x 4] * S[4 x 16], following this wave's
    // slice of the host-built schedule (row blocks with long bands are split along K over several waves)
    STAMP(3);  // commit
    MelSeg fin;
    fin.kind = 0;
    const int nseg = __builtin_amdgcn_readfirstlane(wsched[kMelMaxSeg * 8]);
    for (int i = 0; i < nseg; ++i) {
      MelSeg g;
      {
        const int4 lo = *(const int4*)(wsched + i * 8), hi = *(const int4*)(wsched + 4 + i * 8);
        // the schedule is per wave: make it scalar so the segment loop and its addressing run on the SALU
        g.rb = __builtin_amdgcn_readfirstlane(lo.x); g.cb = __builtin_amdgcn_readfirstlane(lo.y);
        g.kbeg = __builtin_amdgcn_readfirstlane(lo.z); g.nsteps = __builtin_amdgcn_readfirstlane(lo.w);
        g.poff = __builtin_amdgcn_readfirstlane(hi.x); g.kind = __builtin_amdgcn_readfirstlane(hi.y);
        g.slot = __builtin_amdgcn_readfirstlane(hi.z); g.nparts = __builtin_amdgcn_readfirstlane(hi.w);
      }
      const f32x4 acc = mel_segment<NC, FPW, FPI>(regions, a, a.mel_pack, g, lane);
      if (g.kind == 1) {
        *(f32x4*)(slots + (g.slot * 64 + lane) * 4) = acc;
      } else if (g.kind == 2) {
        pend = acc;
        fin = g;
      } else {
        mel_epilogue<FPI>(acc, g, a, lm, f0, T, lane, lmax);
      }
    }
    STAMP(4);  // mel segments
    __syncthreads();  // PCM tile of the next iteration is in place; S regions are free again
    STAMP(5);  // barrier
    if (fin.kind == 2) {  // finish the split row block: add the other waves' partial sums, then log + store
      for (int pidx = 0; pidx < fin.nparts; ++pidx) pend += *(const f32x4*)(slots + ((fin.slot + pidx) * 64 + lane) * 4);
      mel_epilogue<FPI>(pend, fin, a, lm, f0, T, lane, lmax);
    }
  }
  STAMP(6);  // finisher epilogues
  // clip-global maximum of the mel power (for the top_db clamp) and the non-finite flag
  lmax = wave_max(lmax);
  if (lane == 0) misc[wave] = lmax;
  if (PCM == DYSF_PCM_F32 && bad) a.ws_flags[clip] = 1;
  __syncthreads();
  if (tid == 0) {
    float m = misc[0];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, misc[w]);
    atomicMax(a.ws_clipmax + clip, __float_as_uint(fmaxf(m, 0.f)));  // non-negative floats order like their bits
  }
}

// =============================================================================================
// Kernel A-exact: kernel A for the parity presets at n_fft 2048 -- the reference's own parameterisation
// (librosa.feature.mfcc / chroma_stft defaults, pipeline1.py:216-228).  Same arithmetic as kernel A's float64 path (bit-equal
// results), laid out so that TWO workgroups share a CU (kernel A's 140 KB of LDS allow one: 2 waves per SIMD):
//   * no PCM tile: a wave owns one frame, and every lane loads its 16 complex points straight from global memory (the
//     next tile's loads are issued before the mel phase and land during it);
//   * the float64 exchange buffer is HALF a frame (512 doubles per wave): v_permlane32_swap first trades registers 0..7 of
//     lanes 32..63 for registers 8..15 of lanes 0..31, after which registers 0..7 of ALL lanes hold the lower half of the
//     stage's output and go through the buffer in one round, registers 8..15 in a second one;
//   * the last transpose is not one: Z[k] already sits in register k / 64 of lane k % 64, and its un-packing partner
//     Z[NC - k] comes from lane 64 - ls through ds_bpermute_b32 (no LDS memory);
//   * stage-1 twiddles depend on ls % 16 only: 240 table entries instead of 960;
//   * |X|^2 as NumPy computes it is evaluated branch-free on packed float32 pairs; a wave re-does a group of bins with the
//     general sequence only when one of its bins lies outside the range the short sequence is exact for (silent frames);
//   * chroma presets: piptrack's local maxima are picked here from the spectrum row in LDS (kernel C then reads the stored
//     spectrogram once, for the chroma product, instead of twice).
// 512 threads, a tile is 8 frames (one per wave); the mel product, its schedule and the epilogue are kernel A's.
// kExRegion (dysf_kernels.h) floats of LDS per wave: 544 doubles of exchange buffer, later the 1025-bin spectrum row + zero
// tail (% 32 == 8: the eight rows the mel product reads together start on different banks)
constexpr int kExTabFloats = 4 * 1024 + 4 * 513 + 4 * 15 * 16 + 4 * 4 * 64;
int stft_exact2048_table_floats() { return kExTabFloats; }
struct ExTabs {
  const cd* win;   // [1024] (w[2n], w[2n+1]) / 2 (times 2^-15 for int16 PCM)
  const cd* ptw;   // [513] un-packing twiddles
  const cd* t1;    // [15][16] stage-1 twiddles of point p = q + 1, lane ls % 16
  const cd* t2;    // [4][64] stage-2 base twiddles (r = 1) of butterfly b, lane ls
};
// 4-byte-per-lane loads of sample pairs; a clip may start at an odd sample of the flat buffer (2-byte alignment)
typedef int i32_a2 __attribute__((aligned(2)));
typedef int i32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
template <int PCM>
struct FramePcm { int v[PCM == DYSF_PCM_I16 ? 16 : 32]; };

// this lane's 16 complex points (32 samples) of the frame that starts at sample s_t of the clip; zeros outside [0, n)
template <int PCM>
__device__ __forceinline__ void frame_prefetch(FramePcm<PCM>& st, const char* pbase, int n, bool live, int s_t, int ls) {
  constexpr int NC = 1024;
  if (live && s_t >= 0 && s_t + 2 * NC <= n) {   // wave-uniform: the frame lies inside the clip
    if (PCM == DYSF_PCM_I16) {
      const char* p0 = pbase + 2 * (s_t + 2 * ls);
#pragma unroll
      for (int p = 0; p < kPts; ++p) st.v[p] = *(const i32_a2*)(p0 + 256 * p);
    } else {
      const char* p0 = pbase + 4 * (s_t + 2 * ls);
#pragma unroll
      for (int p = 0; p < kPts; ++p) {
        const i32x2_a4 w = *(const i32x2_a4*)(p0 + 512 * p);
        st.v[2 * p] = w[0]; st.v[2 * p + 1] = w[1];
      }
    }
    return;
  }
  // edge frames: clamped (always valid) loads, zeroed afterwards where the position lies outside the clip
#pragma unroll
  for (int p = 0; p < kPts; ++p) {
    const int i0 = s_t + 2 * (ls + 64 * p);
    const bool in0 = live && i0 >= 0 && i0 < n, in1 = live && i0 + 1 >= 0 && i0 + 1 < n;
    const int c0 = min(max(i0, 0), n - 1), c1 = min(max(i0 + 1, 0), n - 1);
    if (PCM == DYSF_PCM_I16) {
      const unsigned short* b = (const unsigned short*)pbase;
      const unsigned lo = b[c0], hi = b[c1];
      st.v[p] = (int)((in0 ? lo : 0u) | ((in1 ? hi : 0u) << 16));
    } else {
      const int* b = (const int*)pbase;
      const int lo = b[c0], hi = b[c1];
      st.v[2 * p] = in0 ? lo : 0;
      st.v[2 * p + 1] = in1 ? hi : 0;
    }
  }
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// lanes 32..63 of a <-> lanes 0..31 of b (v_permlane32_swap_b32 on both halves of the doubles)
__device__ __forceinline__ void swap_lane_halves(double& a, double& b) {
  const u32x2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u32x2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}

// Transpose after stage SPREV (0 or 1) through the half-size buffer.  After the swap register p' < 8 of lane l holds the
// stage's output (source lane l & 31, point p' + 8 (l >> 5)), register p' + 8 the output (source lane 32 + (l & 31), same
// point): output indices below / from NC/2 -- exactly what registers 0..7 / 8..15 of all lanes read back.
template <int SPREV>
__device__ __forceinline__ void ex_exchange(cd* z, double* xb, int lane) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double lo = z[p][h], hi = z[p + 8][h];
      swap_lane_halves(lo, hi);
      z[p][h] = lo; z[p + 8][h] = hi;
    }
  }
  const int sl = lane & 31, pe = 8 * (lane >> 5);
  // padded (pad_idx) position of the lane's first value and the step between its registers
  const int wb = SPREV == 0 ? 17 * sl + pe : (sl >> 4) * 272 + (sl & 15) + 17 * pe;
  constexpr int ws = SPREV == 0 ? 1 : 17;
  const int rb = lane + (lane >> 4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
      for (int p = 0; p < 8; ++p) xb[wb + ws * p] = z[p + 8 * rnd][h];
      wave_lds_sync();
#pragma unroll
      for (int p = 0; p < 8; ++p) z[p + 8 * rnd][h] = xb[rb + 68 * p];
      wave_lds_sync();
    }
  }
}

// np.abs(complex64)**2 of two values at once (the pair of bins a lane un-packs together), NumPy's rounding sequence of
// cabs2_numpy on packed float32; *slow is set when a magnitude lies outside the range div_rn_unit is exact for.
__device__ __forceinline__ cf cabs2_numpy_pair(cf re, cf im, bool* slow) {
#pragma clang fp contract(off)
  const cf ar = __builtin_elementwise_abs(re), ai = __builtin_elementwise_abs(im);
  const cf lg = __builtin_elementwise_max(ar, ai), sm = __builtin_elementwise_min(ai, ar);
  *slow = *slow || !(fminf(lg[0], lg[1]) > 8.7e-19f && fmaxf(lg[0], lg[1]) < 1.1e18f);
  const cf one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
  cf r = {__builtin_amdgcn_rcpf(lg[0]), __builtin_amdgcn_rcpf(lg[1])};
  const cf e = __builtin_elementwise_fma(-lg, r, one);
  r = __builtin_elementwise_fma(e, r, r);
  const cf q = sm * r;
  const cf rem = __builtin_elementwise_fma(-lg, q, sm);
  const cf ratio = __builtin_elementwise_fma(rem, r, q);
  const cf t = __builtin_elementwise_fma(ratio, ratio, one);
  const cf s = {__builtin_amdgcn_rsqf(t[0]), __builtin_amdgcn_rsqf(t[1])};
  cf g = t * s, hh = half * s;
  const cf rr = __builtin_elementwise_fma(-hh, g, half);
  g = __builtin_elementwise_fma(g, rr, g);
  hh = __builtin_elementwise_fma(hh, rr, hh);
  const cf d = __builtin_elementwise_fma(-g, g, t);
  const cf sq = __builtin_elementwise_fma(d, hh, g);
  const cf m = sq * lg;
  return m * m;
}

// librosa.piptrack's candidate bins of one frame, taken from its spectrum row in LDS: local maxima (x[k] > x[k-1],
// x[k] >= x[k+1]) above 0.1 * the frame's largest bin inside [kmin, kmax) -- the same decision kernel C makes on the
// thresholded values (for x0 > ref >= 0 the thresholding of the neighbours cannot change either comparison).  They go to the
// frame's own slots of the workspace (bin, power, neighbours; positions from ballot pop-counts), the count next to them;
// kernel C turns them into (interpolated magnitude, histogram bin) densely.
__device__ __forceinline__ void ex_pick_peaks(const float* srow, float mx, int lane, float4* cand, int* count, int kmin, int kmax) {
  const float ref = mul_rn(0.1f, mx);
  int cnt = 0;   // wave-uniform
  const int nsl = (kmax + 255) >> 8;   // 256-bin slices that hold candidate bins
  for (int i = 0; i < nsl; ++i) {
    const float* row = srow + 256 * i + 4 * lane;   // (the row's zero tail follows bin 1024)
    const float4 c = *(const float4*)row;
    const float left = row[(i == 0 && lane == 0) ? 0 : -1];   // bin 0 has no left neighbour: never a local maximum
    const float right = row[4];
    const float xs[6] = {left, c.x, c.y, c.z, c.w, right};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 256 * i + 4 * lane + j;
      const float x0 = xs[j + 1], xm = xs[j], x1 = xs[j + 2];
      const bool is_peak = (k >= kmin) && (k < kmax) && (x0 > ref) && (x0 > xm) && (x0 >= x1);
      const unsigned long long mask = __ballot(is_peak);
      const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
      if (is_peak) cand[cnt + below] = make_float4(__int_as_float(k), x0, xm, x1);
      cnt += __popcll(mask);
    }
  }
  if (lane == 0) *count = cnt;
}

constexpr int EXG = 3;   // bin pairs un-packed together (9 % EXG == 0): more overlap the reciprocal / square-root chains, fewer fit 128 registers
// One frame: staged samples -> window -> float64 FFT -> complex64 -> |.|^2 row in srow (and in the workspace row grow);
// returns the frame's largest bin in every lane.
template <int PCM>
__device__ __forceinline__ float ex_frame_power(const FramePcm<PCM>& st, double* xb, float* srow, int lane, const ExTabs& tb,
                                                float* grow, int s_stride, bool* bad, Stamper& sp) {
  constexpr int NC = 1024;
  cd z[kPts];
  cf fin = {0.f, 0.f};   // float PCM: one packed fma per sample pair (32 compares would park 32 masks in scalar registers)
#pragma unroll
  for (int p = 0; p < kPts; ++p) {
    cd x;
    if (PCM == DYSF_PCM_I16) {
      x[0] = (double)(short)(st.v[p] & 0xffff); x[1] = (double)(st.v[p] >> 16);
    } else {
      const float f0 = __int_as_float(st.v[2 * p]), f1 = __int_as_float(st.v[2 * p + 1]);
      fin = __builtin_elementwise_fma(cf{f0, f1}, cf{0.f, 0.f}, fin);   // x * 0 summed over the samples is NaN iff one is not finite
      x[0] = (double)f0; x[1] = (double)f1;
    }
    z[p] = x * (tb.win + lane)[64 * p];   // one float64 rounding per sample, as numpy's float64(frame) * window
  }
  if (PCM == DYSF_PCM_F32) *bad = *bad || __ballot(!(fin[0] == 0.f && fin[1] == 0.f)) != 0ull;   // wave-uniform from here on
  fft_butterflies<NC, 0>(z);
  sp.mark(1);   // convert + window + stage 0
  ex_exchange<0>(z, xb, lane);
  sp.mark(2);   // exchange
#pragma unroll
  for (int p = 1; p < kPts; ++p) z[p] = cmul(z[p], tb.t1[(p - 1) * 16 + (lane & 15)]);
  fft_butterflies<NC, 1>(z);
  sp.mark(3);   // stage 1
  ex_exchange<1>(z, xb, lane);
  sp.mark(2);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const cd w1 = tb.t2[b * 64 + lane];
    const cd w2 = cmul(w1, w1), w3 = cmul(w2, w1);
    z[b + 4] = cmul(z[b + 4], w1);
    z[b + 8] = cmul(z[b + 8], w2);
    z[b + 12] = cmul(z[b + 12], w3);
  }
  fft_butterflies<NC, 2>(z);
  sp.mark(4);   // stage 2
  // (the lane index is made opaque again: the un-packing addresses are formed here, after the FFT, not carried across it)
  int ul = lane;
  asm volatile("" : "+v"(ul));
  // register i of lane ls holds Z[ls + 64 i]; Z[NC - k] for k = ls + 64 i is register 15 - i of lane 64 - ls (lane 0:
  // register (16 - i) % 16 of itself)
  const unsigned long long lane0 = __ballot(ul == 0);
  const int src = ((64 - ul) & 63) << 2;
  // the spectrum row's zero tail (the mel product reads up to 35 words past the last bin); the buffer held doubles there
  if (ul < kExRegion - (NC + 1)) srow[NC + 1 + ul] = 0.f;
  if (ul + 64 < kExRegion - (NC + 1)) srow[NC + 1 + 64 + ul] = 0.f;
  float mx = 0.f;
  float* s_up = srow + ul;
  float* s_dn = srow + (NC - 64 * 8) - ul;   // s_dn[64 (8 - i)] = srow[NC - (lane + 64 i)]
  float* g_up = grow + ul;
  float* g_dn = grow + (NC - 64 * 8) - ul;
  // all partner fetches first (32 ds_bpermute_b32 in flight, one wait), then the bin pairs in groups of EXG
  cd zm[9];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double da = z[15 - i][h], db = z[(16 - i) & 15][h];
      int olo, ohi;
      asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(olo) : "v"(__double2loint(da)), "v"(__double2loint(db)), "s"(lane0));
      asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(ohi) : "v"(__double2hiint(da)), "v"(__double2hiint(db)), "s"(lane0));
      zm[i][h] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, ohi), __builtin_amdgcn_ds_bpermute(src, olo));
    }
  }
  zm[8] = z[8];
#pragma unroll
  for (int g0 = 0; g0 < 9; g0 += EXG) {
    cf re[EXG], im[EXG];
#pragma unroll
    for (int u = 0; u < EXG; ++u) {
      const int i = g0 + u;
      cd xk, xm;
      rfft_pair(z[i], zm[i], (tb.ptw + ul)[64 * i], &xk, &xm);   // k = lane + 64 i; i = 8 matters in lane 0 only (k = NC / 2)
      re[u] = cf{(float)xk[0], (float)xm[0]};   // complex64 rounding
      im[u] = cf{(float)xk[1], (float)xm[1]};
    }
    bool slow = false;
    cf pw[EXG];
#pragma unroll
    for (int u = 0; u < EXG; ++u) pw[u] = cabs2_numpy_pair(re[u], im[u], &slow);
    if (__builtin_expect(__ballot(slow) != 0ull, 0)) {   // wave-uniform and rare: a bin at zero / denormal / huge magnitude
#pragma unroll
      for (int u = 0; u < EXG; ++u) pw[u] = cf{cabs2_numpy(re[u][0], im[u][0]), cabs2_numpy(re[u][1], im[u][1])};
    }
#pragma unroll
    for (int u = 0; u < EXG; ++u) {
      const int i = g0 + u;
      // bins k = lane + 64 i (i = 8: k = NC / 2, lane 0 only) and NC - k, addressed from two bases with constant offsets
      if (i < 8 || ul == 0) {
        s_up[64 * i] = pw[u][0];
        s_dn[64 * (8 - i)] = pw[u][1];
        mx = fmaxf(mx, fmaxf(pw[u][0], pw[u][1]));
        if (grow) { g_up[64 * i] = pw[u][0]; g_dn[64 * (8 - i)] = pw[u][1]; }
      }
    }
  }
  if (grow) for (int k = NC + 1 + ul; k < s_stride; k += 64) grow[k] = 0.f;   // padding columns (the chroma GEMM reads them)
  mx = wave_max(mx);
  wave_lds_sync();
  sp.mark(5);   // un-packing, |X|^2, rows
  return mx;
}

template <int PCM>
__global__ __launch_bounds__(512, 4) void dysf_stft_mel_exact2048_kernel(const KArgs a) {
  constexpr int NC = 1024, NT = 512, NW = 8, FPI = 8, FPW = 1;
  extern __shared__ __align__(16) float lds[];
  float* regions = lds + a.lds_region;
  float* misc = lds + a.lds_misc;   // [0..15] wave max, then schedule, split-K slots
  int* imisc = (int*)misc;
  const int tid = threadIdx.x, lane_id = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int clip = blockIdx.y;
  const long long off = a.offsets[clip];
  const long long n = a.offsets[clip + 1] - off;
  const int T = (int)(1 + n / a.hop);
  const int ti0 = blockIdx.x * a.tiles_per_wg;
  if (n <= 0 || (T < 9 && !a.ws_flat) || T > a.t_max || ti0 * FPI >= T) return;
  const int ntiles = min(a.tiles_per_wg, (T - ti0 * FPI + FPI - 1) / FPI);
  Stamper sp;
  sp.init(a);

  float* srow_own = regions + wave * kExRegion;
  double* xb = (double*)srow_own;
  ExTabs tb;
  {
    cd* win = (cd*)(lds + a.lds_tables);
    cd* ptw = win + NC;
    cd* t1 = ptw + (NC / 2 + 1);
    cd* t2 = t1 + 15 * 16;
    for (int i = tid; i < NC; i += NT) win[i] = ((const cd*)a.window64)[i];
    for (int i = tid; i < NC / 2 + 1; i += NT) ptw[i] = ((const cd*)a.post_tw64)[i];
    // stage_tw64: [stage - 1][lane][point]; stage 1 (radix 16) turns point p by W_256^(p (ls % 16)), stage 2 (radix 4) keeps
    // the r = 1 twiddle of each of its four butterflies (points 4 .. 7) and squares / cubes it in registers
    for (int i = tid; i < 15 * 16; i += NT) t1[i] = ((const cd*)a.stage_tw64)[(i % 16) * kPts + 1 + i / 16];
    for (int i = tid; i < 4 * 64; i += NT) t2[i] = ((const cd*)a.stage_tw64)[(size_t)64 * kPts + (i % 64) * kPts + 4 + i / 64];
    tb.win = win; tb.ptw = ptw; tb.t1 = t1; tb.t2 = t2;
  }
  for (int i = tid * 4; i < NW * kExRegion; i += NT * 4) *(float4*)(regions + i) = float4{0.f, 0.f, 0.f, 0.f};
  int* sched = imisc + 32;
  for (int i = tid; i < NW * kMelSchedStride; i += NT) sched[i] = a.mel_sched[i];
  const int* wsched = sched + wave * kMelSchedStride;
  float* slots = misc + 32 + 8 * kMelSchedStride;
  float* lm = a.ws_logmel + (size_t)clip * a.ws_logmel_stride;
  float lmax = 0.f;
  bool bad = false;
  const char* pbase = (const char*)a.pcm + off * (PCM == DYSF_PCM_I16 ? 2 : 4);
  const int n32 = (int)n;
  float* sdst = a.ws_S ? a.ws_S + (size_t)clip * a.ws_S_stride : nullptr;
  FramePcm<PCM> st;
  {
    const int t0 = ti0 * FPI + wave;
    frame_prefetch<PCM>(st, pbase, n32, t0 < T, t0 * a.hop - NC, lane_id);
  }
  __syncthreads();
  sp.mark(0);   // tables, first loads
  for (int j = 0; j < ntiles; ++j) {
    const int f0 = (ti0 + j) * FPI;
    const int t_own = f0 + wave;
    // the lane index is made opaque once per tile: everything derived from it (two dozen LDS / workspace addresses) is then
    // recomputed per frame with one or two integer instructions each instead of being hoisted out of the loop and spilled
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    if (t_own < T) {   // wave-uniform
      float* grow = sdst ? sdst + (size_t)t_own * a.s_stride : nullptr;
      const float mx_own = ex_frame_power<PCM>(st, xb, srow_own, lane, tb, grow, a.s_stride, &bad, sp);
      if (a.ws_rowmax && lane == 0) a.ws_rowmax[(size_t)clip * a.t_max + t_own] = mx_own;
      if (a.ws_cand)
        ex_pick_peaks(srow_own, mx_own, lane, a.ws_cand + ((size_t)clip * a.t_max + t_own) * a.cand_per_frame,
                      a.ws_framecnt + (size_t)clip * a.t_max + t_own, a.pip_kmin, a.pip_kmax);
      if (a.ws_flat) {   // librosa.feature.spectral_flatness of this frame (pipeline1.py:168-174)
        float slog = 0.f, ssum = 0.f;
        for (int k = lane; k <= NC; k += 64) {
          const float v = fmaxf(1e-10f, srow_own[k]);
          slog += logf(v); ssum += v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { slog += __shfl_xor(slog, o); ssum += __shfl_xor(ssum, o); }
        if (lane == 0) a.ws_flat[(size_t)clip * a.t_max + t_own] = div_rn(expf(slog / (float)(NC + 1)), ssum / (float)(NC + 1));
      }
    } else {   // no such frame: the mel product still reads the row
      for (int k = lane * 4; k < kExRegion; k += 256) *(float4*)(srow_own + k) = float4{0.f, 0.f, 0.f, 0.f};
    }
    sp.mark(6);   // local maxima, flatness
    if (j + 1 < ntiles) frame_prefetch<PCM>(st, pbase, n32, t_own + FPI < T, (t_own + FPI) * a.hop - NC, lane);
    sp.mark(7);   // loads of the next frame issued
    __syncthreads();  // S tile complete
    sp.mark(13);  // barrier
    if (a.dbg_S) {
      for (int i = tid; i < FPI * (NC + 1); i += NT) {
        const int f = i / (NC + 1), k = i % (NC + 1);
        if (f0 + f < T) a.dbg_S[(size_t)(f0 + f) * (NC + 1) + k] = regions[f * kExRegion + k];
      }
    }
    MelSeg fin;
    fin.kind = 0;
    f32x4 pend = {0.f, 0.f, 0.f, 0.f};
    const int nseg = __builtin_amdgcn_readfirstlane(wsched[kMelMaxSeg * 8]);
    for (int i = 0; i < nseg; ++i) {
      MelSeg g;
      {
        const int4 lo = *(const int4*)(wsched + i * 8), hi = *(const int4*)(wsched + 4 + i * 8);
        g.rb = __builtin_amdgcn_readfirstlane(lo.x); g.cb = __builtin_amdgcn_readfirstlane(lo.y);
        g.kbeg = __builtin_amdgcn_readfirstlane(lo.z); g.nsteps = __builtin_amdgcn_readfirstlane(lo.w);
        g.poff = __builtin_amdgcn_readfirstlane(hi.x); g.kind = __builtin_amdgcn_readfirstlane(hi.y);
        g.slot = __builtin_amdgcn_readfirstlane(hi.z); g.nparts = __builtin_amdgcn_readfirstlane(hi.w);
      }
      const f32x4 acc = mel_segment<NC, FPW, FPI>(regions, a, a.mel_pack, g, lane);
      if (g.kind == 1) {
        *(f32x4*)(slots + (g.slot * 64 + lane) * 4) = acc;
      } else if (g.kind == 2) {
        pend = acc;
        fin = g;
      } else {
        mel_epilogue<FPI>(acc, g, a, lm, f0, T, lane, lmax);
      }
    }
    sp.mark(14);  // mel product
    __syncthreads();  // the spectrum rows are free again; the split-K partial sums are in place
    sp.mark(13);
    if (fin.kind == 2) {
      for (int pidx = 0; pidx < fin.nparts; ++pidx) pend += *(const f32x4*)(slots + ((fin.slot + pidx) * 64 + lane) * 4);
      mel_epilogue<FPI>(pend, fin, a, lm, f0, T, lane, lmax);
    }
    sp.mark(15);  // finisher
  }
  lmax = wave_max(lmax);
  if (lane_id == 0) misc[wave] = lmax;
  if (PCM == DYSF_PCM_F32 && bad) a.ws_flags[clip] = 1;
  __syncthreads();
  if (tid == 0) {
    float m = misc[0];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, misc[w]);
    atomicMax(a.ws_clipmax + clip, __float_as_uint(fmaxf(m, 0.f)));
  }
}

// =============================================================================================
// Kernel A2: the same PCM -> STFT power -> mel -> dB pass for the throughput presets (n_fft 512, no chroma, no parity
// arithmetic), with nothing shared between the waves of a workgroup but read-only tables: no barrier after start-up.
//   * a wave owns a run of frame quads (four consecutive frames, 16 lanes x 16 complex points each) of one clip;
//   * PCM goes straight from global memory into the registers of the lane that transforms it (one 4-byte load per
//     complex point, the next quad's loads in flight during this quad's FFT): no LDS tile, no commit pass;
//   * the FFT is kernel A's (radix 16 x 16, one transpose through the wave's LDS region, DPP un-packing);
//   * the four spectrum rows overwrite the exchange buffer, and the mel product runs on v_mfma_f32_4x4x1_16B_f32:
//     sixteen independent 4 x 4 blocks, block b = mels 4b .. 4b+3 against the wave's own four frames, every block
//     walking its own band of bins (host-built: KArgs::mel4_*), both operands read 16 bytes at a time;
//   * 10 log10, one 16-byte store per lane into the frame-major log-mel workspace, running max of the mel power.
// grid = (wave groups, clips), 256 threads; wave w of workgroup g takes quads [(4g + w) * quads_per_wave, ...).
typedef __attribute__((address_space(3))) volatile float lds_vfloat;   // volatile float in LDS (a plain volatile cast makes it a flat access)
template <int PCM>
struct QuadPcm { int v[PCM == DYSF_PCM_I16 ? 16 : 32]; };


template <int PCM>
__device__ __forceinline__ void quad_prefetch(QuadPcm<PCM>& st, const char* pbase, int n, int T, int hop, int f0, int q, int ls) {
  constexpr int NC = 256;
  const int s_t = (f0 + q) * hop - NC;  // first sample of this lane's frame (centre padding: may be negative)
  // wave-uniform: all four frames lie inside the clip (true for all but the first and the last one or two quads)
  if (f0 * hop - NC >= 0 && (f0 + 3) * hop + NC <= n) {
    if (PCM == DYSF_PCM_I16) {
      const char* p0 = pbase + 2 * (s_t + 2 * ls);
#pragma unroll
      for (int p = 0; p < kPts; ++p) st.v[p] = *(const i32_a2*)(p0 + 64 * p);
    } else {
      const char* p0 = pbase + 4 * (s_t + 2 * ls);
#pragma unroll
      for (int p = 0; p < kPts; ++p) {
        const i32x2_a4 w = *(const i32x2_a4*)(p0 + 128 * p);
        st.v[2 * p] = w[0]; st.v[2 * p + 1] = w[1];
      }
    }
    return;
  }
  // edge quads: every load goes to a clamped (always valid) sample and is zeroed afterwards if its position lies outside
  // the clip or its frame does not exist -- no divergent branches, few live registers
  const bool live = f0 + q < T;
#pragma unroll
  for (int p = 0; p < kPts; ++p) {
    const int i0 = s_t + 2 * (ls + 16 * p);
    const bool in0 = live && i0 >= 0 && i0 < n, in1 = live && i0 + 1 >= 0 && i0 + 1 < n;
    const int c0 = min(max(i0, 0), n - 1), c1 = min(max(i0 + 1, 0), n - 1);
    if (PCM == DYSF_PCM_I16) {
      const unsigned short* b = (const unsigned short*)pbase;
      const unsigned lo = b[c0], hi = b[c1];
      st.v[p] = (int)((in0 ? lo : 0u) | ((in1 ? hi : 0u) << 16));
    } else {
      const int* b = (const int*)pbase;
      const int lo = b[c0], hi = b[c1];
      st.v[2 * p] = in0 ? lo : 0;
      st.v[2 * p + 1] = in1 ? hi : 0;
    }
  }
}

// Un-packing of one bin pair (k, NC - k) straight to the two powers: zk = Z[k], zm = Z[NC - k], v = V[k] (see
// rfft_pair_power).  Eight packed instructions: the conjugations ride on neg modifiers, and X = e + t, Y = e - t are
// formed as (X.re, Y.re) and (X.im, Y.im) so that both squares and both sums are one v_pk_mul_f32 + one v_pk_fma_f32.
__device__ __forceinline__ cf pair_powers(cf zk, cf zm, cf v) {
  cf e, o, A, B;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(e) : "v"(zk), "v"(zm));   // zk + conj(zm)
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(o) : "v"(zk), "v"(zm));   // zk - conj(zm)
  const cf t = cmul(o, v);
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(A) : "v"(e), "v"(t));  // (e.re + t.re, e.re - t.re)
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(B) : "v"(e), "v"(t));  // (e.im + t.im, e.im - t.im)
  return B * B + A * A;   // (|X[k]|^2, |X[NC-k]|^2)
}

// LDS tables of kernel A2, one row per FFT lane so that a lane fetches its constants with 16-byte reads: rows are an odd
// number of 16-byte slots long, which puts the sixteen lanes of a frame on sixteen different slots (the four frames of a
// wave read the same rows: broadcast).
constexpr int kA2TabRow = 84;   // floats per lane (21 slots): 16 window pairs (w[2n], w[2n+1]), n = ls + 16 p; 16 stage-1
                                // twiddles (p = 0 unused); 9 un-packing twiddles V[ls + 16 i], i < 8, and V[NC/2]
constexpr int kA2TabFloats = 16 * kA2TabRow;

// NW waves per workgroup (4 or 8), OCC waves per SIMD the build is register-budgeted for (4: 128, 5: 96, 6: 80 VGPRs).
template <int PCM, int NW, int OCC>
__global__ __launch_bounds__(64 * NW, OCC) void dysf_stft_mel_wave_kernel(const KArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int NC = 256;
  using G = FftGeom<NC>;
  constexpr int L = G::L;
  static_assert(L == 16 && G::FPW == 4 && G::XSTRIDE == kA2Kpad, "kernel A2 is laid out for n_fft 512");
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, ls = lane & 15;   // FFT roles: frame of the quad, lane of the frame
  STAMP_INIT;
  const int mb = lane >> 2, mr = lane & 3;   // mel roles: block (four mels), A row / B column (mel of the block / frame)
  const int clip = blockIdx.y;
  const long long off = a.offsets[clip];
  const long long n64 = a.offsets[clip + 1] - off;
  const int T = (int)(1 + n64 / a.hop);
  if (n64 <= 0 || T < 9 || T > a.t_max) return;  // rejected clips: kernel B writes their zero rows
  const int n = (int)n64;
  const int NQ = (T + 3) >> 2;
  if (blockIdx.x * NW * a.quads_per_wave >= NQ) return;  // whole workgroup beyond the clip

  float* wregion = lds + a.lds_region + wave * kA2Region;
  float* tabT = lds + a.lds_tables;                          // [16 lanes][kA2TabRow]
  float* melA = lds + a.lds_misc;                            // [rounds][steps/4][2 sets][64][4]
  {
    const float* wsrc = (PCM == DYSF_PCM_I16) ? a.window_i16 : a.window;   // window / 2 (x 2^-15 for int16 PCM)
    for (int i = tid; i < 16 * 32; i += NT) {   // i = (lane l, float j): pair p = j / 2 of the points n = l + 16 p
      const int l = i >> 5, j = i & 31;
      tabT[l * kA2TabRow + j] = wsrc[2 * (l + L * (j >> 1)) + (j & 1)];
      tabT[l * kA2TabRow + 32 + j] = a.stage_tw[(l * kPts + (j >> 1)) * 2 + (j & 1)];
    }
    for (int i = tid; i < 16 * 18; i += NT) {
      const int l = i / 18, j = i % 18, k = (j >> 1) == 8 ? NC / 2 : l + L * (j >> 1);
      tabT[l * kA2TabRow + 64 + j] = a.post_tw[2 * k + (j & 1)];
    }
    const int nm = a.mel4_rounds * a.mel4_steps * 32;        // float4 count
    for (int i = tid; i < nm; i += NT) ((float4*)melA)[i] = ((const float4*)a.mel4_pack)[i];
    // the mel product reads a few words past a row's last bin and the exchange leaves one word per frame untouched
    for (int i = lane * 4; i < kA2Region; i += 256) *(float4*)(wregion + i) = float4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();  // the only barrier: tables are in place
  STAMP(0);  // set-up
  // The waves that share a SIMD come from different workgroups, start together and run the same program: without a
  // stagger they stay in phase -- all in their VALU-heavy passes at once, then all in the exchange / MFMA / store
  // passes with the VALU idle.  Each wave sleeps a quarter, half or three quarters of a quad first, keyed by the wave
  // slot it occupies on its SIMD.
  if (a.stagger_mode) {
    int key;
    if (a.stagger_mode == 1) key = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3;   // HW_ID.wave_id
    else if (a.stagger_mode == 2) key = blockIdx.y & 3;
    else if (a.stagger_mode == 3) key = (blockIdx.y >> 3) & 3;
    else if (a.stagger_mode == 4) key = (blockIdx.y >> 8) & 3;
    else key = (blockIdx.y >> 5) & 3;
    for (int i = 0; i < key * a.stagger_units; ++i) __builtin_amdgcn_s_sleep(8);   // 8 x 64 cycles per unit and key step
  }

  const int q_begin = (blockIdx.x * NW + wave) * a.quads_per_wave;
  const int q_end = min(NQ, q_begin + a.quads_per_wave);
  if (q_begin >= q_end) return;
  const char* pbase = (const char*)a.pcm + off * (PCM == DYSF_PCM_I16 ? 2 : 4);
  float* xb = wregion + q * G::XSTRIDE;
  float* srow_own = wregion + q * kA2Kpad;
  const float4* winr = (const float4*)(tabT + ls * kA2TabRow);   // one base register, constant offsets
  const float4* twr = winr + 8;
  const float4* ptwr = winr + 16;
  float* lm = a.ws_logmel + (size_t)clip * a.ws_logmel_stride;
  const int steps4 = a.mel4_steps >> 2;
  const unsigned long long lane0 = __ballot(ls == 0);
  unsigned lmax = 0u;  // bits of the largest mel power seen (non-negative floats order like their bits)
  float badacc = 0.f;  // float PCM: x * 0 summed over every sample is NaN iff one of them is not finite

  // int16 PCM: the next quad's samples are requested a quad ahead (16 registers); float PCM would need 32, which does not
  // fit beside the FFT, and neither do the 16 at six waves per SIMD: those loads are issued at the top of the quad that
  // uses them (the other waves of the SIMD cover their latency)
  constexpr bool AHEAD = PCM == DYSF_PCM_I16 && OCC <= 5;
  QuadPcm<PCM> st;
  if (AHEAD) quad_prefetch<PCM>(st, pbase, n, T, a.hop, 4 * q_begin, q, ls);
  for (int qi = q_begin; qi < q_end; ++qi) {
    const int f0 = 4 * qi;
    if (!AHEAD) quad_prefetch<PCM>(st, pbase, n, T, a.hop, f0, q, ls);
    cf z[kPts];
#pragma unroll
    for (int p = 0; p < kPts; ++p) {
      if (PCM == DYSF_PCM_I16) {
        z[p][0] = (float)(short)(st.v[p] & 0xffff); z[p][1] = (float)(st.v[p] >> 16);
      } else {
        z[p][0] = __int_as_float(st.v[2 * p]); z[p][1] = __int_as_float(st.v[2 * p + 1]);
        badacc = fmaf(z[p][0], 0.f, badacc); badacc = fmaf(z[p][1], 0.f, badacc);
      }
    }
    // the staging registers are free: the next quad's samples are in flight during this quad's FFT (the fence keeps the
    // loads behind the conversions, so they land in the same registers and no copies are needed at the loop end)
    __builtin_amdgcn_sched_barrier(0);
    if (AHEAD && qi + 1 < q_end) quad_prefetch<PCM>(st, pbase, n, T, a.hop, f0 + 4, q, ls);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 w = winr[j];
      z[2 * j] *= mk(w.x, w.y);
      z[2 * j + 1] *= mk(w.z, w.w);
    }
    fft_stage<NC, 0>(z, (const cf*)nullptr);
    STAMP(1);  // conversion, prefetch issue, window, first radix-16 pass
    // transpose through the frame's exchange buffer, real parts then imaginary parts (volatile: single ds_read_b32s land
    // in the registers of the pairs they belong to; merged ds_read2_b32 results need a v_mov each)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int p = 0; p < kPts; ++p) xb[pad_idx(fft_out_index<NC, 0>(ls, p))] = z[p][h];
      wave_lds_sync();
#pragma unroll
      for (int p = 0; p < kPts; ++p) z[p][h] = ((const lds_vfloat*)xb)[pad_idx(fft_in_index<NC>(ls, p))];
      wave_lds_sync();
    }
    __builtin_amdgcn_sched_barrier(0);   // (table reads hoisted across the phases cost more registers than they hide latency)
    STAMP(2);  // exchange
    // stage-1 twiddles in two batches of four 16-byte reads (a batch is dead once its eight products are formed)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int j = 4 * half; j < 4 * half + 4; ++j) {
        const float4 w = twr[j];
        if (j > 0) z[2 * j] = cmul(z[2 * j], mk(w.x, w.y));
        z[2 * j + 1] = cmul(z[2 * j + 1], mk(w.z, w.w));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    fft_butterflies<NC, 1>(z);
    STAMP(3);  // twiddles + second radix-16 pass
    // un-packing: register p of lane ls holds Z[ls + 16 p]; Z[NC - k] sits in register 15 - i of lane 16 - ls (lane 0
    // pairs with itself, one register on): row_mirror + row_ror:1, as in kernel A
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // two bin pairs per 16-byte read of the lane's un-packing twiddles
      const float4 w = ptwr[j];
#pragma unroll
      for (int i = 2 * j; i < 2 * j + 2; ++i) {
        cf zm;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float offer;
          asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(offer) : "v"(z[15 - i][h]), "v"(z[(16 - i) & 15][h]), "s"(lane0));
          const int m = __builtin_amdgcn_mov_dpp(__float_as_int(offer), 0x140, 0xf, 0xf, true);
          zm[h] = __int_as_float(__builtin_amdgcn_mov_dpp(m, 0x121, 0xf, 0xf, true));
        }
        const cf pw = pair_powers(z[i], zm, (i & 1) ? mk(w.z, w.w) : mk(w.x, w.y));
        const int k = ls + L * i;
        srow_own[k] = pw[0];
        srow_own[NC - k] = pw[1];
      }
      if (j & 1) __builtin_amdgcn_sched_barrier(0);   // (keeps the twiddle reads from being hoisted in one block: registers)
    }
    {  // bin NC/2 pairs with itself: X = 2 conj(Z) (V = -1), |X|^2 = 4 |Z|^2; lane 0 holds it
      const cf d = z[8] + z[8], m = d * d;
      if (ls == 0) srow_own[NC / 2] = m[0] + m[1];
    }
    wave_lds_sync();
    STAMP(4);  // un-packing, powers, row stores
    // mel product: D_b[4 mels x 4 frames] += A_b[4 x 1] * B_b[1 x 4] per K-step, the sixteen blocks in one instruction.
    // Two accumulator sets follow the host's schedule (build_mel4): the long bands of the high mels are cut into pieces
    // that run in otherwise idle blocks; the two sets' instructions alternate, so no MFMA waits for its predecessor.
    for (int rd = 0; rd < a.mel4_rounds; ++rd) {
      const int* meta = a.mel4_kbeg + rd * 48;
      const float4* Ap = (const float4*)melA + (size_t)rd * steps4 * 128 + lane;
      const float4* B0 = (const float4*)(wregion + mr * kA2Kpad + meta[mb]);
      const float4* B1 = (const float4*)(wregion + mr * kA2Kpad + meta[16 + mb]);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < steps4; ++s) {
        const float4 a0 = Ap[(2 * s) * 64], a1 = Ap[(2 * s + 1) * 64], b0 = B0[s], b1 = B1[s];
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, b1.w, acc1, 0, 0, 0);
      }
      STAMP(5);  // mel MFMA loop
      // pieces -> groups: a DPP row is four blocks (banks of four lanes).  row_shl:n reads lane i + n.
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = acc0[i], y = acc1[i];
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xf, 0x5, false));  // blocks 0, 2 += blocks 1, 3
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x108, 0xf, 0x1, false));  // block 0 += block 2
        y += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), 0x104, 0xf, 0x4, false));  // block 2 += block 3
        // block 3 of set 1 takes the row's finished set-0 group (row_shr:12 reads lane i - 12)
        acc1[i] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(y), __float_as_int(x), 0x11c, 0xf, 0x8, false));
      }
      // lane (block mb, column mr) now holds the four mels from meta[32 + mb] on of frame f0 + mr
      const int t = f0 + mr, m0 = meta[32 + mb];
      if (t < T && m0 >= 0) {
        float4 v;
        v.x = power_to_db(acc1[0], a); v.y = power_to_db(acc1[1], a); v.z = power_to_db(acc1[2], a); v.w = power_to_db(acc1[3], a);
        *(float4*)(lm + (size_t)t * a.lm_stride + m0) = v;
        lmax = max(max(lmax, max(__float_as_uint(acc1[0]), __float_as_uint(acc1[1]))),
                   max(__float_as_uint(acc1[2]), __float_as_uint(acc1[3])));
      }
    }
    wave_lds_sync();
    STAMP(6);  // dB + store
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, (unsigned)__shfl_xor((int)lmax, o));
  if (lane == 0) atomicMax(a.ws_clipmax + clip, lmax);
  if (PCM == DYSF_PCM_F32 && !(badacc == 0.f)) a.ws_flags[clip] = 1;
}

// ---------------------------------------------------------------------------------------------
// NumPy's float32 pairwise summation (np.add.reduce over a contiguous axis, umath/loops_utils.h.src), the order
// np.mean / np.std of pipeline1.py:220-221 sum in.  A block of n <= 128 values is summed by 8 strided accumulators
// that are then combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and followed by the n % 8 tail; longer rows are halved
// (left half rounded down to a multiple of 8) recursively.  Eight consecutive lanes play the accumulators of one row.
// pass 0 sums x[i]; pass 1 sums float32(float32(x[i] - mean)**2) (np.std's intermediate array).
__device__ __forceinline__ float pw_leaf(const float* x, int n, int j, int pass, float mean) {
  auto el = [&](int i) -> float {
    const float v = x[i];
    if (!pass) return v;
    const float d = sub_rn(v, mean);
    return mul_rn(d, d);
  };
  float res;
  if (n < 8) {
    res = 0.f;
    for (int i = 0; i < n; ++i) res = add_rn(res, el(i));
    return res;
  }
  float r = el(j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) r = add_rn(r, el(i + j));
  r = add_rn(r, __shfl_xor(r, 1));
  r = add_rn(r, __shfl_xor(r, 2));
  r = add_rn(r, __shfl_xor(r, 4));
  res = r;
  for (; i < n; ++i) res = add_rn(res, el(i));
  return res;
}

// =============================================================================================
// Kernel B: log-mel [T][lm_stride] -> clamp -> DCT (float64 MFMA) -> deltas -> frame rows + pooled statistics.
// One workgroup per clip, chunks of CH = 16*NW - 16 output frames; every wave runs one 16-frame column tile of the
// DCT (8 halo frames each side of the chunk), then owns the MFCC rows c = wave, wave + NW, ... (a lane is a frame).
// Mean / variance are one-pass sums of e = x - x[frame 0] (exactly zero for constant rows): float32 per-lane
// partials across the chunks, one float64 reduction per clip.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kBShift = 0;     // float2[24][3]: frame-0 features of the row pairs
constexpr int kBDct = 144;     // float offset of the float64 DCT operands (8-byte aligned)

// RB = 1..3 fixes the geometry at compile time (n_mels == 16 * NG, dct_rb == RB: no mel masks, no per-block branches,
// constant LDS offsets); RB = 0 reads it from the arguments.
// H16 (throughput presets, n_mels a multiple of 32): the DCT as three float16 MFMA products of a hi / lo split -- coefficients
// D = Dh + Dl 2^-11 (host), mean-free log-mel x' = xh + xl -- Dh xh + Dh xl + 2^-11 Dl xh on v_mfma_f32_16x16x32_f16: float32
// accumulation, operands carried to ~2^-22, 18 matrix instructions of 8 passes per 16-frame tile instead of 48 of 16 passes.
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __fp16 h2v __attribute__((ext_vector_type(2)));   // (the type __builtin_amdgcn_cvt_pkrtz returns)
template <int NT, int NG, bool F32, int RB, bool H16 = false>
__global__ __launch_bounds__(NT, 4) void dysf_mfcc_kernel(const KArgs a) {
  static_assert(!H16 || (F32 && RB > 0 && NG % 2 == 0), "the float16-split DCT needs whole 32-mel K blocks and a fixed geometry");
  constexpr int NW = NT / 64;
  constexpr int COLS = 16 * NW;
  constexpr int CH = COLS - 16;
  // The MFCC chunk sits in LDS as row PAIRS: rows 16p + w and 16p + w + 8 interleaved as float2 per column, pair row
  // 8p + w, CSTR float2 per pair row (CSTR % 8 == 4: the DCT tile writes of the four lane groups miss each other's
  // banks).  The delta pass of wave w reads both rows of a pair with one ds_read_b64 per tap.
  constexpr int CSTR = kMfccCstr;
  static_assert(NW == 8 && CSTR % 8 == 4, "row pairs of the delta pass");
  constexpr int RPW = (48 + NW - 1) / NW;  // rows per wave (n_mfcc <= 48)
  extern __shared__ __align__(16) float lds[];
  STAMP_INIT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row bases and store addresses stay in SGPRs
  __builtin_assume(wave >= 0 && wave < NW);
  const int groups = RB ? NG : a.dct_groups;   // K groups of 16 mels
  const int rbs = RB ? RB : a.dct_rb;          // row blocks of 16 coefficients
  const int clip = blockIdx.x;
  const int C = a.n_mfcc;
  const long long off = a.offsets[clip];
  const long long n = a.offsets[clip + 1] - off;
  const int T = (int)(1 + n / a.hop);
  int st = DYSF_OK;
  if (n <= 0) st = DYSF_EMPTY;
  else if (T < 9) st = DYSF_TOO_SHORT;
  else if (T > a.t_max) st = DYSF_TOO_LONG;  // device entry point called with a max_samples below this clip's length
  else if (a.ws_flags[clip]) st = DYSF_NONFINITE;
  float* prow = a.pooled ? a.pooled + (size_t)clip * a.pooled_dim : nullptr;
  float* frow = a.frames ? a.frames + (size_t)clip * 3 * C * a.frame_stride : nullptr;
  if (tid == 0 && a.status) a.status[clip] = st;
  if (st != DYSF_OK) {  // the reference's "except -> zeros" row (pipeline1.py:237-239)
    if (prow) for (int i = tid; i < a.pooled_dim; i += NT) prow[i] = 0.f;
    if (frow) for (long long i = tid; i < 3LL * C * a.frame_stride; i += NT) frow[i] = 0.f;
    if (tid == 0 && a.tuning) a.tuning[clip] = 50;
    if (a.self_clean) {  // leave the clip's workspace words zero for the next launch (every thread has read them)
      __syncthreads();
      if (tid == 0) { a.ws_clipmax[clip] = 0u; a.ws_flags[clip] = 0; }
    }
    return;
  }
  if (tid == 0 && a.tuning && !a.chroma) a.tuning[clip] = 50;
  const float* lm = a.ws_logmel + (size_t)clip * a.ws_logmel_stride;
  // the clamp starts from float32(10) * float32(log10(largest mel power)): log10 correctly rounded through float64,
  // then the float32 multiply -- the two roundings of librosa.power_to_db.  One ulp here moves every clamped cell.
  const float xmax = __uint_as_float(a.ws_clipmax[clip]);
  const float clip_max = (xmax <= a.amin) ? a.db_amin : mul_rn(10.0f, (float)log10((double)xmax));
  const float floor_db = (a.top_db >= 0.f) ? clip_max - a.top_db : -INFINITY;
  if (a.dbg_logmel)
    for (int i = tid; i < T * a.n_mels; i += NT) a.dbg_logmel[i] = lm[(size_t)(i / a.n_mels) * a.lm_stride + (i % a.n_mels)];

  f2* kshift = (f2*)(lds + kBShift);   // [pair row][x, d, dd] as row pairs, like the chunk
  double* dct = (double*)(lds + kBDct);
  const float* dct32 = lds + kBDct;               // F32: float A operands, then the float64 row sums of the coefficients
  const int nd = rbs * groups * 256;
  const double* rowsum = (const double*)(lds + kBDct + nd);
  float* chunk = lds + kBDct + (F32 ? nd + 2 * 16 * rbs : 2 * nd);
  if (F32) {
    const float* psrc = H16 ? (const float*)a.dct_pack16 : a.dct_pack32;   // (the two packs have the same size)
    for (int i = tid * 4; i < nd; i += NT * 4) *(float4*)(lds + kBDct + i) = *(const float4*)(psrc + i);
    for (int i = tid; i < 16 * rbs; i += NT) ((double*)(lds + kBDct + nd))[i] = (H16 ? a.dct_rowsum16 : a.dct_rowsum)[i];
  } else {
    for (int i = tid * 2; i < nd; i += NT * 2) *(double2*)(dct + i) = *(const double2*)(a.dct_pack + i);
  }
  // parity presets: the frame rows are kept (in the caller's frames output, else in the workspace stash) and summed
  // again in NumPy's order at the end
  float* wrow = frow;
  long long wstr = a.frame_stride;
  if (!wrow && a.parity && prow) { wrow = a.ws_stat + (size_t)clip * 3 * C * a.stat_stride; wstr = a.stat_stride; }
  f2 ps[RPW / 2][6];  // [row pair][sum x, d, dd, then the squares]: lane partials of e = feature - frame-0 feature
#pragma unroll
  for (int i = 0; i < RPW / 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) ps[i][j] = f2{0.f, 0.f};
  // The log-mel rows of a chunk are fetched one chunk ahead: the loads of chunk i+1 fly during the delta pass of chunk i
  // (NG = 16-mel groups held in registers; rows past the clip clamp to the last frame, so the extra fetch is harmless).
  float4 pf[NG];
  auto fetch = [&](int t0n) {
    int t = t0n - 8 + wave * 16 + (lane & 15);
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    if (H16) {   // float16 MFMA operand order: lane (frame = l & 15, kb = l >> 4) holds mels 32 blk + 8 kb + {0..7}
      const float* lrow = lm + (size_t)t * a.lm_stride + 8 * (lane >> 4);
#pragma unroll
      for (int g = 0; g < NG; ++g) pf[g] = *(const float4*)(lrow + 32 * (g >> 1) + 4 * (g & 1));
      return;
    }
    const float* lrow = lm + (size_t)t * a.lm_stride + 4 * (lane >> 4);  // this lane's four mels of every 16-mel group
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (g < groups) pf[g] = *(const float4*)(lrow + 16 * g);  // 16 frames x 64 bytes per wave load
  };
  fetch(0);
  STAMP(8);  // setup
  for (int t0 = 0; t0 < T; t0 += CH) {
    __syncthreads();
    // every thread has read the clip's maximum and flag by now: zero them for the next launch (no memset node then)
    if (a.self_clean && t0 == 0 && tid == 0) { a.ws_clipmax[clip] = 0u; a.ws_flags[clip] = 0; }
    STAMP(9);  // barrier
    // a wave whose 16 columns all lie past the last frame the delta filters can reach has nothing to compute
    // (only in a clip's last chunk; wave-uniform)
    const bool tile_live = t0 - 8 + wave * 16 <= T - 1;
    if (!tile_live) {
    } else if constexpr (H16) {
      float4 xq[NG];
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        xq[g].x = fmaxf(pf[g].x, floor_db); xq[g].y = fmaxf(pf[g].y, floor_db);
        xq[g].z = fmaxf(pf[g].z, floor_db); xq[g].w = fmaxf(pf[g].w, floor_db);
        sum += (xq[g].x + xq[g].y) + (xq[g].z + xq[g].w);
      }
      fetch(t0 + CH);
      sum += __shfl_xor(sum, 16);  // the four lanes of a frame (lane & 15) end up with the same total
      sum += __shfl_xor(sum, 32);
      const float m32 = sum * a.inv_n_mels;
      h8v bh[NG / 2], bl[NG / 2];   // x' = xh + xl per K block of 32 mels
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float x0 = xq[g].x - m32, x1 = xq[g].y - m32, x2 = xq[g].z - m32, x3 = xq[g].w - m32;
        const h2v h01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), h23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
        const h2v l01 = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h01[0], x1 - (float)h01[1]);
        const h2v l23 = __builtin_amdgcn_cvt_pkrtz(x2 - (float)h23[0], x3 - (float)h23[1]);
        const int o = 4 * (g & 1);
        bh[g >> 1][o] = (_Float16)h01[0]; bh[g >> 1][o + 1] = (_Float16)h01[1]; bh[g >> 1][o + 2] = (_Float16)h23[0]; bh[g >> 1][o + 3] = (_Float16)h23[1];
        bl[g >> 1][o] = (_Float16)l01[0]; bl[g >> 1][o + 1] = (_Float16)l01[1]; bl[g >> 1][o + 2] = (_Float16)l23[0]; bl[g >> 1][o + 3] = (_Float16)l23[1];
      }
      const h8v* dp = (const h8v*)(lds + kBDct) + lane;   // [row block][K block][hi, lo][lane]
      f32x4 acc0[RB], acc1[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) { acc0[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int b = 0; b < NG / 2; ++b)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const h8v ah = dp[((rb * (NG / 2) + b) * 2 + 0) * 64], al = dp[((rb * (NG / 2) + b) * 2 + 1) * 64];
          acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[b], acc0[rb], 0, 0, 0);
          acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[b], acc1[rb], 0, 0, 0);
          acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[b], acc0[rb], 0, 0, 0);
        }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // f32 C/D map: row = 4 * (lane >> 4) + reg, col = lane & 15
          const int row = rb * 16 + 4 * (lane >> 4) + r;
          const int ib = 4 * (lane >> 4) + r;
          chunk[((rb * 8 + (ib & 7)) * CSTR + wave * 16 + (lane & 15)) * 2 + (ib >> 3)] =
              (float)((double)fmaf(acc1[rb][r], 1.0f / 2048.0f, acc0[rb][r]) + (double)m32 * rowsum[row]);
        }
    } else if constexpr (F32) {
      // float32 MFMA on mean-free columns: with x' = x - m (m = any per-frame constant near the column mean)
      // D x = D x' + m * rowsum(D) exactly, the sums the MFMA carries stay small (silent / clamped frames give
      // x' = 0 and an exact result) and only the m * rowsum term -- sqrt(n_mels) * m for coefficient 0, ~1e-16 * m
      // for the others -- is added in float64.
      const float* dp = dct32 + lane * 4;
      float4 xq[NG];  // this chunk's rows (the next chunk's fetch below reuses pf)
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g < groups) {
          const int m0 = 16 * g + 4 * (lane >> 4);
          xq[g].x = (RB || m0 + 0 < a.n_mels) ? fmaxf(pf[g].x, floor_db) : 0.f;
          xq[g].y = (RB || m0 + 1 < a.n_mels) ? fmaxf(pf[g].y, floor_db) : 0.f;
          xq[g].z = (RB || m0 + 2 < a.n_mels) ? fmaxf(pf[g].z, floor_db) : 0.f;
          xq[g].w = (RB || m0 + 3 < a.n_mels) ? fmaxf(pf[g].w, floor_db) : 0.f;
          sum += (xq[g].x + xq[g].y) + (xq[g].z + xq[g].w);
        }
      }
      fetch(t0 + CH);
      sum += __shfl_xor(sum, 16);  // the four lanes of a frame (lane & 15) end up with the same total
      sum += __shfl_xor(sum, 32);
      const float m32 = sum * a.inv_n_mels;
      f32x4 acc[3];
#pragma unroll
      for (int rb = 0; rb < 3; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g < groups) {
          const int m0 = 16 * g + 4 * (lane >> 4);
          float bv[4];
          bv[0] = (RB || m0 + 0 < a.n_mels) ? xq[g].x - m32 : 0.f;
          bv[1] = (RB || m0 + 1 < a.n_mels) ? xq[g].y - m32 : 0.f;
          bv[2] = (RB || m0 + 2 < a.n_mels) ? xq[g].z - m32 : 0.f;
          bv[3] = (RB || m0 + 3 < a.n_mels) ? xq[g].w - m32 : 0.f;
#pragma unroll
          for (int rb = 0; rb < 3; ++rb) {
            if (rb < rbs) {
              const float4 a4 = *(const float4*)(dp + (rb * groups + g) * 256);
              acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, bv[0], acc[rb], 0, 0, 0);
              acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, bv[1], acc[rb], 0, 0, 0);
              acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, bv[2], acc[rb], 0, 0, 0);
              acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, bv[3], acc[rb], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int rb = 0; rb < 3; ++rb)
        if (rb < rbs)
#pragma unroll
          for (int r = 0; r < 4; ++r) {  // f32 C/D map: row = 4 * (lane >> 4) + reg, col = lane & 15
            const int row = rb * 16 + 4 * (lane >> 4) + r;
            const int ib = 4 * (lane >> 4) + r;  // row within the block: pair row ib & 7, half ib >> 3
            chunk[((rb * 8 + (ib & 7)) * CSTR + wave * 16 + (lane & 15)) * 2 + (ib >> 3)] =
                (float)((double)acc[rb][r] + (double)m32 * rowsum[row]);
          }
    } else {
        const double* dp = dct + lane * 4;
        f64x4 acc[3];
  #pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb] = f64x4{0.0, 0.0, 0.0, 0.0};
  #pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g < groups) {
            const float4 b4 = pf[g];
            const int m0 = 16 * g + 4 * (lane >> 4);
            double bv[4];
            bv[0] = (RB || m0 + 0 < a.n_mels) ? (double)fmaxf(b4.x, floor_db) : 0.0;
            bv[1] = (RB || m0 + 1 < a.n_mels) ? (double)fmaxf(b4.y, floor_db) : 0.0;
            bv[2] = (RB || m0 + 2 < a.n_mels) ? (double)fmaxf(b4.z, floor_db) : 0.0;
            bv[3] = (RB || m0 + 3 < a.n_mels) ? (double)fmaxf(b4.w, floor_db) : 0.0;
  #pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
              if (rb < rbs) {
                const double2 a01 = *(const double2*)(dp + (rb * groups + g) * 256);
                const double2 a23 = *(const double2*)(dp + (rb * groups + g) * 256 + 2);
                acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, bv[0], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, bv[1], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, bv[2], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, bv[3], acc[rb], 0, 0, 0);
              }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep one group's operands live at a time (the kernel sits at its VGPR cap)
          }
        }
        fetch(t0 + CH);
  #pragma unroll
        for (int rb = 0; rb < 3; ++rb)
          if (rb < rbs)
  #pragma unroll
            for (int r = 0; r < 4; ++r) {  // f64 C/D map: row = (lane >> 4) + 4 * reg, col = lane & 15
              const int ib = (lane >> 4) + 4 * r;
              chunk[((rb * 8 + (ib & 7)) * CSTR + wave * 16 + (lane & 15)) * 2 + (ib >> 3)] = (float)acc[rb][r];
            }
    }
    STAMP(10);  // DCT
    __syncthreads();
    STAMP(11);  // barrier
#pragma unroll
    for (int sub = 0; sub < (CH + 63) / 64; ++sub) {
      const int t = t0 + sub * 64 + lane;
      const bool tv = (sub * 64 + lane < CH) && (t < T);
      int tc = t < 4 ? 4 : (t > T - 5 ? T - 5 : t);
      tc = tv ? tc : 4;
      const int bx = (tv ? t : t0) - (t0 - 8);  // column of frame t in the chunk
      const int bc = tc - (t0 - 8);             // column of the delta centre
#pragma unroll
      for (int p = 0; p < RPW / 2; ++p) {
        // two rows per pass: c0 = 16p + wave and c1 = c0 + 8 are the halves of pair row 8p + wave, so every tap is one
        // ds_read_b64 whose register pair feeds packed float32 math directly.  Rows between n_mfcc and the end of
        // their 16-row block are zero rows of the DCT pack; they ride along unused.
        const int c0 = wave + 2 * p * NW, c1 = c0 + NW;
        if (c0 < 16 * rbs) {
          const f2* rp = (const f2*)chunk + (p * 8 + wave) * CSTR;
          auto ld = [&](int col) { return rp[col]; };
          const f2 x = ld(bx);
          f2 d = f2{0.f, 0.f}, dd = f2{0.f, 0.f};
          if (tv) {
            const f2 xc = ld(bc), x2 = xc + xc;
            const f2 p1 = ld(bc + 1), m1 = ld(bc - 1), p2 = ld(bc + 2), m2 = ld(bc - 2);
            const f2 p3 = ld(bc + 3), m3 = ld(bc - 3), p4 = ld(bc + 4), m4 = ld(bc - 4);
            d = ((p1 - m1) + 2.f * (p2 - m2) + 3.f * (p3 - m3) + 4.f * (p4 - m4)) * (1.0f / 60.0f);
            dd = (28.f * ((p4 + m4) - x2) + 7.f * ((p3 + m3) - x2) - 8.f * ((p2 + m2) - x2) - 17.f * ((p1 + m1) - x2)) *
                 (1.0f / 462.0f);
          }
          f2* ks = kshift + (p * 8 + wave) * 3;
          if (t0 == 0 && sub == 0) {  // shift values = the features of frame 0
            if (lane == 0) { ks[0] = x; ks[1] = d; ks[2] = dd; }
            wave_lds_sync();
          }
          if (tv) {
            const f2 e0 = x - ks[0], e1 = d - ks[1], e2 = dd - ks[2];
            ps[p][0] += e0; ps[p][1] += e1; ps[p][2] += e2;
            ps[p][3] += e0 * e0; ps[p][4] += e1 * e1; ps[p][5] += e2 * e2;
            if (wrow) {
              auto put = [&](int row, float v) { wrow[(size_t)row * wstr + t] = v; };
              if (c0 < C) { put(c0, x[0]); put(C + c0, d[0]); put(2 * C + c0, dd[0]); }
              if (c1 < C) { put(c1, x[1]); put(C + c1, d[1]); put(2 * C + c1, dd[1]); }
            }
          }
        }
      }
    }
  }
  STAMP(12);  // last chunk's deltas (the others are inside slot 9)
  if (frow && a.frame_stride > T) {  // zero-fill the padding columns
    const int padc = (int)(a.frame_stride - T);
    for (int i = tid; i < 3 * C * padc; i += NT) frow[(size_t)(i / padc) * a.frame_stride + T + (i % padc)] = 0.f;
  }
  __syncthreads();  // every wave is done with the MFCC chunk and the DCT operands: the scratch below reuses them
  const bool np_stats = a.parity && prow && T <= kPwMaxFrames;  // statistics in NumPy's summation order (below)
  if (prow && !np_stats) {
    // Cross-lane totals without shuffles: per round the wave parks 18 quantities x 64 lanes in LDS, lane q sums
    // row q in float64 (conflict-free stride 65), lanes with a sum / sum-of-squares pair finish mean and std.
    float* scr = lds + kBDct + wave * (18 * 65 + 3);
    double* tot = (double*)(lds + kBDct + NW * (18 * 65 + 3)) + wave * 18;  // 8-byte aligned: NW and kBDct are even
#pragma unroll
    for (int rnd = 0; rnd < RPW / 3; ++rnd) {
#pragma unroll
      for (int q = 0; q < 18; ++q) scr[q * 65 + lane] = ps[(rnd * 3 + q / 6) / 2][q % 6][(rnd * 3 + q / 6) % 2];
      wave_lds_sync();
      if (lane < 18) {
        double sum = 0.0;
        for (int l = 0; l < 64; ++l) sum += (double)scr[lane * 65 + l];
        tot[lane] = sum;
      }
      wave_lds_sync();
      if (lane < 18 && (lane % 6) < 3) {
        const int i = rnd * 3 + lane / 6, j = lane % 6;
        const int c = wave + i * NW;
        if (c < C) {
          const double m1 = tot[lane] / (double)T, m2 = tot[lane + 3] / (double)T;
          const double var = m2 - m1 * m1;
          prow[j * 2 * C + c] = (float)((double)kshift[((c >> 4) * 8 + (c & 7)) * 3 + j][(c >> 3) & 1] + m1);
          prow[j * 2 * C + C + c] = (float)sqrt(var > 0.0 ? var : 0.0);
        }
      }
      wave_lds_sync();
    }
  }
  STAMP(13);  // final reduction
  if (np_stats) {
    // np.mean / np.std in NumPy's own float32 summation order (clips beyond kPwMaxFrames keep the float64 statistics).
    __syncthreads();  // scratch is free; every frame-row store of this workgroup has been issued and waited for
    int2* prog = (int2*)(lds + kBDct);          // post-order program: (start, len) = push the leaf's sum; (0, 0) = add the top two
    int* bstack = (int*)(prog + kPwProg);       // [40][3] thread 0's build stack
    float* vstack = (float*)(bstack + 120);     // [NT / 8][16] value stacks
    int* plen = (int*)(vstack + (NT / 8) * 16);
    if (tid == 0) {
      int np_ = 0, sp = 1;
      bstack[0] = 0; bstack[1] = T; bstack[2] = 0;
      while (sp > 0) {
        int* e = bstack + 3 * (sp - 1);
        const int s0 = e[0], ln = e[1];
        if (ln <= 128) { prog[np_++] = make_int2(s0, ln); --sp; }
        else if (e[2] == 0) {
          e[2] = 1;
          int n2 = ln / 2;
          n2 -= n2 % 8;
          int* rr = bstack + 3 * sp;       // right half below the left one: the left is summed first
          rr[0] = s0 + n2; rr[1] = ln - n2; rr[2] = 0;
          rr[3] = s0; rr[4] = n2; rr[5] = 0;
          sp += 2;
        } else { prog[np_++] = make_int2(0, 0); --sp; }
      }
      *plen = np_;
    }
    __syncthreads();
    const int nprog = *plen;
    const int j = tid & 7;
    float* vs = vstack + (tid >> 3) * 16;
    const float fT = (float)T;
    for (int r = tid >> 3; r < 3 * C; r += NT / 8) {
      const float* x = wrow + (size_t)r * wstr;
      float mean = 0.f, sd = 0.f;
      if (nprog == 1) {
        // clips of up to 128 frames are one leaf of NumPy's pairwise sum (pw_leaf): the row is read ONCE, all loads in
        // flight together (the accumulator's strided share and the n % 8 tail), and both passes run from registers
        const int n8 = T - (T % 8), nq = n8 >> 3, nt = T - n8;
        float xv[16], tv[7];
#pragma unroll
        for (int q = 0; q < 16; ++q) xv[q] = (q < nq) ? x[j + 8 * q] : 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) tv[i] = (i < nt) ? x[n8 + i] : 0.f;
        auto leaf = [&](int pass, float mu) -> float {
          auto el = [&](float v) -> float {
            if (!pass) return v;
            const float d = sub_rn(v, mu);
            return mul_rn(d, d);
          };
          float acc = el(xv[0]);
#pragma unroll
          for (int q = 1; q < 16; ++q) if (q < nq) acc = add_rn(acc, el(xv[q]));
          acc = add_rn(acc, __shfl_xor(acc, 1));
          acc = add_rn(acc, __shfl_xor(acc, 2));
          acc = add_rn(acc, __shfl_xor(acc, 4));
#pragma unroll
          for (int i = 0; i < 7; ++i) if (i < nt) acc = add_rn(acc, el(tv[i]));
          return acc;
        };
        mean = div_rn(leaf(0, 0.f), fT);
        sd = sqrt_rn(div_rn(leaf(1, mean), fT));
      } else
      for (int pass = 0; pass < 2; ++pass) {
        int sp = 0;
        for (int e = 0; e < nprog; ++e) {
          const int2 pe = prog[e];
          float v;
          if (pe.y == 0) { sp -= 2; v = add_rn(vs[sp], vs[sp + 1]); }
          else v = pw_leaf(x + pe.x, pe.y, j, pass, mean);
          wave_lds_sync();
          vs[sp++] = v;
          wave_lds_sync();
        }
        const float tot = vs[0];
        if (pass == 0) mean = div_rn(tot, fT); else sd = sqrt_rn(div_rn(tot, fT));
      }
      if (j == 0) {
        const int blk = r / C, c = r % C;
        prow[blk * 2 * C + c] = mean;
        prow[blk * 2 * C + C + c] = sd;
      }
    }
  }
}

// =============================================================================================
// Kernel C (chroma presets): tuning decision from the clip's peak list (librosa.estimate_tuning), then
// chroma = bank[tuning] x S over the power spectrogram kernel A left in the workspace, per-frame max-norm, mean / std
// (librosa.feature.chroma_stft + pipeline1.py:228).  One workgroup per clip; a wave owns 16-frame column blocks and
// streams their spectrum rows straight from HBM into MFMA B operands (16 bytes per lane, eight loads in flight):
// K is walked in groups of 16 bins, lane (frame = l & 15, q = l >> 4) holding the four consecutive bins 16g + 4q + s,
// and the bank pack has the same order ([tuning][group][lane][4]), so there is no LDS staging and no second FFT.
// NVT: 256-bin slices of a row held in registers per frame by the peak pass (3 covers the pitch band up to bin 767, which is
// every preset at 16 kHz and above; the full-width instantiation serves lower sample rates).
template <int NC, int NT, int NVT>
__global__ __launch_bounds__(NT, 4) void dysf_chroma_kernel(const KArgs a) {
  constexpr int NW = NT / 64;
  extern __shared__ __align__(16) float4 sm_cand[];   // [2 * NW frames][(pip_kmax - pip_kmin) / 2 + 2] parked local maxima (they cannot be adjacent)
  __shared__ int sm_cnt;
  __shared__ int sm_i[128];                    // [2..3] upper-median pass, [16..115] tuning histogram
  __shared__ int sm_h3[3 * 256];               // radix-select histograms (block_select3)
  __shared__ double sm_red[NW * kChroma * 2];  // per-wave partial sums of the chroma rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int clip = blockIdx.x;
  const int C = a.n_mfcc;
  STAMP_INIT;
  const long long off = a.offsets[clip];
  const long long n = a.offsets[clip + 1] - off;
  const int T = (int)(1 + n / a.hop);
  if (n <= 0 || T < 9 || T > a.t_max || a.ws_flags[clip]) {  // rejected clip: kernel B already wrote the zero row
    if (tid == 0 && a.tuning) a.tuning[clip] = 50;
    return;
  }
  float* pk_mag = a.ws_peak_mag + (size_t)clip * a.ws_peak_stride;
  int* pk_bin = a.ws_peak_bin + (size_t)clip * a.ws_peak_stride;
  int* hist = sm_i + 16;
  const float* S = a.ws_S + (size_t)clip * a.ws_S_stride;
  // ---- librosa.piptrack on the stored power spectrogram: a wave takes a frame per round; local maxima above 0.1 * the
  // frame's largest bin inside [pip_kmin, pip_kmax) are parked in LDS (bin, power, neighbours), then converted densely --
  // one thread per peak -- to (interpolated magnitude, histogram bin) and appended to the clip's list in the workspace.
  int npk = 0;
  if (a.ws_cand) {
    // kernel A-exact picked the local maxima while the rows were in LDS: a wave takes the lists of its frames and converts
    // them, one lane per maximum, into (interpolated magnitude, histogram bin) at the next free places of the clip's list
    const float4* cand = a.ws_cand + (size_t)clip * a.t_max * a.cand_per_frame;
    const int* fcnt = a.ws_framecnt + (size_t)clip * a.t_max;
    // exclusive prefix sums of the per-frame counts (wave 0, 64 frames per step), then every thread finds the frame of its
    // maxima i = tid, tid + NT, ... by bisection: all candidate loads of a round are in flight together
    int* sm_off = (int*)sm_cand;   // [T + 1]
    for (int i = tid; i < 3 * 256; i += NT) sm_h3[i] = 0;
    if (tid < 128) sm_i[tid] = (tid == 3) ? 0x7f800000 : 0;
    if (wave == 0) {
      int run = 0;
      for (int t0 = 0; t0 < T; t0 += 64) {
        const int c = (t0 + lane < T) ? fcnt[t0 + lane] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (t0 + lane < T) sm_off[t0 + lane] = run + inc - c;
        run += __shfl(inc, 63);
      }
      if (lane == 0) sm_off[T] = run;
    }
    __syncthreads();
    npk = min(sm_off[T], (int)a.ws_peak_stride);
    constexpr int KC = 8;
    for (int i0 = 0; i0 < npk; i0 += KC * NT) {
      float4 c[KC];
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const int i = i0 + tid + j * NT;
        if (i < npk) {
          int lo = 0, hi = T;   // sm_off[lo] <= i < sm_off[hi]
          while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sm_off[mid] <= i) lo = mid; else hi = mid; }
          c[j] = cand[(size_t)lo * a.cand_per_frame + (i - sm_off[lo])];
        }
      }
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const int i = i0 + tid + j * NT;
        if (i < npk) {
          float mag; int bin;
          peak_to_mag_bin<NC>(__float_as_int(c[j].x), c[j].y, c[j].z, c[j].w, a, &mag, &bin);
          pk_mag[i] = mag; pk_bin[i] = bin;
        }
      }
    }
    __syncthreads();
    if (tid == 0) a.ws_peak_cnt[clip] = npk;   // (dysf_debug_peaks)
  } else
  {
    // A wave walks its frames t = wave, wave + NW, ... on its own: no workgroup barrier inside the pass.  A lane owns four
    // consecutive bins of every 256-bin slice of the frame's row (one 16-byte load per slice; the next frame's row is
    // requested before this one is looked at), the neighbours of its first and last bin come from the adjacent lanes.
    // Local maxima are parked in the wave's private slice of the LDS list (positions from ballot pop-counts, no atomics);
    // when the slice cannot take another frame, and at the end, the wave converts its maxima -- one lane per peak -- to
    // (interpolated magnitude, histogram bin) and appends them to the clip's list in the workspace (order is immaterial).
    constexpr int NV = NVT;                        // 256-bin slices held per frame (the launcher checks that they reach pip_kmax)
    const float* rowmax = a.ws_rowmax + (size_t)clip * a.t_max;
    const int nv = min(NV, (a.pip_kmax + 1 + 255) / 256);   // slices that reach the pitch band (wave-uniform)
    const int per_frame = (a.pip_kmax - a.pip_kmin) / 2 + 2;   // most local maxima one frame can park
    const int rcap = a.cand_cap / NW;                          // the wave's slice (>= 2 * per_frame: dysf_capi.hip)
    float4* mine = sm_cand + wave * rcap;
    if (tid == 0) sm_cnt = 0;
    for (int i = tid; i < 3 * 256; i += NT) sm_h3[i] = 0;
    if (tid < 128) sm_i[tid] = (tid == 3) ? 0x7f800000 : 0;
    __syncthreads();
    int wcnt = 0;   // maxima parked in the slice (wave-uniform)
    auto flush = [&]() {
      wave_lds_sync();
      int gbase = 0;
      if (lane == 0) gbase = atomicAdd(&sm_cnt, wcnt);
      gbase = __builtin_amdgcn_readfirstlane(gbase);
      for (int i = lane; i < wcnt; i += 64) {
        const float4 c = mine[i];
        float mag; int bin;
        peak_to_mag_bin<NC>(__float_as_int(c.x), c.y, c.z, c.w, a, &mag, &bin);
        if (gbase + i < (int)a.ws_peak_stride) { pk_mag[gbase + i] = mag; pk_bin[gbase + i] = bin; }
      }
      wave_lds_sync();
      wcnt = 0;
    };
    float4 vn[NV];
    float refn = 0.f;
    auto request = [&](int t) {
      const float* row = S + (size_t)(t < T ? t : T - 1) * a.s_stride + 4 * lane;
      refn = rowmax[t < T ? t : T - 1];
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i < nv) vn[i] = (256 * i + 4 * lane + 3 < a.s_stride) ? *(const float4*)(row + 256 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    request(wave);
    for (int t = wave; t < T; t += NW) {
      float4 v[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = vn[i];
      const float ref = mul_rn(0.1f, refn);
      if (t + NW < T) request(t + NW);
      if (wcnt + per_frame > rcap) flush();
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (i >= nv) continue;
        const float4 c = v[i];
        // left neighbour of .x: lane - 1's .w (lane 0: the previous slice's lane 63); right neighbour of .w: lane + 1's .x
        // (whole-wave DPP shifts, zero shifted in at the ends; the slice seams through v_readlane)
        float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c.w), 0x138, 0xf, 0xf, false));    // wave_shr:1
        float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c.x), 0x130, 0xf, 0xf, false));   // wave_shl:1
        if (i > 0) { const float pw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i - 1].w), 63)); if (lane == 0) left = pw; }
        if (i + 1 < NV && i + 1 < nv) { const float nx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i + 1].x), 0)); if (lane == 63) right = nx; }
        const float xs[6] = {left, c.x, c.y, c.z, c.w, right};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 256 * i + 4 * lane + j;
          const float x0 = xs[j + 1], xm = xs[j], x1 = xs[j + 2];
          const float z0 = x0 > ref ? x0 : 0.f, zm = xm > ref ? xm : 0.f, z1 = x1 > ref ? x1 : 0.f;
          const bool is_peak = (k >= a.pip_kmin) && (k < a.pip_kmax) && (z0 > zm) && (z0 >= z1);
          const unsigned long long mask = __ballot(is_peak);
          if (mask != 0ull) {
            const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (is_peak) mine[wcnt + below] = make_float4(__int_as_float(k), x0, xm, x1);
            wcnt += __popcll(mask);
          }
        }
      }
    }
    flush();
    __syncthreads();
    npk = sm_cnt;
    if (tid == 0) a.ws_peak_cnt[clip] = npk;   // (dysf_debug_peaks)
  }
  npk = npk < (int)a.ws_peak_stride ? npk : (int)a.ws_peak_stride;
  STAMP(8);   // peak list
  int tune = 50;
  if (npk > 0) {
    // this thread's share of the peak list in registers (all loads in flight together): the median's four radix passes,
    // the upper-median pass and the histogram pass read it from there; lists beyond KR * NT peaks fall back to memory
    constexpr int KR = 8;
    float rmag[KR];
    int rbin[KR];
#pragma unroll
    for (int j = 0; j < KR; ++j) {
      const int idx = min(tid + j * NT, npk - 1);
      rmag[j] = pk_mag[idx];
      rbin[j] = pk_bin[idx];
    }
    float thr;
    if (npk & 1) {
      thr = block_select3<NT, KR>(pk_mag, rmag, npk, npk / 2, sm_h3);
    } else {
      // np.median of an even count: mean of the two middle order statistics.  The upper one follows from the lower one in a
      // single pass: it is the lower value again if that value fills index npk / 2 too, else the smallest larger element.
      const float lo = block_select3<NT, KR>(pk_mag, rmag, npk, npk / 2 - 1, sm_h3);
      int c_le = 0;
      unsigned mn = 0x7f800000u;   // +inf; the magnitudes that matter are positive floats: their bits order like the values
#pragma unroll
      for (int j = 0; j < KR; ++j)
        if (tid + j * NT < npk) { if (rmag[j] <= lo) ++c_le; else mn = min(mn, __float_as_uint(rmag[j])); }
      for (int i = tid + KR * NT; i < npk; i += NT) {
        const float x = pk_mag[i];
        if (x <= lo) ++c_le; else mn = min(mn, __float_as_uint(x));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { c_le += __shfl_xor(c_le, o); mn = min(mn, (unsigned)__shfl_xor((int)mn, o)); }
      if (lane == 0) { atomicAdd(&sm_i[2], c_le); atomicMin((unsigned*)&sm_i[3], mn); }   // (zero / +inf since the start)
      __syncthreads();
      const float hi = (sm_i[2] > npk / 2) ? lo : __uint_as_float((unsigned)sm_i[3]);
      thr = mul_rn(add_rn(lo, hi), 0.5f);
    }
    STAMP(9);   // median
#pragma unroll
    for (int j = 0; j < KR; ++j)
      if (tid + j * NT < npk && rmag[j] >= thr) atomicAdd(&hist[rbin[j]], 1);   // (hist is zero since the start)
    for (int i = tid + KR * NT; i < npk; i += NT)
      if (pk_mag[i] >= thr) atomicAdd(&hist[pk_bin[i]], 1);
    __syncthreads();
    // first maximum of the histogram: every wave reduces keys (count << 8 | 255 - bin) over its lanes (bins l and l + 64)
    static_assert(kTunings <= 128, "two bins per lane");
    unsigned key = ((unsigned)hist[lane] << 8) | (unsigned)(255 - lane);
    if (lane + 64 < kTunings) key = max(key, ((unsigned)hist[lane + 64] << 8) | (unsigned)(255 - (lane + 64)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, o));
    tune = 255 - (int)(key & 255u);
  }
  if (tid == 0 && a.tuning) a.tuning[clip] = tune;
  STAMP(10);  // histogram
  if (!a.pooled) return;

  const float* bank = a.chroma_pack + (size_t)tune * a.chroma_groups * 256 + lane * 4;
  const int ncb = (T + 15) / 16;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  // K groups [g_lo, g_hi) of the column block cb: 16 bytes of the spectrum rows and of the bank per lane and group, eight
  // groups in flight
  auto product = [&](int cb, int g_lo, int g_hi) -> f32x4 {
    const int t = cb * 16 + (lane & 15);
    const float* srow = S + (size_t)(t < T ? t : T - 1) * a.s_stride + 4 * (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int g0 = g_lo; g0 < g_hi; g0 += 8) {
      float4 av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (g0 + u < g_hi) {
          bv[u] = *(const float4*)(srow + 16 * (g0 + u));
          av[u] = *(const float4*)(bank + (size_t)(g0 + u) * 256);
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (g0 + u < g_hi) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
    }
    return acc;
  };
  // acc[r] = chroma row 4 * (lane >> 4) + r of frame t (rows 12..15 are padding: zero taps).  librosa.util.normalize
  // with norm=inf: divide the frame by its largest magnitude unless that is below float32 tiny.
  auto finish = [&](f32x4 acc, int cb) {
    const int t = cb * 16 + (lane & 15);
    float mx = fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float len = (mx < 1.17549435e-38f) ? 1.0f : mx;
    if (t < T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = (double)div_rn(acc[r], len);
        s1[r] += v; s2[r] += v * v;
      }
    }
  };
  // Short clips have fewer column blocks than the workgroup has waves (3 s at hop 512: six), and one block is a serial
  // chain of K / 8 memory round trips: K is then split KS ways, the partial tiles meet in LDS (the candidate list /
  // prefix sums of the first phase are dead) and are added in a fixed order.
  int KS = 1;
  if (ncb < 2 * NW) KS = max(1, min(min(8, (3 * NW + ncb - 1) / ncb), 32 / ncb));
  const int G = a.chroma_groups;
  if (KS == 1) {
    for (int cb = wave; cb < ncb; cb += NW) finish(product(cb, 0, G), cb);
  } else {
    f32x4* sm_part = (f32x4*)sm_cand;   // [ncb * KS <= 32][64]
    for (int item = wave; item < ncb * KS; item += NW) {
      const int cb = item / KS, ks = item % KS;
      sm_part[item * 64 + lane] = product(cb, ks * G / KS, (ks + 1) * G / KS);
    }
    __syncthreads();
    for (int cb = wave; cb < ncb; cb += NW) {
      f32x4 acc = sm_part[cb * KS * 64 + lane];
      for (int ks = 1; ks < KS; ++ks) acc += sm_part[(cb * KS + ks) * 64 + lane];
      finish(acc, cb);
    }
  }
  STAMP(11);  // chroma product
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s1[r] += __shfl_xor(s1[r], o); s2[r] += __shfl_xor(s2[r], o); }
    const int row = 4 * (lane >> 4) + r;
    if ((lane & 15) == 0 && row < kChroma) { sm_red[(wave * kChroma + row) * 2] = s1[r]; sm_red[(wave * kChroma + row) * 2 + 1] = s2[r]; }
  }
  __syncthreads();
  if (tid < kChroma) {
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < NW; ++w) { t1 += sm_red[(w * kChroma + tid) * 2]; t2 += sm_red[(w * kChroma + tid) * 2 + 1]; }
    const double m1 = t1 / (double)T, var = t2 / (double)T - m1 * m1;
    float* prow = a.pooled + (size_t)clip * a.pooled_dim;
    prow[6 * C + tid] = (float)m1;
    prow[6 * C + kChroma + tid] = (float)sqrt(var > 0.0 ? var : 0.0);
  }
  STAMP(12);  // reduction
}

// =============================================================================================
// Kernel D (rich presets, main.py:738-754): spectral contrast mean / std (7 + 7), mean zero-crossing rate, mean RMS,
// mean spectral centroid.  One workgroup per clip: a third STFT pass (librosa recomputes the spectrogram for every
// descriptor too) in which the lane group that transformed a frame also reduces it -- centroid sums and, per octave
// band, the mean of the n_avg smallest / largest magnitudes by repeated (value, bin) min / max extraction -- then the
// two clip-global top_db clamps and the statistics, then the time-domain descriptors from hop-sized segment sums
// (frame_length 2048 is a whole number of hops; checked on the host).
constexpr int kRichBands = 7;
constexpr int kRichSegCap = 1024;  // hop segments held in LDS per pass of the time-domain part

__device__ __forceinline__ double block_sum(double v, double* scratch, int tid, int nw) {  // scratch: nw doubles
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nw; ++w) t += scratch[w];
  return t;
}

template <int NC, int PCM, int NT>
__global__ __launch_bounds__(NT, 2) void dysf_rich_kernel(const KArgs a) {
  using G = FftGeom<NC>;
  constexpr int L = G::L, FPW = G::FPW;
  constexpr int NW = NT / 64;
  constexpr int FPI = FPW * NW;
  constexpr int EPG = PcmStage<PCM>::EPG;
  extern __shared__ __align__(16) float lds[];
  float* regions = lds + a.lds_region;
  float* tile = lds + a.lds_pcm;
  double* red = (double*)(lds + a.lds_misc);  // [16] block-reduction scratch
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane / L;
  const int ls = (FPW == 4) ? ((lane + ((q & 1) ? L / 2 : 0)) % L) : lane % L;  // FFT role (see kernel A)
  const int li = lane % L;                                                      // role in the per-frame reductions
  const int clip = blockIdx.x;
  const long long off = a.offsets[clip];
  const long long n = a.offsets[clip + 1] - off;
  const int T = (int)(1 + n / a.hop);
  if (n <= 0 || T < 9 || a.ws_flags[clip] || !a.pooled) return;  // rejected clip: kernel B already wrote the zero row
  float* wregion = regions + wave * a.rstride;
  float* xb = wregion + q * G::XSTRIDE;
  float* srow = wregion + q * a.kpad;
  FftTabs tb;
  load_fft_tabs<NC, NT, PCM>(lds + a.lds_tables, a, tb);
  float* prow = a.pooled + (size_t)clip * a.pooled_dim + a.rich_offset;
  float* ws = a.ws_rich + (size_t)clip * 2 * kRichBands * a.ws_rich_stride;  // rows 0..6 peak, 7..13 valley, [T] each
  const int* tab = a.rich_tab;                                                // [band][3] = first bin, end bin, n_avg
  const double hz_per_bin = (double)a.sr / (double)(2 * NC);
  double cent_acc = 0.0;

  const int ntiles = (T + FPI - 1) / FPI;
  const int tile_samples = (FPI - 1) * a.hop + 2 * NC;
  const char* pbase = (const char*)a.pcm + off * (PCM == DYSF_PCM_I16 ? 2 : 4);
  const int n32 = (int)n, off_lo = (int)(off & (EPG - 1));
  for (int ti = 0; ti < ntiles; ++ti) {
    const int f0 = ti * FPI;
    const int cs0 = f0 * a.hop - NC;
    const int cr0 = pcm_tile_origin<PCM>(off_lo, cs0);
    __syncthreads();  // every wave is done with the previous PCM tile
    {
      PcmStage<PCM> stage;
      const int cng = (cs0 + tile_samples - cr0 + EPG - 1) / EPG;
      pcm_prefetch<PCM, NT>(stage, pbase, n32, cr0, cng);
      pcm_commit<PCM, NT>(stage, tile, pbase, n32, cr0, cng);
    }
    __syncthreads();
    wave_frame_power<NC>(tile + (cs0 - cr0), (wave * FPW + q) * a.hop, xb, srow, ls, tb, a);
    // the row belongs to this lane group: no workgroup barrier needed before reading it back
    const int t = f0 + wave * FPW + q;
    const bool valid = t < T;
    {  // spectral centroid: sum_k f_k |X_k| / sum_k |X_k|
      double s0 = 0.0, s1 = 0.0;
      for (int k = li; k <= NC; k += L) {
        const float m = sqrt_rn(srow[k]);
        s0 += (double)m; s1 += (double)m * (double)k;
      }
#pragma unroll
      for (int o = L / 2; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
      if (valid && li == 0) cent_acc += (s0 < 2.2250738585072014e-308) ? 0.0 : s1 * hz_per_bin / s0;
    }
    for (int b = 0; b < kRichBands; ++b) {
      const int lo = tab[3 * b], hi = tab[3 * b + 1], na = tab[3 * b + 2];
      float lv = -INFINITY, hv = INFINITY;  // last extracted (value, bin) from the bottom / from the top
      int lk = -1, hk = 0x7fffffff;
      float sum_lo = 0.f, sum_hi = 0.f;
      for (int pass = 0; pass < na; ++pass) {
        float bv = INFINITY, tv = -INFINITY;
        int bk = 0x7fffffff, tk = -1;
        for (int k = lo + li; k < hi; k += L) {
          const float m = sqrt_rn(srow[k]);
          if ((m > lv || (m == lv && k > lk)) && (m < bv || (m == bv && k < bk))) { bv = m; bk = k; }
          if ((m < hv || (m == hv && k < hk)) && (m > tv || (m == tv && k > tk))) { tv = m; tk = k; }
        }
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) {
          const float ov = __shfl_xor(bv, o); const int ok = __shfl_xor(bk, o);
          if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
          const float pv = __shfl_xor(tv, o); const int pk = __shfl_xor(tk, o);
          if (pv > tv || (pv == tv && pk > tk)) { tv = pv; tk = pk; }
        }
        sum_lo += bv; sum_hi += tv;
        lv = bv; lk = bk; hv = tv; hk = tk;
      }
      if (valid && li == 0) {
        ws[(size_t)b * a.ws_rich_stride + t] = sum_hi / (float)na;
        ws[(size_t)(kRichBands + b) * a.ws_rich_stride + t] = sum_lo / (float)na;
      }
    }
  }
  __syncthreads();  // all peak / valley rows of the clip are written (and visible to the workgroup)

  // ---- power_to_db(peak) - power_to_db(valley), each clamped 80 dB below its own clip-global maximum; statistics
  {
    double mp = -INFINITY, mv = -INFINITY;
    for (int i = tid; i < kRichBands * T; i += NT) {
      const int b = i / T, t = i % T;
      mp = fmax(mp, 10.0 * log10(fmax(1e-10, (double)ws[(size_t)b * a.ws_rich_stride + t])));
      mv = fmax(mv, 10.0 * log10(fmax(1e-10, (double)ws[(size_t)(kRichBands + b) * a.ws_rich_stride + t])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mp = fmax(mp, __shfl_xor(mp, o)); mv = fmax(mv, __shfl_xor(mv, o)); }
    if (lane == 0) { red[wave] = mp; red[NW + wave] = mv; }
    __syncthreads();
    for (int w = 0; w < NW; ++w) { mp = fmax(mp, red[w]); mv = fmax(mv, red[NW + w]); }
    __syncthreads();
    if (wave < kRichBands) {
      const int b = wave;
      const float* pr = ws + (size_t)b * a.ws_rich_stride;
      const float* vr = ws + (size_t)(kRichBands + b) * a.ws_rich_stride;
      auto contrast = [&](int t) {
        const double dp = fmax(10.0 * log10(fmax(1e-10, (double)pr[t])), mp - 80.0);
        const double dv = fmax(10.0 * log10(fmax(1e-10, (double)vr[t])), mv - 80.0);
        return dp - dv;
      };
      const double c0 = contrast(0);
      double s1 = 0.0, s2 = 0.0;
      for (int t = lane; t < T; t += 64) { const double e = contrast(t) - c0; s1 += e; s2 += e * e; }
      s1 = wave_sum(s1); s2 = wave_sum(s2);
      if (lane == 0) {
        const double m1 = s1 / (double)T, var = s2 / (double)T - m1 * m1;
        prow[b] = (float)(c0 + m1);
        prow[kRichBands + b] = (float)sqrt(var > 0.0 ? var : 0.0);
      }
    }
  }

  // ---- zero-crossing rate and RMS over frames of a.rich_fl samples (edge / zero padding of rich_fl/2 at both ends)
  double zsum = 0.0, rsum = 0.0;
  {
    const int FL = a.rich_fl, R = FL / a.hop, half = FL / 2;
    double* seg_sq = (double*)regions;                 // regions are free now
    int* seg_zc = (int*)(seg_sq + kRichSegCap);
    int* seg_first = seg_zc + kRichSegCap;
    auto sample = [&](long long s) -> float {          // y[s] for 0 <= s < n
      return (PCM == DYSF_PCM_I16) ? (float)((const short*)pbase)[s] * (1.0f / 32768.0f) : ((const float*)pbase)[s];
    };
    const int fchunk = kRichSegCap - R + 1;            // frames per pass
    for (int fbeg = 0; fbeg < T; fbeg += fchunk) {
      const int nf = min(fchunk, T - fbeg), nseg = nf + R - 1;
      __syncthreads();
      for (int j = wave; j < nseg; j += NW) {
        double sq = 0.0;
        int zc = 0, first = 0;
        for (int i = lane; i < a.hop; i += 64) {
          const long long pi = (long long)(fbeg + j) * a.hop + i, sidx = pi - half;
          const float yz = (sidx >= 0 && sidx < n) ? sample(sidx) : 0.f;
          const long long c1 = sidx < 0 ? 0 : (sidx > n - 1 ? n - 1 : sidx);
          const long long c0i = sidx - 1 < 0 ? 0 : (sidx - 1 > n - 1 ? n - 1 : sidx - 1);
          const bool cross = (pi >= 1) && ((sample(c1) < -1e-10f) != (sample(c0i) < -1e-10f));
          sq += (double)yz * (double)yz;
          zc += cross ? 1 : 0;
          if (i == 0) first = cross ? 1 : 0;
        }
        sq = wave_sum(sq);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) zc += __shfl_xor(zc, o);
        first = __shfl(first, 0);
        if (lane == 0) { seg_sq[j] = sq; seg_zc[j] = zc; seg_first[j] = first; }
      }
      __syncthreads();
      for (int f = tid; f < nf; f += NT) {
        double sq = 0.0;
        int zc = -seg_first[f];                        // a frame's first sample has no predecessor inside the frame
        for (int r = 0; r < R; ++r) { sq += seg_sq[f + r]; zc += seg_zc[f + r]; }
        zsum += (double)zc / (double)FL;
        rsum += (double)sqrt_rn((float)(sq / (double)FL));
      }
    }
  }
  zsum = block_sum(zsum, red, tid, NW);
  rsum = block_sum(rsum, red, tid, NW);
  cent_acc = block_sum(cent_acc, red, tid, NW);
  if (tid == 0) {
    prow[2 * kRichBands + 0] = (float)(zsum / (double)T);
    prow[2 * kRichBands + 1] = (float)(rsum / (double)T);
    prow[2 * kRichBands + 2] = (float)(cent_acc / (double)T);
  }
}

// =============================================================================================
// Metrics kernel (pipeline1.py:151-174): snr_db from the PCM and the mean of the per-frame spectral flatness kernel A left
// in the workspace.  One workgroup per clip.
//   snr_db: 25 ms frames every 10 ms without padding, float32 frame energies, the frames whose energy lies below the 25th
//   percentile (NumPy's linear interpolation between the two order statistics) are noise;
//   10 * log10(mean energy / (mean noise energy + 1e-10)); 0 when the clip is shorter than a frame or nothing is noise.
template <int NT>
__global__ __launch_bounds__(NT) void dysf_metrics_kernel(const KArgs a) {
  constexpr int NW = NT / 64;
  __shared__ int sm_i[400];
  __shared__ double sm_d[NW * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int clip = blockIdx.x;
  const long long off = a.offsets[clip];
  const long long n = a.offsets[clip + 1] - off;
  const int T = (int)(1 + n / a.hop);
  float* out = a.metrics + (size_t)clip * 3;
  const bool pcm16 = a.pcm_type == DYSF_PCM_I16;
  const int FL = (int)(0.025 * (double)a.sr), HP = (int)(0.010 * (double)a.sr);
  auto sample = [&](long long i) -> float {
    return pcm16 ? (float)((const short*)a.pcm)[off + i] * (1.0f / 32768.0f) : ((const float*)a.pcm)[off + i];
  };
  // ---- spectral_flatness_mean
  float flat = 0.f;
  if (n > 0 && T <= a.t_max && !a.ws_flags[clip]) {
    double s = 0.0;
    for (int t = tid; t < T; t += NT) s += (double)a.ws_flat[(size_t)clip * a.t_max + t];
    s = wave_sum(s);
    if (lane == 0) sm_d[wave] = s;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < NW; ++w) tot += sm_d[w];
    flat = (float)(tot / (double)T);
    __syncthreads();
  }
  // ---- snr_db
  float snr = 0.f;
  const long long nfr = (n >= FL) ? 1 + (n - FL) / HP : 0;
  if (nfr > 0 && nfr <= a.energy_stride && !a.ws_flags[clip]) {
    float* en = a.ws_energy + (size_t)clip * a.energy_stride;
    for (int f = tid; f < (int)nfr; f += NT) {
      // (NumPy sums the float32 squares pairwise in float32; float64 here: the results agree to ~1e-7 relative)
      double acc = 0.0;
      for (int i = 0; i < FL; ++i) { const float x = sample((long long)f * HP + i); acc += (double)mul_rn(x, x); }
      const float tot = (float)acc;
      en[f] = tot;
    }
    __syncthreads();
    const double vidx = 0.25 * (double)(nfr - 1);
    const int lo = (int)floor(vidx);
    const int hi = lo + 1 < (int)nfr ? lo + 1 : (int)nfr - 1;
    const double g = vidx - (double)lo;
    const float e_lo = block_select<NT>(en, (int)nfr, lo, sm_i + 128, sm_i);
    const float e_hi = block_select<NT>(en, (int)nfr, hi, sm_i + 128, sm_i);
    const float d32 = sub_rn(e_hi, e_lo);
    const double p25 = (g >= 0.5) ? (double)e_hi - (double)d32 * (1.0 - g) : (double)e_lo + (double)d32 * g;
    double s_all = 0.0, s_noise = 0.0, c_noise = 0.0;
    for (int f = tid; f < (int)nfr; f += NT) {
      const float e = en[f];
      s_all += (double)e;
      if ((double)e < p25) { s_noise += (double)e; c_noise += 1.0; }
    }
    s_all = wave_sum(s_all); s_noise = wave_sum(s_noise); c_noise = wave_sum(c_noise);
    if (lane == 0) { sm_d[wave] = s_all; sm_d[NW + wave] = s_noise; sm_d[2 * NW + wave] = c_noise; }
    __syncthreads();
    if (tid == 0) {
      double A = 0.0, B = 0.0, Cn = 0.0;
      for (int w = 0; w < NW; ++w) { A += sm_d[w]; B += sm_d[NW + w]; Cn += sm_d[2 * NW + w]; }
      if (Cn > 0.0) {
        const float sig = (float)(A / (double)nfr), noi = (float)(B / Cn);
        const float ratio = div_rn(sig, add_rn(noi, 1e-10f));
        snr = mul_rn(10.0f, (float)log10((double)ratio));
      }
    }
  }
  if (tid == 0) { out[0] = snr; out[1] = flat; }
}

// ---------------------------------------------------------------------------------------------
// The dynamic-LDS attribute belongs to the kernel function (per device), not to a handle or a template instantiation of
// this helper: one process-wide table keyed by function and device, only ever raised.  Steady-state launches are pure
// stream work and can be captured into a hipGraph by the caller.
static hipError_t grant_dynamic_lds(const void* kern, size_t lds_bytes) {
  struct Entry { const void* fn; int dev; size_t bytes; };
  static std::mutex mu;
  static std::vector<Entry> table;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  Entry* e = nullptr;
  for (Entry& x : table) if (x.fn == kern && x.dev == dev) { e = &x; break; }
  if (e && lds_bytes <= e->bytes) return hipSuccess;
  const hipError_t rc = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (rc != hipSuccess) return rc;
  if (e) e->bytes = lds_bytes; else table.push_back({kern, dev, lds_bytes});
  return hipSuccess;
}
template <typename K>
static hipError_t launch_one(K kern, dim3 grid, int threads, size_t lds_bytes, hipStream_t s, const KArgs& a) {
  if (lds_bytes > 0) {
    const hipError_t e = grant_dynamic_lds((const void*)kern, lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds_bytes, s, a);
  return hipGetLastError();
}

template <int NC>
static hipError_t launch_stft_nc(const KArgs& a, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s) {
  constexpr int NT = stft_threads(NC);
  const bool ch = a.chroma || a.dbg_S;
  if (a.parity) {  // parity presets are not the throughput path: one instantiation with everything in it
    if (pcm_type == DYSF_PCM_I16) return launch_one(dysf_stft_mel_kernel<NC, DYSF_PCM_I16, NT, true, true>, grid, NT, lds_bytes, s, a);
    return launch_one(dysf_stft_mel_kernel<NC, DYSF_PCM_F32, NT, true, true>, grid, NT, lds_bytes, s, a);
  }
  if (ch) {
    if (pcm_type == DYSF_PCM_I16) return launch_one(dysf_stft_mel_kernel<NC, DYSF_PCM_I16, NT, false, true>, grid, NT, lds_bytes, s, a);
    return launch_one(dysf_stft_mel_kernel<NC, DYSF_PCM_F32, NT, false, true>, grid, NT, lds_bytes, s, a);
  }
  if (pcm_type == DYSF_PCM_I16) return launch_one(dysf_stft_mel_kernel<NC, DYSF_PCM_I16, NT, false, false>, grid, NT, lds_bytes, s, a);
  return launch_one(dysf_stft_mel_kernel<NC, DYSF_PCM_F32, NT, false, false>, grid, NT, lds_bytes, s, a);
}

template <int NC>
static hipError_t launch_rich_nc(const KArgs& a, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s) {
  if (pcm_type == DYSF_PCM_I16) return launch_one(dysf_rich_kernel<NC, DYSF_PCM_I16, 512>, grid, 512, lds_bytes, s, a);
  return launch_one(dysf_rich_kernel<NC, DYSF_PCM_F32, 512>, grid, 512, lds_bytes, s, a);
}
hipError_t launch_rich(const KArgs& a, int n_fft, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s) {
  switch (n_fft) {
    case 256: return launch_rich_nc<128>(a, pcm_type, grid, lds_bytes, s);
    case 512: return launch_rich_nc<256>(a, pcm_type, grid, lds_bytes, s);
    case 1024: return launch_rich_nc<512>(a, pcm_type, grid, lds_bytes, s);
    case 2048: return launch_rich_nc<1024>(a, pcm_type, grid, lds_bytes, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_stft_mel(const KArgs& a, int n_fft, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s) {
  switch (n_fft) {
    case 256: return launch_stft_nc<128>(a, pcm_type, grid, lds_bytes, s);
    case 512: return launch_stft_nc<256>(a, pcm_type, grid, lds_bytes, s);
    case 1024: return launch_stft_nc<512>(a, pcm_type, grid, lds_bytes, s);
    case 2048: return launch_stft_nc<1024>(a, pcm_type, grid, lds_bytes, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_stft_mel_exact2048(const KArgs& a, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s) {
  if (pcm_type == DYSF_PCM_I16) return launch_one(dysf_stft_mel_exact2048_kernel<DYSF_PCM_I16>, grid, 512, lds_bytes, s, a);
  return launch_one(dysf_stft_mel_exact2048_kernel<DYSF_PCM_F32>, grid, 512, lds_bytes, s, a);
}
template <int NW, int OCC>
static hipError_t launch_a2(const KArgs& a, int pcm_type, dim3 grid, size_t lds_bytes, hipStream_t s) {
  if (pcm_type == DYSF_PCM_I16) return launch_one(dysf_stft_mel_wave_kernel<DYSF_PCM_I16, NW, OCC>, grid, 64 * NW, lds_bytes, s, a);
  // float PCM stages twice the registers per quad: only the 128-register build holds them without scratch
  return launch_one(dysf_stft_mel_wave_kernel<DYSF_PCM_F32, NW, 4>, grid, 64 * NW, lds_bytes, s, a);
}
hipError_t launch_stft_mel_wave(const KArgs& a, int pcm_type, int waves_per_wg, int waves_per_simd, dim3 grid, size_t lds_bytes,
                                hipStream_t s) {
  // builds: (waves per workgroup, waves per SIMD the registers allow) -- 4 workgroups of 4 / 5 / 6 waves per CU, or 8-wave
  // workgroups at 4 or 6 waves per SIMD
  if (waves_per_wg == 8) {
    if (waves_per_simd >= 6) return launch_a2<8, 6>(a, pcm_type, grid, lds_bytes, s);
    return launch_a2<8, 4>(a, pcm_type, grid, lds_bytes, s);
  }
  if (waves_per_wg == 6) return launch_a2<6, 6>(a, pcm_type, grid, lds_bytes, s);
  if (waves_per_wg == 5) return launch_a2<5, 5>(a, pcm_type, grid, lds_bytes, s);
  if (waves_per_simd >= 6) return launch_a2<4, 6>(a, pcm_type, grid, lds_bytes, s);
  if (waves_per_simd == 5) return launch_a2<4, 5>(a, pcm_type, grid, lds_bytes, s);
  return launch_a2<4, 4>(a, pcm_type, grid, lds_bytes, s);
}
hipError_t launch_mfcc(const KArgs& a, dim3 grid, size_t lds_bytes, hipStream_t s) {
  // up to 64 mels: mean-free float32 MFMA.  The eight-group form of it spills its statistics accumulators, so the
  // 128-mel presets stay on the float64 MFMA (same results to ~1e-5, see tests/diagnostics/margin.py).
  const int full = (a.n_mels == 16 * a.dct_groups) ? a.dct_groups : 0;   // every 16-mel group complete: a fixed-geometry build may apply
  if (a.dct_pack32 && a.dct_groups <= 4) {
    if (full == 4 && a.dct_rb == 3 && a.dct_pack16) return launch_one(dysf_mfcc_kernel<512, 4, true, 3, true>, grid, 512, lds_bytes, s, a);
    if (full == 4 && a.dct_rb == 3) return launch_one(dysf_mfcc_kernel<512, 4, true, 3>, grid, 512, lds_bytes, s, a);
    if (full == 4 && a.dct_rb == 1) return launch_one(dysf_mfcc_kernel<512, 4, true, 1>, grid, 512, lds_bytes, s, a);
    return launch_one(dysf_mfcc_kernel<512, 4, true, 0>, grid, 512, lds_bytes, s, a);
  }
  if (a.dct_groups <= 4) return launch_one(dysf_mfcc_kernel<512, 4, false, 0>, grid, 512, lds_bytes, s, a);
  if (full == 8 && a.dct_rb == 3) return launch_one(dysf_mfcc_kernel<512, 8, false, 3>, grid, 512, lds_bytes, s, a);
  if (full == 8 && a.dct_rb == 1) return launch_one(dysf_mfcc_kernel<512, 8, false, 1>, grid, 512, lds_bytes, s, a);
  return launch_one(dysf_mfcc_kernel<512, 8, false, 0>, grid, 512, lds_bytes, s, a);
}
int stft_exact_table_floats(int n_fft) {
  switch (n_fft) {
    case 256: return fft_tabs64_floats<128>();
    case 512: return fft_tabs64_floats<256>();
    case 1024: return fft_tabs64_floats<512>();
    default: return fft_tabs64_floats<1024>();
  }
}

hipError_t launch_metrics(const KArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL(dysf_metrics_kernel<256>, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

template <int NC>
static hipError_t launch_chroma_nc(const KArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  constexpr int NVF = (NC + 1 + 255) / 256;          // slices that cover bins 0 .. NC
  constexpr int NVS = NVF < 3 ? NVF : 3;
  // (the candidate list of a wide pitch band -- low sample rates at n_fft 2048 -- exceeds the 64 KB a kernel gets without
  // the attribute: launch_one raises it)
  if ((a.pip_kmax + 1 + 255) / 256 <= NVS) return launch_one(dysf_chroma_kernel<NC, 512, NVS>, grid, 512, lds, s, a);
  return launch_one(dysf_chroma_kernel<NC, 512, NVF>, grid, 512, lds, s, a);
}
hipError_t launch_chroma(const KArgs& a, int n_fft, dim3 grid, hipStream_t s) {
  // kernel A-exact has picked the maxima: prefix sums of the per-frame counts instead of the list; then 32 partial tiles
  const size_t lds = a.ws_cand ? std::max<size_t>(32768, (size_t)(a.t_max + 2) * 4) : (size_t)a.cand_cap * sizeof(float4);
  switch (n_fft) {
    case 256: return launch_chroma_nc<128>(a, grid, lds, s);
    case 512: return launch_chroma_nc<256>(a, grid, lds, s);
    case 1024: return launch_chroma_nc<512>(a, grid, lds, s);
    case 2048: return launch_chroma_nc<1024>(a, grid, lds, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dysf
