// Host emulation of the wave-level FFT in dysf_fft.hpp: runs the exact per-lane stage functions and
// index maps the kernel uses, with the LDS exchange buffer replaced by an array, and checks the real
// power spectrum against a naive float64 DFT.  Built with g++ and run by tests/test_fft_emulation.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dysf_fft.hpp"

using namespace dysf;

template <int NC, int S>
static void run_stage(std::vector<cf>& z) {
  constexpr int L = FftGeom<NC>::L;
  std::vector<float> bufr(FftGeom<NC>::XSTRIDE + 64), bufi(FftGeom<NC>::XSTRIDE + 64);
  if (S > 0) {
    for (int ls = 0; ls < L; ++ls)
      for (int p = 0; p < kPts; ++p) {
        bufr[pad_idx(fft_out_index<NC, (S > 0 ? S - 1 : 0)>(ls, p))] = z[ls * kPts + p][0];
        bufi[pad_idx(fft_out_index<NC, (S > 0 ? S - 1 : 0)>(ls, p))] = z[ls * kPts + p][1];
      }
    for (int ls = 0; ls < L; ++ls)
      for (int p = 0; p < kPts; ++p) {
        z[ls * kPts + p][0] = bufr[pad_idx(fft_in_index<NC>(ls, p))];
        z[ls * kPts + p][1] = bufi[pad_idx(fft_in_index<NC>(ls, p))];
      }
  }
  for (int ls = 0; ls < L; ++ls) {
    cf tw[kPts];
    for (int p = 0; p < kPts; ++p) {
      int num, den;
      fft_twiddle_frac<NC, S>(ls, p, &num, &den);
      double a = -2.0 * M_PI * (double)num / (double)den;
      tw[p] = mk((float)std::cos(a), (float)std::sin(a));
    }
    fft_stage<NC, S>(&z[ls * kPts], tw);
  }
}

template <int NC>
static double test_one(unsigned seed) {
  constexpr int L = FftGeom<NC>::L;
  constexpr int NS = FftGeom<NC>::NSTAGE;
  const int N = 2 * NC;
  std::vector<double> x(N);
  srand(seed);
  for (int n = 0; n < N; ++n) x[n] = (rand() / (double)RAND_MAX - 0.5) * 2.0;
  std::vector<cf> z(L * kPts);
  for (int ls = 0; ls < L; ++ls)
    for (int p = 0; p < kPts; ++p) {
      int n = fft_in_index<NC>(ls, p);
      z[ls * kPts + p] = mk(0.5f * (float)x[2 * n], 0.5f * (float)x[2 * n + 1]);  // the kernel's window carries the 1/2
    }
  run_stage<NC, 0>(z);
  run_stage<NC, 1>(z);
  if (NS == 3) run_stage<NC, 2>(z);
  // final store, natural order
  std::vector<float> zr(FftGeom<NC>::XSTRIDE + 64), zi(FftGeom<NC>::XSTRIDE + 64);
  for (int ls = 0; ls < L; ++ls)
    for (int p = 0; p < kPts; ++p) {
      int o = (NS == 3) ? fft_out_index<NC, 2>(ls, p) : fft_out_index<NC, 1>(ls, p);
      zr[pad_idx(o)] = z[ls * kPts + p][0];
      zi[pad_idx(o)] = z[ls * kPts + p][1];
    }
  std::vector<float> S(NC + 1, -1.f);
  for (int ls = 0; ls < L; ++ls)
    for (int i = 0; i <= 8; ++i) {
      if (i == 8 && ls != 0) continue;
      int k = (i == 8) ? NC / 2 : ls + L * i;
      int m = (NC - k) % NC;
      double a = -2.0 * M_PI * k / (double)N;
      float pk, pm;
      rfft_pair_power(mk(zr[pad_idx(k)], zi[pad_idx(k)]), mk(zr[pad_idx(m)], zi[pad_idx(m)]),
                      mk((float)std::sin(a), (float)(-std::cos(a))), &pk, &pm);
      S[k] = pk;
      S[NC - k] = pm;
    }
  double worst = 0, scale = 0;
  for (int k = 0; k <= NC; ++k) {
    double sr = 0, si = 0;
    for (int n = 0; n < N; ++n) {
      double a = -2.0 * M_PI * (double)((long long)k * n % N) / (double)N;
      sr += (double)(float)x[n] * std::cos(a);
      si += (double)(float)x[n] * std::sin(a);
    }
    double ref = sr * sr + si * si;
    scale = std::fmax(scale, ref);
    worst = std::fmax(worst, std::fabs(ref - (double)S[k]));
  }
  return worst / scale;
}

int main() {
  int bad = 0;
  double e;
  e = test_one<128>(1);  printf("NC=128  rel_err=%.3e\n", e); bad += !(e < 2e-6);
  e = test_one<256>(2);  printf("NC=256  rel_err=%.3e\n", e); bad += !(e < 2e-6);
  e = test_one<512>(3);  printf("NC=512  rel_err=%.3e\n", e); bad += !(e < 2e-6);
  e = test_one<1024>(4); printf("NC=1024 rel_err=%.3e\n", e); bad += !(e < 2e-6);
  printf(bad ? "FAIL\n" : "OK\n");
  return bad;
}
