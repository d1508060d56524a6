// dysf_svc.hip -- RBF support-vector classifier inference for the classify step.
// The reference trains SVC(probability=True, kernel='rbf') next to the forest and the MLP (pipeline1.py:498,
// main1.py:865) and evaluates predict / predict_proba on standard-scaled rows (pipeline1.py:512-514); main1.py:867
// soft-votes the three heads.  scikit-learn delegates to libsvm, whose published prediction algorithm this follows:
//   k(x, sv) = exp(-gamma * sum_f (x_f - sv_f)^2)                      (Kernel::k_function, dense RBF, float64)
//   dec(i,j) = sum_{sv in i} coef[j-1][sv] k + sum_{sv in j} coef[i][sv] k - rho(i,j)     (svm_predict_values)
//   predict  = class with most pairwise votes, first maximum                              (svm_predict_values)
//   proba    = multiclass_probability over r(i,j) = clip(sigmoid_predict(dec, A, B))      (svm_predict_probability)
// One wave per row: the lanes scale the row with StandardScaler's float32 rounding steps, lane l evaluates the
// kernel for support vectors l, l+64, ... (matrix stored feature-major so the wave reads it coalesced), the pair sums
// are reduced across the wave in float64 and lane 0 runs the tiny vote / Platt / coupling epilogue.
#include <hip/hip_runtime.h>

#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "../../include/dysf_hip.h"

namespace {

constexpr int kMaxClasses = 8;
constexpr int kMaxPairs = kMaxClasses * (kMaxClasses - 1) / 2;

struct SvcDev {
  int n_features, n_classes, n_pairs, n_sv, has_prob;
  double gamma;
  const double* svt;        // [n_features][n_sv]
  const double* pair_coef;  // [n_pairs][n_sv]: the dual coefficient a support vector carries into each class pair
  const double* intercept;  // [n_pairs] = -rho
  const double* prob_a; const double* prob_b;
  const double* mean; const double* scale;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void dysf_svc_kernel(SvcDev m, const float* x, int n, double* decision, double* proba,
                                                       int* pred) {
  extern __shared__ double xs_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  double* xs = xs_all + wave * m.n_features;
  if (row < n) {
    for (int j = lane; j < m.n_features; j += 64) {
      const float centred = (float)((double)x[(size_t)row * m.n_features + j] - m.mean[j]);
      xs[j] = (double)(float)((double)centred / m.scale[j]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (row >= n) return;
  double part[kMaxPairs];
#pragma unroll
  for (int p = 0; p < kMaxPairs; ++p) part[p] = 0.0;
  for (int s = lane; s < m.n_sv; s += 64) {
    double d2 = 0.0;
    for (int f = 0; f < m.n_features; ++f) {
      const double d = xs[f] - m.svt[(size_t)f * m.n_sv + s];
      d2 = __dadd_rn(d2, __dmul_rn(d, d));  // libsvm accumulates without contraction
    }
    const double k = exp(-m.gamma * d2);
#pragma unroll
    for (int p = 0; p < kMaxPairs; ++p)
      if (p < m.n_pairs) part[p] += m.pair_coef[(size_t)p * m.n_sv + s] * k;
  }
  double dec[kMaxPairs];
#pragma unroll
  for (int p = 0; p < kMaxPairs; ++p) dec[p] = (p < m.n_pairs) ? wave_sum_f64(part[p]) + m.intercept[p] : 0.0;
  if (lane != 0) return;
  const int K = m.n_classes;
  if (decision)
    for (int p = 0; p < m.n_pairs; ++p) decision[(size_t)row * m.n_pairs + p] = dec[p];
  if (pred) {
    int votes[kMaxClasses];
    for (int c = 0; c < K; ++c) votes[c] = 0;
    int p = 0;
    for (int i = 0; i < K; ++i)
      for (int j = i + 1; j < K; ++j, ++p) {
        if (dec[p] > 0) ++votes[i]; else ++votes[j];
      }
    int best = 0;
    for (int c = 1; c < K; ++c) if (votes[c] > votes[best]) best = c;
    pred[row] = best;
  }
  if (proba && m.has_prob) {
    double r[kMaxClasses][kMaxClasses], Q[kMaxClasses][kMaxClasses], Qp[kMaxClasses], pr[kMaxClasses];
    int p = 0;
    for (int i = 0; i < K; ++i)
      for (int j = i + 1; j < K; ++j, ++p) {
        const double f = dec[p] * m.prob_a[p] + m.prob_b[p];
        double s = (f >= 0) ? exp(-f) / (1.0 + exp(-f)) : 1.0 / (1.0 + exp(f));
        s = fmin(fmax(s, 1e-7), 1.0 - 1e-7);
        r[i][j] = s; r[j][i] = 1.0 - s;
      }
    for (int t = 0; t < K; ++t) {
      pr[t] = 1.0 / K;
      Q[t][t] = 0.0;
      for (int j = 0; j < t; ++j) { Q[t][t] += r[j][t] * r[j][t]; Q[t][j] = Q[j][t]; }
      for (int j = t + 1; j < K; ++j) { Q[t][t] += r[j][t] * r[j][t]; Q[t][j] = -r[j][t] * r[t][j]; }
    }
    const double eps = 0.005 / K;
    const int max_iter = K > 100 ? K : 100;
    for (int it = 0; it < max_iter; ++it) {
      double pQp = 0.0;
      for (int t = 0; t < K; ++t) {
        Qp[t] = 0.0;
        for (int j = 0; j < K; ++j) Qp[t] += Q[t][j] * pr[j];
        pQp += pr[t] * Qp[t];
      }
      double max_error = 0.0;
      for (int t = 0; t < K; ++t) max_error = fmax(max_error, fabs(Qp[t] - pQp));
      if (max_error < eps) break;
      for (int t = 0; t < K; ++t) {
        const double diff = (-Qp[t] + pQp) / Q[t][t];
        pr[t] += diff;
        pQp = (pQp + diff * (diff * Q[t][t] + 2.0 * Qp[t])) / (1.0 + diff) / (1.0 + diff);
        for (int j = 0; j < K; ++j) { Qp[j] = (Qp[j] + diff * Q[t][j]) / (1.0 + diff); pr[j] /= (1.0 + diff); }
      }
    }
    for (int c = 0; c < K; ++c) proba[(size_t)row * K + c] = pr[c];
  }
}

}  // namespace

struct DysfSvc {
  int device = 0;
  SvcDev d{};
  std::vector<void*> bufs;
};

static thread_local std::string g_serr;
static bool up(DysfSvc* m, const double** dst, const double* src, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(double)) != hipSuccess) return false;
  m->bufs.push_back(p);
  if (hipMemcpy(p, src, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return false;
  *dst = (const double*)p;
  return true;
}

extern "C" {

const char* dysf_svc_last_error(void) { return g_serr.c_str(); }

int32_t dysf_svc_create(int32_t device, int32_t n_features, int32_t n_classes, int32_t n_sv,
                        const double* support_vectors, const double* dual_coef, const double* intercept,
                        const int32_t* n_support, const double* prob_a, const double* prob_b, double gamma,
                        const double* mean, const double* scale, DysfSvc** out) {
  if (!out || !support_vectors || !dual_coef || !intercept || !n_support || !mean || !scale) { g_serr = "null argument"; return DYSF_E_INVALID; }
  *out = nullptr;
  if (n_features < 1 || n_classes < 2 || n_classes > kMaxClasses || n_sv < 1 || !(gamma > 0.0)) { g_serr = "bad model shape"; return DYSF_E_INVALID; }
  if ((prob_a == nullptr) != (prob_b == nullptr)) { g_serr = "prob_a and prob_b go together"; return DYSF_E_INVALID; }
  int64_t tot = 0;
  for (int c = 0; c < n_classes; ++c) { if (n_support[c] < 0) { g_serr = "negative n_support"; return DYSF_E_INVALID; } tot += n_support[c]; }
  if (tot != n_sv) { g_serr = "n_support does not add up to n_sv"; return DYSF_E_INVALID; }
  const int n_pairs = n_classes * (n_classes - 1) / 2;
  // feature-major support vectors; per-pair coefficients (libsvm: pair (i,j) takes row j-1 for class-i vectors and
  // row i for class-j vectors of sv_coef)
  std::vector<double> svt((size_t)n_features * n_sv), pc((size_t)n_pairs * n_sv, 0.0);
  for (int s = 0; s < n_sv; ++s)
    for (int f = 0; f < n_features; ++f) svt[(size_t)f * n_sv + s] = support_vectors[(size_t)s * n_features + f];
  std::vector<int> start(n_classes + 1, 0);
  for (int c = 0; c < n_classes; ++c) start[c + 1] = start[c] + n_support[c];
  int p = 0;
  for (int i = 0; i < n_classes; ++i)
    for (int j = i + 1; j < n_classes; ++j, ++p) {
      for (int s = start[i]; s < start[i + 1]; ++s) pc[(size_t)p * n_sv + s] = dual_coef[(size_t)(j - 1) * n_sv + s];
      for (int s = start[j]; s < start[j + 1]; ++s) pc[(size_t)p * n_sv + s] = dual_coef[(size_t)i * n_sv + s];
    }
  DysfSvc* m = new (std::nothrow) DysfSvc();
  if (!m) { g_serr = "out of host memory"; return DYSF_E_NOMEM; }
  m->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete m; g_serr = "hipSetDevice failed"; return DYSF_E_HIP; }
  m->d.n_features = n_features; m->d.n_classes = n_classes; m->d.n_pairs = n_pairs; m->d.n_sv = n_sv;
  m->d.has_prob = prob_a ? 1 : 0; m->d.gamma = gamma;
  bool ok = up(m, &m->d.svt, svt.data(), svt.size()) && up(m, &m->d.pair_coef, pc.data(), pc.size()) &&
            up(m, &m->d.intercept, intercept, (size_t)n_pairs) && up(m, &m->d.mean, mean, (size_t)n_features) &&
            up(m, &m->d.scale, scale, (size_t)n_features);
  if (ok && prob_a) ok = up(m, &m->d.prob_a, prob_a, (size_t)n_pairs) && up(m, &m->d.prob_b, prob_b, (size_t)n_pairs);
  if (!ok) { for (void* q : m->bufs) (void)hipFree(q); delete m; g_serr = "device upload failed"; return DYSF_E_HIP; }
  *out = m;
  return 0;
}

void dysf_svc_destroy(DysfSvc* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  for (void* p : m->bufs) (void)hipFree(p);
  delete m;
}

int32_t dysf_svc_predict_dev(DysfSvc* m, const float* d_x, int32_t n, double* d_decision, double* d_proba,
                             int32_t* d_pred, void* stream) {
  if (!m || !d_x || n < 0) { g_serr = "bad argument"; return DYSF_E_INVALID; }
  if (d_proba && !m->d.has_prob) { g_serr = "model was created without Platt coefficients"; return DYSF_E_INVALID; }
  if (n == 0) return 0;
  if (hipSetDevice(m->device) != hipSuccess) { g_serr = "hipSetDevice failed"; return DYSF_E_HIP; }
  const size_t lds = (size_t)4 * m->d.n_features * sizeof(double);
  hipLaunchKernelGGL(dysf_svc_kernel, dim3((n + 3) / 4), dim3(256), lds, (hipStream_t)stream, m->d, d_x, n, d_decision,
                     d_proba, d_pred);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_serr = hipGetErrorString(e); return DYSF_E_HIP; }
  return 0;
}

}  // extern "C"
