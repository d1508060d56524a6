// dysf_forest.hip -- RandomForest inference for the classify step.
// The reference's deployed predictor is a RandomForestClassifier on standard-scaled features:
//   feats_s = scaler.transform(feats.reshape(1, -1)); rf.predict(feats_s); rf.predict_proba(feats_s)   main1.py:987-989
//   (trained at pipeline1.py:496, 512-514).
// One wave per row: the lanes scale the row exactly as StandardScaler does on float32 input (float64 subtract,
// round to float32, float64 divide, round to float32), then each lane walks trees lane, lane+64, ... to their leaves;
// the leaf class fractions are summed in float64 in estimator order and divided by the tree count, which is
// bit-for-bit what ForestClassifier.predict_proba does; predict = first arg-max.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/dysf_hip.h"

namespace {

constexpr int kMaxClasses = 8;

struct ForestDev {
  int n_features, n_classes, n_trees;
  const int* feature; const float* thr; const int* left; const int* right; const double* value; const int* root;
  const double* mean; const double* scale;
};

__global__ __launch_bounds__(256) void dysf_forest_kernel(ForestDev f, const float* x, int n, double* proba, int* pred) {
  extern __shared__ float xs_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  float* xs = xs_all + wave * f.n_features;
  if (row < n) {
    for (int j = lane; j < f.n_features; j += 64) {
      const float centred = (float)((double)x[(size_t)row * f.n_features + j] - f.mean[j]);
      xs[j] = (float)((double)centred / f.scale[j]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (row >= n) return;
  // walk: lane t, t+64, ... each follow one tree to its leaf
  int* leaf = (int*)(xs_all + 4 * f.n_features) + wave * f.n_trees;
  for (int t = lane; t < f.n_trees; t += 64) {
    int node = f.root[t];
    int feat = f.feature[node];
    while (feat >= 0) {  // sklearn: go left when X[i, feature] <= threshold
      node = (xs[feat] <= f.thr[node]) ? f.left[node] : f.right[node];
      feat = f.feature[node];
    }
    leaf[t] = node;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // average: lane c adds the leaf fractions of class c in estimator order, exactly like sklearn's accumulation
  double v = -1.0;
  if (lane < f.n_classes) {
    v = 0.0;
    for (int t = 0; t < f.n_trees; ++t) v += f.value[(size_t)leaf[t] * f.n_classes + lane];
    v /= (double)f.n_trees;
    if (proba) proba[(size_t)row * f.n_classes + lane] = v;
  }
  int best = 0;
  double bestv = __shfl(v, 0);
  for (int c = 1; c < f.n_classes; ++c) {
    const double vc = __shfl(v, c);
    if (vc > bestv) { bestv = vc; best = c; }  // first maximum, like np.argmax
  }
  if (lane == 0 && pred) pred[row] = best;
}

}  // namespace

struct DysfForest {
  int device = 0;
  ForestDev d{};
  std::vector<void*> bufs;
};

static thread_local std::string g_ferr;
template <typename T>
static bool up(DysfForest* f, const T** dst, const T* src, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return false;
  f->bufs.push_back(p);
  if (hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
  *dst = (const T*)p;
  return true;
}

extern "C" {

const char* dysf_forest_last_error(void) { return g_ferr.c_str(); }

int32_t dysf_forest_create(int32_t device, int32_t n_features, int32_t n_classes, int32_t n_trees, int64_t n_nodes,
                           const int32_t* feature, const double* threshold, const int32_t* left, const int32_t* right,
                           const double* value, const int32_t* root, const double* mean, const double* scale,
                           DysfForest** out) {
  if (!out || !feature || !threshold || !left || !right || !value || !root || !mean || !scale) { g_ferr = "null argument"; return DYSF_E_INVALID; }
  *out = nullptr;
  if (n_features < 1 || n_classes < 1 || n_classes > kMaxClasses || n_trees < 1 || n_nodes < n_trees) { g_ferr = "bad forest shape"; return DYSF_E_INVALID; }
  for (int64_t i = 0; i < n_nodes; ++i) {
    if (feature[i] >= n_features) { g_ferr = "feature index out of range"; return DYSF_E_INVALID; }
    if (feature[i] >= 0 && (left[i] <= i || right[i] <= i || left[i] >= n_nodes || right[i] >= n_nodes)) {
      g_ferr = "children must follow their parent (guarantees that every walk ends)"; return DYSF_E_INVALID;
    }
  }
  for (int i = 0; i < n_trees; ++i) if (root[i] < 0 || root[i] >= n_nodes) { g_ferr = "root out of range"; return DYSF_E_INVALID; }
  // x <= threshold (float32 x, float64 threshold) == x <= the largest float32 not above the threshold
  std::vector<float> thr32((size_t)n_nodes);
  for (int64_t i = 0; i < n_nodes; ++i) {
    float t = (float)threshold[i];
    if ((double)t > threshold[i]) t = std::nextafterf(t, -INFINITY);
    thr32[(size_t)i] = t;
  }
  DysfForest* f = new (std::nothrow) DysfForest();
  if (!f) { g_ferr = "out of host memory"; return DYSF_E_NOMEM; }
  f->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete f; g_ferr = "hipSetDevice failed"; return DYSF_E_HIP; }
  f->d.n_features = n_features; f->d.n_classes = n_classes; f->d.n_trees = n_trees;
  bool ok = up(f, &f->d.feature, feature, (size_t)n_nodes) && up(f, &f->d.thr, thr32.data(), (size_t)n_nodes) &&
            up(f, &f->d.left, left, (size_t)n_nodes) && up(f, &f->d.right, right, (size_t)n_nodes) &&
            up(f, &f->d.value, value, (size_t)n_nodes * n_classes) && up(f, &f->d.root, root, (size_t)n_trees) &&
            up(f, &f->d.mean, mean, (size_t)n_features) && up(f, &f->d.scale, scale, (size_t)n_features);
  if (!ok) { for (void* p : f->bufs) (void)hipFree(p); delete f; g_ferr = "device upload failed"; return DYSF_E_HIP; }
  *out = f;
  return 0;
}

void dysf_forest_destroy(DysfForest* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  for (void* p : f->bufs) (void)hipFree(p);
  delete f;
}

int32_t dysf_forest_predict_dev(DysfForest* f, const float* d_x, int32_t n, double* d_proba, int32_t* d_pred, void* stream) {
  if (!f || !d_x || n < 0) { g_ferr = "bad argument"; return DYSF_E_INVALID; }
  if (n == 0) return 0;
  if (hipSetDevice(f->device) != hipSuccess) { g_ferr = "hipSetDevice failed"; return DYSF_E_HIP; }
  const size_t lds = (size_t)4 * (f->d.n_features + f->d.n_trees) * sizeof(float);
  hipLaunchKernelGGL(dysf_forest_kernel, dim3((n + 3) / 4), dim3(256), lds, (hipStream_t)stream, f->d, d_x, n, d_proba, d_pred);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_ferr = hipGetErrorString(e); return DYSF_E_HIP; }
  return 0;
}

}  // extern "C"
