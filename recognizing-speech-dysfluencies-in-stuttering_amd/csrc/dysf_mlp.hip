// dysf_mlp.hip -- the classify step's MLP head as ONE kernel: StandardScaler + dense/ReLU layers + logits, softmax
// and arg-max.  Replaces scaler.transform + MLPClassifier.predict / predict_proba (pipeline1.py:470-473, 497,
// 512-514) for batches that are already on the GPU: the eager version is seven tiny launches, which cost more than
// the arithmetic (27 456 MACs per row for 149 -> 128 -> 64 -> 3).
// A workgroup owns kRows rows: the scaled rows sit in LDS, the layer's weights stream through an LDS tile (coalesced
// block copies of up to 32 KB), every thread produces outputs (row, unit) for a fixed unit and a strided
// set of rows: consecutive lanes read consecutive weights and broadcast the activations.  float32 FMA throughout;
// hidden activations ping-pong between two LDS buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "../../include/dysf_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 4;        // rows per workgroup
constexpr int kThreads = 256;
constexpr int kMaxLayers = 4;
constexpr int kMaxWidth = 512;  // widest layer (input or hidden) held in LDS
constexpr int kTileK = 16;     // K block of the unrolled inner product (weight rows are padded to whole blocks)
constexpr int kTileFloats = 8192;  // LDS weight tile (32 KB)
constexpr int kPf = kTileFloats / 4 / kThreads;  // 16-byte prefetch registers per thread

struct MlpDev {
  int n_layers, n_pass, n_features, n_outputs;
  int width;                // LDS row stride of the activation buffers
  const int* layer;         // [n_layers][3] = {din, dout, offset of the bias in params}
  const int* pass;          // [n_pass][4] = {layer, t0, rows, offset in params of row t0 of that layer's matrix}
  const float* params;      // every layer's zero-padded weights (row-major [din_pad][dout]) and biases
  const float* mean; const float* inv_scale;
};

__global__ __launch_bounds__(kThreads) void dysf_mlp_kernel(MlpDev m, const float* x, int n, float* logits, float* proba,
                                                            int* pred) {
  extern __shared__ __align__(16) float act[];  // two activation buffers of [kRows][width], then the weight tile
  float* cur = act;
  float* nxt = act + kRows * m.width;
  float* wt = act + 2 * kRows * m.width;        // [kTileK][dout] slice of the current layer's weights
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * kRows;
  const int d0 = m.n_features;
  // the K loop below runs in whole tiles: the columns it touches past a layer's width must hold finite numbers
  // (they meet zero weights), so both buffers start out as zeros
  for (int i = tid; i < 2 * kRows * m.width; i += kThreads) act[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < kRows * d0; i += kThreads) {
    const int r = i / d0, j = i % d0;
    const int row = row0 + r;
    cur[r * m.width + j] = (row < n) ? (x[(size_t)row * d0 + j] - m.mean[j]) * m.inv_scale[j] : 0.f;
  }
  // Weight tiles stream through registers one pass ahead (also across layer boundaries), so only the very first
  // fetch is exposed: a pass = up to 32 KB of consecutive rows of one layer's matrix.
  f32x4 pf[kPf];
  auto fetch = [&](int p) {
    const int dout = m.layer[3 * m.pass[4 * p] + 1];
    const int nvec = m.pass[4 * p + 2] * dout / 4;
    const f32x4* src = (const f32x4*)(m.params + m.pass[4 * p + 3]);
#pragma unroll
    for (int j = 0; j < kPf; ++j) {  // branch-free (clamped) so the registers stay registers
      const int e = tid + j * kThreads;
      pf[j] = src[e < nvec ? e : nvec - 1];
    }
  };
  fetch(0);
  int p = 0;
  for (int l = 0; l < m.n_layers; ++l) {
    const int dout = m.layer[3 * l + 1];
    const bool last = l + 1 == m.n_layers;
    // outputs (r, u): u fastest across threads; a thread keeps its unit and owns rows rb, rb + rstep, ...
    const int rstep = kThreads / dout;            // dout <= kThreads (checked by dysf_mlp_create)
    const int rpt = (kRows + rstep - 1) / rstep;  // rows per thread (workgroup-uniform)
    const int u = tid % dout, rb = tid / dout;
    const bool owner = rb < rstep && rb < kRows;
    float acc[kRows];
    const float* arow[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      acc[i] = 0.f;
      const int r = rb + i * rstep;
      arow[i] = cur + (r < kRows ? r : kRows - 1) * m.width;
    }
    for (; p < m.n_pass && m.pass[4 * p] == l; ++p) {
      const int t0 = m.pass[4 * p + 1], rows = m.pass[4 * p + 2];
      const int nvec = rows * dout / 4;
      __syncthreads();  // previous tile consumed (and, first time round, the activations are in place)
#pragma unroll
      for (int j = 0; j < kPf; ++j)
        if (tid + j * kThreads < nvec) ((f32x4*)wt)[tid + j * kThreads] = pf[j];
      __syncthreads();
      if (p + 1 < m.n_pass) fetch(p + 1);
      if (owner) {
        for (int sb = 0; sb < rows; sb += kTileK) {
          float wv[kTileK];
#pragma unroll
          for (int kk = 0; kk < kTileK; ++kk) wv[kk] = wt[(sb + kk) * dout + u];
#pragma unroll
          for (int i = 0; i < kRows; ++i)
            if (i < rpt) {
              const float4* a4 = (const float4*)(arow[i] + t0 + sb);
#pragma unroll
              for (int q = 0; q < kTileK / 4; ++q) {
                const float4 av = a4[q];
                acc[i] = fmaf(av.x, wv[4 * q + 0], acc[i]);
                acc[i] = fmaf(av.y, wv[4 * q + 1], acc[i]);
                acc[i] = fmaf(av.z, wv[4 * q + 2], acc[i]);
                acc[i] = fmaf(av.w, wv[4 * q + 3], acc[i]);
              }
            }
        }
      }
    }
    if (owner) {
      const float bias = m.params[m.layer[3 * l + 2] + u];
#pragma unroll
      for (int i = 0; i < kRows; ++i) {
        const int r = rb + i * rstep;
        if (i < rpt && r < kRows) {
          float v = acc[i] + bias;
          if (!last) v = fmaxf(v, 0.f);
          nxt[r * m.width + u] = v;
        }
      }
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
  // epilogue: one thread per row
  const int dout = m.n_outputs;
  if (tid < kRows && row0 + tid < n) {
    const int row = row0 + tid;
    const float* z = cur + tid * m.width;
    float zmax = z[0];
    int best = 0;
    for (int c = 1; c < dout; ++c) if (z[c] > zmax) { zmax = z[c]; best = c; }
    if (logits) for (int c = 0; c < dout; ++c) logits[(size_t)row * dout + c] = z[c];
    if (pred) pred[row] = best;
    if (proba) {
      float s = 0.f;
      for (int c = 0; c < dout; ++c) s += expf(z[c] - zmax);
      for (int c = 0; c < dout; ++c) proba[(size_t)row * dout + c] = expf(z[c] - zmax) / s;
    }
  }
}

}  // namespace

struct DysfMlp {
  int device = 0;
  bool attr_set = false;
  MlpDev d{};
  std::vector<void*> bufs;
};

static thread_local std::string g_merr;
template <typename T>
static bool up(DysfMlp* m, const T** dst, const std::vector<T>& src) {
  void* p = nullptr;
  if (hipMalloc(&p, src.size() * sizeof(T)) != hipSuccess) return false;
  m->bufs.push_back(p);
  if (hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
  *dst = (const T*)p;
  return true;
}

extern "C" {

const char* dysf_mlp_last_error(void) { return g_merr.c_str(); }

int32_t dysf_mlp_create(int32_t device, int32_t n_layers, const int32_t* dims, const double* const* weights,
                        const double* const* biases, const double* mean, const double* scale, DysfMlp** out) {
  if (!out || !dims || !weights || !biases || !mean || !scale) { g_merr = "null argument"; return DYSF_E_INVALID; }
  *out = nullptr;
  if (n_layers < 1 || n_layers > kMaxLayers) { g_merr = "1..4 layers supported"; return DYSF_E_INVALID; }
  int width = 0;
  for (int l = 0; l <= n_layers; ++l) {
    if (dims[l] < 1 || dims[l] > (l == 0 ? kMaxWidth : kThreads)) { g_merr = "width out of range (input 1..512, layers 1..256)"; return DYSF_E_INVALID; }
    width = dims[l] > width ? dims[l] : width;
  }
  for (int l = 0; l < n_layers; ++l) if (!weights[l] || !biases[l]) { g_merr = "null layer"; return DYSF_E_INVALID; }
  DysfMlp* m = new (std::nothrow) DysfMlp();
  if (!m) { g_merr = "out of host memory"; return DYSF_E_NOMEM; }
  m->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete m; g_merr = "hipSetDevice failed"; return DYSF_E_HIP; }
  m->d.n_layers = n_layers; m->d.n_features = dims[0]; m->d.n_outputs = dims[n_layers];
  m->d.width = (width + kTileK - 1) / kTileK * kTileK + 4;  // whole K blocks per row, 16-byte aligned rows
  std::vector<float> params;
  std::vector<int> layer_tab, pass_tab;
  for (int l = 0; l < n_layers; ++l) {
    const int din_pad = (dims[l] + kTileK - 1) / kTileK * kTileK;  // zero rows up to a whole number of K blocks
    const int woff = (int)params.size();
    params.resize(woff + (size_t)din_pad * dims[l + 1], 0.f);
    for (size_t i = 0; i < (size_t)dims[l] * dims[l + 1]; ++i) params[woff + i] = (float)weights[l][i];
    const int boff = (int)params.size();
    for (int i = 0; i < dims[l + 1]; ++i) params.push_back((float)biases[l][i]);
    params.resize((params.size() + 3) / 4 * 4, 0.f);  // the next matrix starts 16-byte aligned
    layer_tab.insert(layer_tab.end(), {dims[l], dims[l + 1], boff});
    int ts = kTileFloats / dims[l + 1] / kTileK * kTileK;
    ts = ts < kTileK ? kTileK : ts;
    for (int t0 = 0; t0 < din_pad; t0 += ts)
      pass_tab.insert(pass_tab.end(), {l, t0, std::min(ts, din_pad - t0), woff + t0 * dims[l + 1]});
  }
  m->d.n_pass = (int)pass_tab.size() / 4;
  std::vector<float> mu((size_t)dims[0]), inv((size_t)dims[0]);
  for (int j = 0; j < dims[0]; ++j) { mu[j] = (float)mean[j]; inv[j] = (float)(1.0 / scale[j]); }
  bool ok = up(m, &m->d.params, params) && up(m, &m->d.mean, mu) && up(m, &m->d.inv_scale, inv) &&
            up(m, &m->d.layer, layer_tab) && up(m, &m->d.pass, pass_tab);
  if (!ok) { for (void* p : m->bufs) (void)hipFree(p); delete m; g_merr = "device upload failed"; return DYSF_E_HIP; }
  *out = m;
  return 0;
}

void dysf_mlp_destroy(DysfMlp* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  for (void* p : m->bufs) (void)hipFree(p);
  delete m;
}

int32_t dysf_mlp_forward_dev(DysfMlp* m, const float* d_x, int32_t n, float* d_logits, float* d_proba, int32_t* d_pred,
                             void* stream) {
  if (!m || !d_x || n < 0) { g_merr = "bad argument"; return DYSF_E_INVALID; }
  if (n == 0) return 0;
  if (hipSetDevice(m->device) != hipSuccess) { g_merr = "hipSetDevice failed"; return DYSF_E_HIP; }
  const size_t lds = ((size_t)2 * kRows * m->d.width + (size_t)kTileFloats) * sizeof(float);
  if (!m->attr_set) {  // more than the default 64 KB of dynamic LDS
    if (hipFuncSetAttribute((const void*)dysf_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      g_merr = "hipFuncSetAttribute failed"; return DYSF_E_HIP;
    }
    m->attr_set = true;
  }
  hipLaunchKernelGGL(dysf_mlp_kernel, dim3((n + kRows - 1) / kRows), dim3(kThreads), lds, (hipStream_t)stream, m->d, d_x,
                     n, d_logits, d_proba, d_pred);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_merr = hipGetErrorString(e); return DYSF_E_HIP; }
  return 0;
}

}  // extern "C"
