// dysf_mlp.hip -- the classify step's MLP head as ONE kernel: StandardScaler + dense/ReLU layers + logits, softmax
// and arg-max.  Replaces scaler.transform + MLPClassifier.predict / predict_proba (pipeline1.py:470-473, 497,
// 512-514) for batches that are already on the GPU: the eager version is seven tiny launches, which cost more than
// the arithmetic (27 456 MACs per row for 149 -> 128 -> 64 -> 3).
// A workgroup of four waves owns 16 rows.  The scaled rows and the hidden activations sit in LDS as [16][width]; every
// layer is out[16 x dout] = act[16 x din] * W on v_mfma_f32_16x16x4_f32: a wave takes the 16-unit output blocks
// w, w + 4, ... and walks K in groups of 16 -- lane (row = l & 15, q = l >> 4) reads its four consecutive activations
// 16g + 4q .. + 3 with one ds_read_b128, and the weights come from a host-made pack in exactly that order
// ([layer][block][group][lane][4], zero padded): one 16-byte load per lane, group and block straight from L2 / HBM, eight
// groups requested before the first is used (in the pipeline the feature kernels have just streamed through L2, so these
// loads are not cheap: per-group loads inside the K loop cost 25 us per launch, a chunked prefetch sized for four blocks
// per wave cost co-resident kernels their registers, hence the template on the blocks a wave really owns).  A copy of the
// pack in LDS was measured too: equal at 1024 rows, twice as slow at 16 384 (one workgroup per CU).
// float32 accumulation in a fixed order: results do not depend on the batch.
#include <hip/hip_runtime.h>
#include <mutex>

#include <algorithm>
#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "../../include/dysf_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 16;       // rows per workgroup (one MFMA tile)
constexpr int kThreads = 256;   // four waves
constexpr int kWaves = kThreads / 64;
constexpr int kMaxLayers = 4;
constexpr int kMaxWidth = 512;  // widest layer (input or hidden) held in LDS
constexpr int kMaxBlocks = 4;   // 16-unit output blocks per wave: layers up to 256 units

struct MlpDev {
  int n_layers, n_features, n_outputs;
  int width;                // LDS row stride of the activation buffers (floats)
  const int* layer;         // [n_layers][4] = {din, dout, offset of the bias in params, offset of the weight pack in params}
  const float* params;      // every layer's weight pack ([block][group][lane][4]) and biases
  const float* mean; const float* inv_scale;
};

// NB: 16-unit output blocks a wave may own (the widest layer decides: 1, 2 or 4) -- sizes the weight registers
template <int NB>
__global__ __launch_bounds__(kThreads) void dysf_mlp_kernel(MlpDev m, const float* x, int n, float* logits, float* proba,
                                                            int* pred) {
  extern __shared__ __align__(16) float act[];  // two activation buffers of [kRows][width]
  float* cur = act;
  float* nxt = act + kRows * m.width;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * kRows;
  const int d0 = m.n_features;
  // K is walked in whole groups of 16: the columns past a layer's width meet zero weights but must hold finite numbers
  for (int i = tid; i < 2 * kRows * m.width; i += kThreads) act[i] = 0.f;
  __syncthreads();
  // scaled inputs: a thread keeps its column(s) j = tid (+ 256) and walks the 16 rows -- no index division, the scaler
  // terms loaded once, and all 16 row loads of a column in flight together (rows past the batch repeat its last row)
  for (int j = tid; j < d0; j += kThreads) {
    const float mu = m.mean[j], is = m.inv_scale[j];
    float xv[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) xv[r] = x[(size_t)min(row0 + r, n - 1) * d0 + j];
#pragma unroll
    for (int r = 0; r < kRows; ++r) cur[r * m.width + j] = (row0 + r < n) ? (xv[r] - mu) * is : 0.f;
  }
  __syncthreads();
  const float* params = m.params;
  const int arow = (lane & 15) * m.width + 4 * (lane >> 4);   // this lane's A operand: row l & 15, K offset 4 * (l >> 4)
  for (int l = 0; l < m.n_layers; ++l) {
    const int din = m.layer[4 * l], dout = m.layer[4 * l + 1];
    const bool last = l + 1 == m.n_layers;
    const int ngrp = (din + 15) / 16, nblk = (dout + 15) / 16;
    const float* wp = params + m.layer[4 * l + 3] + lane * 4;
    f32x4 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // eight K groups at a time: all weight loads of the chunk first (they come from L2 or, after the feature kernels have
    // streamed through it, from HBM -- one exposed latency per chunk instead of one per group), then the MFMAs
    constexpr int GC = 8;
    for (int g0 = 0; g0 < ngrp; g0 += GC) {
      float4 w4[GC][NB];
#pragma unroll
      for (int gi = 0; gi < GC; ++gi)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int blk = wave + i * kWaves;   // wave-uniform
          if (g0 + gi < ngrp && blk < nblk) w4[gi][i] = *(const float4*)(wp + ((size_t)blk * ngrp + g0 + gi) * 256);
        }
#pragma unroll
      for (int gi = 0; gi < GC; ++gi) {
        if (g0 + gi >= ngrp) break;
        const float4 a4 = *(const float4*)(cur + arow + 16 * (g0 + gi));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int blk = wave + i * kWaves;
          if (blk < nblk) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w4[gi][i].x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w4[gi][i].y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w4[gi][i].z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w4[gi][i].w, acc[i], 0, 0, 0);
          }
        }
      }
    }
    // C/D map: acc[i][r] = output unit 16 * blk + (l & 15) of row 4 * (l >> 4) + r
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int blk = wave + i * kWaves;
      const int u = 16 * blk + (lane & 15);
      if (blk < nblk && u < dout) {
        const float bias = params[m.layer[4 * l + 2] + u];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][r] + bias;
          if (!last) v = fmaxf(v, 0.f);
          nxt[(4 * (lane >> 4) + r) * m.width + u] = v;
        }
      }
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
    // (columns past dout keep older, finite values: they only ever meet the zero padding of a weight pack)
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
  int blocks_per_wave = 1;   // of the widest layer: ceil(ceil(dout / 16) / 4)
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

// The dynamic-LDS attribute belongs to the kernel function (per device), not to a handle: one process-wide table keyed by
// function and device, only ever raised (handles on different streams see different batch widths).
static bool grant_dynamic_lds(const void* kern, size_t lds_bytes) {
  struct Entry { const void* fn; int dev; size_t bytes; };
  static std::mutex mu;
  static std::vector<Entry> table;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  Entry* e = nullptr;
  for (Entry& x : table) if (x.fn == kern && x.dev == dev) { e = &x; break; }
  if (e && lds_bytes <= e->bytes) return true;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return false;
  if (e) e->bytes = lds_bytes; else table.push_back({kern, dev, lds_bytes});
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
  m->d.width = (width + 15) / 16 * 16 + 4;  // whole K groups per row, 16-byte aligned rows
  std::vector<float> params;
  std::vector<int> layer_tab;
  for (int l = 0; l < n_layers; ++l) {
    const int din = dims[l], dout = dims[l + 1];
    const int ngrp = (din + 15) / 16, nblk = (dout + 15) / 16;
    const int woff = (int)params.size();
    params.resize(woff + (size_t)nblk * ngrp * 256, 0.f);
    // pack[block][group][lane][s] = W[16 * group + 4 * (lane >> 4) + s][16 * block + (lane & 15)]  (weights[l] is [din][dout])
    for (int blk = 0; blk < nblk; ++blk)
      for (int g = 0; g < ngrp; ++g)
        for (int ln = 0; ln < 64; ++ln)
          for (int sidx = 0; sidx < 4; ++sidx) {
            const int k = 16 * g + 4 * (ln >> 4) + sidx, u = 16 * blk + (ln & 15);
            if (k < din && u < dout) params[woff + (((size_t)blk * ngrp + g) * 64 + ln) * 4 + sidx] = (float)weights[l][(size_t)k * dout + u];
          }
    const int boff = (int)params.size();
    for (int i = 0; i < dout; ++i) params.push_back((float)biases[l][i]);
    params.resize((params.size() + 3) / 4 * 4, 0.f);  // the next pack starts 16-byte aligned
    layer_tab.insert(layer_tab.end(), {din, dout, boff, woff});
    m->blocks_per_wave = std::max(m->blocks_per_wave, (nblk + kWaves - 1) / kWaves);
  }
  std::vector<float> mu((size_t)dims[0]), inv((size_t)dims[0]);
  for (int j = 0; j < dims[0]; ++j) { mu[j] = (float)mean[j]; inv[j] = (float)(1.0 / scale[j]); }
  bool ok = up(m, &m->d.params, params) && up(m, &m->d.mean, mu) && up(m, &m->d.inv_scale, inv) &&
            up(m, &m->d.layer, layer_tab);
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
  const size_t lds = (size_t)2 * kRows * m->d.width * sizeof(float);   // <= 66 KB
  auto launch = [&](auto kernel) -> bool {
    if (lds > 64 * 1024 && !grant_dynamic_lds((const void*)kernel, lds)) return false;  // more than the default 64 KB of dynamic LDS
    hipLaunchKernelGGL(kernel, dim3((n + kRows - 1) / kRows), dim3(kThreads), lds, (hipStream_t)stream, m->d, d_x, n, d_logits,
                       d_proba, d_pred);
    return true;
  };
  const bool ok = m->blocks_per_wave <= 1 ? launch(dysf_mlp_kernel<1>) : m->blocks_per_wave == 2 ? launch(dysf_mlp_kernel<2>)
                                                                                                  : launch(dysf_mlp_kernel<4>);
  if (!ok) { g_merr = "hipFuncSetAttribute failed"; return DYSF_E_HIP; }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_merr = hipGetErrorString(e); return DYSF_E_HIP; }
  return 0;
}

}  // extern "C"
