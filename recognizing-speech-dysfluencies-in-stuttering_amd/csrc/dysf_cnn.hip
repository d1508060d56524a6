// dysf_cnn.hip -- the benchmark CNN head of BASELINE.json configs[2] / SURVEY.md 8d "Config 3" on the frame-level
// features:  Conv2d(1->32, 3x3, pad 1) -> ReLU -> MaxPool 2x2 -> Conv2d(32->64, 3x3, pad 1) -> ReLU ->
// global average pool -> Linear(64 -> n_classes).  (The reference has no CNN; the architecture is SURVEY's.)
// Two launches on the caller's stream (three with DYSF_CNN_SPLIT=1, which keeps conv1 as its own kernel and the
// activations in an HBM workspace; the default computes conv1 straight into conv2's LDS tile):
//   conv1 + ReLU + max-pool   float32 FMA, one thread per pooled pixel, all 32 channels; the weights are wave-uniform
//                             (scalar operands).  Writes the activations channel-last as float16 into a workspace
//                             with a zero border: act1[clip][62][152][32].
//   conv2 + ReLU + pooled sum implicit GEMM on v_mfma_f32_16x16x32_f16: M = 16 output channels per wave (its 18 A
//                             operands -- one v_mfma_f32_16x16x32_f16 step per tap, K = all 32 input channels -- stay
//                             in registers), N = 16 consecutive pixels of a row, K = 288, two pixel tiles in flight.  A workgroup stages kRowTile + 2 activation rows
//                             in LDS (pixel pitch 80 bytes: aligned, conflict-light 16-byte B-operand reads), float32
//                             accumulation, per-channel sums reduced in registers into the workgroup's slot of sums[clip][tile][64].
//   mean + Linear             one thread per (clip, class).
// conv2 is 92 % of the arithmetic (332 of 353 MFLOP per clip); float16 operands with float32 accumulation is the
// precision BASELINE configs[4] asks of the CNN ("fp16 MFMA CNN").  A handle created with DYSF_CNN_F32 runs conv2 on
// v_mfma_f32_16x16x4_f32 instead (float32 activations in LDS, an exact fmaf chain per output): the float32 result
// configs[2] is checked against (arg-max equal to the float32 CPU module on every clip), at 1/16 of the MFMA rate.
// Clips of different lengths share one launch: cols[clip] valid columns per clip (BASELINE configs[4]), everything
// past them -- conv padding, pooling, the global average -- behaves as if the clip had been run alone.
#include <hip/hip_runtime.h>
#include <mutex>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "../../include/dysf_hip.h"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kC1 = 32, kC2 = 64;
#ifndef DYSF_CNN_ROWS
#define DYSF_CNN_ROWS 4
#endif
constexpr int kRowTile = DYSF_CNN_ROWS;  // conv2 output rows per workgroup.  conv1 is computed for kRowTile + 2 rows (1.5x its arithmetic, and
                                     // conv1 is three quarters of the kernel's vector instructions); 6 or 8 rows need column tiles to keep two
                                     // workgroups per CU and measured 2-3 % slower (tools/cnn_rows.sh)
constexpr int kPixPitch = 40;        // float16 per pixel in LDS (32 channels + 8 of padding = 80 bytes: 16-byte aligned,
                                     // neighbouring pixels 20 banks apart)

struct CnnDev {
  int H, W, stride;                  // input rows (3*n_mfcc), valid columns (frames; the batch maximum), row pitch in floats
  int PH, PW;                        // pooled size: H/2, W/2 (batch maximum)
  const int* cols;                   // [n] valid columns of every clip (null: all W)
  int AW;                            // act1 row pitch in pixels (PW + 2, rounded up to a multiple of 16 + 2)
  int CW, TW, ctiles;                // conv2 workgroup tile: CW pooled columns (a multiple of 16), LDS row pitch TW = CW + 2
                                     // pixels, column tiles per row tile (wide maps are cut so that the tile fits in LDS)
  int n_classes;
  const float* w1;                   // conv1: [16 channel pairs][9 taps + bias][2]
  const _Float16* w2p;               // [4 oc tiles][9 taps][64 lanes][8] A operands of conv2
  const float* w2p32;                // float32 handles: [4 oc tiles][9 taps][2 groups of 16 ic][64 lanes][4 steps]
  const float* b2;                   // [64]
  const float* wfc; const float* bfc;  // [n_classes][64], [n_classes]
  _Float16* act1;                    // [clip][PH + 2][AW][32]
  float* sums;                       // [clip][row tiles][2][64] per-wave-group partial sums (added in order by the last kernel)
  int n_tiles;
};

// ---------------------------------------------------------------------------------------------------------------
// conv1 + ReLU + 2x2 max-pool of one pooled pixel (py, px), all 32 channels, written as float16 to dst[0..31]
// acc + p.{lo | hi} * w as ONE v_pk_fma_f32: two output channels per instruction, the patch value broadcast to both halves
// by op_sel, the weight pair a scalar operand.  (v_pk_fma_f32 is not affected by the co-execution hazard of the packed add /
// multiply forms, see dysf_fft.hpp.)
template <int HALF>
__device__ __forceinline__ f2 pk_fma_bcast(f2 p, f2 w, f2 acc) {
  if (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(p), "s"(w));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(p), "s"(w));
  return acc;
}

// the 4 x 4 input patch of pooled pixel (py, px): rows 2py-1 .. 2py+2, columns 2px-1 .. 2px+2 as pairs, zero outside the
// [H x W] image
__device__ __forceinline__ void conv1_load_patch(const CnnDev& m, const float* in, int W, int py, int px, f2 (&p)[4][2]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = 2 * py - 1 + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int x = 2 * px - 1 + c;
      // (guarded loads: clamping the coordinates instead and zeroing afterwards measured 10 % slower on the whole kernel)
      p[r][c >> 1][c & 1] = (y >= 0 && y < m.H && x >= 0 && x < W) ? in[(size_t)y * m.stride + x] : 0.f;
    }
  }
}

template <typename OT>
__device__ __forceinline__ void conv1_pool_compute(const CnnDev& m, const f2 (&p)[4][2], OT* dst) {
  // two channels at a time on packed float32 FMAs (the same fmaf chain per channel as a scalar loop: bias, then the taps in
  // row-major order); the weight pairs are wave-uniform: scalar operands
#pragma unroll 1
#if defined(DYSF_CNN_EXP) && DYSF_CNN_EXP == 3   // diagnostic build: half of conv1's channels, no conv2 product
  for (int c8 = 0; c8 < kC1 / 16; ++c8) {
#else
  for (int c8 = 0; c8 < kC1 / 8; ++c8) {   // eight channels per round: one 16-byte store of float16 (two of float32)
#endif
    float o[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f2* w = (const f2*)m.w1 + (c8 * 4 + q) * 10;  // [channel pair][9 taps + bias][2]
      f2 sp[4];   // the four positions of the 2 x 2 pooling window
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          f2 sacc = w[9];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            // taps kx = 0, 1, 2 read columns dx, dx + 1, dx + 2
            if (dx == 0) {
              sacc = pk_fma_bcast<0>(p[dy + ky][0], w[ky * 3 + 0], sacc);
              sacc = pk_fma_bcast<1>(p[dy + ky][0], w[ky * 3 + 1], sacc);
              sacc = pk_fma_bcast<0>(p[dy + ky][1], w[ky * 3 + 2], sacc);
            } else {
              sacc = pk_fma_bcast<1>(p[dy + ky][0], w[ky * 3 + 0], sacc);
              sacc = pk_fma_bcast<0>(p[dy + ky][1], w[ky * 3 + 1], sacc);
              sacc = pk_fma_bcast<1>(p[dy + ky][1], w[ky * 3 + 2], sacc);
            }
          }
          sp[2 * dy + dx] = sacc;
        }
      // max-pool and ReLU (they commute) as two v_max3_f32 per channel; written as instructions because fmaxf() on the
      // results of the packed-fma statements makes the compiler quiet every operand first (a v_max x, x per value)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#if defined(DYSF_CNN_OLD_POOL)   // diagnostic build: the fmaxf() form (tools/cnn_ab.sh)
        o[2 * q + h] = fmaxf(fmaxf(fmaxf(sp[0][h], sp[1][h]), fmaxf(sp[2][h], sp[3][h])), 0.f);
#else
        float t, r;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(sp[0][h]), "v"(sp[1][h]), "v"(sp[2][h]));
        asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(t), "v"(sp[3][h]));
        o[2 * q + h] = r;
#endif
      }
    }
    if (sizeof(OT) == 2) {
      h8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (_Float16)o[i];
      *(h8*)((_Float16*)dst + 8 * c8) = v;
    } else {
      *(f4*)((float*)dst + 8 * c8) = f4{o[0], o[1], o[2], o[3]};
      *(f4*)((float*)dst + 8 * c8 + 4) = f4{o[4], o[5], o[6], o[7]};
    }
  }
}

template <typename OT>
__device__ __forceinline__ void conv1_pool_pixel(const CnnDev& m, const float* in, int W, int py, int px, OT* dst) {
  f2 p[4][2];
  conv1_load_patch(m, in, W, py, px, p);
  conv1_pool_compute(m, p, dst);
}

__global__ __launch_bounds__(256) void dysf_cnn_conv1_kernel(CnnDev m, const float* frames, int n) {
  const int clip = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= m.PH * m.PW) return;
  const int py = pix / m.PW, px = pix % m.PW;
  const int W = m.cols ? min(max(m.cols[clip], 0), m.W) : m.W;
  _Float16* dst = m.act1 + (((size_t)clip * (m.PH + 2) + (py + 1)) * m.AW + (px + 1)) * kC1;
  if (px < W / 2) {
    conv1_pool_pixel(m, frames + (size_t)clip * m.H * m.stride, W, py, px, dst);
  } else {  // past this clip's pooled width: conv2's zero padding
#pragma unroll
    for (int q = 0; q < 4; ++q) *(int4*)(dst + 8 * q) = make_int4(0, 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <bool FUSED>
__global__ __launch_bounds__(512, 4) void dysf_cnn_conv2_kernel(CnnDev m, const float* frames, int n) {
  extern __shared__ __align__(16) _Float16 tile[];  // [kRowTile + 2][AW][kPixPitch]
  const int clip = blockIdx.y;
  const int y0 = (blockIdx.x / m.ctiles) * kRowTile;  // first output row of this workgroup
  const int cx0 = (blockIdx.x % m.ctiles) * m.CW;     // first output column
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3, half = wave8 >> 2;   // output-channel tile, and which half of the pixel-tile pairs
  const int rows = min(kRowTile, m.PH - y0);
  const int W = m.cols ? min(max(m.cols[clip], 0), m.W) : m.W;  // this clip's valid columns and pooled width
  const int PW = W / 2;
  const int nx = min(PW - cx0, m.CW);                 // output columns of this workgroup (<= 0: nothing to do)
  // rows y0 .. y0 + rows + 1 of the bordered activation map (row r of the map is pooled row r - 1, column c pooled
  // column c - 1; the border and the columns past PW are zero)
  if (FUSED) {
    // conv1 + pool computed straight into the tile: no activation round trip through HBM (the halo rows are computed
    // by both neighbours: 1.5x the conv1 arithmetic, which overlaps other workgroups' MFMA phase)
    const float* in = frames + (size_t)clip * m.H * m.stride;
    const int npix = (rows + 2) * m.TW;
    for (int i = tid; i < npix; i += 512) {
      const int j = i / m.TW, c = i % m.TW;
      const int py = y0 + j - 1, px = cx0 + c - 1;
      _Float16* dst = tile + (size_t)i * kPixPitch;
#if defined(DYSF_CNN_EXP) && DYSF_CNN_EXP == 1   // diagnostic build: no conv1 arithmetic (tools/cnn_split.sh)
      if (false) {
#else
      if (py >= 0 && py < m.PH && px >= 0 && px < PW) {
#endif
        conv1_pool_pixel(m, in, W, py, px, dst);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) *(int4*)(dst + 8 * q) = make_int4(0, 0, 0, 0);
      }
    }
  } else {
    const _Float16* src = m.act1 + ((size_t)clip * (m.PH + 2) + y0) * m.AW * kC1;
    const int nchunks = (rows + 2) * m.TW * 4;  // 16-byte chunks: 4 per pixel
    for (int i = tid; i < nchunks; i += 512) {
      const int pixel = i >> 2, part = i & 3;
      const int j = pixel / m.TW, c = cx0 + pixel % m.TW;   // row and column of the bordered map
      const int4 v = (c < m.AW) ? *(const int4*)(src + ((size_t)j * m.AW + c) * kC1 + part * 8) : make_int4(0, 0, 0, 0);
      *(int4*)(tile + (size_t)pixel * kPixPitch + part * 8) = v;
    }
  }
  // this wave's A operands: output channels 16*wave .. +15, one K = 32 step (all input channels) per tap
  h8 wa[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wa[t] = *(const h8*)(m.w2p + (((size_t)wave * 9 + t) * 64 + lane) * 8);
  __syncthreads();
  const int ncol = lane & 15, kg = lane >> 4;
  float bias[4], rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = m.b2[16 * wave + 4 * kg + r];
  const int xtiles = (max(nx, 0) + 15) / 16;   // this clip's pixel tiles: the columns past its pooled width cost nothing
  const int pairs = (xtiles + 1) / 2;
#if defined(DYSF_CNN_EXP) && DYSF_CNN_EXP >= 2   // diagnostic build: no conv2 product
  for (int item = half; item < 0; item += 2) {
#else
  for (int item = half; item < rows * pairs; item += 2) {  // the two waves of an output-channel tile alternate
#endif
    const int ry = item / pairs;
    {
      const int xt = 2 * (item % pairs);            // two pixel tiles per pass: two independent MFMA chains
      const int x = 16 * xt + ncol;                 // output pixel column of this lane's B / D column (first tile)
      const bool two = xt + 1 < xtiles;             // wave-uniform
      f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      // bordered coordinates: output (y, x) reads map rows y+ky, columns x+kx; lane kg supplies channels 8kg .. 8kg+7.
      // Branch-free and a kernel row ahead: the six B operands of row ky + 1 are requested before the six MFMAs of row ky
      // are issued (an odd last tile multiplies its own pixels a second time and drops the result).
      const _Float16* bp = tile + ((size_t)ry * m.TW + x) * kPixPitch + 8 * kg;
      const int off1 = two ? 16 * kPixPitch : 0;
      h8 b0[3], b1[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) { b0[kx] = *(const h8*)(bp + kx * kPixPitch); b1[kx] = *(const h8*)(bp + kx * kPixPitch + off1); }
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every read next to its MFMA: one read in flight)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        h8 n0[3], n1[3];
        if (ky < 2) {
          const _Float16* np = bp + (size_t)(ky + 1) * m.TW * kPixPitch;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) { n0[kx] = *(const h8*)(np + kx * kPixPitch); n1[kx] = *(const h8*)(np + kx * kPixPitch + off1); }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[3 * ky + kx], b0[kx], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[3 * ky + kx], b1[kx], acc1, 0, 0, 0);
        }
        if (ky < 2) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) { b0[kx] = n0[kx]; b1[kx] = n1[kx]; }
        }
      }
      if (x < nx) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[r] += fmaxf(acc0[r] + bias[r], 0.f);
      }
      if (two && x + 16 < nx) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[r] += fmaxf(acc1[r] + bias[r], 0.f);
      }
    }
  }
  // sum over the 16 pixel columns (lanes with equal lane >> 4); the workgroup's partial sums go to its own slot
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = rs[r];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (ncol == 0) m.sums[(((size_t)clip * m.n_tiles + blockIdx.x) * 2 + half) * kC2 + 16 * wave + 4 * kg + r] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// float32 form of conv2 (handles created with DYSF_CNN_F32): conv1 + pool straight into a float32 LDS tile, conv2 on
// v_mfma_f32_16x16x4_f32.  K = (tap, input channel) is walked per tap in two groups of 16 channels; lane (pixel = l & 15,
// q = l >> 4) reads channels 16G + 4q .. + 3 of its pixel with one ds_read_b128 and feeds them to four MFMA steps, the
// weight pack being laid out to match ([oc tile][tap][G][lane][step] -> input channel 16G + 4q + step).  The 72 A operands
// of the wave's output-channel tile stay in registers.
constexpr int kPixPitch32 = 36;      // floats per pixel in LDS: 32 channels + 4 of padding (144 bytes, 16-byte aligned)

__global__ __launch_bounds__(512, 2) void dysf_cnn_conv2_f32_kernel(CnnDev m, const float* frames, int n) {
  extern __shared__ __align__(16) float tile32[];  // [kRowTile + 2][AW][kPixPitch32]
  const int clip = blockIdx.y;
  const int y0 = (blockIdx.x / m.ctiles) * kRowTile;
  const int cx0 = (blockIdx.x % m.ctiles) * m.CW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3, half = wave8 >> 2;
  const int rows = min(kRowTile, m.PH - y0);
  const int W = m.cols ? min(max(m.cols[clip], 0), m.W) : m.W;
  const int PW = W / 2;
  const int nx = min(PW - cx0, m.CW);
  {
    const float* in = frames + (size_t)clip * m.H * m.stride;
    const int npix = (rows + 2) * m.TW;
    for (int i = tid; i < npix; i += 512) {
      const int j = i / m.TW, c = i % m.TW;
      const int py = y0 + j - 1, px = cx0 + c - 1;
      float* dst = tile32 + (size_t)i * kPixPitch32;
#if defined(DYSF_CNN_EXP) && DYSF_CNN_EXP == 1   // diagnostic build: no conv1 arithmetic (tools/cnn_split.sh)
      if (false) {
#else
      if (py >= 0 && py < m.PH && px >= 0 && px < PW) {
#endif
        conv1_pool_pixel(m, in, W, py, px, dst);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) *(f4*)(dst + 4 * q) = f4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  f4 wa[18];  // [tap][G]: four MFMA steps each
#pragma unroll
  for (int t = 0; t < 18; ++t) wa[t] = *(const f4*)(m.w2p32 + (((size_t)wave * 18 + t) * 64 + lane) * 4);
  __syncthreads();
  const int ncol = lane & 15, kg = lane >> 4;
  float bias[4], rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = m.b2[16 * wave + 4 * kg + r];
  const int xtiles = (max(nx, 0) + 15) / 16;
  const int pairs = (xtiles + 1) / 2;
  for (int item = half; item < rows * pairs; item += 2) {
    const int ry = item / pairs;
    const int xt = 2 * (item % pairs);
    const int x = 16 * xt + ncol;
    const bool two = xt + 1 < xtiles;  // wave-uniform
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ky = t / 3, kx = t % 3;
      const float* bp = tile32 + ((size_t)(ry + ky) * m.TW + (x + kx)) * kPixPitch32 + 4 * kg;
#pragma unroll
      for (int G = 0; G < 2; ++G) {
        const f4 a4 = wa[2 * t + G];
        const f4 b0 = *(const f4*)(bp + 16 * G);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], b0[0], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], b0[1], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], b0[2], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], b0[3], acc0, 0, 0, 0);
        if (two) {
          const f4 b1 = *(const f4*)(bp + 16 * kPixPitch32 + 16 * G);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], b1[0], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], b1[1], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], b1[2], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], b1[3], acc1, 0, 0, 0);
        }
      }
    }
    if (x < nx) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[r] += fmaxf(acc0[r] + bias[r], 0.f);
    }
    if (two && x + 16 < nx) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[r] += fmaxf(acc1[r] + bias[r], 0.f);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = rs[r];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (ncol == 0) m.sums[(((size_t)clip * m.n_tiles + blockIdx.x) * 2 + half) * kC2 + 16 * wave + 4 * kg + r] = v;
  }
}

// one wave per clip: lane c owns channel c, adds the workgroups' partial sums in tile order (reproducible), then the
// classes are 64-term dot products reduced across the wave
__global__ __launch_bounds__(256) void dysf_cnn_fc_kernel(CnnDev m, int n, float* logits) {
  const int lane = threadIdx.x & 63;
  const int clip = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (clip >= n) return;
  float tot = 0.f;
  for (int t = 0; t < 2 * m.n_tiles; ++t) tot += m.sums[((size_t)clip * 2 * m.n_tiles + t) * kC2 + lane];
  const int W = m.cols ? min(max(m.cols[clip], 0), m.W) : m.W;
  const int npix = m.PH * (W / 2);  // the clip's own pooled size
  const float mean = npix > 0 ? tot * (1.0f / (float)npix) : 0.f;
  for (int j = 0; j < m.n_classes; ++j) {
    float v = m.wfc[j * kC2 + lane] * mean;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) logits[(size_t)clip * m.n_classes + j] = v + m.bfc[j];
  }
}

}  // namespace

struct DysfCnn {
  int device = 0;
  CnnDev d{};
  std::vector<void*> bufs;
  void* act1 = nullptr; size_t act1_clips = 0;
  void* sums = nullptr; size_t sums_clips = 0;
  int act_h = 0, act_w = 0;  // geometry the workspace was laid out for
  int precision = DYSF_CNN_F16;
  bool fused = true;  // DYSF_CNN_SPLIT=1: conv1 as its own kernel with the activations in HBM (float16 handles only)
};

static thread_local std::string g_cerr;
template <typename T>
static bool up(DysfCnn* m, const T** dst, const std::vector<T>& src) {
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

const char* dysf_cnn_last_error(void) { return g_cerr.c_str(); }

int32_t dysf_cnn_create(int32_t device, int32_t n_classes, int32_t precision, const float* conv1_w, const float* conv1_b,
                        const float* conv2_w, const float* conv2_b, const float* fc_w, const float* fc_b,
                        DysfCnn** out) {
  if (precision != DYSF_CNN_F16 && precision != DYSF_CNN_F32) { g_cerr = "precision must be DYSF_CNN_F16 or DYSF_CNN_F32"; return DYSF_E_INVALID; }
  if (!out || !conv1_w || !conv1_b || !conv2_w || !conv2_b || !fc_w || !fc_b) { g_cerr = "null argument"; return DYSF_E_INVALID; }
  *out = nullptr;
  if (n_classes < 1 || n_classes > 64) { g_cerr = "n_classes out of range"; return DYSF_E_INVALID; }
  DysfCnn* m = new (std::nothrow) DysfCnn();
  if (!m) { g_cerr = "out of host memory"; return DYSF_E_NOMEM; }
  m->device = device;
  m->precision = precision;
  if (hipSetDevice(device) != hipSuccess) { delete m; g_cerr = "hipSetDevice failed"; return DYSF_E_HIP; }
  // float32 conv2 A operands: [oc tile][tap][G][lane][step]: oc = 16*tile + (lane & 15), ic = 16*G + 4*(lane >> 4) + step
  std::vector<float> w2p32((size_t)4 * 9 * 2 * 64 * 4);
  for (int tile = 0; tile < 4; ++tile)
    for (int tap = 0; tap < 9; ++tap)
      for (int G = 0; G < 2; ++G)
        for (int l = 0; l < 64; ++l)
          for (int st = 0; st < 4; ++st) {
            const int oc = 16 * tile + (l & 15), ic = 16 * G + 4 * (l >> 4) + st;
            w2p32[((((size_t)tile * 9 + tap) * 2 + G) * 64 + l) * 4 + st] = conv2_w[((size_t)oc * kC1 + ic) * 9 + tap];
          }
  // conv2 A operands: [oc tile][tap][lane][j]: oc = 16*tile + (lane & 15), ic = 8*(lane >> 4) + j
  std::vector<_Float16> w2p((size_t)4 * 9 * 64 * 8);
  for (int tile = 0; tile < 4; ++tile)
    for (int tap = 0; tap < 9; ++tap)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int oc = 16 * tile + (l & 15), ic = 8 * (l >> 4) + j;
          w2p[(((size_t)tile * 9 + tap) * 64 + l) * 8 + j] = (_Float16)conv2_w[((size_t)oc * kC1 + ic) * 9 + tap];
        }
  // conv1: [channel][tap 0..8, bias]
  std::vector<float> w1p((size_t)kC1 * 10);
  for (int c = 0; c < kC1; ++c) {   // [channel pair][9 taps + bias][channel of the pair]
    for (int k = 0; k < 9; ++k) w1p[((size_t)(c >> 1) * 10 + k) * 2 + (c & 1)] = conv1_w[c * 9 + k];
    w1p[((size_t)(c >> 1) * 10 + 9) * 2 + (c & 1)] = conv1_b[c];
  }
  std::vector<float> b2(conv2_b, conv2_b + kC2), wfc(fc_w, fc_w + (size_t)n_classes * kC2), bfc(fc_b, fc_b + n_classes);
  m->d.n_classes = n_classes;
  if (const char* e = std::getenv("DYSF_CNN_SPLIT")) m->fused = std::atoi(e) == 0;
  if (precision == DYSF_CNN_F32) m->fused = true;
  const bool ok = up(m, &m->d.w1, w1p) && up(m, &m->d.w2p, w2p) && up(m, &m->d.w2p32, w2p32) && up(m, &m->d.b2, b2) &&
                  up(m, &m->d.wfc, wfc) && up(m, &m->d.bfc, bfc);
  if (!ok) { for (void* p : m->bufs) (void)hipFree(p); delete m; g_cerr = "device upload failed"; return DYSF_E_HIP; }
  *out = m;
  return 0;
}

void dysf_cnn_destroy(DysfCnn* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  for (void* p : m->bufs) (void)hipFree(p);
  if (m->act1) (void)hipFree(m->act1);
  if (m->sums) (void)hipFree(m->sums);
  delete m;
}

int32_t dysf_cnn_forward_dev(DysfCnn* m, const float* d_frames, int32_t n, int32_t rows, int32_t cols, const int32_t* d_cols,
                             int64_t row_stride, float* d_logits, void* stream_) {
  if (!m || !d_frames || !d_logits || n < 0) { g_cerr = "bad argument"; return DYSF_E_INVALID; }
  if (rows < 2 || cols < 2 || row_stride < cols || rows > 4096 || cols > 65536) { g_cerr = "bad image shape"; return DYSF_E_INVALID; }
  if (n == 0) return 0;
  if (n > 65535) { g_cerr = "at most 65535 clips per call"; return DYSF_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  if (hipSetDevice(m->device) != hipSuccess) { g_cerr = "hipSetDevice failed"; return DYSF_E_HIP; }
  CnnDev& d = m->d;
  d.H = rows; d.W = cols; d.stride = (int)row_stride; d.PH = rows / 2; d.PW = cols / 2; d.cols = d_cols;
  d.AW = (d.PW + 15) / 16 * 16 + 2;  // the last x tile reads up to column 16*xtiles + 1 of the bordered map
  const size_t act_per_clip = (size_t)(d.PH + 2) * d.AW * kC1 * sizeof(_Float16);
  if (!m->fused && ((size_t)n > m->act1_clips || rows != m->act_h || cols != m->act_w)) {
    if (m->act1) { (void)hipFree(m->act1); m->act1 = nullptr; m->act1_clips = 0; }
    if (hipMalloc(&m->act1, act_per_clip * n) != hipSuccess) { g_cerr = "hipMalloc of the activation workspace failed"; return DYSF_E_NOMEM; }
    // the border (and the columns past PW) is zero and is never written afterwards
    if (hipMemsetAsync(m->act1, 0, act_per_clip * n, stream) != hipSuccess) { g_cerr = "hipMemset failed"; return DYSF_E_HIP; }
    m->act1_clips = (size_t)n; m->act_h = rows; m->act_w = cols;
  }
  const bool f32 = m->precision == DYSF_CNN_F32;
  // conv2 workgroup tile: kRowTile rows x CW columns (+ the halo) of activations in LDS; wide maps are cut into column tiles
  // column tiles sized so that the float16 tile leaves room for two workgroups per CU (78 KB each), the float32 tile for one
  const size_t lds_budget = f32 ? (size_t)150 * 1024 : (size_t)78 * 1024;
  const size_t pix_bytes = f32 ? kPixPitch32 * sizeof(float) : kPixPitch * sizeof(_Float16);
  int cw_cap = std::max(16, ((int)(lds_budget / ((kRowTile + 2) * pix_bytes)) - 2) / 16 * 16);
  if (const char* e = std::getenv("DYSF_CNN_CW")) { const int v = std::atoi(e); if (v >= 16) cw_cap = v / 16 * 16; }   // tile-width experiments
  {
    const int ct = std::max(1, (d.PW + cw_cap - 1) / cw_cap);          // column tiles, then equal widths (no ragged last tile)
    d.CW = std::min(cw_cap, ((d.PW + ct - 1) / ct + 15) / 16 * 16);
  }
  if (d.CW < 16) d.CW = 16;
  d.TW = d.CW + 2;
  d.ctiles = std::max(1, (d.PW + d.CW - 1) / d.CW);
  d.n_tiles = ((d.PH + kRowTile - 1) / kRowTile) * d.ctiles;
  if ((size_t)n * d.n_tiles * 2 > m->sums_clips) {
    if (m->sums) { (void)hipFree(m->sums); m->sums = nullptr; }
    if (hipMalloc(&m->sums, (size_t)n * d.n_tiles * 2 * kC2 * sizeof(float)) != hipSuccess) { g_cerr = "hipMalloc failed"; return DYSF_E_NOMEM; }
    m->sums_clips = (size_t)n * d.n_tiles * 2;
  }
  d.act1 = (_Float16*)m->act1; d.sums = (float*)m->sums;
  const size_t lds = f32 ? (size_t)(kRowTile + 2) * d.TW * kPixPitch32 * sizeof(float)
                         : (size_t)(kRowTile + 2) * d.TW * kPixPitch * sizeof(_Float16);
  if (lds > 160 * 1024) { g_cerr = "feature map too wide for the LDS tile"; return DYSF_E_INVALID; }
  size_t lds_launch = lds;
  if (const char* e = std::getenv("DYSF_CNN_LDS")) lds_launch = std::max(lds, (size_t)std::atoll(e));   // co-residency experiments
  {
    const void* fn = f32 ? (const void*)dysf_cnn_conv2_f32_kernel
                         : (m->fused ? (const void*)dysf_cnn_conv2_kernel<true> : (const void*)dysf_cnn_conv2_kernel<false>);
    if (!grant_dynamic_lds(fn, lds_launch)) { g_cerr = "hipFuncSetAttribute failed"; return DYSF_E_HIP; }
  }
  const dim3 grid2(d.n_tiles, n);
  if (f32) {
    hipLaunchKernelGGL(dysf_cnn_conv2_f32_kernel, grid2, dim3(512), lds_launch, stream, d, d_frames, n);
  } else if (m->fused) {
    hipLaunchKernelGGL(dysf_cnn_conv2_kernel<true>, grid2, dim3(512), lds_launch, stream, d, d_frames, n);
  } else {
    hipLaunchKernelGGL(dysf_cnn_conv1_kernel, dim3((d.PH * d.PW + 255) / 256, n), dim3(256), 0, stream, d, d_frames, n);
    hipLaunchKernelGGL(dysf_cnn_conv2_kernel<false>, grid2, dim3(512), lds, stream, d, d_frames, n);
  }
  hipLaunchKernelGGL(dysf_cnn_fc_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, d, n, d_logits);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_cerr = hipGetErrorString(e); return DYSF_E_HIP; }
  return 0;
}

}  // extern "C"
