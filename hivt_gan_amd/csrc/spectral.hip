// Spectral normalisation of the critics' linear layers: one power iteration and the division by the
// estimated largest singular value for ALL layers of a critic in one launch, and the matching
// gradient in one launch.
//
// The framework's hook (one per layer and per forward call) spends about a dozen tiny launches on
// vectors of at most 256 entries; a critic is called nine times per training step.  Here one block
// owns one layer: W (at most 512 x 512 floats) is read three times out of L2.
#include "common.h"

#include <math.h>

namespace hivt {
namespace {

constexpr int kMaxLayers = HIVT_SPECTRAL_MAX_LAYERS;
constexpr int kMaxDim = 512;  // rows / columns of one weight matrix (vectors and partial sums live in LDS)
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / kWave;

struct SpectralArgs {
  const float* w[kMaxLayers];    // [rows, cols] row-major, un-normalised
  float* u[kMaxLayers];          // [rows] running left vector  (updated in training mode)
  float* v[kMaxLayers];          // [cols] running right vector (updated in training mode)
  float* wn[kMaxLayers];         // out: w / sigma
  float* u_used[kMaxLayers];     // out: the vectors sigma was formed with (constants of the gradient)
  float* v_used[kMaxLayers];
  float* sigma;                  // out: [layers]
  int32_t rows[kMaxLayers];
  int32_t cols[kMaxLayers];
  int32_t power_iteration;
  float eps;
};

struct SpectralBwdArgs {
  const float* d_wn[kMaxLayers];
  const float* w[kMaxLayers];
  const float* u_used[kMaxLayers];
  const float* v_used[kMaxLayers];
  const float* sigma;
  float* d_w[kMaxLayers];
  int32_t rows[kMaxLayers];
  int32_t cols[kMaxLayers];
};

// Sum over the block; every thread gets the result.  `red` holds one float per wave.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();  // protect `red` against the previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) s += red[w];
  return s;
}

// One workgroup of 16 waves per layer.  The matrix is streamed three times (W^T u, W v, W / sigma), a
// wave taking whole rows so that its lanes read consecutive columns; rows are interleaved over the
// waves and four rows are in flight per wave.
__global__ __launch_bounds__(kThreads) void spectral_fwd_kernel(const SpectralArgs a) {
  __shared__ float su[kMaxDim];
  __shared__ float sv[kMaxDim];
  __shared__ float ss[kMaxDim];
  __shared__ float part[kWaves][kMaxDim];
  __shared__ float red[kWaves];
  const int L = blockIdx.x;
  const float* __restrict__ w = a.w[L];
  const int R = a.rows[L], C = a.cols[L];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int i = tid; i < R; i += kThreads) su[i] = a.u[L][i];
  for (int j = tid; j < C; j += kThreads) sv[j] = a.v[L][j];
  __syncthreads();

  if (a.power_iteration) {
    // t = W^T u: every wave sums its rows into a private copy of t, the copies are added afterwards
    for (int j0 = 0; j0 < C; j0 += kWave) {
      const int j = j0 + lane;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
      if (j < C) {
        int i = wave;
        for (; i + 3 * kWaves < R; i += 4 * kWaves) {
          const float a0 = w[(int64_t)i * C + j], a1 = w[(int64_t)(i + kWaves) * C + j];
          const float a2 = w[(int64_t)(i + 2 * kWaves) * C + j], a3 = w[(int64_t)(i + 3 * kWaves) * C + j];
          t0 = fmaf(a0, su[i], t0);
          t1 = fmaf(a1, su[i + kWaves], t1);
          t2 = fmaf(a2, su[i + 2 * kWaves], t2);
          t3 = fmaf(a3, su[i + 3 * kWaves], t3);
        }
        for (; i < R; i += kWaves) t0 = fmaf(w[(int64_t)i * C + j], su[i], t0);
        part[wave][j] = (t0 + t1) + (t2 + t3);
      }
    }
    __syncthreads();
    float sq = 0.0f;
    for (int j = tid; j < C; j += kThreads) {
      float t = 0.0f;
#pragma unroll
      for (int q = 0; q < kWaves; ++q) t += part[q][j];
      sv[j] = t;
      sq = fmaf(t, t, sq);
    }
    const float nt = fmaxf(sqrtf(block_sum(sq, red)), a.eps);
    for (int j = tid; j < C; j += kThreads) sv[j] = sv[j] / nt;
    __syncthreads();
  }

  // s = W v: one wave per row, lanes over columns, two rows in flight
  for (int i = wave; i < R; i += 2 * kWaves) {
    const int i2 = i + kWaves;
    float t0 = 0.0f, t1 = 0.0f;
    for (int j = lane; j < C; j += kWave) {
      t0 = fmaf(w[(int64_t)i * C + j], sv[j], t0);
      if (i2 < R) t1 = fmaf(w[(int64_t)i2 * C + j], sv[j], t1);
    }
    t0 = wave_sum(t0);
    t1 = wave_sum(t1);
    if (lane == 0) {
      ss[i] = t0;
      if (i2 < R) ss[i2] = t1;
    }
  }
  __syncthreads();

  if (a.power_iteration) {
    float sq = 0.0f;
    for (int i = tid; i < R; i += kThreads) sq = fmaf(ss[i], ss[i], sq);
    const float ns = fmaxf(sqrtf(block_sum(sq, red)), a.eps);
    for (int i = tid; i < R; i += kThreads) su[i] = ss[i] / ns;
    __syncthreads();
  }

  float dot = 0.0f;
  for (int i = tid; i < R; i += kThreads) dot = fmaf(su[i], ss[i], dot);
  const float sigma = block_sum(dot, red);

  for (int i = tid; i < R; i += kThreads) {
    if (a.power_iteration) a.u[L][i] = su[i];
    a.u_used[L][i] = su[i];
  }
  for (int j = tid; j < C; j += kThreads) {
    if (a.power_iteration) a.v[L][j] = sv[j];
    a.v_used[L][j] = sv[j];
  }
  if (tid == 0) a.sigma[L] = sigma;
  const int64_t total = (int64_t)R * C;
  float* __restrict__ wn = a.wn[L];
  if ((total & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(wn)) & 15) == 0) {
    for (int64_t e = tid; e < total / 4; e += kThreads) {
      const float4 x = reinterpret_cast<const float4*>(w)[e];
      reinterpret_cast<float4*>(wn)[e] = make_float4(x.x / sigma, x.y / sigma, x.z / sigma, x.w / sigma);
    }
  } else {
    for (int64_t e = tid; e < total; e += kThreads) wn[e] = w[e] / sigma;
  }
}

// wn = w / sigma, sigma = u^T w v with u, v constants:
//   d_w = d_wn / sigma - (<d_wn, w> / sigma^2) u v^T
__global__ __launch_bounds__(kThreads) void spectral_bwd_kernel(const SpectralBwdArgs a) {
  __shared__ float red[kWaves];
  const int L = blockIdx.x;
  const int R = a.rows[L], C = a.cols[L];
  const float* __restrict__ g = a.d_wn[L];
  const float* __restrict__ w = a.w[L];
  const float* __restrict__ u = a.u_used[L];
  const float* __restrict__ v = a.v_used[L];
  float* __restrict__ d = a.d_w[L];
  const int64_t total = (int64_t)R * C;
  const bool vec = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(w) |
                                     reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  float acc = 0.0f;
  if (vec) {
    for (int64_t e = threadIdx.x; e < total / 4; e += kThreads) {
      const float4 x = reinterpret_cast<const float4*>(g)[e], y = reinterpret_cast<const float4*>(w)[e];
      acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
  } else {
    for (int64_t e = threadIdx.x; e < total; e += kThreads) acc = fmaf(g[e], w[e], acc);
  }
  const float sigma = a.sigma[L];
  const float coef = block_sum(acc, red) / (sigma * sigma);
  const float inv = 1.0f / sigma;
  if (vec) {
    const int c4 = C / 4;
    for (int64_t e = threadIdx.x; e < total / 4; e += kThreads) {
      const int i = (int)(e / c4), j4 = (int)(e - (int64_t)i * c4);
      const float4 x = reinterpret_cast<const float4*>(g)[e];
      const float4 vv = reinterpret_cast<const float4*>(v)[j4];
      const float cu = coef * u[i];
      reinterpret_cast<float4*>(d)[e] =
          make_float4(x.x * inv - cu * vv.x, x.y * inv - cu * vv.y, x.z * inv - cu * vv.z, x.w * inv - cu * vv.w);
    }
  } else {
    for (int64_t e = threadIdx.x; e < total; e += kThreads) {
      const int i = (int)(e / C), j = (int)(e - (int64_t)i * C);
      d[e] = g[e] * inv - coef * u[i] * v[j];
    }
  }
}

}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" int hivt_spectral_norm_fwd(int32_t layers, const float* const* w, float* const* u, float* const* v,
                                      const int32_t* rows, const int32_t* cols, int32_t power_iteration, float eps,
                                      float* const* wn, float* const* u_used, float* const* v_used, float* sigma,
                                      hivt_stream_t stream) {
  const char* what = "hivt_spectral_norm_fwd";
  HIVT_REQUIRE(layers >= 0 && layers <= kMaxLayers, HIVT_ERR_ARG, "%s: %d layers (at most %d per call)", what, layers,
               kMaxLayers);
  if (layers == 0) return HIVT_OK;
  HIVT_REQUIRE(w && u && v && rows && cols && wn && u_used && v_used && sigma, HIVT_ERR_ARG, "%s: null pointer", what);
  SpectralArgs a{};
  for (int l = 0; l < layers; ++l) {
    HIVT_REQUIRE(w[l] && u[l] && v[l] && wn[l] && u_used[l] && v_used[l], HIVT_ERR_ARG, "%s: null pointer (layer %d)",
                 what, l);
    HIVT_REQUIRE(rows[l] >= 1 && rows[l] <= kMaxDim && cols[l] >= 1 && cols[l] <= kMaxDim, HIVT_ERR_ARG,
                 "%s: layer %d is %d x %d (each side must be in [1,%d])", what, l, rows[l], cols[l], kMaxDim);
    a.w[l] = w[l]; a.u[l] = u[l]; a.v[l] = v[l]; a.wn[l] = wn[l]; a.u_used[l] = u_used[l]; a.v_used[l] = v_used[l];
    a.rows[l] = rows[l]; a.cols[l] = cols[l];
  }
  a.sigma = sigma; a.power_iteration = power_iteration; a.eps = eps;
  hipLaunchKernelGGL(spectral_fwd_kernel, dim3((unsigned)layers), dim3(kThreads), 0, (hipStream_t)stream, a);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_spectral_norm_bwd(int32_t layers, const float* const* d_wn, const float* const* w,
                                      const float* const* u_used, const float* const* v_used, const float* sigma,
                                      const int32_t* rows, const int32_t* cols, float* const* d_w,
                                      hivt_stream_t stream) {
  const char* what = "hivt_spectral_norm_bwd";
  HIVT_REQUIRE(layers >= 0 && layers <= kMaxLayers, HIVT_ERR_ARG, "%s: %d layers (at most %d per call)", what, layers,
               kMaxLayers);
  if (layers == 0) return HIVT_OK;
  HIVT_REQUIRE(d_wn && w && u_used && v_used && sigma && rows && cols && d_w, HIVT_ERR_ARG, "%s: null pointer", what);
  SpectralBwdArgs a{};
  for (int l = 0; l < layers; ++l) {
    HIVT_REQUIRE(d_wn[l] && w[l] && u_used[l] && v_used[l] && d_w[l], HIVT_ERR_ARG, "%s: null pointer (layer %d)", what,
                 l);
    HIVT_REQUIRE(rows[l] >= 1 && cols[l] >= 1, HIVT_ERR_ARG, "%s: empty layer %d", what, l);
    a.d_wn[l] = d_wn[l]; a.w[l] = w[l]; a.u_used[l] = u_used[l]; a.v_used[l] = v_used[l]; a.d_w[l] = d_w[l];
    a.rows[l] = rows[l]; a.cols[l] = cols[l];
  }
  a.sigma = sigma;
  hipLaunchKernelGGL(spectral_bwd_kernel, dim3((unsigned)layers), dim3(kThreads), 0, (hipStream_t)stream, a);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}
