// Fused supervised loss: per-mode displacement error, argmin over modes, Laplace NLL of the best
// mode and soft-target cross entropy of the mode scores, in one pass over y_hat.
// One wave per agent: lane h owns future step h (Hf <= 64), modes are looped.  A second
// single-block launch reduces the per-agent partials in a fixed order (deterministic) and
// applies the two 'mean' reductions; nothing is copied to the host.
#include "common.h"

#include <math.h>

namespace hivt {
namespace {

constexpr int kMaxModes = 16;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kEps = 1e-6f;

__global__ __launch_bounds__(256) void laplace_node_kernel(const float* __restrict__ y_hat,
                                                           const float* __restrict__ pi, const float* __restrict__ y,
                                                           const uint8_t* __restrict__ pm, int64_t pm_stride,
                                                           int32_t t_hist, int32_t F, int32_t N, int32_t Hf,
                                                           int64_t* __restrict__ best_out, float* __restrict__ soft_out,
                                                           float* __restrict__ part) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int h = threadIdx.x & 63;
  const bool in = h < Hf;
  const bool valid = in && pm[(int64_t)n * pm_stride + t_hist + h] == 0;
  float2 yy = make_float2(0.f, 0.f);
  if (in) yy = *reinterpret_cast<const float2*>(y + ((int64_t)n * Hf + h) * 2);

  float l2[kMaxModes];
#pragma unroll
  for (int f = 0; f < kMaxModes; ++f) {
    float d = 0.0f;
    if (f < F && valid) {
      const float4 p = *reinterpret_cast<const float4*>(y_hat + (((int64_t)f * N + n) * Hf + h) * 4);
      const float dx = p.x - yy.x, dy = p.y - yy.y;
      d = __fsqrt_rn(__fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
    }
    l2[f] = f < F ? wave_sum(d) : 0.0f;
  }
  int best = 0;
  float bestv = l2[0];
#pragma unroll
  for (int f = 1; f < kMaxModes; ++f) {
    if (f < F && l2[f] < bestv) {
      bestv = l2[f];
      best = f;
    }
  }
  const float cnt = wave_sum(valid ? 1.0f : 0.0f);

  float nll = 0.0f;
  if (valid) {
    const float4 p = *reinterpret_cast<const float4*>(y_hat + (((int64_t)best * N + n) * Hf + h) * 4);
    const float sx = fminf(fmaxf(p.z, -10.0f), 10.0f), sy = fminf(fmaxf(p.w, -10.0f), 10.0f);
    nll = (sx + kLn2 + fabsf(yy.x - p.x) / (expf(sx) + kEps)) + (sy + kLn2 + fabsf(yy.y - p.y) / (expf(sy) + kEps));
  }
  nll = wave_sum(nll);

  // soft target over modes and cross entropy with log_softmax(pi[n]) (every lane redundantly)
  float ce = 0.0f;
  if (cnt > 0.0f) {
    const float inv = 1.0f / (cnt + kEps);
    float amax = -INFINITY, pmax = -INFINITY;
    float pv[kMaxModes];
#pragma unroll
    for (int f = 0; f < kMaxModes; ++f) {
      pv[f] = f < F ? pi[(int64_t)n * F + f] : -INFINITY;
      if (f < F) {
        amax = fmaxf(amax, -l2[f] * inv);
        pmax = fmaxf(pmax, pv[f]);
      }
    }
    float asum = 0.0f, psum = 0.0f;
#pragma unroll
    for (int f = 0; f < kMaxModes; ++f) {
      if (f < F) {
        asum += expf(-l2[f] * inv - amax);
        psum += expf(pv[f] - pmax);
      }
    }
    const float lse = pmax + logf(psum);
#pragma unroll
    for (int f = 0; f < kMaxModes; ++f) {
      if (f < F) {
        const float st = expf(-l2[f] * inv - amax) / asum;
        ce -= st * (pv[f] - lse);
        if (h == 0) soft_out[(int64_t)n * F + f] = st;
      }
    }
  } else if (h < F) {
    soft_out[(int64_t)n * F + h] = 0.0f;
  }
  if (h == 0) {
    best_out[n] = best;
    part[n] = nll;
    part[(int64_t)N + n] = cnt;
    part[2 * (int64_t)N + n] = ce;
  }
}

__global__ __launch_bounds__(1024) void laplace_finalize_kernel(const float* __restrict__ part, int32_t N,
                                                                float* __restrict__ losses,
                                                                float* __restrict__ stats) {
  __shared__ float red[4][16];
  float nll = 0.f, cnt = 0.f, ce = 0.f, rows = 0.f;
  for (int i = threadIdx.x; i < N; i += 1024) {
    nll += part[i];
    const float c = part[(int64_t)N + i];
    cnt += c;
    if (c > 0.0f) {
      ce += part[2 * (int64_t)N + i];
      rows += 1.0f;
    }
  }
  nll = wave_sum(nll);
  cnt = wave_sum(cnt);
  ce = wave_sum(ce);
  rows = wave_sum(rows);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = nll;
    red[1][w] = cnt;
    red[2][w] = ce;
    red[3][w] = rows;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 16; ++j)
      for (int q = 0; q < 4; ++q) t[q] += red[q][j];
    losses[0] = t[1] > 0.0f ? t[0] / (2.0f * t[1]) : 0.0f;
    losses[1] = t[3] > 0.0f ? t[2] / t[3] : 0.0f;
    stats[0] = t[1];
    stats[1] = t[3];
    stats[2] = 0.0f;
    stats[3] = 0.0f;
  }
}

__global__ __launch_bounds__(256) void laplace_bwd_kernel(const float* __restrict__ grad_losses,
                                                          const float* __restrict__ y_hat,
                                                          const float* __restrict__ pi, const float* __restrict__ y,
                                                          const uint8_t* __restrict__ pm, int64_t pm_stride,
                                                          int32_t t_hist, int32_t F, int32_t N, int32_t Hf,
                                                          const int64_t* __restrict__ best,
                                                          const float* __restrict__ soft,
                                                          const float* __restrict__ stats, float* __restrict__ d_y_hat,
                                                          float* __restrict__ d_pi) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int h = threadIdx.x & 63;
  const float m_cnt = stats[0], rows = stats[1];
  const float g_reg = m_cnt > 0.0f ? grad_losses[0] / (2.0f * m_cnt) : 0.0f;
  const float g_cls = rows > 0.0f ? grad_losses[1] / rows : 0.0f;
  const int b = (int)best[n];
  if (h < Hf) {
    const bool valid = pm[(int64_t)n * pm_stride + t_hist + h] == 0;
    for (int f = 0; f < F; ++f) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      const int64_t off = (((int64_t)f * N + n) * Hf + h) * 4;
      if (f == b && valid) {
        const float4 p = *reinterpret_cast<const float4*>(y_hat + off);
        const float2 yy = *reinterpret_cast<const float2*>(y + ((int64_t)n * Hf + h) * 2);
        const float rx = yy.x - p.x, ry = yy.y - p.y;
        const float sx = fminf(fmaxf(p.z, -10.0f), 10.0f), sy = fminf(fmaxf(p.w, -10.0f), 10.0f);
        const float ex = expf(sx), ey = expf(sy);
        const float ix = 1.0f / (ex + kEps), iy = 1.0f / (ey + kEps);
        const float sgx = rx > 0.f ? 1.f : (rx < 0.f ? -1.f : 0.f);
        const float sgy = ry > 0.f ? 1.f : (ry < 0.f ? -1.f : 0.f);
        g.x = -sgx * ix * g_reg;
        g.y = -sgy * iy * g_reg;
        g.z = (p.z >= -10.0f && p.z <= 10.0f) ? (1.0f - fabsf(rx) * ex * ix * ix) * g_reg : 0.0f;
        g.w = (p.w >= -10.0f && p.w <= 10.0f) ? (1.0f - fabsf(ry) * ey * iy * iy) * g_reg : 0.0f;
      }
      *reinterpret_cast<float4*>(d_y_hat + off) = g;
    }
  }
  if (h < F) {
    // d/dpi of -sum_f soft_f * log_softmax(pi)_f = softmax(pi) * sum(soft) - soft
    float pmax = -INFINITY, ssum = 0.0f;
    for (int f = 0; f < F; ++f) {
      pmax = fmaxf(pmax, pi[(int64_t)n * F + f]);
      ssum += soft[(int64_t)n * F + f];
    }
    float psum = 0.0f;
    for (int f = 0; f < F; ++f) psum += expf(pi[(int64_t)n * F + f] - pmax);
    const float sm = expf(pi[(int64_t)n * F + h] - pmax) / psum;
    d_pi[(int64_t)n * F + h] = (sm * ssum - soft[(int64_t)n * F + h]) * g_cls;
  }
}

int check(const char* what, int32_t F, int32_t N, int32_t Hf) {
  HIVT_REQUIRE(F >= 1 && F <= kMaxModes, HIVT_ERR_ARG, "%s: num_modes %d not in [1,%d]", what, F, kMaxModes);
  HIVT_REQUIRE(Hf >= 1 && Hf <= kWave, HIVT_ERR_ARG, "%s: future_steps %d not in [1,64]", what, Hf);
  HIVT_REQUIRE(N >= 0, HIVT_ERR_ARG, "%s: negative N", what);
  return HIVT_OK;
}

}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" size_t hivt_laplace_workspace_bytes(int32_t N) {
  return align_up((size_t)3 * (size_t)(N > 0 ? N : 1) * sizeof(float), 256);
}

extern "C" int hivt_laplace_minmode_fwd(const float* y_hat, const float* pi, const float* y,
                                        const uint8_t* padding_mask, int64_t pm_stride, int32_t t_hist, int32_t F,
                                        int32_t N, int32_t Hf, float* losses_out, int64_t* best_out, float* soft_out,
                                        float* stats_out, void* workspace, size_t workspace_bytes,
                                        hivt_stream_t stream) {
  const char* what = "hivt_laplace_minmode_fwd";
  int rc = check(what, F, N, Hf);
  if (rc != HIVT_OK) return rc;
  HIVT_REQUIRE(losses_out && stats_out, HIVT_ERR_ARG, "%s: null output", what);
  HIVT_REQUIRE(N == 0 || (y_hat && pi && y && padding_mask && best_out && soft_out), HIVT_ERR_ARG, "%s: null pointer",
               what);
  HIVT_REQUIRE(workspace != nullptr && workspace_bytes >= hivt_laplace_workspace_bytes(N), HIVT_ERR_WORKSPACE,
               "%s: workspace too small", what);
  float* part = reinterpret_cast<float*>(workspace);
  if (N > 0) {
    hipLaunchKernelGGL(laplace_node_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y_hat, pi,
                       y, padding_mask, pm_stride, t_hist, F, N, Hf, best_out, soft_out, part);
    HIVT_CHECK_LAUNCH(what);
  }
  hipLaunchKernelGGL(laplace_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, N, losses_out,
                     stats_out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_laplace_minmode_bwd(const float* grad_losses, const float* y_hat, const float* pi, const float* y,
                                        const uint8_t* padding_mask, int64_t pm_stride, int32_t t_hist, int32_t F,
                                        int32_t N, int32_t Hf, const int64_t* best, const float* soft,
                                        const float* stats, float* d_y_hat, float* d_pi, hivt_stream_t stream) {
  const char* what = "hivt_laplace_minmode_bwd";
  int rc = check(what, F, N, Hf);
  if (rc != HIVT_OK) return rc;
  if (N == 0) return HIVT_OK;
  HIVT_REQUIRE(grad_losses && y_hat && pi && y && padding_mask && best && soft && stats && d_y_hat && d_pi,
               HIVT_ERR_ARG, "%s: null pointer", what);
  hipLaunchKernelGGL(laplace_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, grad_losses,
                     y_hat, pi, y, padding_mask, pm_stride, t_hist, F, N, Hf, best, soft, stats, d_y_hat, d_pi);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}


// ---- guards of the training step ------------------------------------------------------------------
namespace hivt {
namespace {

__global__ void guard_init_kernel(const uint8_t* __restrict__ extra, int32_t* __restrict__ flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) flag[0] = (extra != nullptr && extra[0] != 0) ? 1 : 0;
}

// Both scans of a step's guards in one launch: the first `nan_blocks` workgroups look for NaN in x_nan, the others
// for NaN or infinity in x_fin.  A wave that finds one ORs 1 into the flag (an idempotent integer atomic: the
// result does not depend on arrival order).
__global__ __launch_bounds__(256) void guard_scan_kernel(const float* __restrict__ x_nan, int64_t n_nan, int nan_blocks,
                                                         const float* __restrict__ x_fin, int64_t n_fin,
                                                         int32_t* __restrict__ flag) {
  const bool finite = (int)blockIdx.x >= nan_blocks;
  const float* __restrict__ x = finite ? x_fin : x_nan;
  const int64_t n = finite ? n_fin : n_nan;
  const int64_t block = finite ? (int)blockIdx.x - nan_blocks : (int)blockIdx.x;
  const int64_t stride = (int64_t)(finite ? (int)gridDim.x - nan_blocks : nan_blocks) * blockDim.x;
  bool bad = false;
  for (int64_t i = block * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    bad |= finite ? !(fabsf(v) <= 3.402823466e38f) : (v != v);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace
}  // namespace hivt

extern "C" int hivt_step_guards(const float* nan_check, int64_t n_nan, const float* finite_check, int64_t n_finite,
                                const uint8_t* extra, int32_t* flag, hivt_stream_t stream_) {
  const char* what = "hivt_step_guards";
  hipStream_t stream = (hipStream_t)stream_;
  HIVT_REQUIRE(flag != nullptr && n_nan >= 0 && n_finite >= 0, HIVT_ERR_ARG, "%s: bad arguments", what);
  HIVT_REQUIRE((n_nan == 0 || nan_check) && (n_finite == 0 || finite_check), HIVT_ERR_ARG, "%s: null pointer", what);
  hipLaunchKernelGGL(guard_init_kernel, dim3(1), dim3(64), 0, stream, extra, flag);
  HIVT_CHECK_LAUNCH(what);
  auto blocks = [](int64_t n) { int64_t b = (n + 1023) / 1024; return (int)(b > 1024 ? 1024 : b); };  // 0 for n = 0
  const int nan_blocks = blocks(n_nan), fin_blocks = blocks(n_finite);
  if (nan_blocks + fin_blocks > 0) {
    hipLaunchKernelGGL(guard_scan_kernel, dim3((unsigned)(nan_blocks + fin_blocks)), dim3(256), 0, stream, nan_check, n_nan,
                       nan_blocks, finite_check, n_finite, flag);
    HIVT_CHECK_LAUNCH(what);
  }
  return HIVT_OK;
}
