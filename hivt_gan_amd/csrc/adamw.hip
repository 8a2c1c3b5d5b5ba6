// AdamW over a list of parameter tensors in few launches.
//
// The update of the reference's two optimizers (torch.optim.AdamW, models/hivt.py:444,453) touches a few
// hundred tensors, most of them a few hundred bytes.  Every launch here takes up to kAdamTensors tensors
// by value in its argument block and cuts them into 4096-element pieces, one workgroup per piece, so the
// 2.9 M parameters of HiVT-128 are five or six launches of a few microseconds each.
#include "common.h"

#include <math.h>

namespace hivt {
namespace {

constexpr int kAdamTensors = 96;    // tensors per launch: 96 * 40 bytes of descriptors stay under the 4 KB argument limit
constexpr int kAdamPiece = 4096;    // elements per workgroup

struct AdamBatch {
  float* p[kAdamTensors];
  const float* g[kAdamTensors];
  float* m[kAdamTensors];
  float* v[kAdamTensors];
  int32_t n[kAdamTensors];           // elements of tensor i
  int32_t first_piece[kAdamTensors + 1];  // prefix sum of the tensors' piece counts
  int32_t count;
  const float* step;       // device scalar: number of this update minus step_offset
  float step_offset;       // 0: the caller has advanced the counter; 1: the call advances it after the update
  const float* found_inf;  // optional device scalar: 1 -> leave everything as it is
  const float* lr_ptr;     // optional device scalar that replaces lr
  float lr, beta1, beta2, eps, weight_decay;
};

__global__ __launch_bounds__(256) void adamw_kernel(const AdamBatch a) {
  if (a.found_inf != nullptr && *a.found_inf == 1.0f) return;
  // which tensor this workgroup's piece belongs to (uniform over the workgroup)
  const int piece = blockIdx.x;
  int t = 0;
  while (t + 1 < a.count && a.first_piece[t + 1] <= piece) ++t;
  const int begin = (piece - a.first_piece[t]) * kAdamPiece;
  const int n = a.n[t];
  const int end = min(n, begin + kAdamPiece);
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const float step = *a.step + a.step_offset;
  const float lr = a.lr_ptr != nullptr ? *a.lr_ptr : a.lr;
  const float bc1 = 1.0f - powf(a.beta1, step);
  const float bc2_sqrt = sqrtf(1.0f - powf(a.beta2, step));
  const float step_size = lr / bc1;
  const float decay = 1.0f - lr * a.weight_decay;
  auto update = [&](float& pi, float grad, float& mi, float& vi) {
    const float param = pi * decay;
    mi = a.beta1 * mi + (1.0f - a.beta1) * grad;
    vi = a.beta2 * vi + (1.0f - a.beta2) * grad * grad;
    const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
    pi = param - step_size * (mi / denom);
  };
  const uintptr_t bits = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                         reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v);
  if ((bits & 15) == 0 && (n & 3) == 0) {  // whole tensor in 16-byte quads (pieces start at multiples of 4096)
    for (int i = begin + 4 * threadIdx.x; i < end; i += 4 * 256) {
      float4 p4 = *reinterpret_cast<float4*>(p + i), m4 = *reinterpret_cast<float4*>(m + i);
      float4 v4 = *reinterpret_cast<float4*>(v + i);
      const float4 g4 = *reinterpret_cast<const float4*>(g + i);
      update(p4.x, g4.x, m4.x, v4.x);
      update(p4.y, g4.y, m4.y, v4.y);
      update(p4.z, g4.z, m4.z, v4.z);
      update(p4.w, g4.w, m4.w, v4.w);
      *reinterpret_cast<float4*>(m + i) = m4;
      *reinterpret_cast<float4*>(v + i) = v4;
      *reinterpret_cast<float4*>(p + i) = p4;
    }
    return;
  }
  for (int i = begin + threadIdx.x; i < end; i += 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    update(pi, g[i], mi, vi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

// step += 1 unless the update was skipped
__global__ void advance_step_kernel(float* step, const float* found_inf) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(found_inf != nullptr && *found_inf == 1.0f)) *step += 1.0f;
}

}  // namespace
}  // namespace hivt

using namespace hivt;

static int adamw_impl(const char* what, const uint64_t* params, const uint64_t* grads, const uint64_t* exp_avgs,
                      const uint64_t* exp_avg_sqs, const int64_t* numels, int32_t count, const float* step,
                      float step_offset, const float* found_inf, const float* lr_ptr, float lr, float beta1, float beta2,
                      float eps, float weight_decay, hivt_stream_t stream);

extern "C" int hivt_adamw_step(const uint64_t* params, const uint64_t* grads, const uint64_t* exp_avgs,
                               const uint64_t* exp_avg_sqs, const int64_t* numels, int32_t count, const float* step,
                               const float* found_inf, float lr, float beta1, float beta2, float eps,
                               float weight_decay, hivt_stream_t stream) {
  return adamw_impl("hivt_adamw_step", params, grads, exp_avgs, exp_avg_sqs, numels, count, step, 0.0f, found_inf, nullptr,
                    lr, beta1, beta2, eps, weight_decay, stream);
}

extern "C" int hivt_adamw_update(const uint64_t* params, const uint64_t* grads, const uint64_t* exp_avgs,
                                 const uint64_t* exp_avg_sqs, const int64_t* numels, int32_t count, float* step,
                                 const float* found_inf, const float* lr_ptr, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, hivt_stream_t stream) {
  const char* what = "hivt_adamw_update";
  const int rc = adamw_impl(what, params, grads, exp_avgs, exp_avg_sqs, numels, count, step, 1.0f, found_inf, lr_ptr, lr,
                            beta1, beta2, eps, weight_decay, stream);
  if (rc != HIVT_OK || count == 0) return rc;
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, found_inf);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

static int adamw_impl(const char* what, const uint64_t* params, const uint64_t* grads, const uint64_t* exp_avgs,
                      const uint64_t* exp_avg_sqs, const int64_t* numels, int32_t count, const float* step,
                      float step_offset, const float* found_inf, const float* lr_ptr, float lr, float beta1, float beta2,
                      float eps, float weight_decay, hivt_stream_t stream) {
  HIVT_REQUIRE(count >= 0, HIVT_ERR_ARG, "%s: negative tensor count", what);
  if (count == 0) return HIVT_OK;
  HIVT_REQUIRE(params && grads && exp_avgs && exp_avg_sqs && numels && step, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f, HIVT_ERR_ARG,
               "%s: betas must be in [0,1) and eps non-negative", what);
  for (int32_t i = 0; i < count; ++i) {
    HIVT_REQUIRE(numels[i] >= 0 && numels[i] < ((int64_t)1 << 31), HIVT_ERR_ARG, "%s: tensor %d has %lld elements", what,
                 (int)i, (long long)numels[i]);
    HIVT_REQUIRE(numels[i] == 0 || (params[i] && grads[i] && exp_avgs[i] && exp_avg_sqs[i]), HIVT_ERR_ARG,
                 "%s: tensor %d has a null pointer", what, (int)i);
  }
  hipStream_t s = (hipStream_t)stream;
  AdamBatch b;
  b.step = step; b.step_offset = step_offset; b.found_inf = found_inf; b.lr_ptr = lr_ptr;
  b.lr = lr; b.beta1 = beta1; b.beta2 = beta2; b.eps = eps; b.weight_decay = weight_decay;
  int32_t i = 0;
  while (i < count) {
    int k = 0, pieces = 0;
    for (; i < count && k < kAdamTensors; ++i) {
      if (numels[i] == 0) continue;
      b.p[k] = reinterpret_cast<float*>(params[i]);
      b.g[k] = reinterpret_cast<const float*>(grads[i]);
      b.m[k] = reinterpret_cast<float*>(exp_avgs[i]);
      b.v[k] = reinterpret_cast<float*>(exp_avg_sqs[i]);
      b.n[k] = (int32_t)numels[i];
      b.first_piece[k] = pieces;
      pieces += (int)((numels[i] + kAdamPiece - 1) / kAdamPiece);
      ++k;
    }
    if (k == 0) break;
    b.first_piece[k] = pieces;
    for (int j = k; j < kAdamTensors; ++j) {  // unused slots: defined values, never selected
      b.p[j] = nullptr; b.g[j] = nullptr; b.m[j] = nullptr; b.v[j] = nullptr; b.n[j] = 0;
      b.first_piece[j + 1] = pieces;
    }
    b.count = k;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)pieces), dim3(256), 0, s, b);
    HIVT_CHECK_LAUNCH(what);
  }
  return HIVT_OK;
}
