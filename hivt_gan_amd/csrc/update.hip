// Elementwise tail of the gated update that follows the aggregation:
//   out = agg + sigmoid(ih + hh) * (self - agg)
// and its gradients.  Pure HBM streaming over [N, D] operands, one float4 per lane.
#include "common.h"

#include <math.h>

namespace hivt {
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gated_update_fwd_kernel(const float* __restrict__ agg, int64_t agg_s,
                                                               const float* __restrict__ ih, int64_t ih_s,
                                                               const float* __restrict__ hh, int64_t hh_s,
                                                               const float* __restrict__ sf, int64_t sf_s, int64_t N,
                                                               int32_t D4, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D4) return;
  const int64_t r = i / D4;
  const int c = (int)(i - r * D4) * 4;
  const float4 m = *reinterpret_cast<const float4*>(agg + r * agg_s + c);
  const float4 a = *reinterpret_cast<const float4*>(ih + r * ih_s + c);
  const float4 b = *reinterpret_cast<const float4*>(hh + r * hh_s + c);
  const float4 s = *reinterpret_cast<const float4*>(sf + r * sf_s + c);
  float4 o;
  o.x = m.x + sigmoidf_(a.x + b.x) * (s.x - m.x);
  o.y = m.y + sigmoidf_(a.y + b.y) * (s.y - m.y);
  o.z = m.z + sigmoidf_(a.z + b.z) * (s.z - m.z);
  o.w = m.w + sigmoidf_(a.w + b.w) * (s.w - m.w);
  *reinterpret_cast<float4*>(out + r * (int64_t)D4 * 4 + c) = o;
}

__device__ __forceinline__ void bwd1(float go, float m, float a, float b, float s, float& dm, float& dg, float& dsf) {
  const float g = sigmoidf_(a + b);
  dm = go * (1.0f - g);
  dsf = go * g;
  dg = go * (s - m) * g * (1.0f - g);
}

__global__ __launch_bounds__(256) void gated_update_bwd_kernel(const float* __restrict__ d_out,
                                                               const float* __restrict__ agg, int64_t agg_s,
                                                               const float* __restrict__ ih, int64_t ih_s,
                                                               const float* __restrict__ hh, int64_t hh_s,
                                                               const float* __restrict__ sf, int64_t sf_s, int64_t N,
                                                               int32_t D4, float* __restrict__ d_agg,
                                                               float* __restrict__ d_ih, float* __restrict__ d_hh,
                                                               int64_t d_hh_s, float* __restrict__ d_self,
                                                               int64_t d_self_s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D4) return;
  const int64_t r = i / D4;
  const int c = (int)(i - r * D4) * 4;
  const int64_t o = r * (int64_t)D4 * 4 + c;
  const float4 go = *reinterpret_cast<const float4*>(d_out + o);
  const float4 m = *reinterpret_cast<const float4*>(agg + r * agg_s + c);
  const float4 a = *reinterpret_cast<const float4*>(ih + r * ih_s + c);
  const float4 b = *reinterpret_cast<const float4*>(hh + r * hh_s + c);
  const float4 s = *reinterpret_cast<const float4*>(sf + r * sf_s + c);
  float4 dm, dg, ds;
  bwd1(go.x, m.x, a.x, b.x, s.x, dm.x, dg.x, ds.x);
  bwd1(go.y, m.y, a.y, b.y, s.y, dm.y, dg.y, ds.y);
  bwd1(go.z, m.z, a.z, b.z, s.z, dm.z, dg.z, ds.z);
  bwd1(go.w, m.w, a.w, b.w, s.w, dm.w, dg.w, ds.w);
  *reinterpret_cast<float4*>(d_agg + o) = dm;
  *reinterpret_cast<float4*>(d_ih + o) = dg;
  if (d_hh) *reinterpret_cast<float4*>(d_hh + r * d_hh_s + c) = dg;
  *reinterpret_cast<float4*>(d_self + r * d_self_s + c) = ds;
}

bool strides_ok(int32_t D, int64_t s0, int64_t s1, int64_t s2, int64_t s3) {
  auto ok = [D](int64_t s) { return s >= D && s % 4 == 0; };
  return ok(s0) && ok(s1) && ok(s2) && ok(s3);
}

}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" int hivt_gated_update_fwd(const float* agg, int64_t agg_stride, const float* ih, int64_t ih_stride,
                                     const float* hh, int64_t hh_stride, const float* self_, int64_t self_stride,
                                     int64_t N, int32_t D, float* out, hivt_stream_t stream) {
  const char* what = "hivt_gated_update_fwd";
  HIVT_REQUIRE(N >= 0 && D > 0 && D % 4 == 0, HIVT_ERR_ARG, "%s: D must be a positive multiple of 4", what);
  if (N == 0) return HIVT_OK;
  HIVT_REQUIRE(agg && ih && hh && self_ && out, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(strides_ok(D, agg_stride, ih_stride, hh_stride, self_stride), HIVT_ERR_ARG,
               "%s: row strides must be >= D and multiples of 4", what);
  const int64_t total = N * (D / 4);
  hipLaunchKernelGGL(gated_update_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     agg, agg_stride, ih, ih_stride, hh, hh_stride, self_, self_stride, N, D / 4, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_gated_update_bwd(const float* d_out, const float* agg, int64_t agg_stride, const float* ih,
                                     int64_t ih_stride, const float* hh, int64_t hh_stride, const float* self_,
                                     int64_t self_stride, int64_t N, int32_t D, float* d_agg, float* d_ih,
                                     float* d_hh, int64_t d_hh_stride, float* d_self, int64_t d_self_stride,
                                     hivt_stream_t stream) {
  const char* what = "hivt_gated_update_bwd";
  HIVT_REQUIRE(N >= 0 && D > 0 && D % 4 == 0, HIVT_ERR_ARG, "%s: D must be a positive multiple of 4", what);
  if (N == 0) return HIVT_OK;
  HIVT_REQUIRE(d_out && agg && ih && hh && self_ && d_agg && d_ih && d_self, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(strides_ok(D, agg_stride, ih_stride, hh_stride, self_stride) && d_self_stride >= D &&
                   d_self_stride % 4 == 0 && (d_hh == nullptr || (d_hh_stride >= D && d_hh_stride % 4 == 0)),
               HIVT_ERR_ARG, "%s: row strides must be >= D and multiples of 4", what);
  const int64_t total = N * (D / 4);
  hipLaunchKernelGGL(gated_update_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     d_out, agg, agg_stride, ih, ih_stride, hh, hh_stride, self_, self_stride, N, D / 4, d_agg,
                     d_ih, d_hh, d_hh_stride, d_self, d_self_stride);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}
