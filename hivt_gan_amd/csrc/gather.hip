// Kernel (a): per-edge gather of the source feature and the edge vector, rotated into the target
// node's frame, written in CSR order as one 16-byte record per edge.
//
// HBM-bound index/byte work: per edge it reads 12 B of indices, 8 B of edge vector, and the
// source pair / target rotation (node rows are re-used by every edge of the node and stay in
// L2), and writes one coalesced float4.
#include "common.h"

namespace hivt {
namespace {

__global__ __launch_bounds__(256) void gather_rotate_fwd_kernel(
    const float* __restrict__ x_src, int64_t x_stride, const float* __restrict__ attr,
    const float* __restrict__ rot, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
    const int32_t* __restrict__ perm, int32_t rot_mod, int64_t E, const int32_t* __restrict__ count,
    float* __restrict__ out, int32_t* __restrict__ src_csr, int32_t* __restrict__ dst_csr) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (count != nullptr && e >= *count) {  // padding row of a fixed-capacity list: inert record
    *reinterpret_cast<float4*>(out + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src_csr) src_csr[e] = 0;
    if (dst_csr) dst_csr[e] = 0;
    return;
  }
  const int64_t p = perm ? perm[e] : e;
  const int32_t s = src[p], d = dst[p];
  const float2 xs = *reinterpret_cast<const float2*>(x_src + (int64_t)s * x_stride);
  const float2 a = *reinterpret_cast<const float2*>(attr + 2 * p);
  float4 o;
  if (rot != nullptr) {
    const float4 r = *reinterpret_cast<const float4*>(rot + 4 * (int64_t)(d % rot_mod));
    // row vector times 2x2: [x y] @ [[r.x r.y],[r.z r.w]]
    o.x = xs.x * r.x + xs.y * r.z;
    o.y = xs.x * r.y + xs.y * r.w;
    o.z = a.x * r.x + a.y * r.z;
    o.w = a.x * r.y + a.y * r.w;
  } else {
    o = make_float4(xs.x, xs.y, a.x, a.y);
  }
  *reinterpret_cast<float4*>(out + 4 * e) = o;
  if (src_csr) src_csr[e] = s;
  if (dst_csr) dst_csr[e] = d;
}

__global__ __launch_bounds__(256) void gather_rotate_bwd_kernel(
    const float* __restrict__ d_out, const float* __restrict__ rot, const int32_t* __restrict__ src,
    const int32_t* __restrict__ dst, const int32_t* __restrict__ perm, int32_t rot_mod, int64_t E,
    const int32_t* __restrict__ count, float* __restrict__ d_attr, float* __restrict__ d_x_edge) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (count != nullptr && e >= *count) {
    if (d_x_edge) *reinterpret_cast<float4*>(d_x_edge + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int64_t p = perm ? perm[e] : e;
  const float4 g = *reinterpret_cast<const float4*>(d_out + 4 * e);
  float2 gx = make_float2(g.x, g.y), ga = make_float2(g.z, g.w);
  if (rot != nullptr) {
    const float4 r = *reinterpret_cast<const float4*>(rot + 4 * (int64_t)(dst[p] % rot_mod));
    // v = u @ R  =>  du = dv @ R^T
    gx = make_float2(g.x * r.x + g.y * r.y, g.x * r.z + g.y * r.w);
    ga = make_float2(g.z * r.x + g.w * r.y, g.z * r.z + g.w * r.w);
  }
  if (d_attr) *reinterpret_cast<float2*>(d_attr + 2 * p) = ga;
  // the contribution of this edge to the gradient of its source row; the caller adds the rows of one
  // source in a fixed order (hivt_segment_sum_rows over a CSR by source), so no float atomics are needed
  if (d_x_edge) *reinterpret_cast<float4*>(d_x_edge + 4 * e) = make_float4(gx.x, gx.y, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void relpose_fwd_kernel(const float* __restrict__ attr, const float* __restrict__ rot,
                                                          const float* __restrict__ angles,
                                                          const int32_t* __restrict__ src,
                                                          const int32_t* __restrict__ dst,
                                                          const int32_t* __restrict__ perm, int64_t E,
                                                          const int32_t* __restrict__ count,
                                                          float* __restrict__ out, int32_t* __restrict__ src_csr,
                                                          int32_t* __restrict__ dst_csr) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (count != nullptr && e >= *count) {
    *reinterpret_cast<float4*>(out + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src_csr) src_csr[e] = 0;
    if (dst_csr) dst_csr[e] = 0;
    return;
  }
  const int64_t p = perm ? perm[e] : e;
  const int32_t s = src[p], d = dst[p];
  const float2 a = *reinterpret_cast<const float2*>(attr + 2 * p);
  const float4 r = *reinterpret_cast<const float4*>(rot + 4 * (int64_t)d);
  const float dth = angles[s] - angles[d];
  float4 o;
  o.x = a.x * r.x + a.y * r.z;
  o.y = a.x * r.y + a.y * r.w;
  o.z = cosf(dth);
  o.w = sinf(dth);
  *reinterpret_cast<float4*>(out + 4 * e) = o;
  if (src_csr) src_csr[e] = s;
  if (dst_csr) dst_csr[e] = d;
}

}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" int hivt_edge_gather_rotate_fwd(const float* x_src, int64_t x_stride, const float* attr, const float* rot,
                                           const int32_t* src, const int32_t* dst, const int32_t* perm,
                                           int32_t rot_mod, int64_t E, const int32_t* count, float* out,
                                           int32_t* src_csr, int32_t* dst_csr, hivt_stream_t stream) {
  const char* what = "hivt_edge_gather_rotate_fwd";
  HIVT_REQUIRE(E >= 0, HIVT_ERR_ARG, "%s: negative edge count", what);
  if (E == 0) return HIVT_OK;
  HIVT_REQUIRE(x_src && attr && src && dst && out, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(x_stride >= 2 && x_stride % 2 == 0, HIVT_ERR_ARG, "%s: x_stride must be even and >= 2", what);
  HIVT_REQUIRE(rot == nullptr || rot_mod > 0, HIVT_ERR_ARG, "%s: rot_mod must be positive", what);
  hipLaunchKernelGGL(gather_rotate_fwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x_src, x_stride, attr, rot, src, dst, perm, rot_mod > 0 ? rot_mod : 1, E, count, out, src_csr,
                     dst_csr);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_edge_gather_rotate_bwd(const float* d_out, const float* rot, const int32_t* src,
                                           const int32_t* dst, const int32_t* perm, int32_t rot_mod, int64_t E,
                                           const int32_t* count, float* d_attr, float* d_x_edge,
                                           hivt_stream_t stream) {
  const char* what = "hivt_edge_gather_rotate_bwd";
  HIVT_REQUIRE(E >= 0, HIVT_ERR_ARG, "%s: negative edge count", what);
  if (E == 0 || (d_attr == nullptr && d_x_edge == nullptr)) return HIVT_OK;
  HIVT_REQUIRE(d_out && src && dst, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(rot == nullptr || rot_mod > 0, HIVT_ERR_ARG, "%s: rot_mod must be positive", what);
  hipLaunchKernelGGL(gather_rotate_bwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     d_out, rot, src, dst, perm, rot_mod > 0 ? rot_mod : 1, E, count, d_attr, d_x_edge);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_edge_relpose_fwd(const float* attr, const float* rot, const float* angles, const int32_t* src,
                                     const int32_t* dst, const int32_t* perm, int64_t E, const int32_t* count,
                                     float* out, int32_t* src_csr, int32_t* dst_csr, hivt_stream_t stream) {
  const char* what = "hivt_edge_relpose_fwd";
  HIVT_REQUIRE(E >= 0, HIVT_ERR_ARG, "%s: negative edge count", what);
  if (E == 0) return HIVT_OK;
  HIVT_REQUIRE(attr && rot && angles && src && dst && out, HIVT_ERR_ARG, "%s: null pointer", what);
  hipLaunchKernelGGL(relpose_fwd_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, attr,
                     rot, angles, src, dst, perm, E, count, out, src_csr, dst_csr);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}
