// First stage of the embedding MLPs: h_b = ReLU(LayerNorm_b(W_b f_b + bias_b)) for one or two
// two-component inputs f_b (rotated displacement / relative position pairs), written side by side as
// one [M, B*D] row.  Replaces Linear(2, D) + LayerNorm + ReLU of every branch and the concatenation of
// the branch outputs (models/embedding.py:12-19,36-44,49-52).
//
// With K = 2 there is no product worth a matrix core: the kernel is a pure stream, 16 input bytes and
// B*D*4 output bytes per row.  A (row, branch) pair is covered by D/4 lanes with one float4 of columns
// each; mean and variance are xor-shuffle reductions inside that lane group.  Backward recomputes the
// pre-norm values from the 8 input bytes instead of keeping a [M, B*D] copy of them, and reduces the
// parameter gradients per block in a fixed order (no atomics).
#include "common.h"

#include <math.h>

namespace hivt {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;

struct EmbedArgs {
  const float* f;      // [M, f_ld]; branch b reads columns 2b, 2b+1
  int64_t f_ld;
  const float* w[2];   // [D, 2]
  const float* bias[2];
  const float* gamma[2];
  const float* beta[2];
  float eps;
  int64_t M;
  int32_t B;           // branches: 1 or 2
  float* h;            // forward out [M, B*D]
  // backward
  const float* dh;     // [M, B*D]
  float* partial;      // [grid][B][5*D]: dW (D x 2 interleaved) | dbias | dgamma | dbeta
  float* df;           // optional [M, 2*B]
};

template <int D>
struct Lane {
  static constexpr int LPR = D / 4;          // lanes per (row, branch)
  static constexpr int SLOTS = kWave / LPR;  // pairs per wave instruction
  float4 w0, w1, bias, gamma, beta;

  __device__ __forceinline__ void load(const EmbedArgs& a, int b, int c) {
    const float4 lo = *reinterpret_cast<const float4*>(a.w[b] + 2 * c);      // w[c][0], w[c][1], w[c+1][0], w[c+1][1]
    const float4 hi = *reinterpret_cast<const float4*>(a.w[b] + 2 * c + 4);  // columns c+2, c+3
    w0 = make_float4(lo.x, lo.z, hi.x, hi.z);
    w1 = make_float4(lo.y, lo.w, hi.y, hi.w);
    bias = *reinterpret_cast<const float4*>(a.bias[b] + c);
    gamma = *reinterpret_cast<const float4*>(a.gamma[b] + c);
    beta = *reinterpret_cast<const float4*>(a.beta[b] + c);
  }

  // pre-norm values, mean and rstd of one (row, branch) pair
  __device__ __forceinline__ void prenorm(float2 f, float eps, float4& xh, float& rstd) const {
    float4 z;
    z.x = fmaf(w1.x, f.y, fmaf(w0.x, f.x, bias.x));
    z.y = fmaf(w1.y, f.y, fmaf(w0.y, f.x, bias.y));
    z.z = fmaf(w1.z, f.y, fmaf(w0.z, f.x, bias.z));
    z.w = fmaf(w1.w, f.y, fmaf(w0.w, f.x, bias.w));
    const float mean = group_sum<LPR>((z.x + z.y) + (z.z + z.w)) * (1.0f / D);
    const float4 d = make_float4(z.x - mean, z.y - mean, z.z - mean, z.w - mean);
    const float var = group_sum<LPR>((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / D);
    rstd = rsqrtf(var + eps);
    xh = make_float4(d.x * rstd, d.y * rstd, d.z * rstd, d.w * rstd);
  }
};

template <int D>
__global__ __launch_bounds__(kThreads) void embed2_fwd_kernel(const EmbedArgs a) {
  using L = Lane<D>;
  constexpr int U = 2;  // pairs in flight per lane group
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / L::LPR, c = (lane % L::LPR) * 4;
  const int b = slot % a.B;  // SLOTS is a multiple of B, so a lane group always serves the same branch
  L k;
  k.load(a, b, c);
  const int64_t pairs = a.M * a.B;
  const int64_t stride = (int64_t)gridDim.x * kWaves * L::SLOTS;
  for (int64_t p0 = ((int64_t)blockIdx.x * kWaves + wave) * L::SLOTS + slot; p0 < pairs; p0 += U * stride) {
    float2 f[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * stride;
      ok[u] = p < pairs;
      const int64_t row = (ok[u] ? p : p0) / a.B;
      f[u] = *reinterpret_cast<const float2*>(a.f + row * a.f_ld + 2 * b);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 xh;
      float rstd;
      k.prenorm(f[u], a.eps, xh, rstd);
      if (ok[u]) {
        const int64_t row = (p0 + u * stride) / a.B;
        float4 y;
        y.x = fmaxf(fmaf(xh.x, k.gamma.x, k.beta.x), 0.0f);
        y.y = fmaxf(fmaf(xh.y, k.gamma.y, k.beta.y), 0.0f);
        y.z = fmaxf(fmaf(xh.z, k.gamma.z, k.beta.z), 0.0f);
        y.w = fmaxf(fmaf(xh.w, k.gamma.w, k.beta.w), 0.0f);
        *reinterpret_cast<float4*>(a.h + row * ((int64_t)a.B * D) + (int64_t)b * D + c) = y;
      }
    }
  }
}

template <int D>
__global__ __launch_bounds__(kThreads) void embed2_bwd_kernel(const EmbedArgs a) {
  using L = Lane<D>;
  constexpr int U = 2;
  constexpr int GROUPS = kWaves * L::SLOTS;  // lane groups per block
  __shared__ float red[GROUPS][5 * D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / L::LPR, c = (lane % L::LPR) * 4;
  const int b = slot % a.B;
  L k;
  k.load(a, b, c);
  float4 aw0 = {0.f, 0.f, 0.f, 0.f}, aw1 = aw0, ab = aw0, ag = aw0, abe = aw0;
  const int64_t pairs = a.M * a.B;
  const int64_t stride = (int64_t)gridDim.x * kWaves * L::SLOTS;
  for (int64_t p0 = ((int64_t)blockIdx.x * kWaves + wave) * L::SLOTS + slot; p0 < pairs; p0 += U * stride) {
    float2 f[U];
    float4 g[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * stride;
      ok[u] = p < pairs;
      const int64_t row = (ok[u] ? p : p0) / a.B;
      f[u] = *reinterpret_cast<const float2*>(a.f + row * a.f_ld + 2 * b);
      g[u] = *reinterpret_cast<const float4*>(a.dh + row * ((int64_t)a.B * D) + (int64_t)b * D + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 xh;
      float rstd;
      k.prenorm(f[u], a.eps, xh, rstd);
      // gradient at the LayerNorm output, masked by the ReLU and by the validity of the pair
      float4 dy;
      dy.x = (ok[u] && fmaf(xh.x, k.gamma.x, k.beta.x) > 0.0f) ? g[u].x : 0.0f;
      dy.y = (ok[u] && fmaf(xh.y, k.gamma.y, k.beta.y) > 0.0f) ? g[u].y : 0.0f;
      dy.z = (ok[u] && fmaf(xh.z, k.gamma.z, k.beta.z) > 0.0f) ? g[u].z : 0.0f;
      dy.w = (ok[u] && fmaf(xh.w, k.gamma.w, k.beta.w) > 0.0f) ? g[u].w : 0.0f;
      ag.x += dy.x * xh.x; ag.y += dy.y * xh.y; ag.z += dy.z * xh.z; ag.w += dy.w * xh.w;
      abe.x += dy.x; abe.y += dy.y; abe.z += dy.z; abe.w += dy.w;
      const float4 gg = make_float4(dy.x * k.gamma.x, dy.y * k.gamma.y, dy.z * k.gamma.z, dy.w * k.gamma.w);
      const float s1 = group_sum<L::LPR>((gg.x + gg.y) + (gg.z + gg.w)) * (1.0f / D);
      const float s2 =
          group_sum<L::LPR>((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * (1.0f / D);
      float4 dz;
      dz.x = rstd * (gg.x - s1 - xh.x * s2);
      dz.y = rstd * (gg.y - s1 - xh.y * s2);
      dz.z = rstd * (gg.z - s1 - xh.z * s2);
      dz.w = rstd * (gg.w - s1 - xh.w * s2);
      aw0.x = fmaf(dz.x, f[u].x, aw0.x); aw0.y = fmaf(dz.y, f[u].x, aw0.y);
      aw0.z = fmaf(dz.z, f[u].x, aw0.z); aw0.w = fmaf(dz.w, f[u].x, aw0.w);
      aw1.x = fmaf(dz.x, f[u].y, aw1.x); aw1.y = fmaf(dz.y, f[u].y, aw1.y);
      aw1.z = fmaf(dz.z, f[u].y, aw1.z); aw1.w = fmaf(dz.w, f[u].y, aw1.w);
      ab.x += dz.x; ab.y += dz.y; ab.z += dz.z; ab.w += dz.w;
      if (a.df != nullptr) {
        const float d0 = group_sum<L::LPR>((dz.x * k.w0.x + dz.y * k.w0.y) + (dz.z * k.w0.z + dz.w * k.w0.w));
        const float d1 = group_sum<L::LPR>((dz.x * k.w1.x + dz.y * k.w1.y) + (dz.z * k.w1.z + dz.w * k.w1.w));
        if (ok[u] && c == 0) {
          const int64_t row = (p0 + u * stride) / a.B;
          *reinterpret_cast<float2*>(a.df + row * (2 * (int64_t)a.B) + 2 * b) = make_float2(d0, d1);
        }
      }
    }
  }
  // block reduction in a fixed order: lane group -> LDS row, then the groups of one branch are summed
  float* mine = red[wave * L::SLOTS + slot];
  mine[2 * c + 0] = aw0.x; mine[2 * c + 1] = aw1.x; mine[2 * c + 2] = aw0.y; mine[2 * c + 3] = aw1.y;
  mine[2 * c + 4] = aw0.z; mine[2 * c + 5] = aw1.z; mine[2 * c + 6] = aw0.w; mine[2 * c + 7] = aw1.w;
  *reinterpret_cast<float4*>(mine + 2 * D + c) = ab;
  *reinterpret_cast<float4*>(mine + 3 * D + c) = ag;
  *reinterpret_cast<float4*>(mine + 4 * D + c) = abe;
  __syncthreads();
  for (int i = threadIdx.x; i < a.B * 5 * D; i += kThreads) {
    const int bb = i / (5 * D), col = i % (5 * D);
    float t = 0.0f;
    for (int gidx = bb; gidx < GROUPS; gidx += a.B) t += red[gidx][col];  // groups gidx % B == bb serve branch bb
    a.partial[((int64_t)blockIdx.x * a.B + bb) * (5 * D) + col] = t;
  }
}

int grid_for(int64_t pairs, int slots) {
  int64_t blocks = (pairs + (int64_t)kWaves * slots * 2 - 1) / ((int64_t)kWaves * slots * 2);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

int check(const char* what, int64_t M, int32_t D, int32_t B, int64_t f_ld, const float* f) {
  HIVT_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), HIVT_ERR_ARG, "%s: bad row count", what);
  HIVT_REQUIRE(D == 64 || D == 128, HIVT_ERR_ARG, "%s: width %d not in {64,128}", what, D);
  HIVT_REQUIRE(B == 1 || B == 2, HIVT_ERR_ARG, "%s: %d branches (1 or 2)", what, B);
  HIVT_REQUIRE(f_ld >= 2 * B && f_ld % 2 == 0 && (reinterpret_cast<uintptr_t>(f) & 7) == 0, HIVT_ERR_ARG,
               "%s: input rows must hold 2 floats per branch, 8-byte aligned", what);
  return HIVT_OK;
}

}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" size_t hivt_embed2_bwd_workspace_bytes(int64_t M, int32_t D, int32_t B) {
  if (D != 64 && D != 128) return 256;
  const int blocks = grid_for(M * B, kWave / (D / 4));
  return align_up((size_t)blocks * B * 5 * D * sizeof(float), 256);
}

extern "C" int hivt_embed2_fwd(const float* f, int64_t f_ld, int32_t branches, const float* const* w,
                               const float* const* bias, const float* const* gamma, const float* const* beta,
                               float eps, int64_t M, int32_t D, float* h, hivt_stream_t stream) {
  const char* what = "hivt_embed2_fwd";
  int rc = check(what, M, D, branches, f_ld, f);
  if (rc != HIVT_OK) return rc;
  if (M == 0) return HIVT_OK;
  HIVT_REQUIRE(f && w && bias && gamma && beta && h, HIVT_ERR_ARG, "%s: null pointer", what);
  EmbedArgs a{};
  a.f = f; a.f_ld = f_ld; a.eps = eps; a.M = M; a.B = branches; a.h = h;
  for (int b = 0; b < branches; ++b) {
    HIVT_REQUIRE(w[b] && bias[b] && gamma[b] && beta[b], HIVT_ERR_ARG, "%s: null parameter (branch %d)", what, b);
    HIVT_REQUIRE(((reinterpret_cast<uintptr_t>(w[b]) | reinterpret_cast<uintptr_t>(bias[b]) |
                   reinterpret_cast<uintptr_t>(gamma[b]) | reinterpret_cast<uintptr_t>(beta[b])) & 15) == 0,
                 HIVT_ERR_ARG, "%s: parameters must be 16-byte aligned", what);
    a.w[b] = w[b]; a.bias[b] = bias[b]; a.gamma[b] = gamma[b]; a.beta[b] = beta[b];
  }
  HIVT_REQUIRE((reinterpret_cast<uintptr_t>(h) & 15) == 0, HIVT_ERR_ARG, "%s: output must be 16-byte aligned", what);
  const int grid = grid_for(M * branches, kWave / (D / 4));
  if (D == 128)
    hipLaunchKernelGGL((embed2_fwd_kernel<128>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((embed2_fwd_kernel<64>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_embed2_bwd(const float* dh, const float* f, int64_t f_ld, int32_t branches,
                               const float* const* w, const float* const* bias, const float* const* gamma,
                               const float* const* beta, float eps, int64_t M, int32_t D, float* dparams, float* df,
                               void* workspace, size_t workspace_bytes, hivt_stream_t stream) {
  const char* what = "hivt_embed2_bwd";
  int rc = check(what, M, D, branches, f_ld, f);
  if (rc != HIVT_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int cols = branches * 5 * D;
  if (M == 0) {
    HIVT_REQUIRE(dparams != nullptr, HIVT_ERR_ARG, "%s: dparams is null (no rows: nothing to leave as partial sums)", what);
    zero_words(dparams, cols, s);
    HIVT_CHECK_LAUNCH(what);
    return HIVT_OK;
  }
  HIVT_REQUIRE(dh && f && w && bias && gamma && beta, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(workspace && workspace_bytes >= hivt_embed2_bwd_workspace_bytes(M, D, branches), HIVT_ERR_WORKSPACE,
               "%s: workspace too small", what);
  EmbedArgs a{};
  a.f = f; a.f_ld = f_ld; a.eps = eps; a.M = M; a.B = branches; a.dh = dh; a.df = df;
  a.partial = reinterpret_cast<float*>(workspace);
  for (int b = 0; b < branches; ++b) {
    HIVT_REQUIRE(w[b] && bias[b] && gamma[b] && beta[b], HIVT_ERR_ARG, "%s: null parameter (branch %d)", what, b);
    HIVT_REQUIRE(((reinterpret_cast<uintptr_t>(w[b]) | reinterpret_cast<uintptr_t>(bias[b]) |
                   reinterpret_cast<uintptr_t>(gamma[b]) | reinterpret_cast<uintptr_t>(beta[b])) & 15) == 0,
                 HIVT_ERR_ARG, "%s: parameters must be 16-byte aligned", what);
    a.w[b] = w[b]; a.bias[b] = bias[b]; a.gamma[b] = gamma[b]; a.beta[b] = beta[b];
  }
  HIVT_REQUIRE((reinterpret_cast<uintptr_t>(dh) & 15) == 0 && (df == nullptr || (reinterpret_cast<uintptr_t>(df) & 7) == 0),
               HIVT_ERR_ARG, "%s: gradients must be 16-byte (dh) / 8-byte (df) aligned", what);
  const int grid = grid_for(M * branches, kWave / (D / 4));
  if (D == 128)
    hipLaunchKernelGGL((embed2_bwd_kernel<128>), dim3(grid), dim3(kThreads), 0, s, a);
  else
    hipLaunchKernelGGL((embed2_bwd_kernel<64>), dim3(grid), dim3(kThreads), 0, s, a);
  HIVT_CHECK_LAUNCH(what);
  if (dparams == nullptr) return HIVT_OK;  // the caller sums workspace[hivt_embed2_bwd_partial_rows][branches * 5 D]
  reduce_partials(a.partial, grid, cols, dparams, s);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" size_t hivt_embed2_bwd_partial_rows(int64_t M, int32_t D, int32_t branches) {
  if ((D != 64 && D != 128) || M <= 0) return 0;
  return (size_t)grid_for(M * branches, kWave / (D / 4));
}
