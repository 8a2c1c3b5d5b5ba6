// One trajectory critic, whole passes per call.
//
// A critic is an MLP over B flattened trajectories: L hidden layers h_l = Dropout(LeakyReLU(h_{l-1} W_l^T
// + b_l)) and a one-wide output layer D = h_L w_o^T + b_o (models/critics/base_critic.py:19-44).  The GAN
// step calls each critic nine times and needs, for the real samples, the squared norm of dD/dx as a
// differentiable quantity (R1 penalty, losses/adv_losses.py:36-47 and models/hivt.py:244-252).  Driven
// from the framework that is several hundred small operator launches per step and a double backward.
//
// Here a forward (optionally with the input-gradient chain) or a backward is ONE host call that issues
// the whole sequence of launches.  The R1 term needs no double backward: with the activation / dropout
// factors c_l = d h_l / d z_l (piecewise constant, read off the sign of the stored h_l) the input
// gradient is itself a short chain of products,
//     G_L = w_o * c_L,   g_{l-1} = G_l W_l,   G_{l-1} = g_{l-1} * c_{l-1},   dD/dx = g_0,
// whose gradient with respect to the weights is a second chain of the same shape:
//     e_0 = d r / d g_0,   dW_l += G_l^T e_{l-1},   e_l = (e_{l-1} W_l^T) * c_l,   dw_o += colsum(e_L).
// All products run on the fp32-MFMA linear kernels of this library.
#include "common.h"

#include <math.h>

namespace hivt {
namespace {

constexpr int kMaxHidden = HIVT_CRITIC_MAX_LAYERS;  // hidden layers per critic

// d h / d z of Dropout(LeakyReLU(z)) recovered from h: kept and positive, kept and negative, dropped.
__device__ __forceinline__ float act_factor(float h, float keep_scale, float neg_slope) {
  return h > 0.0f ? keep_scale : (h < 0.0f ? keep_scale * neg_slope : 0.0f);
}

// out[i][c] = (row ? row[i] : 1) * (vec ? vec[c] : 1) * (in ? in[i][c] : 1) * act_factor(h[i][c])
__global__ __launch_bounds__(256) void masked_product_kernel(const float* __restrict__ in, const float* __restrict__ row,
                                                             const float* __restrict__ vec,
                                                             const float* __restrict__ h, float keep_scale,
                                                             float neg_slope, int64_t B, int W,
                                                             float* __restrict__ out) {
  const int64_t total = B * (int64_t)W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / W;
    const int c = (int)(i - r * W);
    float v = act_factor(h[i], keep_scale, neg_slope);
    if (in != nullptr) v *= in[i];
    if (row != nullptr) v *= row[r];
    if (vec != nullptr) v *= vec[c];
    out[i] = v;
  }
}

// out[i][c] = scale * row[i] * in[i][c]
__global__ __launch_bounds__(256) void row_scale_kernel(const float* __restrict__ in, const float* __restrict__ row,
                                                        float scale, int64_t B, int W, float* __restrict__ out) {
  const int64_t total = B * (int64_t)W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = scale * row[i / W] * in[i];
}

// out[i] = sum_c in[i][c]^2, one wave per row
__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ in, int64_t B, int W,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < B; r += (int64_t)gridDim.x * 4) {
    float s = 0.0f;
    for (int c = lane; c < W; c += kWave) {
      const float v = in[r * W + c];
      s = fmaf(v, v, s);
    }
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
  }
}

inline unsigned blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

struct Shape {
  int64_t B;
  int32_t L;
  int32_t w[kMaxHidden + 1];  // w[0] = input width, w[l] = width of hidden layer l
  int64_t h_off[kMaxHidden + 2];  // float offsets of h_1..h_L in `saved`
  int64_t g_off[kMaxHidden + 2];  // float offsets of G_1..G_L
  int64_t g0_off;
  int64_t total_plain, total_r1;
  int32_t max_w;
};

int make_shape(const char* what, int64_t B, int32_t layers, const int32_t* widths, Shape& s) {
  HIVT_REQUIRE(B >= 0 && B < ((int64_t)1 << 31), HIVT_ERR_ARG, "%s: bad batch size", what);
  HIVT_REQUIRE(layers >= 1 && layers <= kMaxHidden && widths != nullptr, HIVT_ERR_ARG,
               "%s: %d hidden layers (1..%d)", what, layers, kMaxHidden);
  s.B = B;
  s.L = layers;
  s.max_w = 0;
  int64_t off = 0;
  for (int l = 0; l <= layers; ++l) {
    HIVT_REQUIRE(widths[l] >= 1 && widths[l] <= 4096, HIVT_ERR_ARG, "%s: layer width %d", what, widths[l]);
    s.w[l] = widths[l];
    if (widths[l] > s.max_w) s.max_w = widths[l];
  }
  for (int l = 1; l <= layers; ++l) {
    s.h_off[l] = off;
    off += align_up((size_t)(B * s.w[l]), 64);
  }
  s.total_plain = off;
  for (int l = 1; l <= layers; ++l) {
    s.g_off[l] = off;
    off += align_up((size_t)(B * s.w[l]), 64);
  }
  s.g0_off = off;
  off += align_up((size_t)(B * s.w[0]), 64);
  s.total_r1 = off;
  return HIVT_OK;
}

// scratch: the backward chains' intermediate rows (dz_1..dz_L of the score chain, e_0..e_L of the penalty
// chain: all of them are operands of the weight-gradient products issued together at the end) + the workspace
// of that grouped call
struct Scratch {
  float* dz[kMaxHidden + 1];  // dz[l]: [B, w_l], l = 1..L
  float* e[kMaxHidden + 1];   // e[l]: [B, w_l], l = 0..L
  void* wgrad_ws;
  size_t wgrad_bytes;
};

size_t act_bytes(const Shape& s, int l) {
  return align_up((size_t)(s.B > 0 ? s.B : 1) * s.w[l] * sizeof(float), 256);
}

// upper bound of the grouped call's workspace: every weight takes at most two terms
size_t group_bytes(const Shape& s) {
  size_t total = 256;
  for (int l = 1; l <= s.L; ++l)
    total += 2 * align_up(hivt_linear_wgrad_workspace_bytes(s.B, s.w[l], s.w[l - 1]), 256);
  total += align_up(hivt_linear_wgrad_workspace_bytes(s.B, 1, s.w[s.L]), 256);
  return total;
}

size_t scratch_bytes(const Shape& s) {
  size_t total = group_bytes(s);
  for (int l = 1; l <= s.L; ++l) total += act_bytes(s, l);
  for (int l = 0; l <= s.L; ++l) total += act_bytes(s, l);
  return total;
}

Scratch carve(const Shape& s, void* ws) {
  Scratch k{};
  char* p = reinterpret_cast<char*>(ws);
  for (int l = 1; l <= s.L; ++l) {
    k.dz[l] = reinterpret_cast<float*>(p);
    p += act_bytes(s, l);
  }
  for (int l = 0; l <= s.L; ++l) {
    k.e[l] = reinterpret_cast<float*>(p);
    p += act_bytes(s, l);
  }
  k.wgrad_ws = p;
  k.wgrad_bytes = group_bytes(s);
  return k;
}

// one row of the item table of hivt_linear_wgrad_group
struct Terms {
  int64_t row[2 * (kMaxHidden + 1)][10];
  int32_t count = 0;
  void add(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, int64_t M, int32_t N, int32_t K, float* dw,
           float* dbias, bool accumulate, bool joins) {
    int64_t* r = row[count++];
    r[0] = (int64_t)(uintptr_t)dy; r[1] = dy_ld; r[2] = (int64_t)(uintptr_t)x; r[3] = x_ld; r[4] = M; r[5] = N; r[6] = K;
    r[7] = (int64_t)(uintptr_t)dw; r[8] = (int64_t)(uintptr_t)dbias; r[9] = (accumulate ? 1 : 0) | (joins ? 2 : 0);
  }
};

#define HIVT_TRY(expr)            \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != HIVT_OK) return rc__; \
  } while (0)

// Block-wide sum in a fixed order (pairwise tree over the 256 partial sums): independent of scheduling.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int half = 128; half > 0; half >>= 1) {
    if ((int)threadIdx.x < half) red[threadIdx.x] += red[threadIdx.x + half];
    __syncthreads();
  }
  const float total = red[0];
  __syncthreads();
  return total;
}

// out[0] = mean_w relu(1 - s_real) + mean_w relu(1 + s_fake) + lambda * r1,  out[1] = r1 = mean_w grad_norm2,
// mean_w(v) = sum_i w_i v_i / count.  One block; each thread owns a fixed strided subset.
__global__ __launch_bounds__(256) void hinge_r1_fwd_kernel(const float* __restrict__ s_real,
                                                           const float* __restrict__ grad_norm2,
                                                           const float* __restrict__ s_fake,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ count, int64_t B, float lambda,
                                                           float* __restrict__ out) {
  __shared__ float red[256];
  float a = 0.0f, b = 0.0f, c = 0.0f;
  for (int64_t i = threadIdx.x; i < B; i += 256) {
    const float wi = w[i];
    a += fmaxf(1.0f - s_real[i], 0.0f) * wi;
    b += fmaxf(1.0f + s_fake[i], 0.0f) * wi;
    c += grad_norm2[i] * wi;
  }
  a = block_sum_256(a, red);
  b = block_sum_256(b, red);
  c = block_sum_256(c, red);
  if (threadIdx.x == 0) {
    const float cnt = *count;
    const float r1 = c / cnt;
    out[0] = (a / cnt + b / cnt) + lambda * r1;
    out[1] = r1;
  }
}

__global__ __launch_bounds__(256) void hinge_r1_bwd_kernel(const float* __restrict__ g_loss,
                                                           const float* __restrict__ s_real,
                                                           const float* __restrict__ s_fake,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ count, int64_t B, float lambda,
                                                           float* __restrict__ g_real, float* __restrict__ g_norm2,
                                                           float* __restrict__ g_fake) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  const float unit = *g_loss * w[i] / *count;
  g_real[i] = (1.0f - s_real[i] > 0.0f) ? -unit : 0.0f;
  g_fake[i] = (1.0f + s_fake[i] > 0.0f) ? unit : 0.0f;
  g_norm2[i] = lambda * unit;
}

// out = scale * sum_i w_i s_i / count, and its gradient
__global__ __launch_bounds__(256) void weighted_mean_fwd_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                                const float* __restrict__ count, int64_t B, float scale,
                                                                float* __restrict__ out) {
  __shared__ float red[256];
  float a = 0.0f;
  for (int64_t i = threadIdx.x; i < B; i += 256) a += s[i] * w[i];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) out[0] = scale * (a / *count);
}

__global__ __launch_bounds__(256) void weighted_mean_bwd_kernel(const float* __restrict__ g_out,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ count, int64_t B, float scale,
                                                                float* __restrict__ g_s) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < B) g_s[i] = scale * *g_out * w[i] / *count;
}


}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" size_t hivt_critic_saved_floats(int64_t B, int32_t layers, const int32_t* widths, int32_t with_input_grad) {
  Shape s;
  if (make_shape("hivt_critic_saved_floats", B, layers, widths, s) != HIVT_OK) return 0;
  const int64_t n = with_input_grad ? s.total_r1 : s.total_plain;
  return (size_t)(n > 0 ? n : 1);
}

extern "C" size_t hivt_critic_workspace_bytes(int64_t B, int32_t layers, const int32_t* widths) {
  Shape s;
  if (make_shape("hivt_critic_workspace_bytes", B, layers, widths, s) != HIVT_OK) return 0;
  return scratch_bytes(s);
}

extern "C" int hivt_critic_fwd(const float* x, int64_t x_ld, int64_t B, int32_t layers, const int32_t* widths,
                               const float* const* w, const float* const* bias, float neg_slope, float dropout_p,
                               uint64_t seed, const uint64_t* seed_ptr, int32_t with_input_grad, float* saved,
                               float* scores, float* grad_norm2, hivt_stream_t stream) {
  const char* what = "hivt_critic_fwd";
  Shape s;
  HIVT_TRY(make_shape(what, B, layers, widths, s));
  HIVT_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, HIVT_ERR_ARG, "%s: dropout_p must be in [0,1)", what);
  if (B == 0) return HIVT_OK;
  HIVT_REQUIRE(x && w && bias && saved && scores, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(!with_input_grad || grad_norm2, HIVT_ERR_ARG, "%s: grad_norm2 is null", what);
  HIVT_REQUIRE(x_ld >= s.w[0], HIVT_ERR_ARG, "%s: x_ld smaller than the input width", what);
  for (int l = 0; l <= layers; ++l)
    HIVT_REQUIRE(w[l] && bias[l], HIVT_ERR_ARG, "%s: null parameter (layer %d)", what, l);
  hipStream_t st = (hipStream_t)stream;
  const float keep_scale = 1.0f / (1.0f - dropout_p);
  // hidden layers (every layer draws its own dropout stream)
  const float* in = x;
  int64_t in_ld = x_ld;
  for (int l = 1; l <= layers; ++l) {
    float* h = saved + s.h_off[l];
    HIVT_TRY(hivt_linear_fwd(in, in_ld, w[l - 1], s.w[l - 1], bias[l - 1], nullptr, 0, nullptr, 0, nullptr, nullptr,
                             0.0f, 0 /* plain */, neg_slope, dropout_p, seed + 0x51ED27ull * (uint64_t)l, seed_ptr,
                             nullptr, 0, B, s.w[l], s.w[l - 1], h, s.w[l], nullptr, 0, nullptr, stream));
    in = h;
    in_ld = s.w[l];
  }
  // output layer: one column
  HIVT_TRY(hivt_linear_fwd(in, in_ld, w[layers], s.w[layers], bias[layers], nullptr, 0, nullptr, 0, nullptr, nullptr,
                           0.0f, 0, 1.0f, 0.0f, 0, nullptr, nullptr, 0, B, 1, s.w[layers], scores, 1, nullptr, 0,
                           nullptr, stream));
  if (!with_input_grad) return HIVT_OK;
  // input-gradient chain
  float* G = saved + s.g_off[layers];
  hipLaunchKernelGGL(masked_product_kernel, dim3(blocks_for(B * s.w[layers])), dim3(256), 0, st, (const float*)nullptr,
                     (const float*)nullptr, w[layers], saved + s.h_off[layers], keep_scale, neg_slope, B, s.w[layers],
                     G);
  HIVT_CHECK_LAUNCH(what);
  for (int l = layers; l >= 1; --l) {
    float* g_prev = l > 1 ? saved + s.g_off[l - 1] : saved + s.g0_off;
    if (l > 1)  // through the activation of layer l - 1, inside the product's epilogue
      HIVT_TRY(linear_bwd_data_gated(saved + s.g_off[l], s.w[l], w[l - 1], s.w[l - 1], B, s.w[l], s.w[l - 1],
                                     saved + s.h_off[l - 1], s.w[l - 1], keep_scale, keep_scale * neg_slope, g_prev,
                                     s.w[l - 1], stream));
    else
      HIVT_TRY(hivt_linear_bwd_data(saved + s.g_off[l], s.w[l], w[l - 1], s.w[l - 1], B, s.w[l], s.w[l - 1], nullptr, 0,
                                    1.0f, nullptr, 0, g_prev, s.w[l - 1], stream));
  }
  hipLaunchKernelGGL(row_sumsq_kernel, dim3((unsigned)((B + 3) / 4 > 2048 ? 2048 : (B + 3) / 4)), dim3(256), 0, st,
                     (const float*)(saved + s.g0_off), B, s.w[0], grad_norm2);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_critic_bwd(const float* g_scores, const float* g_norm2, const float* x, int64_t x_ld, int64_t B,
                               int32_t layers, const int32_t* widths, const float* const* w, float neg_slope,
                               float dropout_p, const float* saved, float* const* dw, float* const* dbias, float* dx,
                               void* workspace, size_t workspace_bytes, hivt_stream_t stream) {
  const char* what = "hivt_critic_bwd";
  Shape s;
  HIVT_TRY(make_shape(what, B, layers, widths, s));
  // dw == dbias == null: the parameters are frozen, only dx (the gradient of the scores) is wanted
  const bool params = dw != nullptr || dbias != nullptr;
  HIVT_REQUIRE(!params || (dw && dbias), HIVT_ERR_ARG, "%s: dw and dbias go together", what);
  HIVT_REQUIRE(params || (g_norm2 == nullptr && dx != nullptr), HIVT_ERR_ARG,
               "%s: without parameter gradients the call computes dx from g_scores only", what);
  if (params)
    for (int l = 0; l <= layers; ++l)
      HIVT_REQUIRE(dw[l] && dbias[l], HIVT_ERR_ARG, "%s: null gradient (layer %d)", what, l);
  hipStream_t st = (hipStream_t)stream;
  if (!params && (B == 0 || g_scores == nullptr)) {
    if (B > 0) zero_words(dx, B * s.w[0], st);
    HIVT_CHECK_LAUNCH(what);
    return HIVT_OK;
  }
  if (B == 0 || (g_scores == nullptr && g_norm2 == nullptr)) {
    for (int l = 1; l <= layers; ++l) {
      zero_words(dw[l - 1], (int64_t)s.w[l] * s.w[l - 1], st);
      zero_words(dbias[l - 1], s.w[l], st);
    }
    zero_words(dw[layers], s.w[layers], st);
    zero_words(dbias[layers], 1, st);
    if (dx != nullptr && B > 0) zero_words(dx, B * s.w[0], st);
    HIVT_CHECK_LAUNCH(what);
    return HIVT_OK;
  }
  HIVT_REQUIRE(x && w && saved, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(workspace && workspace_bytes >= scratch_bytes(s), HIVT_ERR_WORKSPACE, "%s: workspace too small", what);
  for (int l = 0; l <= layers; ++l) HIVT_REQUIRE(w[l], HIVT_ERR_ARG, "%s: null weight (layer %d)", what, l);
  const Scratch k = carve(s, workspace);
  const float keep_scale = 1.0f / (1.0f - dropout_p);

  // The weight gradients of both chains are leaves: their products are collected here and issued in one
  // grouped call at the end, each weight as one entry (score-chain term, joined by the penalty-chain term).
  const bool scores_chain = g_scores != nullptr, penalty_chain = g_norm2 != nullptr;
  if (scores_chain) {
    // ---- gradient of the scores: ordinary backward through the layers ----
    // dz_L = (g w_o) * c_L
    hipLaunchKernelGGL(masked_product_kernel, dim3(blocks_for(B * s.w[layers])), dim3(256), 0, st, (const float*)nullptr,
                       g_scores, w[layers], saved + s.h_off[layers], keep_scale, neg_slope, B, s.w[layers],
                       k.dz[layers]);
    HIVT_CHECK_LAUNCH(what);
    for (int l = layers; l >= 1; --l) {
      if (l > 1) {
        HIVT_TRY(linear_bwd_data_gated(k.dz[l], s.w[l], w[l - 1], s.w[l - 1], B, s.w[l], s.w[l - 1],
                                       saved + s.h_off[l - 1], s.w[l - 1], keep_scale, keep_scale * neg_slope,
                                       k.dz[l - 1], s.w[l - 1], stream));
      } else if (dx != nullptr) {
        HIVT_TRY(hivt_linear_bwd_data(k.dz[l], s.w[l], w[l - 1], s.w[l - 1], B, s.w[l], s.w[l - 1], nullptr, 0, 1.0f,
                                      nullptr, 0, dx, s.w[l - 1], stream));
      }
    }
  } else {
    for (int l = 1; l <= layers; ++l) zero_words(dbias[l - 1], s.w[l], st);
    zero_words(dbias[layers], 1, st);
    if (dx != nullptr) zero_words(dx, B * s.w[0], st);
    HIVT_CHECK_LAUNCH(what);
  }

  if (penalty_chain) {
    // ---- gradient of ||dD/dx||^2: the second chain (weights only; the factors c_l are constants) ----
    hipLaunchKernelGGL(row_scale_kernel, dim3(blocks_for(B * s.w[0])), dim3(256), 0, st,
                       (const float*)(saved + s.g0_off), g_norm2, 2.0f, B, s.w[0], k.e[0]);
    HIVT_CHECK_LAUNCH(what);
    // e_l = (e_{l-1} W_l^T) * c_l
    for (int l = 1; l <= layers; ++l)
      HIVT_TRY(linear_fwd_gated(k.e[l - 1], s.w[l - 1], w[l - 1], s.w[l - 1], B, s.w[l], s.w[l - 1],
                                saved + s.h_off[l], s.w[l], keep_scale, keep_scale * neg_slope, k.e[l], s.w[l], stream));
  }

  if (params) {
    Terms t;
    // output layer: dw_o = g^T h_L, db_o = sum g
    if (scores_chain)
      t.add(g_scores, 1, saved + s.h_off[layers], s.w[layers], B, 1, s.w[layers], dw[layers], dbias[layers], false, false);
    for (int l = layers; l >= 1; --l) {
      const float* h_prev = l > 1 ? saved + s.h_off[l - 1] : x;
      const int64_t h_prev_ld = l > 1 ? s.w[l - 1] : x_ld;
      if (scores_chain)  // dW_l = dz_l^T h_{l-1}, db_l = column sums of dz_l
        t.add(k.dz[l], s.w[l], h_prev, h_prev_ld, B, s.w[l], s.w[l - 1], dw[l - 1], dbias[l - 1], false, false);
      if (penalty_chain)  // dW_l (+)= G_l^T e_{l-1}
        t.add(saved + s.g_off[l], s.w[l], k.e[l - 1], s.w[l - 1], B, s.w[l], s.w[l - 1], dw[l - 1], nullptr, false,
              scores_chain);
    }
    HIVT_TRY(hivt_linear_wgrad_group(&t.row[0][0], t.count, k.wgrad_ws, k.wgrad_bytes, stream));
    if (penalty_chain) {
      // dw_o (+)= column sums of e_L (rows summed in a fixed order)
      reduce_partials(k.e[layers], (int)B, s.w[layers], dw[layers], st, scores_chain);
      HIVT_CHECK_LAUNCH(what);
    }
  }
  return HIVT_OK;
}

extern "C" int hivt_hinge_r1_fwd(const float* s_real, const float* grad_norm2, const float* s_fake, const float* w,
                                 const float* count, int64_t B, float lambda_r1, float* out, hivt_stream_t stream) {
  const char* what = "hivt_hinge_r1_fwd";
  HIVT_REQUIRE(B >= 0, HIVT_ERR_ARG, "%s: bad size", what);
  HIVT_REQUIRE(count && out && (B == 0 || (s_real && grad_norm2 && s_fake && w)), HIVT_ERR_ARG, "%s: null pointer",
               what);
  hipLaunchKernelGGL(hinge_r1_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, s_real, grad_norm2, s_fake, w,
                     count, B, lambda_r1, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_hinge_r1_bwd(const float* g_loss, const float* s_real, const float* s_fake, const float* w,
                                 const float* count, int64_t B, float lambda_r1, float* g_real, float* g_norm2,
                                 float* g_fake, hivt_stream_t stream) {
  const char* what = "hivt_hinge_r1_bwd";
  HIVT_REQUIRE(B >= 0, HIVT_ERR_ARG, "%s: bad size", what);
  if (B == 0) return HIVT_OK;
  HIVT_REQUIRE(g_loss && s_real && s_fake && w && count && g_real && g_norm2 && g_fake, HIVT_ERR_ARG,
               "%s: null pointer", what);
  hipLaunchKernelGGL(hinge_r1_bwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_loss,
                     s_real, s_fake, w, count, B, lambda_r1, g_real, g_norm2, g_fake);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_weighted_mean_fwd(const float* s, const float* w, const float* count, int64_t B, float scale,
                                      float* out, hivt_stream_t stream) {
  const char* what = "hivt_weighted_mean_fwd";
  HIVT_REQUIRE(B >= 0, HIVT_ERR_ARG, "%s: bad size", what);
  HIVT_REQUIRE(count && out && (B == 0 || (s && w)), HIVT_ERR_ARG, "%s: null pointer", what);
  hipLaunchKernelGGL(weighted_mean_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, s, w, count, B, scale, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_weighted_mean_bwd(const float* g_out, const float* w, const float* count, int64_t B, float scale,
                                      float* g_s, hivt_stream_t stream) {
  const char* what = "hivt_weighted_mean_bwd";
  HIVT_REQUIRE(B >= 0, HIVT_ERR_ARG, "%s: bad size", what);
  if (B == 0) return HIVT_OK;
  HIVT_REQUIRE(g_out && w && count && g_s, HIVT_ERR_ARG, "%s: null pointer", what);
  hipLaunchKernelGGL(weighted_mean_bwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     g_out, w, count, B, scale, g_s);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}
