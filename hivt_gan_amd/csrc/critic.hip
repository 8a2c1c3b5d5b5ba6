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
#include "linear_core.h"

#include <math.h>

#include <vector>

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
  float* out_part[2];         // output layer's gradient per 64-row tile, [tiles][out_part_pitch]: score / penalty chain
  void* wgrad_ws;
  size_t wgrad_bytes;
};

constexpr int kOutTileRows = 64;  // = the rows a workgroup of the layer chain carries
int64_t out_part_pitch(const Shape& s) { return (int64_t)(s.w[s.L] + 1 + 3) / 4 * 4; }
int64_t out_part_tiles(const Shape& s) { return ((s.B > 0 ? s.B : 1) + kOutTileRows - 1) / kOutTileRows; }
size_t out_part_bytes(const Shape& s) {
  return align_up((size_t)(out_part_tiles(s) * out_part_pitch(s)) * sizeof(float), 256);
}

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
  total += 2 * out_part_bytes(s);
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
  for (int i = 0; i < 2; ++i) {
    k.out_part[i] = reinterpret_cast<float*>(p);
    p += out_part_bytes(s);
  }
  k.wgrad_ws = p;
  k.wgrad_bytes = group_bytes(s);
  return k;
}

// one row of the item table of hivt_linear_wgrad_group
struct Terms {
  int64_t row[2 * (kMaxHidden + 1) + 8][10];
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

// All critic scales at once (losses/adv_losses.py:36-58 summed over the scales as models/hivt.py:305-310 does):
// out[0] = sum_k loss_k (added in scale order), out[1 + 2k] = loss_k, out[2 + 2k] = r1_k.  One block.
constexpr int kMaxScales = 4;

struct ScaleRows {
  const float* s_real[kMaxScales];
  const float* grad_norm2[kMaxScales];
  const float* s_fake[kMaxScales];
  float* g_real[kMaxScales];
  float* g_norm2[kMaxScales];
  float* g_fake[kMaxScales];
  int32_t count;
};

// Sums of K values per thread over a workgroup of kLossThreads threads, in a fixed order: xor-shuffle tree inside
// every wave, then the waves' sums added in wave order.  One barrier pair per call.
constexpr int kLossThreads = 1024;

template <int K>
__device__ __forceinline__ void block_sums(float (&v)[K], float (*smem)[K]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) smem[wave][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float t = smem[0][k];
    for (int w = 1; w < kLossThreads / 64; ++w) t += smem[w][k];
    v[k] = t;
  }
  __syncthreads();
}

// The number of kept rows: the caller's device scalar, or max(sum of w, 1) computed here.
__device__ __forceinline__ float kept_rows(const float* __restrict__ count, const float* __restrict__ w, int64_t B,
                                           float (*smem)[1]) {
  if (count != nullptr) return *count;
  float c[1] = {0.0f};
  for (int64_t i = threadIdx.x; i < B; i += kLossThreads) c[0] += w[i];
  block_sums<1>(c, smem);
  return fmaxf(c[0], 1.0f);
}

// out[0] = sum_k loss_k, out[1 + 2k] = loss_k, out[2 + 2k] = r1_k, out[1 + 2K] = the number of kept rows
__global__ __launch_bounds__(kLossThreads) void hinge_r1_multi_fwd_kernel(const ScaleRows p, const float* __restrict__ w,
                                                                          const float* __restrict__ count, int64_t B,
                                                                          float lambda, float* __restrict__ out) {
  __shared__ float smem3[kLossThreads / 64][3];
  __shared__ float smem1[kLossThreads / 64][1];
  const float cnt = kept_rows(count, w, B, smem1);
  float total = 0.0f;
  for (int k = 0; k < p.count; ++k) {
    float v[3] = {0.0f, 0.0f, 0.0f};
    for (int64_t i = threadIdx.x; i < B; i += kLossThreads) {
      const float wi = w[i];
      v[0] += fmaxf(1.0f - p.s_real[k][i], 0.0f) * wi;
      v[1] += fmaxf(1.0f + p.s_fake[k][i], 0.0f) * wi;
      v[2] += p.grad_norm2[k][i] * wi;
    }
    block_sums<3>(v, smem3);
    const float r1 = v[2] / cnt;
    const float loss = (v[0] / cnt + v[1] / cnt) + lambda * r1;
    total = k == 0 ? loss : total + loss;
    if (threadIdx.x == 0) {
      out[1 + 2 * k] = loss;
      out[2 + 2 * k] = r1;
    }
  }
  if (threadIdx.x == 0) {
    out[0] = total;
    out[1 + 2 * p.count] = cnt;
  }
}

__global__ __launch_bounds__(256) void hinge_r1_multi_bwd_kernel(const ScaleRows p, const float* __restrict__ g_loss,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ count, int64_t B, float lambda) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= B) return;
  const float unit = *g_loss * w[i] / *count;
  p.g_real[k][i] = (1.0f - p.s_real[k][i] > 0.0f) ? -unit : 0.0f;
  p.g_fake[k][i] = (1.0f + p.s_fake[k][i] > 0.0f) ? unit : 0.0f;
  p.g_norm2[k][i] = lambda * unit;
}

// Generator objective (models/hivt.py:317-324 with losses/adv_losses.py:91-98): out[0] = (losses[0] + losses[1]) +
// out[1], out[1] = lambda_adv * sum_k out[2 + k], out[2 + k] = -sum_i w_i s_k[i] / count, out[2 + K] = the number
// of kept rows.  One workgroup.
__global__ __launch_bounds__(kLossThreads) void generator_total_fwd_kernel(const ScaleRows p, const float* __restrict__ losses,
                                                                           const float* __restrict__ w,
                                                                           const float* __restrict__ count, int64_t B,
                                                                           float lambda_adv, float* __restrict__ out) {
  __shared__ float smem4[kLossThreads / 64][kMaxScales];
  __shared__ float smem1[kLossThreads / 64][1];
  const float cnt = kept_rows(count, w, B, smem1);
  float v[kMaxScales] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int64_t i = threadIdx.x; i < B; i += kLossThreads) {
    const float wi = w[i];
#pragma unroll
    for (int k = 0; k < kMaxScales; ++k)
      if (k < p.count) v[k] += p.s_fake[k][i] * wi;
  }
  block_sums<kMaxScales>(v, smem4);
  if (threadIdx.x == 0) {
    float total = 0.0f;
    for (int k = 0; k < p.count; ++k) {
      const float g = -1.0f * (v[k] / cnt);
      total = k == 0 ? g : total + g;
      out[2 + k] = g;
    }
    const float adv = lambda_adv * total;
    out[1] = adv;
    out[0] = (losses[0] + losses[1]) + adv;
    out[2 + p.count] = cnt;
  }
}

__global__ __launch_bounds__(256) void generator_total_bwd_kernel(const ScaleRows p, const float* __restrict__ g_out,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ count, int64_t B,
                                                                  float lambda_adv, float* __restrict__ g_losses) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  const float g = *g_out;
  if (blockIdx.x == 0 && k == 0 && threadIdx.x < 2) g_losses[threadIdx.x] = g;
  if (i >= B) return;
  const float go = lambda_adv * g;
  p.g_fake[k][i] = -1.0f * go * w[i] / *count;
}

// ---------------------------------------------------------------------------------------------
// Layer chains.  A critic pass over a few thousand rows is a chain of small products in which every row
// depends on its own row of the layer before only.  One workgroup therefore carries kChainRows rows of a pass
// through ALL layers of a chain (the layer outputs go to global memory as the backward pass needs them anyway,
// and come back through L2), and the workgroups of up to kChainPasses passes -- the real and the generated
// trajectories of all critics -- share the launch: one launch where the layer-by-layer route issues a dozen
// per pass, each of them a few microseconds of latency for microseconds of work.
// ---------------------------------------------------------------------------------------------
constexpr int kChainPasses = 6;
constexpr int kChainLayers = 4;   // hidden layers a chain is built for (the reference's critics have 2, 3, 4)
constexpr int kChainRows = 64;

struct ChainPass {
  const float* x;
  int64_t x_ld;
  int32_t B, L;
  int32_t w[kChainLayers + 1];
  const float* weight[kChainLayers + 1];  // weight[l - 1] = W_l [w_l, w_{l-1}]; weight[L] = w_o [1, w_L]
  const float* bias[kChainLayers + 1];    // forward only
  float* saved;                           // h_1..h_L, then G_1..G_L and g_0 (laid out as make_shape says)
  int64_t h_off[kChainLayers + 1], g_off[kChainLayers + 1], g0_off;  // float offsets inside `saved`
  uint64_t seed;
  float* scores;                          // forward
  float* grad_norm2;                      // forward: non-null = the input-gradient chain runs too
  const float* g_scores;                  // backward (either may be null)
  const float* g_norm2;
  float* dz[kChainLayers + 1];            // backward: dz[l] [B, w_l], l = 1..L
  float* e[kChainLayers + 1];             // backward: e[l] [B, w_l], l = 0..L
  float* dx;                              // backward, optional
  float* out_part[2];                     // backward, optional: this pass's per-tile sums of the output layer's
  int64_t out_pitch;                      // gradient, [tiles][out_pitch]: [0] g^T h_L | sum g, [1] column sums of e_L | 0
};

struct ChainArgs {
  ChainPass pass[kChainPasses];
  int32_t first_block[kChainPasses + 1];
  int32_t count;
  float neg_slope, dropout_p;
  const uint64_t* seed_ptr;
};

__device__ __forceinline__ int64_t chain_align(int64_t v) { return (v + 63) / 64 * 64; }

// Writes of this workgroup to global memory become visible to all of its waves: the barrier carries a
// workgroup-scope release / acquire (the waves of a workgroup share their compute unit's vector cache), which is
// all a chain needs; a device-scope fence here would flush and invalidate caches a dozen times per workgroup.
__device__ __forceinline__ void chain_sync() { __syncthreads(); }

// out[rows of the tile][N] = in[.., K] W[N, K]^T, then either Dropout(LeakyReLU(. + bias)) (gate == null) or the
// gate of the backward chains (no bias)
template <int PREC>
__device__ __forceinline__ void chain_fwd_layer(const float* in, int64_t in_ld, const float* W, int K, int N, int B, int row0,
                                            const float* bias, float neg_slope, float drop_p, uint64_t seed,
                                            const uint64_t* seed_ptr, const float* gate, float pos_scale,
                                            float neg_scale, float* out, float* lds) {
  using C = Core<kChainRows, 128, true, true, true, false, 1, PREC>;
  Operand a, b;
  a.p = in; a.ld = in_ld; a.rows = B; a.red = K; a.vec_ok = 1;
  b.p = W; b.ld = K; b.rows = N; b.red = K; b.vec_ok = 1;
  Tail t;
  if (gate != nullptr) {
    t.residual = gate;
    t.res_ld = N;
    t.residual_is_mask = true;
    t.mask_scale = pos_scale;
    t.mask_neg_scale = neg_scale;
  } else {
    t.bias = bias;
    t.neg_slope = neg_slope;
    t.drop_p = drop_p;
    t.seed = seed;
    t.seed_ptr = seed_ptr;
  }
  for (int col0 = 0; col0 < N; col0 += 128) {
    f32x16 acc0, acc1;
    const float2 bias2 = C::bias_pair(t, col0, N);
    C::accumulate(a, b, row0, col0, 0, K, lds, acc0, acc1);
    C::template store_direct<false>(acc0, acc1, row0, col0, B, N, out, N, t, bias2);
  }
}

// out[rows of the tile][K] = (in[.., N] W[N, K]) gated by `gate` (null: no gate)
template <int PREC>
__device__ __forceinline__ void chain_bwd_layer(const float* in, int N, const float* W, int K, int B, int row0,
                                            const float* gate, float pos_scale, float neg_scale, float* out,
                                            float* lds) {
  using C = Core<kChainRows, 128, true, false, true, false, 1, PREC>;
  Operand a, b;
  a.p = in; a.ld = N; a.rows = B; a.red = N; a.vec_ok = 1;
  b.p = W; b.ld = K; b.rows = K; b.red = N; b.vec_ok = 1;
  Tail t;
  if (gate != nullptr) {
    t.residual = gate;
    t.res_ld = K;
    t.residual_is_mask = true;
    t.mask_scale = pos_scale;
    t.mask_neg_scale = neg_scale;
  }
  for (int col0 = 0; col0 < K; col0 += 128) {
    f32x16 acc0, acc1;
    C::accumulate(a, b, row0, col0, 0, N, lds, acc0, acc1);
    C::template store_direct<false>(acc0, acc1, row0, col0, B, K, out, K, t, make_float2(0.0f, 0.0f));
  }
}

// out[r][c] = (row ? row[r] : 1) * vec[c] * act_factor(h[r][c]) for the tile's rows
__device__ __forceinline__ void chain_gate_init(const float* row, const float* vec, const float* h, int W, int B,
                                                int row0, float keep_scale, float neg_slope, float* out) {
  for (int i = threadIdx.x; i < kChainRows * W; i += kThreads) {
    const int r = row0 + i / W, c = i % W;
    if (r >= B) break;
    const int64_t at = (int64_t)r * W + c;
    float v = act_factor(h[at], keep_scale, neg_slope) * vec[c];
    if (row != nullptr) v *= row[r];
    out[at] = v;
  }
}

// part[c] = sum over the tile's rows r of (row ? row[r] : 1) * m[r][c], c < W; part[W] = sum of row[r] (0 without
// `row`).  A wave per row, rows in ascending order, the four waves' sums added in wave order.
__device__ __forceinline__ void chain_column_sums(const float* row, const float* m, int W, int B, int row0, float* part,
                                                  float* lds) {
  static_assert(kThreads == 256 && kChainRows == 64, "four waves over 64 rows");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int end = min(B, row0 + kChainRows);
  for (int cb = 0; cb < W; cb += 64) {
    const int c = cb + lane;
    float s = 0.0f;
    if (c < W)
      for (int r = row0 + wave; r < end; r += 4) s += (row != nullptr ? row[r] : 1.0f) * m[(int64_t)r * W + c];
    lds[wave * 64 + lane] = s;
    __syncthreads();
    if (wave == 0 && c < W) part[c] = ((lds[lane] + lds[64 + lane]) + lds[128 + lane]) + lds[192 + lane];
    __syncthreads();
  }
  if (wave == 0) {
    float s = (row != nullptr && row0 + lane < end) ? row[row0 + lane] : 0.0f;
    s = wave_sum(s);
    if (lane == 0) part[W] = s;
  }
}

template <int PF, int PD>
__global__ __launch_bounds__(kThreads) void critic_chain_kernel(const ChainArgs g, int backward) {
  using CF = Core<kChainRows, 128, true, true, true, false, 1, PF>;
  using CD = Core<kChainRows, 128, true, false, true, false, 1, PD>;
  constexpr int LDS = CF::LDS_FLOATS > CD::LDS_FLOATS ? CF::LDS_FLOATS : CD::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS];
  int pi = 0;
  while (pi + 1 < g.count && g.first_block[pi + 1] <= (int)blockIdx.x) ++pi;
  const ChainPass& p = g.pass[pi];
  const int row0 = ((int)blockIdx.x - g.first_block[pi]) * kChainRows;
  const int B = p.B, L = p.L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float keep_scale = 1.0f / (1.0f - g.dropout_p);
  const float* w_o = p.weight[L];

  if (!backward) {
    const float* in = p.x;
    int64_t in_ld = p.x_ld;
    for (int l = 1; l <= L; ++l) {  // h_l = Dropout(LeakyReLU(h_{l-1} W_l^T + b_l)); every layer draws its own stream
      float* h = p.saved + p.h_off[l];
      chain_fwd_layer<PF>(in, in_ld, p.weight[l - 1], p.w[l - 1], p.w[l], B, row0, p.bias[l - 1], g.neg_slope, g.dropout_p,
                          p.seed + 0x51ED27ull * (uint64_t)l, g.seed_ptr, nullptr, 0.0f, 0.0f, h, lds);
      chain_sync();
      in = h;
      in_ld = p.w[l];
    }
    // D = h_L . w_o + b_o: a wave per row
    const int WL = p.w[L];
    const float b_o = p.bias[L][0];
    for (int r = row0 + wave; r < min(B, row0 + kChainRows); r += kThreads / 64) {
      float s = 0.0f;
      for (int c = lane; c < WL; c += 64) s += in[(int64_t)r * WL + c] * w_o[c];
      s = wave_sum(s);
      if (lane == 0) p.scores[r] = s + b_o;
    }
    if (p.grad_norm2 == nullptr) return;
    // input gradient: G_L = w_o * c_L, g_{l-1} = G_l W_l, G_{l-1} = g_{l-1} * c_{l-1}
    chain_gate_init(nullptr, w_o, p.saved + p.h_off[L], WL, B, row0, keep_scale, g.neg_slope, p.saved + p.g_off[L]);
    chain_sync();
    for (int l = L; l >= 1; --l) {
      float* out = l > 1 ? p.saved + p.g_off[l - 1] : p.saved + p.g0_off;
      chain_bwd_layer<PD>(p.saved + p.g_off[l], p.w[l], p.weight[l - 1], p.w[l - 1], B, row0,
                          l > 1 ? p.saved + p.h_off[l - 1] : nullptr, keep_scale, keep_scale * g.neg_slope, out, lds);
      chain_sync();
    }
    const int W0 = p.w[0];
    const float* g0 = p.saved + p.g0_off;
    for (int r = row0 + wave; r < min(B, row0 + kChainRows); r += kThreads / 64) {
      float s = 0.0f;
      for (int c = lane; c < W0; c += 64) {
        const float v = g0[(int64_t)r * W0 + c];
        s += v * v;
      }
      s = wave_sum(s);
      if (lane == 0) p.grad_norm2[r] = s;
    }
    return;
  }

  // ---- backward ----
  const int64_t tile = (int)blockIdx.x - g.first_block[pi];
  if (p.g_scores != nullptr) {
    // the output layer's own gradient, this tile's share: dw_o = g^T h_L, db_o = sum g
    if (p.out_part[0] != nullptr)
      chain_column_sums(p.g_scores, p.saved + p.h_off[L], p.w[L], B, row0, p.out_part[0] + tile * p.out_pitch, lds);
    // dz_L = (g w_o) * c_L, then dz_{l-1} = (dz_l W_l) * c_{l-1}
    chain_gate_init(p.g_scores, w_o, p.saved + p.h_off[L], p.w[L], B, row0, keep_scale, g.neg_slope, p.dz[L]);
    chain_sync();
    for (int l = L; l >= 1; --l) {
      if (l > 1) {
        chain_bwd_layer<PD>(p.dz[l], p.w[l], p.weight[l - 1], p.w[l - 1], B, row0, p.saved + p.h_off[l - 1], keep_scale,
                            keep_scale * g.neg_slope, p.dz[l - 1], lds);
        chain_sync();
      } else if (p.dx != nullptr) {
        chain_bwd_layer<PD>(p.dz[l], p.w[l], p.weight[l - 1], p.w[l - 1], B, row0, nullptr, 0.0f, 0.0f, p.dx, lds);
      }
    }
  }
  if (p.g_norm2 != nullptr) {
    // e_0 = 2 g_norm2 g_0, then e_l = (e_{l-1} W_l^T) * c_l
    const int W0 = p.w[0];
    const float* g0 = p.saved + p.g0_off;
    for (int i = threadIdx.x; i < kChainRows * W0; i += kThreads) {
      const int r = row0 + i / W0, c = i % W0;
      if (r >= B) break;
      p.e[0][(int64_t)r * W0 + c] = 2.0f * p.g_norm2[r] * g0[(int64_t)r * W0 + c];
    }
    chain_sync();
    for (int l = 1; l <= L; ++l) {
      chain_fwd_layer<PF>(p.e[l - 1], p.w[l - 1], p.weight[l - 1], p.w[l - 1], p.w[l], B, row0, nullptr, 1.0f, 0.0f, 0, nullptr,
                          p.saved + p.h_off[l], keep_scale, keep_scale * g.neg_slope, p.e[l], lds);
      chain_sync();
    }
    // the penalty's share of dw_o: the column sums of e_L (no bias term)
    if (p.out_part[1] != nullptr)
      chain_column_sums(nullptr, p.e[L], p.w[L], B, row0, p.out_part[1] + tile * p.out_pitch, lds);
  }
}

// Whether a pass can go through the chain kernel: widths and strides that allow 16-byte loads everywhere.
bool chain_ok(const Shape& s, const float* x, int64_t x_ld, const float* const* w, const float* saved) {
  if (s.L > kChainLayers || s.B <= 0) return false;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al(x) || x_ld % 4 != 0 || !al(saved)) return false;
  for (int l = 0; l <= s.L; ++l)
    if (s.w[l] % 4 != 0 || !al(w[l])) return false;
  return true;
}

template <int PF, int PD>
void launch_chain(const ChainArgs& a, int backward, hipStream_t st) {
  hipLaunchKernelGGL((critic_chain_kernel<PF, PD>), dim3((unsigned)a.first_block[a.count]), dim3(kThreads), 0, st, a,
                     backward);
}

void launch_chain_for_mask(const ChainArgs& a, int backward, hipStream_t st) {
  const int mask = matmul_bf16_mask();
  const int pf = (mask & HIVT_BF16_FWD_SPLIT) ? PREC_SPLIT : ((mask & HIVT_BF16_FWD) ? PREC_BF16 : PREC_F32);
  const bool pd = (mask & HIVT_BF16_DGRAD) != 0;
  if (pf == PREC_SPLIT && pd) launch_chain<PREC_SPLIT, PREC_BF16>(a, backward, st);
  else if (pf == PREC_SPLIT) launch_chain<PREC_SPLIT, PREC_F32>(a, backward, st);
  else if (pf == PREC_BF16 && pd) launch_chain<PREC_BF16, PREC_BF16>(a, backward, st);
  else if (pf == PREC_BF16) launch_chain<PREC_BF16, PREC_F32>(a, backward, st);
  else if (pd) launch_chain<PREC_F32, PREC_BF16>(a, backward, st);
  else launch_chain<PREC_F32, PREC_F32>(a, backward, st);
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

extern "C" int hivt_hinge_r1_multi_fwd(const float* const* s_real, const float* const* grad_norm2,
                                       const float* const* s_fake, int32_t scales, const float* w, const float* count,
                                       int64_t B, float lambda_r1, float* out, hivt_stream_t stream) {
  const char* what = "hivt_hinge_r1_multi_fwd";
  HIVT_REQUIRE(B >= 0 && scales >= 1 && scales <= kMaxScales, HIVT_ERR_ARG, "%s: 1..%d scales", what, kMaxScales);
  HIVT_REQUIRE(out && s_real && grad_norm2 && s_fake && (B == 0 || w), HIVT_ERR_ARG, "%s: null pointer", what);
  ScaleRows p{};
  p.count = scales;
  for (int k = 0; k < scales; ++k) {
    HIVT_REQUIRE(B == 0 || (s_real[k] && grad_norm2[k] && s_fake[k]), HIVT_ERR_ARG, "%s: null rows (scale %d)", what, k);
    p.s_real[k] = s_real[k]; p.grad_norm2[k] = grad_norm2[k]; p.s_fake[k] = s_fake[k];
  }
  hipLaunchKernelGGL(hinge_r1_multi_fwd_kernel, dim3(1), dim3(kLossThreads), 0, (hipStream_t)stream, p, w, count, B,
                     lambda_r1, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_hinge_r1_multi_bwd(const float* g_loss, const float* const* s_real, const float* const* s_fake,
                                       int32_t scales, const float* w, const float* count, int64_t B, float lambda_r1,
                                       float* const* g_real, float* const* g_norm2, float* const* g_fake,
                                       hivt_stream_t stream) {
  const char* what = "hivt_hinge_r1_multi_bwd";
  HIVT_REQUIRE(B >= 0 && scales >= 1 && scales <= kMaxScales, HIVT_ERR_ARG, "%s: 1..%d scales", what, kMaxScales);
  if (B == 0) return HIVT_OK;
  HIVT_REQUIRE(g_loss && s_real && s_fake && w && count && g_real && g_norm2 && g_fake, HIVT_ERR_ARG, "%s: null pointer",
               what);
  ScaleRows p{};
  p.count = scales;
  for (int k = 0; k < scales; ++k) {
    HIVT_REQUIRE(s_real[k] && s_fake[k] && g_real[k] && g_norm2[k] && g_fake[k], HIVT_ERR_ARG, "%s: null rows (scale %d)",
                 what, k);
    p.s_real[k] = s_real[k]; p.s_fake[k] = s_fake[k];
    p.g_real[k] = g_real[k]; p.g_norm2[k] = g_norm2[k]; p.g_fake[k] = g_fake[k];
  }
  hipLaunchKernelGGL(hinge_r1_multi_bwd_kernel, dim3((unsigned)((B + 255) / 256), (unsigned)scales), dim3(256), 0,
                     (hipStream_t)stream, p, g_loss, w, count, B, lambda_r1);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_generator_total_fwd(const float* losses, const float* const* s_fake, int32_t scales, const float* w,
                                        const float* count, int64_t B, float lambda_adv, float* out,
                                        hivt_stream_t stream) {
  const char* what = "hivt_generator_total_fwd";
  HIVT_REQUIRE(B >= 0 && scales >= 1 && scales <= kMaxScales, HIVT_ERR_ARG, "%s: 1..%d scales", what, kMaxScales);
  HIVT_REQUIRE(losses && out && s_fake && (B == 0 || w), HIVT_ERR_ARG, "%s: null pointer", what);
  ScaleRows p{};
  p.count = scales;
  for (int k = 0; k < scales; ++k) {
    HIVT_REQUIRE(B == 0 || s_fake[k], HIVT_ERR_ARG, "%s: null rows (scale %d)", what, k);
    p.s_fake[k] = s_fake[k];
  }
  hipLaunchKernelGGL(generator_total_fwd_kernel, dim3(1), dim3(kLossThreads), 0, (hipStream_t)stream, p, losses, w, count,
                     B, lambda_adv, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_generator_total_bwd(const float* g_out, int32_t scales, const float* w, const float* count, int64_t B,
                                        float lambda_adv, float* g_losses, float* const* g_s, hivt_stream_t stream) {
  const char* what = "hivt_generator_total_bwd";
  HIVT_REQUIRE(B >= 0 && scales >= 1 && scales <= kMaxScales, HIVT_ERR_ARG, "%s: 1..%d scales", what, kMaxScales);
  HIVT_REQUIRE(g_out && count && g_losses && g_s && (B == 0 || w), HIVT_ERR_ARG, "%s: null pointer", what);
  ScaleRows p{};
  p.count = scales;
  for (int k = 0; k < scales; ++k) {
    HIVT_REQUIRE(B == 0 || g_s[k], HIVT_ERR_ARG, "%s: null rows (scale %d)", what, k);
    p.g_fake[k] = g_s[k];
  }
  const unsigned blocks = (unsigned)((B + 255) / 256 > 0 ? (B + 255) / 256 : 1);
  hipLaunchKernelGGL(generator_total_bwd_kernel, dim3(blocks, (unsigned)scales), dim3(256), 0, (hipStream_t)stream, p,
                     g_out, w, count, B, lambda_adv, g_losses);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

// ---------------------------------------------------------------------------------------------
// Several passes per call (all critics, real and generated trajectories): one chain launch for up to six passes.
// ---------------------------------------------------------------------------------------------
namespace {

struct FwdPass {
  const float* x; int64_t x_ld; int64_t B; int32_t layers; const int32_t* widths; const float* const* w;
  const float* const* bias; uint64_t seed; int32_t with_input_grad; float* saved; float* scores; float* grad_norm2;
};

struct BwdPass {
  const float* g_scores; const float* g_norm2; const float* x; int64_t x_ld; int64_t B; int32_t layers;
  const int32_t* widths; const float* const* w; const float* saved; float* const* dw; float* const* dbias; float* dx;
};

template <class T> T* as_ptr(int64_t v) { return reinterpret_cast<T*>((uintptr_t)v); }

size_t chain_act_bytes(const Shape& s) {
  size_t total = 0;
  for (int l = 1; l <= s.L; ++l) total += act_bytes(s, l);
  for (int l = 0; l <= s.L; ++l) total += act_bytes(s, l);
  return total;
}

void fill_common(ChainPass& c, const Shape& s, const float* x, int64_t x_ld, const float* const* w, float* saved) {
  c.x = x; c.x_ld = x_ld; c.B = (int32_t)s.B; c.L = s.L;
  for (int l = 0; l <= s.L; ++l) {
    c.w[l] = s.w[l];
    c.weight[l] = w[l];
  }
  c.saved = saved;
  for (int l = 1; l <= s.L; ++l) {
    c.h_off[l] = s.h_off[l];
    c.g_off[l] = s.g_off[l];
  }
  c.g0_off = s.g0_off;
}

}  // namespace

extern "C" int hivt_critics_fwd(const int64_t* passes, int32_t count, float neg_slope, float dropout_p,
                                const uint64_t* seed_ptr, hivt_stream_t stream) {
  const char* what = "hivt_critics_fwd";
  HIVT_REQUIRE(count >= 0 && (count == 0 || passes != nullptr), HIVT_ERR_ARG, "%s: bad pass table", what);
  HIVT_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, HIVT_ERR_ARG, "%s: dropout_p must be in [0,1)", what);
  hipStream_t st = (hipStream_t)stream;
  ChainArgs a{};
  a.neg_slope = neg_slope; a.dropout_p = dropout_p; a.seed_ptr = seed_ptr;
  auto flush = [&]() {
    if (a.count > 0) launch_chain_for_mask(a, 0, st);
    a.count = 0;
    a.first_block[0] = 0;
  };
  for (int32_t i = 0; i < count; ++i) {
    const int64_t* t = passes + (size_t)i * 12;
    FwdPass f{as_ptr<const float>(t[0]), t[1], t[2], (int32_t)t[3], as_ptr<const int32_t>(t[4]),
              as_ptr<const float* const>(t[5]), as_ptr<const float* const>(t[6]), (uint64_t)t[7], (int32_t)t[8],
              as_ptr<float>(t[9]), as_ptr<float>(t[10]), as_ptr<float>(t[11])};
    Shape s;
    HIVT_TRY(make_shape(what, f.B, f.layers, f.widths, s));
    if (f.B == 0) continue;
    HIVT_REQUIRE(f.x && f.w && f.bias && f.saved && f.scores, HIVT_ERR_ARG, "%s: pass %d: null pointer", what, (int)i);
    HIVT_REQUIRE(!f.with_input_grad || f.grad_norm2, HIVT_ERR_ARG, "%s: pass %d: grad_norm2 is null", what, (int)i);
    HIVT_REQUIRE(f.x_ld >= s.w[0], HIVT_ERR_ARG, "%s: pass %d: x_ld smaller than the input width", what, (int)i);
    for (int l = 0; l <= f.layers; ++l)
      HIVT_REQUIRE(f.w[l] && f.bias[l], HIVT_ERR_ARG, "%s: pass %d: null parameter (layer %d)", what, (int)i, l);
    if (!chain_ok(s, f.x, f.x_ld, f.w, f.saved)) {  // the layer-by-layer route
      HIVT_TRY(hivt_critic_fwd(f.x, f.x_ld, f.B, f.layers, f.widths, f.w, f.bias, neg_slope, dropout_p, f.seed, seed_ptr,
                               f.with_input_grad, f.saved, f.scores, f.grad_norm2, stream));
      continue;
    }
    ChainPass& c = a.pass[a.count];
    c = ChainPass{};
    fill_common(c, s, f.x, f.x_ld, f.w, f.saved);
    for (int l = 0; l <= s.L; ++l) c.bias[l] = f.bias[l];
    c.seed = f.seed;
    c.scores = f.scores;
    c.grad_norm2 = f.with_input_grad ? f.grad_norm2 : nullptr;
    a.first_block[a.count + 1] = a.first_block[a.count] + (int32_t)((s.B + kChainRows - 1) / kChainRows);
    if (++a.count == kChainPasses) flush();
  }
  flush();
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" size_t hivt_critics_bwd_workspace_bytes(const int64_t* passes, int32_t count) {
  if (count < 0 || (count > 0 && passes == nullptr)) return 0;
  size_t total = 256;
  for (int32_t i = 0; i < count; ++i) {
    const int64_t* t = passes + (size_t)i * 12;
    Shape s;
    if (make_shape("hivt_critics_bwd_workspace_bytes", t[4], (int32_t)t[5], as_ptr<const int32_t>(t[6]), s) != HIVT_OK)
      return 0;
    total += scratch_bytes(s) + group_bytes(s);  // the pass's own region + its share of the grouped call's workspace
  }
  return total;
}

extern "C" int hivt_critics_bwd(const int64_t* passes, int32_t count, float neg_slope, float dropout_p, void* workspace,
                                size_t workspace_bytes, hivt_stream_t stream) {
  const char* what = "hivt_critics_bwd";
  HIVT_REQUIRE(count >= 0 && (count == 0 || passes != nullptr), HIVT_ERR_ARG, "%s: bad pass table", what);
  HIVT_REQUIRE(workspace_bytes >= hivt_critics_bwd_workspace_bytes(passes, count) && (count == 0 || workspace),
               HIVT_ERR_WORKSPACE, "%s: workspace too small", what);
  hipStream_t st = (hipStream_t)stream;
  char* ws = reinterpret_cast<char*>(workspace);
  struct Done { Shape s; BwdPass b; Scratch k; bool chained, scores_chain, penalty_chain, params; };
  std::vector<Done> done;
  std::vector<char> was_chained((size_t)(count > 0 ? count : 0), 0);
  ChainArgs a{};
  a.neg_slope = neg_slope; a.dropout_p = dropout_p;
  auto flush = [&]() {
    if (a.count > 0) launch_chain_for_mask(a, 1, st);
    a.count = 0;
    a.first_block[0] = 0;
  };
  for (int32_t i = 0; i < count; ++i) {
    const int64_t* t = passes + (size_t)i * 12;
    BwdPass b{as_ptr<const float>(t[0]), as_ptr<const float>(t[1]), as_ptr<const float>(t[2]), t[3], t[4], (int32_t)t[5],
              as_ptr<const int32_t>(t[6]), as_ptr<const float* const>(t[7]), as_ptr<const float>(t[8]),
              as_ptr<float* const>(t[9]), as_ptr<float* const>(t[10]), as_ptr<float>(t[11])};
    Shape s;
    HIVT_TRY(make_shape(what, b.B, b.layers, b.widths, s));
    char* region = ws;
    ws += scratch_bytes(s);
    const bool params = b.dw != nullptr || b.dbias != nullptr;
    const bool trivial = b.B == 0 || (b.g_scores == nullptr && b.g_norm2 == nullptr) || (!params && b.g_scores == nullptr);
    if (trivial || !(b.x && b.w && b.saved) || !chain_ok(s, b.x, b.x_ld, b.w, b.saved)) {
      // nothing to chain (or not chainable): the single-pass entry point covers every corner
      HIVT_TRY(hivt_critic_bwd(b.g_scores, b.g_norm2, b.x, b.x_ld, b.B, b.layers, b.widths, b.w, neg_slope, dropout_p,
                               b.saved, b.dw, b.dbias, b.dx, region, scratch_bytes(s), stream));
      continue;
    }
    HIVT_REQUIRE(!params || (b.dw && b.dbias), HIVT_ERR_ARG, "%s: pass %d: dw and dbias go together", what, (int)i);
    HIVT_REQUIRE(params || (b.g_norm2 == nullptr && b.dx != nullptr), HIVT_ERR_ARG,
                 "%s: pass %d: without parameter gradients the call computes dx from g_scores only", what, (int)i);
    if (params)
      for (int l = 0; l <= s.L; ++l)
        HIVT_REQUIRE(b.dw[l] && b.dbias[l], HIVT_ERR_ARG, "%s: pass %d: null gradient (layer %d)", what, (int)i, l);
    Done d;
    d.s = s; d.b = b; d.k = carve(s, region); d.chained = true; d.params = params;
    d.scores_chain = b.g_scores != nullptr; d.penalty_chain = b.g_norm2 != nullptr;
    if (!d.scores_chain && b.dx != nullptr) zero_words(b.dx, b.B * s.w[0], st);  // dx carries the score chain only
    ChainPass& c = a.pass[a.count];
    c = ChainPass{};
    fill_common(c, s, b.x, b.x_ld, b.w, const_cast<float*>(b.saved));
    c.g_scores = b.g_scores; c.g_norm2 = b.g_norm2; c.dx = b.dx;
    for (int l = 1; l <= s.L; ++l) c.dz[l] = d.k.dz[l];
    for (int l = 0; l <= s.L; ++l) c.e[l] = d.k.e[l];
    if (params) {  // the output layer's gradient leaves the chain as per-tile sums
      c.out_part[0] = d.scores_chain ? d.k.out_part[0] : nullptr;
      c.out_part[1] = d.penalty_chain ? d.k.out_part[1] : nullptr;
      c.out_pitch = out_part_pitch(s);
    }
    a.first_block[a.count + 1] = a.first_block[a.count] + (int32_t)((s.B + kChainRows - 1) / kChainRows);
    done.push_back(d);
    was_chained[(size_t)i] = 1;
    if (++a.count == kChainPasses) flush();
  }
  flush();
  HIVT_CHECK_LAUNCH(what);
  // Passes that were handed the SAME gradient arrays (the passes of one critic over different inputs) sum their
  // parameter gradients: such passes form a set, found by the address of their first weight gradient.
  std::vector<std::vector<size_t>> sets;
  for (size_t i = 0; i < done.size(); ++i) {
    if (!done[i].params) continue;
    bool placed = false;
    for (auto& set : sets)
      if (done[set[0]].b.dw[0] == done[i].b.dw[0]) {
        const Done& f = done[set[0]];
        HIVT_REQUIRE(f.s.L == done[i].s.L, HIVT_ERR_ARG, "%s: passes that share gradients differ in depth", what);
        for (int l = 0; l <= f.s.L; ++l)
          HIVT_REQUIRE(f.b.dw[l] == done[i].b.dw[l] && f.b.dbias[l] == done[i].b.dbias[l] && f.s.w[l] == done[i].s.w[l],
                       HIVT_ERR_ARG, "%s: passes that share gradients must share all of them", what);
        set.push_back(i);
        placed = true;
        break;
      }
    if (!placed) sets.push_back({i});
  }
  // the passes that took the single-pass route wrote their gradients on their own: they cannot be part of a set
  for (int32_t i = 0; i < count; ++i) {
    const int64_t* t = passes + (size_t)i * 12;
    float* const* dw = as_ptr<float* const>(t[9]);
    if (dw == nullptr) continue;
    int sharers = 0;
    for (int32_t j = 0; j < count; ++j) {
      float* const* other = as_ptr<float* const>((passes + (size_t)j * 12)[9]);
      if (other != nullptr && other[0] == dw[0]) ++sharers;
    }
    if (sharers < 2 || t[4] == 0) continue;
    HIVT_REQUIRE(was_chained[(size_t)i] != 0, HIVT_ERR_UNSUPPORTED,
                 "%s: pass %d shares its gradient arrays with another pass but is not built for the layer chain "
                 "(at most %d hidden layers, widths that are multiples of 4, a non-empty batch)", what, (int)i,
                 kChainLayers);
  }
  // the weight gradients of every chained pass in ONE grouped call; its workspace is what lies behind the passes'
  // own regions (hivt_critics_bwd_workspace_bytes reserves every pass's share there).  Per weight: the score-chain
  // terms of the set's passes (the first is the item, the others join it and add their bias sums), then the
  // penalty-chain terms (no bias).
  std::vector<int64_t> table;
  int32_t terms = 0;
  for (const auto& set : sets) {
    const Done& f = done[set[0]];
    const Shape& s = f.s;
    bool any_scores = false;
    for (size_t i : set) any_scores |= done[i].scores_chain;
    if (!any_scores) {  // penalty gradients only: the bias gradients are zero
      for (int l = 1; l <= s.L; ++l) zero_words(f.b.dbias[l - 1], s.w[l], st);
      zero_words(f.b.dbias[s.L], 1, st);
    }
    Terms t;
    auto flush_terms = [&]() {
      table.insert(table.end(), &t.row[0][0], &t.row[0][0] + (size_t)t.count * 10);
      terms += t.count;
      t.count = 0;
    };
    // (output layer: dw_o = g^T h_L + column sums of e_L and db_o = sum g come from the chain's per-tile sums, below)
    for (int l = s.L; l >= 1; --l) {
      bool head = true;
      for (size_t i : set) {
        const Done& d = done[i];
        if (!d.scores_chain) continue;
        const float* h_prev = l > 1 ? d.b.saved + s.h_off[l - 1] : d.b.x;
        const int64_t h_prev_ld = l > 1 ? s.w[l - 1] : d.b.x_ld;
        t.add(d.k.dz[l], s.w[l], h_prev, h_prev_ld, d.b.B, s.w[l], s.w[l - 1], f.b.dw[l - 1], f.b.dbias[l - 1], false, !head);
        if (!head) t.row[t.count - 1][9] |= 8;  // its bias sums count too
        head = false;
      }
      for (size_t i : set) {
        const Done& d = done[i];
        if (!d.penalty_chain) continue;
        t.add(d.b.saved + s.g_off[l], s.w[l], d.k.e[l - 1], s.w[l - 1], d.b.B, s.w[l], s.w[l - 1], f.b.dw[l - 1], nullptr,
              false, !head);
        head = false;
      }
      flush_terms();
    }
  }
  if (terms > 0) {
    const size_t left = workspace_bytes - (size_t)(ws - reinterpret_cast<char*>(workspace));
    HIVT_TRY(hivt_linear_wgrad_group(table.data(), terms, ws, left, stream));
  }
  // the output layers of all sets in one launch: per set the score chains' tiles, then the penalty chains' tiles,
  // four terms per entry (a longer set continues in entries that add to what the ones before them wrote, issued
  // in launches of their own so that they run afterwards)
  std::vector<std::vector<ColumnSums>> waves;
  for (const auto& set : sets) {
    const Done& f = done[set[0]];
    const Shape& s = f.s;
    std::vector<std::pair<const float*, int32_t>> terms_of;
    for (size_t i : set)
      if (done[i].scores_chain) terms_of.push_back({done[i].k.out_part[0], (int32_t)out_part_tiles(done[i].s)});
    for (size_t i : set)
      if (done[i].penalty_chain) terms_of.push_back({done[i].k.out_part[1], (int32_t)out_part_tiles(done[i].s)});
    for (size_t first = 0, wave = 0; first < terms_of.size(); first += 4, ++wave) {
      ColumnSums c{};
      for (size_t t = 0; t < 4 && first + t < terms_of.size(); ++t) {
        c.term[t] = terms_of[first + t].first;
        c.rows[t] = terms_of[first + t].second;
      }
      c.pitch = out_part_pitch(s);
      c.out = f.b.dw[s.L];
      c.out1 = f.b.dbias[s.L];
      c.cols = s.w[s.L] + 1;
      c.split = s.w[s.L];
      c.accumulate = wave > 0 ? 1 : 0;
      if (waves.size() <= wave) waves.resize(wave + 1);
      waves[wave].push_back(c);
    }
  }
  for (const auto& w : waves) HIVT_TRY(reduce_column_sums_group(w.data(), (int)w.size(), st));
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}
