// Causal multi-head self-attention over a SHORT time axis (S <= 32 steps, here 20 history steps +
// 1 class token) for every agent at once: the attention core of the temporal encoder.
//
// One 256-thread block per agent.  The agent's S rows of the fused q|k|v projection ([S, 3D],
// 1.5 KB contiguous per row at D = 128) are staged once into LDS with coalesced 16-byte loads; a
// thread owns one (head, query step) pair, walks the keys 0..i with an online softmax reading K / V
// rows as LDS broadcasts, and the result is written back through LDS as whole 512-byte rows.
// Backward keeps dK / dV per (head, key step) thread and dQ per (head, query step) thread, so no
// atomics and no cross-block reduction are needed.  The kernel is bound by the single pass over
// q|k|v (and dO) in HBM; arithmetic is ~S*S*d*4 flops per head.
#include "common.h"

#include <math.h>

namespace hivt {
namespace {

constexpr int kMaxS = 32;

__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

struct TAttnArgs {
  const float* qkv;   // [S, N, 3D]
  float* out;         // [S, N, D]
  float* lse;         // [N, H, S]  log-sum-exp of the scaled scores per query
  const float* d_out; // backward
  float* d_qkv;       // backward [S, N, 3D]
  int32_t d_qkv_half; // d_qkv points at bfloat16 elements (values rounded to nearest even)
  int32_t S, N, H;
  float scale;
  float dropout_p;
  uint64_t seed;
  const uint64_t* seed_ptr;  // optional device-side seed added to `seed`
};

__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

template <int D, int HD>  // model width, head width
__global__ __launch_bounds__(256) void temporal_attn_fwd_kernel(const TAttnArgs a) {
  constexpr int ROW = 3 * D;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][3D]
  const int n = blockIdx.x;
  const int S = a.S, N = a.N;
  // stage q|k|v rows of this agent
  for (int i = threadIdx.x; i < S * (ROW / 4); i += 256) {
    const int s = i / (ROW / 4), c = (i % (ROW / 4)) * 4;
    *reinterpret_cast<float4*>(lds + s * ROW + c) =
        *reinterpret_cast<const float4*>(a.qkv + ((int64_t)s * N + n) * ROW + c);
  }
  __syncthreads();
  const bool drop = a.dropout_p > 0.0f;
  const float keep_scale = drop ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  const uint64_t seed = a.seed + ((drop && a.seed_ptr) ? *a.seed_ptr : 0ull);
  // (head, query) pairs: H * S of them, 256 threads per pass
  for (int t = threadIdx.x; t < a.H * kMaxS; t += 256) {
    const int h = t / kMaxS, i = t % kMaxS;
    if (i >= S) continue;
    float q[HD], acc[HD];
#pragma unroll
    for (int x = 0; x < HD; ++x) {
      q[x] = lds[i * ROW + h * HD + x] * a.scale;
      acc[x] = 0.0f;
    }
    float m = -1.0e30f, l = 0.0f;
    for (int j = 0; j <= i; ++j) {
      const float* kr = lds + j * ROW + D + h * HD;
      const float* vr = lds + j * ROW + 2 * D + h * HD;
      float s = 0.0f;
#pragma unroll
      for (int x = 0; x < HD; ++x) s += q[x] * kr[x];
      const float mn = fmaxf(m, s);
      const float corr = fexp(m - mn);
      float p = fexp(s - mn);
      l = l * corr + p;
      if (drop) {
        const uint32_t idx = (((uint32_t)n * a.H + h) * kMaxS + i) * kMaxS + j;
        p = uniform01_u32(seed, idx) >= a.dropout_p ? p * keep_scale : 0.0f;
      }
#pragma unroll
      for (int x = 0; x < HD; ++x) acc[x] = acc[x] * corr + p * vr[x];
      m = mn;
    }
    const float inv = 1.0f / l;
    a.lse[((int64_t)n * a.H + h) * S + i] = m + logf(l);
    // the q slot of (i, h) is read by this thread only: reuse it for the output row
#pragma unroll
    for (int x = 0; x < HD; ++x) lds[i * ROW + h * HD + x] = acc[x] * inv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S * (D / 4); i += 256) {
    const int s = i / (D / 4), c = (i % (D / 4)) * 4;
    *reinterpret_cast<float4*>(a.out + ((int64_t)s * N + n) * D + c) = *reinterpret_cast<const float4*>(lds + s * ROW + c);
  }
}

template <int D, int HD>
__global__ __launch_bounds__(256) void temporal_attn_bwd_kernel(const TAttnArgs a) {
  constexpr int ROW = 3 * D;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = blockIdx.x;
  const int S = a.S, N = a.N, H = a.H;
  float* qkv = lds;                 // [S][3D]
  float* go = lds + S * ROW;        // [S][D]   dO
  float* aux = go + S * D;          // [H][kMaxS][2]  lse, delta per (head, query)
  uint32_t* keep_bits = reinterpret_cast<uint32_t*>(aux + H * kMaxS * 2);  // [H][kMaxS]: bit j = score (i, j) kept
  for (int i = threadIdx.x; i < S * (ROW / 4); i += 256) {
    const int s = i / (ROW / 4), c = (i % (ROW / 4)) * 4;
    *reinterpret_cast<float4*>(qkv + s * ROW + c) =
        *reinterpret_cast<const float4*>(a.qkv + ((int64_t)s * N + n) * ROW + c);
  }
  for (int i = threadIdx.x; i < S * (D / 4); i += 256) {
    const int s = i / (D / 4), c = (i % (D / 4)) * 4;
    *reinterpret_cast<float4*>(go + s * D + c) = *reinterpret_cast<const float4*>(a.d_out + ((int64_t)s * N + n) * D + c);
  }
  __syncthreads();
  const bool drop = a.dropout_p > 0.0f;
  const float keep_scale = drop ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  const uint64_t seed = a.seed + ((drop && a.seed_ptr) ? *a.seed_ptr : 0ull);
  for (int t = threadIdx.x; t < H * kMaxS; t += 256) {
    const int h = t / kMaxS, i = t % kMaxS;
    if (i >= S) continue;
    // delta = dO . O over the head's slice; O is read once, straight from global memory (keeping it out
    // of LDS is what lets a third block share the compute unit)
    float d = 0.0f;
    const float* orow = a.out + ((int64_t)i * N + n) * D + h * HD;
#pragma unroll
    for (int x = 0; x < HD; x += 4) {
      const float4 o4 = *reinterpret_cast<const float4*>(orow + x);
      d += go[i * D + h * HD + x] * o4.x + go[i * D + h * HD + x + 1] * o4.y + go[i * D + h * HD + x + 2] * o4.z +
           go[i * D + h * HD + x + 3] * o4.w;
    }
    aux[(h * kMaxS + i) * 2 + 0] = a.lse[((int64_t)n * H + h) * S + i];
    aux[(h * kMaxS + i) * 2 + 1] = d;
    // the forward's dropout decisions of query i, drawn once and shared by both roles below
    uint32_t bits = 0xffffffffu;
    if (drop) {
      bits = 0u;
      for (int j = 0; j <= i; ++j) {
        const uint32_t idx = (((uint32_t)n * H + h) * kMaxS + i) * kMaxS + j;
        bits |= (uniform01_u32(seed, idx) >= a.dropout_p ? 1u : 0u) << j;
      }
    }
    keep_bits[h * kMaxS + i] = bits;
  }
  __syncthreads();
  // Every thread owns index r of head h twice: as a QUERY (dQ_r) and as a KEY (dK_r, dV_r).
  // A pass covers 256 / kMaxS whole heads; all threads run every pass so that the barrier between
  // the last read and the in-place overwrite of a pass's q|k|v slots is reached by everyone.
  const int passes = (H * kMaxS + 255) / 256;
  for (int pass = 0; pass < passes; ++pass) {
    const int t = pass * 256 + threadIdx.x;
    const int h = t / kMaxS, r = t % kMaxS;
    const bool active = h < H && r < S;
    float dq[HD], dk[HD], dv[HD];
#pragma unroll
    for (int x = 0; x < HD; ++x) dq[x] = dk[x] = dv[x] = 0.0f;
    if (active) {
      float qr[HD], kr[HD], vr[HD], gr[HD];
#pragma unroll
      for (int x = 0; x < HD; ++x) {
        qr[x] = qkv[r * ROW + h * HD + x];
        kr[x] = qkv[r * ROW + D + h * HD + x];
        vr[x] = qkv[r * ROW + 2 * D + h * HD + x];
        gr[x] = go[r * D + h * HD + x];
      }
      // as query r over keys j <= r
      const float lse_r = aux[(h * kMaxS + r) * 2], delta_r = aux[(h * kMaxS + r) * 2 + 1];
      const uint32_t bits_r = keep_bits[h * kMaxS + r];
      for (int j = 0; j <= r; ++j) {
        const float* kj = qkv + j * ROW + D + h * HD;
        const float* vj = qkv + j * ROW + 2 * D + h * HD;
        float s = 0.0f, dp = 0.0f;
#pragma unroll
        for (int x = 0; x < HD; ++x) {
          s += qr[x] * kj[x];
          dp += gr[x] * vj[x];
        }
        const float p = fexp(s * a.scale - lse_r);
        const float ms = ((bits_r >> j) & 1u) ? keep_scale : 0.0f;
        const float ds = p * (dp * ms - delta_r) * a.scale;
#pragma unroll
        for (int x = 0; x < HD; ++x) dq[x] += ds * kj[x];
      }
      // as key r over queries i >= r.  Every lane walks ALL queries in the same order and masks the
      // ones before its key: the lanes of a head then read one LDS address per step (a broadcast),
      // where per-lane start rows would hit one bank S ways (row pitch 3D is a multiple of 64 words).
      for (int i = 0; i < S; ++i) {
        if (i < r) continue;
        const float* qi = qkv + i * ROW + h * HD;
        const float* gi = go + i * D + h * HD;
        float s = 0.0f, dp = 0.0f;
#pragma unroll
        for (int x = 0; x < HD; ++x) {
          s += qi[x] * kr[x];
          dp += gi[x] * vr[x];
        }
        const float p = fexp(s * a.scale - aux[(h * kMaxS + i) * 2]);
        const float ms = ((keep_bits[h * kMaxS + i] >> r) & 1u) ? keep_scale : 0.0f;
        const float ds = p * (dp * ms - aux[(h * kMaxS + i) * 2 + 1]) * a.scale;
        const float pm = p * ms;
#pragma unroll
        for (int x = 0; x < HD; ++x) {
          dk[x] += ds * qi[x];
          dv[x] += pm * gi[x];
        }
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int x = 0; x < HD; ++x) {
        qkv[r * ROW + h * HD + x] = dq[x];
        qkv[r * ROW + D + h * HD + x] = dk[x];
        qkv[r * ROW + 2 * D + h * HD + x] = dv[x];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S * (ROW / 4); i += 256) {
    const int s = i / (ROW / 4), c = (i % (ROW / 4)) * 4;
    const float4 g = *reinterpret_cast<const float4*>(qkv + s * ROW + c);
    const int64_t at = ((int64_t)s * N + n) * ROW + c;
    if (a.d_qkv_half) {
      uint2 h;
      h.x = (unsigned)bf16_bits(g.x) | ((unsigned)bf16_bits(g.y) << 16);
      h.y = (unsigned)bf16_bits(g.z) | ((unsigned)bf16_bits(g.w) << 16);
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.d_qkv) + at) = h;
    } else {
      *reinterpret_cast<float4*>(a.d_qkv + at) = g;
    }
  }
}

template <int D, int HD>
int launch(bool backward, const TAttnArgs& a, hipStream_t stream) {
  const size_t fwd_lds = (size_t)a.S * 3 * D * sizeof(float);
  const size_t bwd_lds = ((size_t)a.S * 4 * D + (size_t)a.H * kMaxS * 3) * sizeof(float);
  const size_t need = backward ? bwd_lds : fwd_lds;
  if (need > 160 * 1024) return -1;
  if (need > 48 * 1024) {
    // dynamic LDS beyond the default window has to be granted per kernel
    hipError_t e = backward ? hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_bwd_kernel<D, HD>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)need)
                            : hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_fwd_kernel<D, HD>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
    if (e != hipSuccess) return -1;
  }
  if (backward)
    hipLaunchKernelGGL((temporal_attn_bwd_kernel<D, HD>), dim3((unsigned)a.N), dim3(256), bwd_lds, stream, a);
  else
    hipLaunchKernelGGL((temporal_attn_fwd_kernel<D, HD>), dim3((unsigned)a.N), dim3(256), fwd_lds, stream, a);
  return 0;
}

int dispatch(int D, int H, bool backward, const TAttnArgs& a, hipStream_t stream) {
  const int hd = D / H;
  if (D == 128 && hd == 16) return launch<128, 16>(backward, a, stream);
  if (D == 64 && hd == 8) return launch<64, 8>(backward, a, stream);
  if (D == 32 && hd == 4) return launch<32, 4>(backward, a, stream);
  if (D == 16 && hd == 8) return launch<16, 8>(backward, a, stream);
  if (D == 256 && hd == 32) return launch<256, 32>(backward, a, stream);
  return -1;
}

int check(const char* what, int32_t S, int32_t N, int32_t D, int32_t H, float p) {
  HIVT_REQUIRE(S >= 1 && S <= kMaxS, HIVT_ERR_ARG, "%s: sequence length %d not in [1,%d]", what, S, kMaxS);
  HIVT_REQUIRE(N >= 0 && H >= 1 && D % H == 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  HIVT_REQUIRE(H * kMaxS % 256 == 0 || H * kMaxS <= 256, HIVT_ERR_ARG, "%s: unsupported head count %d", what, H);
  HIVT_REQUIRE(p >= 0.0f && p < 1.0f, HIVT_ERR_ARG, "%s: dropout_p must be in [0,1)", what);
  HIVT_REQUIRE((int64_t)N * H * kMaxS * kMaxS < ((int64_t)1 << 32), HIVT_ERR_ARG,
               "%s: %d agents x %d heads exceed the 32-bit dropout index", what, N, H);
  return HIVT_OK;
}

}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" int hivt_temporal_attn_fwd(const float* qkv, int32_t S, int32_t N, int32_t D, int32_t H, float dropout_p,
                                      uint64_t seed, const uint64_t* seed_ptr, float* out, float* lse,
                                      hivt_stream_t stream) {
  const char* what = "hivt_temporal_attn_fwd";
  int rc = check(what, S, N, D, H, dropout_p);
  if (rc != HIVT_OK) return rc;
  if (N == 0) return HIVT_OK;
  HIVT_REQUIRE(qkv && out && lse, HIVT_ERR_ARG, "%s: null pointer", what);
  TAttnArgs a{};
  a.qkv = qkv; a.out = out; a.lse = lse; a.S = S; a.N = N; a.H = H;
  a.scale = 1.0f / sqrtf((float)(D / H)); a.dropout_p = dropout_p; a.seed = seed; a.seed_ptr = seed_ptr;
  HIVT_REQUIRE(dispatch(D, H, false, a, (hipStream_t)stream) == 0, HIVT_ERR_ARG, "%s: no kernel for D=%d H=%d", what,
               D, H);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

static int temporal_attn_bwd_impl(const float* d_out, const float* qkv, const float* out, const float* lse, int32_t S,
                                  int32_t N, int32_t D, int32_t H, float dropout_p, uint64_t seed,
                                  const uint64_t* seed_ptr, float* d_qkv, bool d_qkv_half, hivt_stream_t stream) {
  const char* what = "hivt_temporal_attn_bwd";
  int rc = check(what, S, N, D, H, dropout_p);
  if (rc != HIVT_OK) return rc;
  if (N == 0) return HIVT_OK;
  HIVT_REQUIRE(d_out && qkv && out && lse && d_qkv, HIVT_ERR_ARG, "%s: null pointer", what);
  TAttnArgs a{};
  a.qkv = qkv; a.out = const_cast<float*>(out); a.lse = const_cast<float*>(lse); a.d_out = d_out; a.d_qkv = d_qkv;
  a.d_qkv_half = d_qkv_half ? 1 : 0;
  a.S = S; a.N = N; a.H = H;
  a.scale = 1.0f / sqrtf((float)(D / H)); a.dropout_p = dropout_p; a.seed = seed; a.seed_ptr = seed_ptr;
  HIVT_REQUIRE(dispatch(D, H, true, a, (hipStream_t)stream) == 0, HIVT_ERR_ARG, "%s: no kernel for D=%d H=%d", what, D,
               H);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_temporal_attn_bwd(const float* d_out, const float* qkv, const float* out, const float* lse,
                                      int32_t S, int32_t N, int32_t D, int32_t H, float dropout_p, uint64_t seed,
                                      const uint64_t* seed_ptr, float* d_qkv, hivt_stream_t stream) {
  return temporal_attn_bwd_impl(d_out, qkv, out, lse, S, N, D, H, dropout_p, seed, seed_ptr, d_qkv, false, stream);
}

extern "C" int hivt_temporal_attn_bwd_half(const float* d_out, const float* qkv, const float* out, const float* lse,
                                           int32_t S, int32_t N, int32_t D, int32_t H, float dropout_p, uint64_t seed,
                                           const uint64_t* seed_ptr, void* d_qkv, hivt_stream_t stream) {
  return temporal_attn_bwd_impl(d_out, qkv, out, lse, S, N, D, H, dropout_p, seed, seed_ptr,
                                reinterpret_cast<float*>(d_qkv), true, stream);
}
