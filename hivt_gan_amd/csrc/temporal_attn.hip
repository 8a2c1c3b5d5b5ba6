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

// Rows of one agent, `rows` of them `stride` floats apart in memory and W floats long, into LDS rows of pitch P.  All
// loads are issued before the first LDS store (a loop of load / store pairs would wait for every load in turn: a
// dozen memory latencies in a row per workgroup).  Loads beyond the last row re-read the last element (their
// registers are never stored), which keeps every buffer register defined.
typedef float tf32x4 __attribute__((ext_vector_type(4)));

template <int W, int ITER>
__device__ __forceinline__ void load_rows(const float* __restrict__ src, int64_t stride, int rows, tf32x4 (&v)[ITER]) {
  constexpr int PER = W / 4;
  const int last = rows * PER - 1;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int i = min((int)threadIdx.x + it * 256, last);
    v[it] = *reinterpret_cast<const tf32x4*>(src + (int64_t)(i / PER) * stride + (i % PER) * 4);
  }
}

template <int W, int P, int ITER>
__device__ __forceinline__ void store_rows(const tf32x4 (&v)[ITER], int rows, float* dst) {
  constexpr int PER = W / 4;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int i = threadIdx.x + it * 256;
    if (i < rows * PER) *reinterpret_cast<tf32x4*>(dst + (i / PER) * P + (i % PER) * 4) = v[it];
  }
}

template <int W, int P>
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int64_t stride, int rows, float* dst) {
  constexpr int ITER = (kMaxS * (W / 4) + 255) / 256;
  tf32x4 v[ITER];
  load_rows<W, ITER>(src, stride, rows, v);
  store_rows<W, P, ITER>(v, rows, dst);
}

// Two such row sets, all loads of both first.
template <int W0, int P0, int W1, int P1>
__device__ __forceinline__ void stage_rows2(const float* __restrict__ src0, int64_t stride0, float* dst0,
                                            const float* __restrict__ src1, int64_t stride1, float* dst1, int rows) {
  constexpr int ITER0 = (kMaxS * (W0 / 4) + 255) / 256, ITER1 = (kMaxS * (W1 / 4) + 255) / 256;
  tf32x4 v0[ITER0], v1[ITER1];
  load_rows<W0, ITER0>(src0, stride0, rows, v0);
  load_rows<W1, ITER1>(src1, stride1, rows, v1);
  store_rows<W0, P0, ITER0>(v0, rows, dst0);
  store_rows<W1, P1, ITER1>(v1, rows, dst1);
}


template <int D, int HD>  // model width, head width
__global__ __launch_bounds__(256) void temporal_attn_fwd_kernel(const TAttnArgs a) {
  constexpr int ROW = 3 * D;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][3D]
  const int n = blockIdx.x;
  const int S = a.S, N = a.N;
  // stage q|k|v rows of this agent
  stage_rows<ROW, ROW>(a.qkv + (int64_t)n * ROW, (int64_t)N * ROW, S, lds);
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
  stage_rows2<ROW, ROW, D, D>(a.qkv + (int64_t)n * ROW, (int64_t)N * ROW, qkv, a.d_out + (int64_t)n * D, (int64_t)N * D, go, S);
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


// ---------------------------------------------------------------------------------------------
// The same attention on the matrix cores, for D = 128 / 16-wide heads while the dense products run on split bfloat16
// operands.  One workgroup per agent as above, but a WAVE per head: the scores of a head are one 32 x 32 tile
// (S <= 32 queries x keys, reduction over the 16 components of the head = one v_mfma_f32_32x32x16_bf16), computed
// TRANSPOSED (keys along the rows, queries along the columns), so that a lane holds, for ITS query, sixteen of the
// keys -- the softmax is a reduction inside the lane plus one exchange with the lane 32 further on -- and the
// probabilities sit in registers exactly as the second product wants its B operand: P V is evaluated as
// O^T = V^T P^T with the key index permuted the same way on both operands (a sum does not care), no transpose
// through memory.  Every fp32 value enters as the pair (hi, lo) of bfloat16 numbers and a product is hi*hi + hi*lo +
// lo*hi, like the split products of the dense layers.
typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef float tf32x16 __attribute__((ext_vector_type(16)));

struct Split8 { tbf16x8 h, l; };

__device__ __forceinline__ Split8 split8(const float (&v)[8]) {
  Split8 p;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const __bf16 h = (__bf16)v[t];
    p.h[t] = h;
    p.l[t] = (__bf16)(v[t] - (float)h);
  }
  return p;
}

__device__ __forceinline__ tf32x16 mfma3(const Split8& a, const Split8& b, tf32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
}

// operands rounded to bfloat16, one product (what the dense layers' gradient products do in this mode)
__device__ __forceinline__ tbf16x8 round8(const float (&v)[8]) {
  tbf16x8 p;
#pragma unroll
  for (int t = 0; t < 8; ++t) p[t] = (__bf16)v[t];
  return p;
}

__device__ __forceinline__ tf32x16 mfma1(const tbf16x8& a, const tbf16x8& b, tf32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ tf32x16 zero16() {
  tf32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.0f;
  return z;
}

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Row index of accumulator element r of a 32 x 32 tile in lane half `hi` (the column is the lane's low five bits),
// and the index that reduction slot t of the lane's half stands for in the g-th of two products over 32 indices
// whose B operand is such an accumulator: slot t of product g is element 8 g + t.
__device__ __forceinline__ int tile_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int D, int HD>
__global__ __launch_bounds__(256) void temporal_attn_fwd_mfma_kernel(const TAttnArgs a) {
  static_assert(HD == 16, "one 32x32x16 product covers the head");
  constexpr int ROW = 3 * D, PITCH = ROW + 4;  // (the padded pitch spreads the rows of an operand read over the banks)
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][PITCH]
  const int n = blockIdx.x;
  const int S = a.S, N = a.N, H = a.H;
  stage_rows<ROW, PITCH>(a.qkv + (int64_t)n * ROW, (int64_t)N * ROW, S, lds);
  __syncthreads();
  const bool drop = a.dropout_p > 0.0f;
  const float keep_scale = drop ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  const uint64_t seed = a.seed + ((drop && a.seed_ptr) ? *a.seed_ptr : 0ull);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  for (int h = wave; h < H; h += 4) {
    float kf[8], qf[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) kf[t] = qf[t] = 0.0f;
    if (lo < S) {
      load8(lds + lo * PITCH + D + h * HD + 8 * hi, kf);
      load8(lds + lo * PITCH + h * HD + 8 * hi, qf);
#pragma unroll
      for (int t = 0; t < 8; ++t) qf[t] *= a.scale;
    }
    // scores, transposed: row = key, column = query
    const tf32x16 st = mfma3(split8(kf), split8(qf), zero16());
    const int i = lo;
    float m = -1.0e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (tile_row(r, hi) <= i && i < S) m = fmaxf(m, st[r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float p[16], l = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = (tile_row(r, hi) <= i && i < S) ? fexp(st[r] - m) : 0.0f;
      l += p[r];
    }
    l += __shfl_xor(l, 32);
    if (hi == 0 && i < S) a.lse[((int64_t)n * H + h) * S + i] = m + logf(l);
    if (drop) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = tile_row(r, hi);
        if (j <= i && i < S) {
          const uint32_t idx = (((uint32_t)n * H + h) * kMaxS + i) * kMaxS + j;
          p[r] = uniform01_u32(seed, idx) >= a.dropout_p ? p[r] * keep_scale : 0.0f;
        }
      }
    }
    // O^T[x][i] = sum_j V^T[x][j] P^T[j][i]: reduction slot t of product g is key tile_row(8 g + t, hi) on both
    // operands; the rows of the result are the head's components x (16 of the 32 rows are used)
    tf32x16 o = zero16();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float pf[8], vf[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        pf[t] = p[8 * g + t];
        const int j = tile_row(8 * g + t, hi);
        vf[t] = (lo < HD && j < S) ? lds[j * PITCH + 2 * D + h * HD + lo] : 0.0f;
      }
      o = mfma3(split8(vf), split8(pf), o);
    }
    // the q slot of (i, h) was read by this wave only, above: it takes the output row
    if (i < S) {
      const float inv = 1.0f / l;
#pragma unroll
      for (int r = 0; r < 8; ++r) lds[i * PITCH + h * HD + tile_row(r, hi)] = o[r] * inv;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S * (D / 4); i += 256) {
    const int s = i / (D / 4), c = (i % (D / 4)) * 4;
    *reinterpret_cast<float4*>(a.out + ((int64_t)s * N + n) * D + c) = *reinterpret_cast<const float4*>(lds + s * PITCH + c);
  }
}

// Backward on the matrix cores.  The gradients need the score tile in BOTH orientations (dQ sums over keys, dK and
// dV over queries), so scores and dP = dO V^T are computed twice, once per orientation, each orientation is turned
// into dS (and the dropped probabilities) in registers, and the three results come out transposed ([component][row])
// from products whose B operands are those registers: dV^T = dO^T Pm, dK^T = Q^T dS, dQ^T = K^T dS^T.
template <int D, int HD>
__global__ __launch_bounds__(256, 3) void temporal_attn_bwd_mfma_kernel(const TAttnArgs a) {
  static_assert(HD == 16, "one 32x32x16 product covers the head");
  constexpr int ROW = 3 * D, PITCH = ROW + 4, GP = D + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = blockIdx.x;
  const int S = a.S, N = a.N, H = a.H;
  float* qkv = lds;                 // [S][PITCH]
  float* go = lds + S * PITCH;      // [S][GP]   dO
  float* aux = go + S * GP;         // [H][kMaxS][2]  lse, delta per (head, query)
  uint32_t* keep_word = reinterpret_cast<uint32_t*>(aux + H * kMaxS * 2);  // [4 waves][kMaxS]: bit j = score (i, j) kept
  // every load of the prologue is issued before the first of them is waited for: the agent's q|k|v and dO rows,
  // and this thread's (head, query) slice of the forward output with its log-sum-exp (H * kMaxS = 256 pairs)
  const int ph = threadIdx.x / kMaxS, pi = threadIdx.x % kMaxS;
  float4 o4[HD / 4];
  float lse_pi = 0.0f;
  if (pi < S) {
    const float* orow = a.out + ((int64_t)pi * N + n) * D + ph * HD;
#pragma unroll
    for (int x = 0; x < HD / 4; ++x) o4[x] = *reinterpret_cast<const float4*>(orow + 4 * x);
    lse_pi = a.lse[((int64_t)n * H + ph) * S + pi];
  }
  stage_rows2<ROW, PITCH, D, GP>(a.qkv + (int64_t)n * ROW, (int64_t)N * ROW, qkv, a.d_out + (int64_t)n * D, (int64_t)N * D,
                                 go, S);
  __syncthreads();
  const bool drop = a.dropout_p > 0.0f;
  const float keep_scale = drop ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  const uint64_t seed = a.seed + ((drop && a.seed_ptr) ? *a.seed_ptr : 0ull);
  if (pi < S) {  // delta = dO . O over the head's slice
    float d = 0.0f;
#pragma unroll
    for (int x = 0; x < HD / 4; ++x) {
      const float4 g4 = *reinterpret_cast<const float4*>(go + pi * GP + ph * HD + 4 * x);
      d += g4.x * o4[x].x + g4.y * o4[x].y + g4.z * o4[x].z + g4.w * o4[x].w;
    }
    aux[(ph * kMaxS + pi) * 2 + 0] = lse_pi;
    aux[(ph * kMaxS + pi) * 2 + 1] = d;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  for (int h = wave; h < H; h += 4) {
    float qf[8], kf[8], vf[8], gf[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) qf[t] = kf[t] = vf[t] = gf[t] = 0.0f;
    if (lo < S) {
      load8(qkv + lo * PITCH + h * HD + 8 * hi, qf);
      load8(qkv + lo * PITCH + D + h * HD + 8 * hi, kf);
      load8(qkv + lo * PITCH + 2 * D + h * HD + 8 * hi, vf);
      load8(go + lo * GP + h * HD + 8 * hi, gf);
    }
    const Split8 Q = split8(qf), K = split8(kf), V = split8(vf), G = split8(gf);
    const tf32x16 s_t = mfma3(K, Q, zero16());   // [key][query]
    const tf32x16 s_n = mfma3(Q, K, zero16());   // [query][key]
    const tf32x16 dp_t = mfma3(V, G, zero16());  // [key][query]
    const tf32x16 dp_n = mfma3(G, V, zero16());  // [query][key]
    float ds_t[16], ds_n[16], pm[16];
    {  // this lane's query is lo, its keys tile_row(r, hi)
      const int i = lo;
      const float lse_i = i < S ? aux[(h * kMaxS + i) * 2] : 0.0f, delta_i = i < S ? aux[(h * kMaxS + i) * 2 + 1] : 0.0f;
      // the forward's dropout decisions of query i, drawn once: this half's keys here, the other half's by the lane
      // 32 further on; the whole word goes to LDS for the other orientation below
      uint32_t bits = 0xffffffffu;
      if (drop) {
        bits = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = tile_row(r, hi);
          if (j <= i && i < S) {
            const uint32_t idx = (((uint32_t)n * H + h) * kMaxS + i) * kMaxS + j;
            bits |= (uniform01_u32(seed, idx) >= a.dropout_p ? 1u : 0u) << j;
          }
        }
        bits |= (uint32_t)__shfl_xor((int)bits, 32);
        if (hi == 0) keep_word[wave * kMaxS + i] = bits;
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = tile_row(r, hi);
        float v = 0.0f;
        if (j <= i && i < S) {
          const float p = fexp(s_t[r] * a.scale - lse_i);
          const float ms = ((bits >> j) & 1u) ? keep_scale : 0.0f;
          v = p * (dp_t[r] * ms - delta_i) * a.scale;
        }
        ds_t[r] = v;
      }
    }
    {  // this lane's key is lo, its queries tile_row(r, hi)
      const int j = lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = tile_row(r, hi);
        float v = 0.0f, w = 0.0f;
        if (j <= i && i < S) {
          const float2 ld = *reinterpret_cast<const float2*>(aux + (h * kMaxS + i) * 2);
          const float p = fexp(s_n[r] * a.scale - ld.x);
          const uint32_t word = drop ? keep_word[wave * kMaxS + i] : 0xffffffffu;
          const float ms = ((word >> j) & 1u) ? keep_scale : 0.0f;
          w = p * ms;
          v = p * (dp_n[r] * ms - ld.y) * a.scale;
        }
        ds_n[r] = v;
        pm[r] = w;
      }
    }
    tf32x16 dv = zero16(), dk = zero16(), dq = zero16();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // A operands: row = component x = lo (16 of 32 used), reduction slot t = row tile_row(8 g + t, hi) of dO / Q / K
      float gt[8], qt[8], kt[8], b0[8], b1[8], b2[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = tile_row(8 * g + t, hi);
        const bool ok = lo < HD && row < S;
        gt[t] = ok ? go[row * GP + h * HD + lo] : 0.0f;
        qt[t] = ok ? qkv[row * PITCH + h * HD + lo] : 0.0f;
        kt[t] = ok ? qkv[row * PITCH + D + h * HD + lo] : 0.0f;
        b0[t] = pm[8 * g + t];
        b1[t] = ds_n[8 * g + t];
        b2[t] = ds_t[8 * g + t];
      }
      // (the three gradient products take operands rounded to bfloat16, like the dense layers' gradient products
      // in this mode; the scores and dP above, which the softmax is rebuilt from, keep the split operands)
      dv = mfma1(round8(gt), round8(b0), dv);  // dV^T[x][key]
      dk = mfma1(round8(qt), round8(b1), dk);  // dK^T[x][key]
      dq = mfma1(round8(kt), round8(b2), dq);  // dQ^T[x][query]
    }
    // the head's q | k | v slots of row lo were read by this wave only, all reads are issued: they take the results
    if (lo < S) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int x = tile_row(r, hi);
        qkv[lo * PITCH + h * HD + x] = dq[r];
        qkv[lo * PITCH + D + h * HD + x] = dk[r];
        qkv[lo * PITCH + 2 * D + h * HD + x] = dv[r];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S * (ROW / 4); i += 256) {
    const int s = i / (ROW / 4), c = (i % (ROW / 4)) * 4;
    const float4 g = *reinterpret_cast<const float4*>(qkv + s * PITCH + c);
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

// Whether the matrix-core kernels apply: the 128-wide model with 16-wide heads, while the forward dense products
// run on split bfloat16 operands (the fp32 mode keeps the scalar kernels, whose sums are exact fp32).
bool mfma_route(int D, int H) {
  return D == 128 && H == 8 && (matmul_bf16_mask() & HIVT_BF16_FWD_SPLIT) != 0;
}

int launch_mfma(bool backward, const TAttnArgs& a, hipStream_t stream) {
  constexpr int D = 128;
  const size_t fwd_lds = (size_t)a.S * (3 * D + 4) * sizeof(float);
  const size_t bwd_lds = ((size_t)a.S * (3 * D + 4 + D + 4) + (size_t)a.H * kMaxS * 2 + 4 * kMaxS) * sizeof(float);
  const size_t need = backward ? bwd_lds : fwd_lds;
  if (need > 160 * 1024) return -1;
  if (need > 48 * 1024) {
    hipError_t e = backward ? hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_bwd_mfma_kernel<128, 16>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)need)
                            : hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_fwd_mfma_kernel<128, 16>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
    if (e != hipSuccess) return -1;
  }
  if (backward)
    hipLaunchKernelGGL((temporal_attn_bwd_mfma_kernel<128, 16>), dim3((unsigned)a.N), dim3(256), bwd_lds, stream, a);
  else
    hipLaunchKernelGGL((temporal_attn_fwd_mfma_kernel<128, 16>), dim3((unsigned)a.N), dim3(256), fwd_lds, stream, a);
  return 0;
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
  if (mfma_route(D, H)) return launch_mfma(backward, a, stream);
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
