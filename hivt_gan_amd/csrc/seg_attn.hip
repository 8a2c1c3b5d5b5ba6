// Kernels (b)+(c): per-edge QK dot, segment softmax over CSR rows and weighted sum of V, and the
// matching backward.  Both are single-pass streaming kernels bound by HBM bandwidth: forward reads
// each K and V row once; backward reads K, V once and writes dK, dV once.
//
// Mapping (wave64): a K/V row of D floats is covered by LPR = D/4 lanes holding one float4 each,
// so one wave instruction loads EPI = 64/LPR edges (2 at D=128, 4 at D=64) as full 512-byte /
// 256-byte contiguous rows.  A head spans LPH = D/(4H) adjacent lanes; the QK dot is finished with
// log2(LPH) xor-shuffles.  Each lane slot runs an online softmax over its share of the row's edges
// (4 edges in flight per slot, 2 in short-row mode), slots are merged with xor-shuffles, and for long rows the 4 waves
// of a block split one row and merge through LDS.  No atomics: every target row is owned by one
// wave or one block, so results are deterministic.
#include "common.h"

#include <hip/hip_ext.h>
#include <math.h>

namespace hivt {
namespace {

constexpr float kNegSentinel = -1.0e30f;  // "no edge seen yet"; exp(kNegSentinel - x) == 0
constexpr int kUnrollLong = 4;   // edges in flight per lane slot
constexpr int kUnrollWide = 8;   // rows split over four waves: a 64-edge row is one batch of loads per wave
constexpr int kUnrollShort = 2;  // short-row mode: rows average under two edges, a wider batch is mostly masked work

struct SegAttnArgs {
  const float* q;
  int64_t q_stride, dq_stride;
  const float* k;
  const float* v;
  int64_t kv_stride;
  const float* k_node;
  const float* v_node;
  int64_t node_stride;
  const int32_t* src;
  const int32_t* rowptr;
  const int32_t* row_order;  // optional [N]: the row that the g-th row group of the launch serves (one row per group)
  int32_t N;
  int32_t H;
  // 0: edge e of the CSR order is row e of k / v / dk / dv.  Otherwise the rows of a target are interleaved
  // ("step-major": the j-th edge of target n is row j * row_interleave + n), the layout of a [steps, targets,
  // width] tensor whose targets attend over their own steps.
  int32_t row_interleave;
  int32_t seg_len;  // with row_interleave: rows per target in k / v (E / N); a target reads no more than that
  float scale_inv;
  float dropout_p;
  uint64_t seed;
  const uint64_t* seed_ptr;  // optional device-side seed added to `seed` (graph replays vary it)
  // forward outputs / backward saved
  float* out;
  float* m;
  float* den;
  // backward only
  const float* d_out;
  float* dq;
  float* dk;
  float* dv;
  int64_t dkv_stride;
  int32_t dkv_half;  // dk / dv point at bfloat16 elements (dkv_stride counts them); values rounded to nearest even
  int32_t zero_tail;   // backward: rows [rowptr[N], E_cap) of dk / dv are zeroed by extra blocks of the same launch
  int32_t row_blocks;  // backward: blocks that serve targets; the blocks after them zero the tail
  int64_t E_cap;
  int32_t narrow;      // every K / V / dK / dV row starts below 4 GiB: 32-bit byte offsets
  // optional: events that receive the kernel's own start / stop time stamps
  hipEvent_t ev_start;
  hipEvent_t ev_stop;
};

// exp through the hardware exp2: relative error ~1e-7 for arguments near 0 (the terms that carry
// the softmax weight) and below 6e-8 * |x| elsewhere (terms whose weight is e^x itself).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
__device__ __forceinline__ void fma4(float4& acc, float s, const float4& v) {
  acc.x += s * v.x;
  acc.y += s * v.y;
  acc.z += s * v.z;
  acc.w += s * v.w;
}
__device__ __forceinline__ void scale4(float4& a, float s) {
  a.x *= s;
  a.y *= s;
  a.z *= s;
  a.w *= s;
}
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 shfl_xor4(const float4& a, int o) {
  return make_float4(__shfl_xor(a.x, o, 64), __shfl_xor(a.y, o, 64), __shfl_xor(a.z, o, 64), __shfl_xor(a.w, o, 64));
}

// Byte offsets into the K / V / dK / dV rows.  NARROW: every row of those tensors starts below 4 GiB (the entry
// points check it), so an offset is one 32-bit register added to the tensor's base address by the load itself,
// and consecutive edges of a batch differ by a wave-uniform step -- a few integer instructions per edge where 64-bit
// row * stride products took a dozen.
template <bool NARROW>
struct OffsetType {
  using type = uint64_t;
};
template <>
struct OffsetType<true> {
  using type = uint32_t;
};
template <typename Off>
__device__ __forceinline__ float4 ldg4_at(const float* base, Off off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}
template <typename Off>
__device__ __forceinline__ char* byte_at(float* base, Off off) {
  return reinterpret_cast<char*>(base) + off;
}

// Dropout decision of (edge, head) number idx: the 24 random bits of the 32-bit mix against p * 2^24 rounded up,
// which is the comparison uniform >= p done on integers.
template <typename Idx>
__device__ __forceinline__ bool keep_draw(uint64_t seed, Idx idx, uint32_t threshold) {
  if constexpr (sizeof(Idx) == 8) seed += (uint64_t)(idx >> 32) * 0x9E3779B97F4A7C15ull;
  return uniform_bits24(seed, (uint32_t)idx) >= threshold;
}

// Merge two online-softmax states (m, l, acc) <- (m, l, acc) (+) (m2, l2, acc2).
__device__ __forceinline__ void merge_state(float& m, float& l, float4& acc, float m2, float l2, const float4& acc2) {
  const float mn = fmaxf(m, m2);
  const float c1 = fast_exp(m - mn), c2 = fast_exp(m2 - mn);
  l = l * c1 + l2 * c2;
  acc.x = acc.x * c1 + acc2.x * c2;
  acc.y = acc.y * c1 + acc2.y * c2;
  acc.z = acc.z * c1 + acc2.z * c2;
  acc.w = acc.w * c1 + acc2.w * c2;
  m = mn;
}

// KIND = where the K / V rows of a target come from (a compile-time choice: the three kinds are distinct kernels,
//        so a profile tells them apart by name)
constexpr int kRowsPlain = 0;        // edge e of the CSR order is row e
constexpr int kRowsNodeTerms = 1;    // row e plus the row of the edge's source node (gathered through src)
constexpr int kRowsInterleaved = 2;  // step-major rows: the j-th edge of target n is row j * N + n

// WPN  = waves that share one target row (1, or 4 for long rows, merged through LDS)
// SPLIT = short-row mode: every lane slot (LPR lanes) of a wave owns its OWN target row, so a wave
//         serves 64/LPR rows at once and no slot merge is needed (requires WPN == 1)
template <int D, int LPH, int WPN, int KIND, bool SPLIT, int UNROLL, bool NARROW>
__global__ __launch_bounds__(256) void seg_attn_fwd_kernel(const SegAttnArgs a) {
  constexpr bool GLOBAL = KIND == kRowsNodeTerms;
  constexpr bool INTERLEAVED = KIND == kRowsInterleaved;
  constexpr int LPR = D / 4;
  constexpr int EPI = kWave / LPR;
  constexpr int NPB = 4 / WPN;  // row groups per 256-thread block
  constexpr int RPW = SPLIT ? EPI : 1;
  constexpr int kUnroll = UNROLL;
  using Off = typename OffsetType<NARROW>::type;
  static_assert(!SPLIT || WPN == 1, "short-row mode is single-wave");
  __shared__ float smem[WPN > 1 ? 4 * LPR * 6 : 1];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / LPR;
  const int c = lane % LPR;
  const int h = c / LPH;
  const int win = wave % WPN;
  int n = (blockIdx.x * NPB + wave / WPN) * RPW + (SPLIT ? sub : 0);
  if (!SPLIT && a.row_order != nullptr && n < a.N) n = a.row_order[n];
  if (!SPLIT && WPN == 1 && n >= a.N) return;  // with WPN > 1 every wave stays for the barrier (row_ok gates the work)
  const bool row_ok = n < a.N;
  const int nc = row_ok ? n : a.N - 1;

  // A row's memory latencies in a row are two, not three: its bounds, then its first batch of K / V rows.  The query
  // is requested with the bounds where most rows have edges (every mode but the short-row one, which asks only for
  // rows that have edges: there most rows are empty), and it is first used after the first batch has been requested.
  const int beg = a.rowptr[nc];
  int end = row_ok ? a.rowptr[nc + 1] : beg;
  float4 qs = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (!SPLIT) qs = ldg4(a.q + (int64_t)nc * a.q_stride + 4 * c);
  if (INTERLEAVED) end = min(end, beg + a.seg_len);
  const bool drop = a.dropout_p > 0.0f;
  const float keep_scale = drop ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  const uint32_t threshold = (uint32_t)ceilf(a.dropout_p * 16777216.0f);
  const uint64_t seed = a.seed + ((drop && a.seed_ptr) ? *a.seed_ptr : 0ull);

  float m = kNegSentinel, l = 0.0f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

  constexpr int STRIDE = SPLIT ? 1 : WPN * EPI;
  const Off row_bytes = (Off)a.kv_stride * 4;
  const Off step = (INTERLEAVED ? row_bytes * (Off)a.row_interleave : row_bytes) * STRIDE;  // to the next edge of a slot
  float4 kk[kUnroll], vv[kUnroll];
  bool ok[kUnroll];
  int e0 = SPLIT ? beg : beg + win * EPI + sub;
  auto request_batch = [&]() {
    const Off r0 = INTERLEAVED ? (Off)(e0 - beg) * (Off)a.row_interleave + (Off)nc : (Off)e0;
    const Off at0 = r0 * row_bytes + 16 * c;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      ok[u] = e0 + u * STRIDE < end;
      const Off at = ok[u] ? at0 + (Off)u * step : at0;
      kk[u] = ldg4_at(a.k, at);
      vv[u] = ldg4_at(a.v, at);
      if (GLOBAL) {
        const int64_t j = a.src[ok[u] ? e0 + u * STRIDE : e0];
        kk[u] = add4(kk[u], ldg4(a.k_node + j * a.node_stride + 4 * c));
        vv[u] = add4(vv[u], ldg4(a.v_node + j * a.node_stride + 4 * c));
      }
    }
  };
  bool more = e0 < end;
  if (more) {
    if constexpr (SPLIT) qs = ldg4(a.q + (int64_t)nc * a.q_stride + 4 * c);
    request_batch();
  }
  scale4(qs, a.scale_inv);
  while (more) {
    float s[kUnroll];
    float smax = m;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const float part = group_sum<LPH>(dot4(qs, kk[u]));
      s[u] = ok[u] ? part : kNegSentinel;
      smax = fmaxf(smax, s[u]);
    }
    const float corr = fast_exp(m - smax);
    l *= corr;
    scale4(acc, corr);
    m = smax;
    const Off idx0 = (Off)e0 * (Off)a.H + (Off)h;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      float p = ok[u] ? fast_exp(s[u] - m) : 0.0f;
      l += p;
      if (drop) p = keep_draw<Off>(seed, idx0 + (Off)(u * STRIDE) * (Off)a.H, threshold) ? p * keep_scale : 0.0f;
      fma4(acc, p, vv[u]);
    }
    e0 += STRIDE * kUnroll;
    more = e0 < end;
    if (more) request_batch();
  }

  if (!SPLIT) {
    // merge the EPI lane slots of this wave
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) {
      const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
      const float4 acc2 = shfl_xor4(acc, o);
      merge_state(m, l, acc, m2, l2, acc2);
    }
  }

  if (WPN > 1) {
    if (sub == 0) {
      float* slot = smem + (wave * LPR + c) * 6;
      slot[0] = m;
      slot[1] = l;
      slot[2] = acc.x;
      slot[3] = acc.y;
      slot[4] = acc.z;
      slot[5] = acc.w;
    }
    __syncthreads();
    if (win != 0) return;
#pragma unroll
    for (int w = 1; w < WPN; ++w) {
      const float* slot = smem + ((wave + w) * LPR + c) * 6;
      merge_state(m, l, acc, slot[0], slot[1], make_float4(slot[2], slot[3], slot[4], slot[5]));
    }
  }

  if ((SPLIT || sub == 0) && row_ok) {
    const float den = l + 1e-16f;
    const float inv = 1.0f / den;
    *reinterpret_cast<float4*>(a.out + (int64_t)n * D + 4 * c) =
        make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    if (c % LPH == 0) {
      a.m[(int64_t)n * a.H + h] = m;
      a.den[(int64_t)n * a.H + h] = den;
    }
  }
}

// four values as bfloat16 (nearest even) in 8 bytes
typedef __bf16 bf16_pair __attribute__((ext_vector_type(2)));
typedef float f32_pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 pack_bf16x4(float x, float y, float z, float w) {
  const f32_pair lo = {x, y}, hi = {z, w};
  return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16_pair)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16_pair)));
}

// Blocks past the row groups zero rows [rowptr[N], E_cap) of dk / dv, which belong to no target (fixed-capacity edge
// lists): everything upstream of the projection then sees exact zeros for the padding.
constexpr int kTailBlocks = 128;
template <int D, bool HALF>
__device__ __forceinline__ void zero_tail_rows(const SegAttnArgs& a, int block, int blocks) {
  constexpr int QUADS = D / 4;
  const int64_t live = a.rowptr[a.N];
  const int64_t total = (a.E_cap - live) * QUADS;
  for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < total; i += (int64_t)blocks * 256) {
    const int64_t e = live + i / QUADS;
    const int c = (int)(i % QUADS) * 4;
    if constexpr (HALF) {  // bfloat16 rows: four zeros are 8 bytes
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.dk) + e * a.dkv_stride + c) = make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.dv) + e * a.dkv_stride + c) = make_uint2(0u, 0u);
    } else {
      *reinterpret_cast<float4*>(a.dk + e * a.dkv_stride + c) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(a.dv + e * a.dkv_stride + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// HALF: dk / dv are bfloat16 rows (a compile-time switch: the fp32 kernels keep their arithmetic as it was)
template <int D, int LPH, int WPN, int KIND, bool SPLIT, int UNROLL, bool HALF, bool NARROW>
__global__ __launch_bounds__(256) void seg_attn_bwd_kernel(const SegAttnArgs a) {
  constexpr bool GLOBAL = KIND == kRowsNodeTerms;
  constexpr bool INTERLEAVED = KIND == kRowsInterleaved;
  constexpr int LPR = D / 4;
  constexpr int EPI = kWave / LPR;
  constexpr int NPB = 4 / WPN;
  constexpr int RPW = SPLIT ? EPI : 1;
  constexpr int kUnroll = UNROLL;
  using Off = typename OffsetType<NARROW>::type;
  static_assert(!SPLIT || WPN == 1, "short-row mode is single-wave");
  __shared__ float smem[WPN > 1 ? 4 * LPR * 4 : 1];

  if ((int)blockIdx.x >= a.row_blocks) {  // whole blocks: none of them reaches a barrier
    zero_tail_rows<D, HALF>(a, (int)blockIdx.x - a.row_blocks, (int)gridDim.x - a.row_blocks);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / LPR;
  const int c = lane % LPR;
  const int h = c / LPH;
  const int win = wave % WPN;
  int n = (blockIdx.x * NPB + wave / WPN) * RPW + (SPLIT ? sub : 0);
  if (!SPLIT && a.row_order != nullptr && n < a.N) n = a.row_order[n];
  if (!SPLIT && WPN == 1 && n >= a.N) return;
  const bool row_ok = n < a.N;
  const int nc = row_ok ? n : a.N - 1;

  // As in the forward kernel: the row's own operands travel with its bounds (or, in the short-row mode, with its
  // first batch) and are first used once that batch has been requested.
  const int beg = a.rowptr[nc];
  int end = row_ok ? a.rowptr[nc + 1] : beg;
  float4 qs = make_float4(0.f, 0.f, 0.f, 0.f), go = qs, o4 = qs;
  float mh = 0.0f, den = 1.0f;
  auto request_row = [&]() {
    qs = ldg4(a.q + (int64_t)nc * a.q_stride + 4 * c);
    go = ldg4(a.d_out + (int64_t)nc * D + 4 * c);
    o4 = ldg4(a.out + (int64_t)nc * D + 4 * c);
    mh = a.m[(int64_t)nc * a.H + h];
    den = a.den[(int64_t)nc * a.H + h];
  };
  if constexpr (!SPLIT) request_row();
  if (INTERLEAVED) end = min(end, beg + a.seg_len);
  const bool drop = a.dropout_p > 0.0f;
  const float keep_scale = drop ? 1.0f / (1.0f - a.dropout_p) : 1.0f;
  const uint32_t threshold = (uint32_t)ceilf(a.dropout_p * 16777216.0f);
  const uint64_t seed = a.seed + ((drop && a.seed_ptr) ? *a.seed_ptr : 0ull);

  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);

  constexpr int STRIDE = SPLIT ? 1 : WPN * EPI;
  const Off row_bytes = (Off)a.kv_stride * 4;
  const Off grad_bytes = (Off)a.dkv_stride * (HALF ? 2 : 4);
  const Off interleave = INTERLEAVED ? (Off)a.row_interleave : (Off)1;
  const Off step = row_bytes * interleave * STRIDE, grad_step = grad_bytes * interleave * STRIDE;
  float4 kk[kUnroll], vv[kUnroll];
  bool ok[kUnroll];
  Off r0 = 0;
  int e0 = SPLIT ? beg : beg + win * EPI + sub;
  auto request_batch = [&]() {
    r0 = INTERLEAVED ? (Off)(e0 - beg) * interleave + (Off)nc : (Off)e0;
    const Off at0 = r0 * row_bytes + 16 * c;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      ok[u] = e0 + u * STRIDE < end;
      const Off at = ok[u] ? at0 + (Off)u * step : at0;
      kk[u] = ldg4_at(a.k, at);
      vv[u] = ldg4_at(a.v, at);
      if (GLOBAL) {
        const int64_t j = a.src[ok[u] ? e0 + u * STRIDE : e0];
        kk[u] = add4(kk[u], ldg4(a.k_node + j * a.node_stride + 4 * c));
        vv[u] = add4(vv[u], ldg4(a.v_node + j * a.node_stride + 4 * c));
      }
    }
  };
  bool more = e0 < end;
  if (more) {
    if constexpr (SPLIT) request_row();
    request_batch();
  }
  scale4(qs, a.scale_inv);
  const float delta = group_sum<LPH>(dot4(go, o4));
  const float inv_den = 1.0f / den;
  while (more) {
    const Off idx0 = (Off)e0 * (Off)a.H + (Off)h;
    const Off grad0 = r0 * grad_bytes + (HALF ? 8 : 16) * c;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const float s = group_sum<LPH>(dot4(qs, kk[u]));
      const float dav = group_sum<LPH>(dot4(go, vv[u]));
      if (ok[u]) {
        const float alpha = fast_exp(s - mh) * inv_den;
        float ms = 1.0f;
        if (drop) ms = keep_draw<Off>(seed, idx0 + (Off)(u * STRIDE) * (Off)a.H, threshold) ? keep_scale : 0.0f;
        const float am = alpha * ms;
        const float ds = alpha * (dav * ms - delta);
        const Off at = grad0 + (Off)u * grad_step;
        if constexpr (HALF) {
          const uint2 pv = pack_bf16x4(am * go.x, am * go.y, am * go.z, am * go.w);
          const uint2 pk = pack_bf16x4(ds * qs.x, ds * qs.y, ds * qs.z, ds * qs.w);
          *reinterpret_cast<uint2*>(byte_at(a.dv, at)) = pv;
          *reinterpret_cast<uint2*>(byte_at(a.dk, at)) = pk;
        } else {
          *reinterpret_cast<float4*>(byte_at(a.dv, at)) = make_float4(am * go.x, am * go.y, am * go.z, am * go.w);
          *reinterpret_cast<float4*>(byte_at(a.dk, at)) = make_float4(ds * qs.x, ds * qs.y, ds * qs.z, ds * qs.w);
        }
        fma4(dq, ds, kk[u]);
      }
    }
    e0 += STRIDE * kUnroll;
    more = e0 < end;
    if (more) request_batch();
  }

  if (!SPLIT) {
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) dq = add4(dq, shfl_xor4(dq, o));
  }

  if (WPN > 1) {
    if (sub == 0) *reinterpret_cast<float4*>(smem + (wave * LPR + c) * 4) = dq;
    __syncthreads();
    if (win != 0) return;
#pragma unroll
    for (int w = 1; w < WPN; ++w) dq = add4(dq, *reinterpret_cast<const float4*>(smem + ((wave + w) * LPR + c) * 4));
  }
  if ((SPLIT || sub == 0) && row_ok) {
    scale4(dq, a.scale_inv);
    *reinterpret_cast<float4*>(a.dq + (int64_t)n * a.dq_stride + 4 * c) = dq;
  }
}

template <typename Kernel>
void launch_kernel(Kernel kernel, unsigned grid, const SegAttnArgs& a, hipStream_t stream) {
  if (a.ev_start != nullptr) {
    // the kernel's argument block does not need the host-side event handles
    SegAttnArgs k = a;
    k.ev_start = nullptr;
    k.ev_stop = nullptr;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, k);
  } else {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, stream, a);
  }
}

template <int D, int LPH, int WPN, int KIND, bool SPLIT, int UNROLL, bool NARROW>
int launch_one(bool backward, SegAttnArgs a, hipStream_t stream) {
  constexpr int NPB = 4 / WPN;
  constexpr int RPW = SPLIT ? kWave / (D / 4) : 1;
  const unsigned grid = (unsigned)((a.N + NPB * RPW - 1) / (NPB * RPW));
  if (!backward) {
    launch_kernel(seg_attn_fwd_kernel<D, LPH, WPN, KIND, SPLIT, UNROLL, NARROW>, grid, a, stream);
    return 0;
  }
  a.row_blocks = (int32_t)grid;
  const unsigned all = grid + (a.zero_tail ? kTailBlocks : 0);
  if (a.dkv_half) {
    // bfloat16 dk / dv: built for plain rows (no node terms, not interleaved)
    if constexpr (KIND == kRowsPlain)
      launch_kernel(seg_attn_bwd_kernel<D, LPH, WPN, KIND, SPLIT, UNROLL, true, NARROW>, all, a, stream);
    else
      return -1;
  } else {
    launch_kernel(seg_attn_bwd_kernel<D, LPH, WPN, KIND, SPLIT, UNROLL, false, NARROW>, all, a, stream);
  }
  return 0;
}

template <int D, int LPH, int WPN, bool SPLIT, int UNROLL>
int launch_g(bool backward, bool global, const SegAttnArgs& a, hipStream_t stream) {
#define HIVT_KIND(KIND_)                                                                           \
  return a.narrow ? launch_one<D, LPH, WPN, KIND_, SPLIT, UNROLL, true>(backward, a, stream)       \
                  : launch_one<D, LPH, WPN, KIND_, SPLIT, UNROLL, false>(backward, a, stream);
  if (global) { HIVT_KIND(kRowsNodeTerms) }
  if (a.row_interleave) { HIVT_KIND(kRowsInterleaved) }
  HIVT_KIND(kRowsPlain)
#undef HIVT_KIND
}

// Launch shapes.  `waves`: 0 = one target row per lane slot (short rows), 1 / 2 / 4 = waves of a block that
// share one row; `unroll` = edges in flight per lane slot.
struct Shape {
  int waves;
  int unroll;
};

// By average row length (measured on 1.3-, 18- and 54-edge rows at D = 128): rows shorter than two unrolled
// passes of a lane slot go one per slot (two edges in flight forward, four backward); long rows are split over the 4 waves of a block with 8 edges in
// flight per slot; rows in between take one wave and a short batch forward, but four waves backward, where
// every edge also costs two stores and the wider split keeps more of them in flight.
Shape default_shape(int64_t E, int32_t N, bool backward) {
  const int64_t avg = N > 0 ? E / N : 0;
  if (avg >= 48) return {4, kUnrollWide};
  if (avg < 8) return {0, backward ? 4 : kUnrollShort};
  return backward ? Shape{4, 4} : Shape{1, 2};
}

template <int D, int LPH>
int launch_dl(bool backward, Shape sh, bool global, const SegAttnArgs& a, hipStream_t stream) {
  if constexpr (LPH <= D / 4) {
    if (sh.waves == 0 && D >= 256) sh = {1, kUnrollLong};
#define HIVT_SHAPE(W_, SPLIT_, U_) \
  if (sh.waves == (SPLIT_ ? 0 : W_) && sh.unroll == U_) return launch_g<D, LPH, W_, SPLIT_, U_>(backward, global, a, stream);
    // the three default shapes exist for every row width
    HIVT_SHAPE(4, false, kUnrollWide)
    HIVT_SHAPE(1, true, kUnrollShort)
    HIVT_SHAPE(1, true, 4)
    HIVT_SHAPE(1, false, kUnrollLong)
    HIVT_SHAPE(4, false, 4)
    HIVT_SHAPE(1, false, 2)
    if constexpr (D == 128) {  // further shapes of the benchmarked width, reachable through the shape hint
      HIVT_SHAPE(4, false, 2)
      HIVT_SHAPE(2, false, 2)
      HIVT_SHAPE(2, false, 4)
      HIVT_SHAPE(2, false, 8)
      HIVT_SHAPE(1, false, 8)
      HIVT_SHAPE(1, true, 8)
    }
#undef HIVT_SHAPE
    return -1;
  } else {
    return -1;
  }
}

template <int D>
int launch_d(int lph, bool backward, Shape sh, bool global, const SegAttnArgs& a, hipStream_t stream) {
  switch (lph) {
    case 1: return launch_dl<D, 1>(backward, sh, global, a, stream);
    case 2: return launch_dl<D, 2>(backward, sh, global, a, stream);
    case 4: return launch_dl<D, 4>(backward, sh, global, a, stream);
    case 8: return launch_dl<D, 8>(backward, sh, global, a, stream);
    default: return -1;
  }
}

// hint: bits 8-11 of the dtype argument = waves code (0 auto, 1 = one row per lane slot, 2 / 3 / 4 = 1 / 2 / 4
// waves per row), bits 12-15 = edges in flight per lane slot (0 auto)
int dispatch(int D, int H, bool backward, int64_t E, bool global, int hint, const SegAttnArgs& a, hipStream_t stream) {
  const int lph = D / (4 * H);
  Shape sh = default_shape(E, a.N, backward);
  const int wcode = (hint >> 8) & 15, ucode = (hint >> 12) & 15;
  if (wcode != 0) {
    static const int kWaves[5] = {0, 0, 1, 2, 4};
    if (wcode > 4) return -1;
    sh.waves = kWaves[wcode];
    sh.unroll = sh.waves == 0 ? kUnrollShort : (sh.waves == 4 ? kUnrollWide : kUnrollLong);
  }
  if (ucode != 0) sh.unroll = ucode;
  switch (D) {
    case 16: return launch_d<16>(lph, backward, sh, global, a, stream);
    case 32: return launch_d<32>(lph, backward, sh, global, a, stream);
    case 64: return launch_d<64>(lph, backward, sh, global, a, stream);
    case 128: return launch_d<128>(lph, backward, sh, global, a, stream);
    case 256: return launch_d<256>(lph, backward, sh, global, a, stream);
    default: return -1;
  }
}

// rows * stride elements of `bytes` each end below 4 GiB, and so does the (edge, head) counter of the dropout draw
// (heads * 4 <= stride)
int fits_32bit(int64_t rows, int64_t stride, int bytes) {
  return rows * stride * bytes <= (int64_t)0xFFFFFFFFll ? 1 : 0;
}

int check_shape(const char* what, int32_t N, int64_t E, int32_t D, int32_t H, int64_t kv_stride, int dtype,
                float dropout_p) {
  HIVT_REQUIRE((dtype & 255) == HIVT_F32, HIVT_ERR_UNSUPPORTED, "%s: only HIVT_F32 is built", what);
  HIVT_REQUIRE(N >= 0 && E >= 0 && E < (int64_t)1 << 31, HIVT_ERR_ARG, "%s: bad sizes N=%d E=%lld", what, N,
               (long long)E);
  HIVT_REQUIRE(D == 16 || D == 32 || D == 64 || D == 128 || D == 256, HIVT_ERR_ARG,
               "%s: embed_dim %d not in {16,32,64,128,256}", what, D);
  HIVT_REQUIRE(H > 0 && D % (4 * H) == 0, HIVT_ERR_ARG, "%s: head dim D/H must be a multiple of 4 (D=%d H=%d)", what,
               D, H);
  const int lph = D / (4 * H);
  HIVT_REQUIRE(lph == 1 || lph == 2 || lph == 4 || lph == 8, HIVT_ERR_ARG,
               "%s: head dim %d not in {4,8,16,32}", what, D / H);
  HIVT_REQUIRE(kv_stride >= D && kv_stride % 4 == 0, HIVT_ERR_ARG, "%s: kv_stride %lld must be >= D and 16-B aligned",
               what, (long long)kv_stride);
  HIVT_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, HIVT_ERR_ARG, "%s: dropout_p must be in [0,1)", what);
  return HIVT_OK;
}

}  // namespace
}  // namespace hivt

using namespace hivt;

static int seg_attn_fwd_impl(const float* q, int64_t q_stride, const float* k, const float* v, int64_t kv_stride, const float* k_node,
                             const float* v_node, int64_t node_stride, const int32_t* src, const int32_t* rowptr,
                             const int32_t* row_order, int32_t N, int64_t E, int32_t D, int32_t H, float dropout_p,
                             uint64_t seed, const uint64_t* seed_ptr, float* out, float* m_out, float* den_out, int dtype,
                             hivt_stream_t stream, void* ev_start, void* ev_stop) {
  const char* what = "hivt_seg_attn_fwd";
  HIVT_REQUIRE((ev_start == nullptr) == (ev_stop == nullptr), HIVT_ERR_ARG, "%s: need both events or none", what);
  int rc = check_shape(what, N, E, D, H, kv_stride, dtype, dropout_p);
  if (rc != HIVT_OK) return rc;
  if (N == 0) return HIVT_OK;
  const bool global = k_node != nullptr;
  HIVT_REQUIRE(q && rowptr && out && m_out && den_out, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(E == 0 || (k && v), HIVT_ERR_ARG, "%s: null k/v with E > 0", what);
  HIVT_REQUIRE(!global || (v_node && src && node_stride >= D && node_stride % 4 == 0), HIVT_ERR_ARG,
               "%s: node terms need v_node, src and a 16-B aligned node_stride >= D", what);
  HIVT_REQUIRE(q_stride >= D && q_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0, HIVT_ERR_ARG,
               "%s: q_stride must be >= D and the query rows 16-B aligned", what);
  SegAttnArgs a{};
  a.q = q; a.q_stride = q_stride; a.k = k; a.v = v; a.kv_stride = kv_stride;
  a.k_node = k_node; a.v_node = v_node; a.node_stride = node_stride; a.src = src;
  a.rowptr = rowptr; a.row_order = row_order; a.N = N; a.H = H;
  if (dtype & HIVT_ROWS_INTERLEAVED) {
    HIVT_REQUIRE(!global && E % N == 0, HIVT_ERR_ARG,
                 "%s: interleaved rows need E a multiple of N and no node terms (E=%lld N=%d)", what, (long long)E, N);
    a.row_interleave = N; a.seg_len = (int32_t)(E / N);
  }
  a.scale_inv = 1.0f / sqrtf((float)(D / H));
  a.dropout_p = dropout_p; a.seed = seed; a.seed_ptr = seed_ptr;
  a.out = out; a.m = m_out; a.den = den_out;
  a.narrow = fits_32bit(E, kv_stride, 4);
  a.ev_start = (hipEvent_t)ev_start; a.ev_stop = (hipEvent_t)ev_stop;
  HIVT_REQUIRE(dispatch(D, H, false, E, global, dtype, a, (hipStream_t)stream) == 0, HIVT_ERR_ARG,
               "%s: no kernel for D=%d H=%d (launch-shape hint 0x%x)", what, D, H, dtype >> 8);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_seg_attn_fwd(const float* q, int64_t q_stride, const float* k, const float* v, int64_t kv_stride,
                                 const float* k_node, const float* v_node, int64_t node_stride, const int32_t* src,
                                 const int32_t* rowptr, const int32_t* row_order, int32_t N, int64_t E, int32_t D,
                                 int32_t H, float dropout_p, uint64_t seed, const uint64_t* seed_ptr, float* out,
                                 float* m_out, float* den_out, int dtype, hivt_stream_t stream) {
  return seg_attn_fwd_impl(q, q_stride, k, v, kv_stride, k_node, v_node, node_stride, src, rowptr, row_order, N, E, D, H,
                           dropout_p, seed, seed_ptr, out, m_out, den_out, dtype, stream, nullptr, nullptr);
}

extern "C" int hivt_seg_attn_fwd_timed(const float* q, int64_t q_stride, const float* k, const float* v,
                                       int64_t kv_stride, const float* k_node, const float* v_node,
                                       int64_t node_stride, const int32_t* src, const int32_t* rowptr,
                                       const int32_t* row_order, int32_t N, int64_t E, int32_t D, int32_t H,
                                       float dropout_p, uint64_t seed, const uint64_t* seed_ptr, float* out,
                                       float* m_out, float* den_out, int dtype, hivt_stream_t stream,
                                       void* start_event, void* stop_event) {
  return seg_attn_fwd_impl(q, q_stride, k, v, kv_stride, k_node, v_node, node_stride, src, rowptr, row_order, N, E, D, H,
                           dropout_p, seed, seed_ptr, out, m_out, den_out, dtype, stream, start_event, stop_event);
}

static int seg_attn_bwd_impl(const float* d_out, const float* q, int64_t q_stride, const float* k, const float* v, int64_t kv_stride,
                             const float* k_node, const float* v_node, int64_t node_stride, const int32_t* src,
                             const int32_t* rowptr, const int32_t* row_order, const float* out, const float* m,
                             const float* den, int32_t N, int64_t E, int32_t D, int32_t H, float dropout_p,
                             uint64_t seed, const uint64_t* seed_ptr, float* dq, int64_t dq_stride, float* dk,
                             float* dv, int64_t dkv_stride, int32_t zero_tail, int dtype, hivt_stream_t stream,
                             void* ev_start, void* ev_stop) {
  const char* what = "hivt_seg_attn_bwd";
  HIVT_REQUIRE((ev_start == nullptr) == (ev_stop == nullptr), HIVT_ERR_ARG, "%s: need both events or none", what);
  int rc = check_shape(what, N, E, D, H, kv_stride, dtype, dropout_p);
  if (rc != HIVT_OK) return rc;
  if (N == 0) return HIVT_OK;
  const bool global = k_node != nullptr;
  HIVT_REQUIRE(d_out && q && rowptr && out && m && den && dq, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(E == 0 || (k && v && dk && dv), HIVT_ERR_ARG, "%s: null k/v/dk/dv with E > 0", what);
  HIVT_REQUIRE(dkv_stride >= D && dkv_stride % 4 == 0, HIVT_ERR_ARG, "%s: dkv_stride must be >= D and 16-B aligned",
               what);
  HIVT_REQUIRE(!global || (v_node && src && node_stride >= D && node_stride % 4 == 0), HIVT_ERR_ARG,
               "%s: node terms need v_node, src and a 16-B aligned node_stride >= D", what);
  HIVT_REQUIRE(q_stride >= D && q_stride % 4 == 0 && dq_stride >= D && dq_stride % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(dq)) & 15) == 0,
               HIVT_ERR_ARG, "%s: q_stride / dq_stride must be >= D and the rows 16-B aligned", what);
  SegAttnArgs a{};
  a.q = q; a.q_stride = q_stride; a.dq_stride = dq_stride; a.k = k; a.v = v; a.kv_stride = kv_stride;
  a.k_node = k_node; a.v_node = v_node; a.node_stride = node_stride; a.src = src;
  a.rowptr = rowptr; a.row_order = row_order; a.N = N; a.H = H;
  if (dtype & HIVT_ROWS_INTERLEAVED) {
    HIVT_REQUIRE(!global && E % N == 0 && !zero_tail, HIVT_ERR_ARG,
                 "%s: interleaved rows need E a multiple of N, no node terms and no padding (E=%lld N=%d)", what,
                 (long long)E, N);
    a.row_interleave = N; a.seg_len = (int32_t)(E / N);
  }
  a.scale_inv = 1.0f / sqrtf((float)(D / H));
  a.dropout_p = dropout_p; a.seed = seed; a.seed_ptr = seed_ptr;
  a.out = const_cast<float*>(out); a.m = const_cast<float*>(m); a.den = const_cast<float*>(den);
  a.d_out = d_out; a.dq = dq; a.dk = dk; a.dv = dv; a.dkv_stride = dkv_stride;
  a.dkv_half = (dtype & HIVT_DKV_BF16) ? 1 : 0;
  HIVT_REQUIRE(!a.dkv_half || (!global && !(dtype & HIVT_ROWS_INTERLEAVED)), HIVT_ERR_UNSUPPORTED,
               "%s: bfloat16 dk / dv are built for plain rows only (no node terms, not interleaved)", what);
  HIVT_REQUIRE(!a.dkv_half || E == 0 || ((reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) & 7) == 0,
               HIVT_ERR_ARG, "%s: bfloat16 dk / dv rows must start on 8-byte boundaries", what);
  a.zero_tail = (zero_tail && E > 0) ? 1 : 0;
  a.E_cap = E;
  a.narrow = fits_32bit(E, kv_stride, 4) && fits_32bit(E, dkv_stride, a.dkv_half ? 2 : 4);
  a.ev_start = (hipEvent_t)ev_start; a.ev_stop = (hipEvent_t)ev_stop;
  HIVT_REQUIRE(dispatch(D, H, true, E, global, dtype, a, (hipStream_t)stream) == 0, HIVT_ERR_ARG,
               "%s: no kernel for D=%d H=%d (launch-shape hint 0x%x)", what, D, H, dtype >> 8);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_seg_attn_bwd(const float* d_out, const float* q, int64_t q_stride, const float* k, const float* v,
                                 int64_t kv_stride, const float* k_node, const float* v_node, int64_t node_stride,
                                 const int32_t* src, const int32_t* rowptr, const int32_t* row_order,
                                 const float* out, const float* m, const float* den, int32_t N, int64_t E, int32_t D, int32_t H, float dropout_p,
                                 uint64_t seed, const uint64_t* seed_ptr, float* dq, int64_t dq_stride, float* dk,
                                 float* dv, int64_t dkv_stride, int32_t zero_tail, int dtype, hivt_stream_t stream) {
  return seg_attn_bwd_impl(d_out, q, q_stride, k, v, kv_stride, k_node, v_node, node_stride, src, rowptr, row_order,
                           out, m, den,
                           N, E, D, H, dropout_p, seed, seed_ptr, dq, dq_stride, dk, dv, dkv_stride, zero_tail, dtype,
                           stream, nullptr, nullptr);
}

extern "C" int hivt_seg_attn_bwd_timed(const float* d_out, const float* q, int64_t q_stride, const float* k,
                                       const float* v, int64_t kv_stride, const float* k_node, const float* v_node,
                                       int64_t node_stride, const int32_t* src, const int32_t* rowptr,
                                       const int32_t* row_order, const float* out, const float* m, const float* den,
                                       int32_t N, int64_t E,
                                       int32_t D, int32_t H, float dropout_p, uint64_t seed,
                                       const uint64_t* seed_ptr, float* dq, int64_t dq_stride, float* dk, float* dv,
                                       int64_t dkv_stride, int32_t zero_tail, int dtype, hivt_stream_t stream,
                                       void* start_event, void* stop_event) {
  return seg_attn_bwd_impl(d_out, q, q_stride, k, v, kv_stride, k_node, v_node, node_stride, src, rowptr, row_order,
                           out, m, den,
                           N, E, D, H, dropout_p, seed, seed_ptr, dq, dq_stride, dk, dv, dkv_stride, zero_tail, dtype,
                           stream, start_event, stop_event);
}
