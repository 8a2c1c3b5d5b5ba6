// Dense projections of the edge-attention path on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
//   forward      Y = epilogue(X W^T + b [+ add])   with bias / residual / LayerNorm / ReLU fused
//   data grad    dX = dY W
//   weight grad  dW = dY^T X, db = colsum(dY)      split over rows, reduced in a fixed order
//   LayerNorm(+ReLU) backward for the fused forward epilogue
//
// One core serves all three products.  A block of 256 threads (4 waves, one per SIMD) owns a
// BM x BN output tile; every wave accumulates a 32 x 64 sub-tile in two 32x32 accumulators.  The
// reduction dimension is walked in steps of 32 through a double-buffered LDS image laid out
// [k][row] for both operands, so a wave's operand read is 32 consecutive floats per half-wave
// (conflict-free) and one read feeds a whole MFMA.  Global loads of step k+1 are issued before
// the MFMAs of step k and written to the other LDS buffer after them (one barrier per step).
// fp32 in, fp32 accumulate: results equal an fmaf chain over k, which keeps the 1e-5 parity bar.
#include "linear_core.h"

#include <algorithm>
#include <vector>

#include <math.h>

#include <atomic>

namespace hivt {
namespace {

// ---------------------------------------------------------------------------------------------
// forward: Y = epilogue(X W^T + bias [+ add])
// ---------------------------------------------------------------------------------------------
struct FwdArgs {
  Operand x, w;
  Tail t;
  const float* gamma;
  const float* beta;
  float eps;
  float* y;
  int64_t y_ld;
  float* z;       // pre-LayerNorm values (LN epilogues, optional)
  int64_t z_ld;
  float* stats;   // [M,2] mean, rstd (LN epilogues)
  int32_t M, N;
};

// TIGHT: registers for three workgroups per compute unit (the split-precision image of 48 KB allows three; the
// budget costs a 64-byte spill, which short reductions repay and long ones do not)
template <int BM, int BN, int EPI, bool VEC, int DEEP = 0, int PREC = PREC_F32, bool TIGHT = false>
__global__ __launch_bounds__(kThreads, TIGHT ? 3 : 1) void linear_fwd_kernel(const FwdArgs a) {
  using C = Core<BM, BN, true, true, VEC, false, DEEP, PREC>;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  int bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.y > 1) xcd_walk(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x, gridDim.y, bx, by);  // column tiles share x rows
  const int row0 = bx * BM, col0 = by * BN;
  Operand xa = a.x, wa = a.w;
  f32x16 acc0, acc1;
  const float2 bias2 = C::bias_pair(a.t, col0, a.N);
  HIVT_STAMP(0);
#ifdef HIVT_PROBE
  if (threadIdx.x == 0 && d_probe != nullptr) {  // where the block ran: HW_ID and XCC_ID registers
    long long* slot = d_probe + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;  // by dispatch id
    slot[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    slot[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  }
#endif
  C::accumulate(xa, wa, row0, col0, 0, a.x.red, lds, acc0, acc1);
  HIVT_STAMP(2);
  if (EPI == EPI_NONE || EPI == EPI_RELU) {
    C::template store_direct<EPI == EPI_RELU>(acc0, acc1, row0, col0, a.M, a.N, a.y, a.y_ld, a.t, bias2);
    HIVT_STAMP(3);
    return;
  }
  // LayerNorm over the full output row: the row is spread over the waves, so it goes through LDS;
  // a wave then owns whole rows, two at a time, with the per-column constants held in registers.
  C::spill(lds, acc0, acc1);
  constexpr int CPL = BN / 64;  // columns per lane: lane, lane + 64
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float bias[CPL], gam[CPL], bet[CPL];
  bool okc[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int gc = col0 + lane + 64 * j;
    okc[j] = gc < a.N;
    bias[j] = (okc[j] && a.t.bias != nullptr) ? a.t.bias[gc] : 0.0f;
    gam[j] = okc[j] ? a.gamma[gc] : 0.0f;
    bet[j] = okc[j] ? a.beta[gc] : 0.0f;
  }
  const float inv_n = 1.0f / (float)a.N;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool wide = BN == 128 && a.N == 128 && col0 == 0 && al16(a.y) && a.y_ld % 4 == 0 &&
                    (a.z == nullptr || (al16(a.z) && a.z_ld % 4 == 0));
  if (wide) {
    // Full 128-column rows with 16-byte aligned outputs: half a wave per row (these epilogues are bound by the
    // issue of their stores and lane exchanges; a wave per row spends two 4-byte stores per lane and output).
    // A lane first completes four adjacent columns of the row (bias, added rows) and leaves them in LDS, then
    // takes columns l, l + 32, l + 64, l + 96 for the statistics: (v[l] + v[l+64]) + (v[l+32] + v[l+96])
    // followed by the butterfly over 32 lanes is the tree a whole wave builds from v[l] + v[l+64] per lane, so
    // the statistics have the bits of the wave-per-row route below; the row then leaves with 16-byte stores.
    const int q = threadIdx.x & 31, half = threadIdx.x >> 5;
    const int c4 = 4 * q;
    float gam4[4], bet4[4], bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gam4[j] = a.gamma[c4 + j];
      bet4[j] = a.beta[c4 + j];
      bias4[j] = a.t.bias != nullptr ? a.t.bias[c4 + j] : 0.0f;
    }
    constexpr int U = 2;  // two rows per half wave in flight
#pragma unroll 1
    for (int r = half; r < BM; r += 8 * U) {
      float v[U][4], mean[U], rstd[U], part[U];
      bool okr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = r + 8 * u;
        const int64_t gr = row0 + rr;
        okr[u] = gr < a.M;
        float4 t = *reinterpret_cast<const float4*>(lds + rr * C::C_PITCH + c4);
        v[u][0] = t.x + bias4[0]; v[u][1] = t.y + bias4[1]; v[u][2] = t.z + bias4[2]; v[u][3] = t.w + bias4[3];
        if (okr[u] && a.t.add != nullptr) {
          const int64_t ar = a.t.add_index != nullptr ? min(max(a.t.add_index[gr], 0), a.t.add_rows - 1) : gr;
          const float4 ad = *reinterpret_cast<const float4*>(a.t.add + ar * a.t.add_ld + c4);
          v[u][0] += ad.x; v[u][1] += ad.y; v[u][2] += ad.z; v[u][3] += ad.w;
        }
        if (!okr[u]) v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.0f;
        *reinterpret_cast<float4*>(lds + rr * C::C_PITCH + c4) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
      }
      float s[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* row = lds + (r + 8 * u) * C::C_PITCH + q;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = row[32 * j];
        float lo = 0.0f, hi = 0.0f;
        lo += s[u][0]; lo += s[u][2];
        hi += s[u][1]; hi += s[u][3];
        part[u] = lo + hi;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) mean[u] = group_sum<32>(part[u]) * inv_n;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float lo = 0.0f, hi = 0.0f;
        { const float d = s[u][0] - mean[u]; lo += d * d; }
        { const float d = s[u][2] - mean[u]; lo += d * d; }
        { const float d = s[u][1] - mean[u]; hi += d * d; }
        { const float d = s[u][3] - mean[u]; hi += d * d; }
        part[u] = lo + hi;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) rstd[u] = rsqrtf(group_sum<32>(part[u]) * inv_n + a.eps);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!okr[u]) continue;
        const int64_t gr = row0 + r + 8 * u;
        if (q == 0 && a.stats) {
          a.stats[2 * gr] = mean[u];
          a.stats[2 * gr + 1] = rstd[u];
        }
        if (a.z) *reinterpret_cast<float4*>(a.z + gr * a.z_ld + c4) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = (v[u][j] - mean[u]) * rstd[u] * gam4[j] + bet4[j];
          if (EPI == EPI_LN_RELU) o[j] = fmaxf(o[j], 0.0f);
        }
        *reinterpret_cast<float4*>(a.y + gr * a.y_ld + c4) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    return;
  }
  for (int r = wave; r < BM; r += 8) {
    constexpr int U = 2;  // rows r and r + 4 together: two independent reduction chains
    float vals[U][CPL], sum[U];
    bool okr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = r + 4 * u, gr = row0 + rr;
      okr[u] = rr < BM && gr < a.M;
      sum[u] = 0.0f;
      const int64_t ar = (okr[u] && a.t.add != nullptr && a.t.add_index != nullptr)
                             ? min(max(a.t.add_index[gr], 0), a.t.add_rows - 1)
                             : gr;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        float v = 0.0f;
        if (okr[u] && okc[j]) {
          v = lds[rr * C::C_PITCH + lane + 64 * j] + bias[j];
          if (a.t.add) v += a.t.add[ar * a.t.add_ld + col0 + lane + 64 * j];
        }
        vals[u][j] = v;
        sum[u] += v;
      }
    }
    float mean[U], rstd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) mean[u] = wave_sum(sum[u]) * inv_n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float sq = 0.0f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const float d = okc[j] ? vals[u][j] - mean[u] : 0.0f;
        sq += d * d;
      }
      sum[u] = sq;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) rstd[u] = rsqrtf(wave_sum(sum[u]) * inv_n + a.eps);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!okr[u]) continue;
      const int64_t gr = row0 + r + 4 * u;
      if (lane == 0 && a.stats) {
        a.stats[2 * gr] = mean[u];
        a.stats[2 * gr + 1] = rstd[u];
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        if (!okc[j]) continue;
        const int gc = col0 + lane + 64 * j;
        if (a.z) a.z[gr * a.z_ld + gc] = vals[u][j];
        float o = (vals[u][j] - mean[u]) * rstd[u] * gam[j] + bet[j];
        if (EPI == EPI_LN_RELU) o = fmaxf(o, 0.0f);
        a.y[gr * a.y_ld + gc] = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// data gradient: dX[M,K] = dY[M,N] W[N,K]   (optionally masked by act > 0 of a ReLU output)
// ---------------------------------------------------------------------------------------------
struct BwdDataArgs {
  Operand dy, w;
  float* dx;
  int64_t dx_ld;
  int32_t M, K;
  const float* mask;  // optional [M,K]: dx *= mask > 0 ? mask_scale : mask < 0 ? mask_neg_scale : 0
  int64_t mask_ld;
  float mask_scale, mask_neg_scale;
  const float* add;   // optional [M,K], exclusive with mask: dx += add
  int64_t add_ld;
  int32_t dx_half;    // dx is stored as bfloat16
  int32_t mask_bits;  // mask points at one bit per element (32-bit words, mask_ld words per row), set = mask > 0
};

// HY: dy holds bfloat16 values in memory
template <int BM, int BN, bool VEC, int DEEP = 0, int PREC = PREC_F32, bool HY = false>
__global__ __launch_bounds__(kThreads) void linear_bwd_data_kernel(const BwdDataArgs a) {
  using C = Core<BM, BN, true, false, VEC, false, DEEP, PREC, HY>;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  int bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.y > 1) xcd_walk(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x, gridDim.y, bx, by);  // column tiles share dy rows
  const int row0 = bx * BM, col0 = by * BN;
  f32x16 acc0, acc1;
  C::accumulate(a.dy, a.w, row0, col0, 0, a.dy.red, lds, acc0, acc1);
  Tail t{};
  if (a.mask != nullptr && a.mask_bits) {
    t.gate_bits = reinterpret_cast<const uint32_t*>(a.mask);
    t.bits_ld = (int32_t)a.mask_ld;
    t.mask_scale = a.mask_scale;
  } else if (a.mask != nullptr) {
    t.residual = a.mask;
    t.res_ld = a.mask_ld;
    t.residual_is_mask = true;
    t.mask_scale = a.mask_scale;
    t.mask_neg_scale = a.mask_neg_scale;
  } else {
    t.residual = a.add;
    t.res_ld = a.add_ld;
  }
  t.out_half = a.dx_half != 0;
  C::template store_direct<false>(acc0, acc1, row0, col0, a.M, a.K, a.dx, a.dx_ld, t, make_float2(0.0f, 0.0f));
}

// ---------------------------------------------------------------------------------------------
// weight gradient: partial[chunk][n][k] = sum_{m in chunk} dY[m][n] X[m][k]; bias partials likewise
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  Operand dy, x;   // both [red = m][row]: dy rows = n, x rows = k
  float* partial;  // [chunks][N][K]
  float* bias_partial;  // start of chunk 0's bias sums (each chunk: N*K weight sums, then N bias sums) or null
  int64_t pitch;        // floats per chunk: N*K (+ N with a bias)
  int32_t N, K, M, chunk_rows;
  int32_t dy_half;      // dy is bfloat16 in memory (its leading dimension counts bfloat16 elements)
};

// One workgroup's share: the [BM x BN] tile (tile_n, tile_k) of the partial sums of row chunk `chunk`.
// zero_bias: the chunk's bias slots are cleared instead of filled (a term without a bias gradient that shares
// its partial buffer with one that has it).
template <int BM, int BN, bool VEC, int DEEP, int PREC, bool HY = false>
__device__ __forceinline__ void wgrad_tile(const WgradArgs& a, int tile_n, int tile_k, int chunk, float* lds,
                                           float (*bsum)[BM], bool zero_bias) {
  using C = Core<BM, BN, false, false, VEC, true, DEEP, PREC, HY>;
  const int row0 = tile_n * BM, col0 = tile_k * BN;
  const int m_begin = chunk * a.chunk_rows;
  const int m_end = min(a.M, m_begin + a.chunk_rows);
  Operand dy = a.dy, x = a.x;
  dy.red = m_end;
  x.red = m_end;
  float4 colsum;
  f32x16 acc0, acc1;
  C::accumulate(dy, x, row0, col0, m_begin, m_end, lds, acc0, acc1, &colsum);
  C::spill(lds, acc0, acc1);
  float* out = a.partial + (int64_t)chunk * a.pitch;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < BM; r += 4) {
    const int gr = row0 + r;
    if (gr >= a.N) break;
#pragma unroll
    for (int j = 0; j < BN / 64; ++j) {
      const int c = lane + 64 * j, gc = col0 + c;
      if (gc < a.K) out[(int64_t)gr * a.K + gc] = lds[r * C::C_PITCH + c];
    }
  }
  // bias gradient of this chunk = column sums of the dY rows this block staged (first column tile
  // only): thread t staged columns 4*(t % (BM/4)) .. +3 of every row it touched
  if (a.bias_partial != nullptr && tile_k == 0) {
    constexpr int QUADS = BM / 4, GROUPS = kThreads / QUADS;
    const int q = threadIdx.x % QUADS, g = threadIdx.x / QUADS;
    bsum[g][4 * q + 0] = colsum.x;
    bsum[g][4 * q + 1] = colsum.y;
    bsum[g][4 * q + 2] = colsum.z;
    bsum[g][4 * q + 3] = colsum.w;
    __syncthreads();
    if (threadIdx.x < BM && row0 + threadIdx.x < a.N) {
      float t = 0.0f;
#pragma unroll
      for (int i = 0; i < GROUPS; ++i) t += bsum[i][threadIdx.x];
      a.bias_partial[(int64_t)chunk * a.pitch + row0 + threadIdx.x] = zero_bias ? 0.0f : t;
    }
  }
}

template <int BM, int BN, bool VEC, int DEEP = 0, int PREC = PREC_F32, bool HY = false>
__global__ __launch_bounds__(kThreads) void linear_wgrad_kernel(const WgradArgs a) {
  using C = Core<BM, BN, false, false, VEC, true, DEEP, PREC>;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  __shared__ float bsum[kThreads / (BM / 4)][BM];
  // the tiles of a row chunk on one XCD (see the grouped kernel)
  const int tiles = gridDim.x * gridDim.y;
  int chunk, tile;
  xcd_walk(blockIdx.x + (blockIdx.y + blockIdx.z * gridDim.y) * gridDim.x, gridDim.z, tiles, chunk, tile);
  wgrad_tile<BM, BN, VEC, DEEP, PREC, HY>(a, tile % gridDim.x, tile / gridDim.x, chunk, lds, bsum, false);
}

// Several small weight gradients in one launch: workgroup b belongs to the term whose block range holds b
// and computes one (tile, chunk) of it exactly as linear_wgrad_kernel would.
constexpr int kGroupMax = 16;
constexpr int kGroupChunkRows = 2048;  // rows per chunk of a long term inside a group

struct WgradGroup {
  WgradArgs item[kGroupMax];
  int32_t first_block[kGroupMax + 1];
  int32_t zero_bias[kGroupMax];
  int32_t count;
};

template <int BM, int BN, int PREC>
__global__ __launch_bounds__(kThreads) void linear_wgrad_group_kernel(const WgradGroup g) {
  using C = Core<BM, BN, false, false, true, true, true, PREC>;
  __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
  __shared__ float bsum[kThreads / (BM / 4)][BM];
  int i = 0;
  while (i + 1 < g.count && g.first_block[i + 1] <= (int)blockIdx.x) ++i;
  const WgradArgs& a = g.item[i];
  const int local = (int)blockIdx.x - g.first_block[i];
  const int tiles_n = (a.N + BM - 1) / BM, tiles_k = (a.K + BN - 1) / BN;
  // The tiles of one row chunk read the same rows of both operands.  Workgroups are dealt round-robin over
  // the eight XCDs (each with an L2 of its own), so eight chunks are walked at a time, one per XCD: the tiles
  // of a chunk get workgroup ids that agree modulo 8 and find the chunk's rows in their L2 instead of fetching
  // them once per tile.  (Placement is an observation, not a contract: only speed depends on it.)
  const int tiles = tiles_n * tiles_k;
  const int chunks = (a.M + a.chunk_rows - 1) / a.chunk_rows;
  int tile, chunk;
  xcd_walk(local, chunks, tiles, chunk, tile);
  if (PREC == PREC_BF16 && a.dy_half)
    wgrad_tile<BM, BN, true, true, PREC, PREC == PREC_BF16>(a, tile % tiles_n, tile / tiles_n, chunk, lds, bsum,
                                                          g.zero_bias[i] != 0);
  else
    wgrad_tile<BM, BN, true, true, PREC>(a, tile % tiles_n, tile / tiles_n, chunk, lds, bsum, g.zero_bias[i] != 0);
}

// The reductions over the chunks of a group, in the association of reduce_rows_kernel<64, 4>: per term four
// row slices summed in ascending order and combined in slice order; the terms of an entry are added in term
// order.  One thread per output column.
constexpr int kJoinMax = 4;  // an item and up to three terms that join it

struct ReduceEntry {
  const float* term_base[kJoinMax];  // partial sums of up to kJoinMax terms, [chunks][pitch] each
  int64_t pitch;
  float* out;
  float* out1;      // columns at or beyond `split`
  int32_t cols, split;
  int32_t term_chunks[kJoinMax];  // term_chunks[t] = 0: no such term
  int32_t accumulate;
  int32_t K, ld;    // columns of a weight-gradient row and the distance between rows in `out` (ld >= K)
  int32_t vec;      // four columns per thread
};

struct ReduceGroup {
  ReduceEntry e[kGroupMax];
  int32_t first_block[kGroupMax + 1];
  int32_t count;
};

// W columns per thread (4 where every term's rows allow 16-byte loads), four chunk rows requested at a time; a
// column's sum keeps its association whatever W is.
template <int W>
__device__ __forceinline__ void reduce_entry_columns(const ReduceEntry& e, int c) {
  if (c >= e.cols) return;
  typedef float vec_t __attribute__((ext_vector_type(W)));
  float total[W];
#pragma unroll
  for (int term = 0; term < kJoinMax; ++term) {
    const int R = e.term_chunks[term];
    if (R == 0) continue;
    const float* base = e.term_base[term] + c;
    const int per = (R + 3) / 4;
    float t[W];
    for (int sl = 0; sl < 4; ++sl) {
      const int r0 = min(R, sl * per), r1 = min(R, r0 + per);
      float s[W];
#pragma unroll
      for (int j = 0; j < W; ++j) s[j] = 0.0f;
      int r = r0;
      for (; r + 4 <= r1; r += 4) {
        vec_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const vec_t*>(base + (int64_t)(r + u) * e.pitch);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < W; ++j) s[j] += v[u][j];
      }
      for (; r < r1; ++r) {
        const vec_t v = *reinterpret_cast<const vec_t*>(base + (int64_t)r * e.pitch);
#pragma unroll
        for (int j = 0; j < W; ++j) s[j] += v[j];
      }
#pragma unroll
      for (int j = 0; j < W; ++j) t[j] = sl == 0 ? s[j] : t[j] + s[j];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) total[j] = term == 0 ? t[j] : total[j] + t[j];
  }
  if (c >= e.split && e.out1 == nullptr) return;  // bias columns of an entry without a bias gradient
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int cj = c + j;  // (split is a multiple of W: the W columns lie on one side of it)
    float* dst = cj >= e.split ? e.out1 + (cj - e.split)
                               : (e.ld == e.K ? e.out + cj : e.out + (int64_t)(cj / e.K) * e.ld + cj % e.K);
    *dst = e.accumulate ? *dst + total[j] : total[j];
  }
}

__global__ __launch_bounds__(256) void reduce_group_kernel(const ReduceGroup g) {
  int i = 0;
  while (i + 1 < g.count && g.first_block[i + 1] <= (int)blockIdx.x) ++i;
  const ReduceEntry& e = g.e[i];
  const int slot = ((int)blockIdx.x - g.first_block[i]) * 256 + (int)threadIdx.x;
  if (e.vec)
    reduce_entry_columns<4>(e, slot * 4);
  else
    reduce_entry_columns<1>(e, slot);
}

// Blocks of an entry, after deciding whether its columns go four to a thread.
inline int reduce_entry_blocks(ReduceEntry& e) {
  bool vec = e.cols % 4 == 0 && e.split % 4 == 0 && e.pitch % 4 == 0;
  for (int t = 0; t < kJoinMax; ++t)
    if (e.term_chunks[t] != 0 && (reinterpret_cast<uintptr_t>(e.term_base[t]) & 15) != 0) vec = false;
  e.vec = vec ? 1 : 0;
  const int per_block = vec ? 1024 : 256;
  return (e.cols + per_block - 1) / per_block;
}

// out[c] = sum_r partial[r * pitch + c], c < C.  Fixed association: SLICES row slices are summed
// independently (ascending r inside a slice) and then combined in slice order, so the result does
// not depend on scheduling.  One block covers COLS columns; each thread sums one slice of one column.
// Columns at or beyond `split` go to out1[c - split] (two result vectors from one partial buffer).
template <int COLS, int SLICES>
__device__ __forceinline__ void reduce_rows_body(float* part /*[SLICES * COLS]*/, int col_block,
                                                 const float* __restrict__ partial, int R, int64_t pitch, int C,
                                                 float* __restrict__ out, int split, float* __restrict__ out1,
                                                 int accumulate) {
  static_assert(COLS * SLICES == 256, "one thread per (column, slice)");
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const int c = col_block * COLS + col;
  const int per = (R + SLICES - 1) / SLICES;
  const int r0 = min(R, sl * per), r1 = min(R, r0 + per);
  float s = 0.0f;
  if (c < C) {
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[(int64_t)(r + u) * pitch + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; r < r1; ++r) s += partial[(int64_t)r * pitch + c];
  }
  part[sl * COLS + col] = s;
  __syncthreads();
  if (sl == 0 && c < C) {
    float t = part[col];
#pragma unroll
    for (int i = 1; i < SLICES; ++i) t += part[i * COLS + col];
    float* dst = c < split ? out + c : out1 + (c - split);
    *dst = accumulate ? *dst + t : t;  // accumulate: the sum is added to what the output holds
  }
}

template <int COLS, int SLICES>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partial, int R, int64_t pitch,
                                                          int C, float* __restrict__ out, int split,
                                                          float* __restrict__ out1, int accumulate) {
  __shared__ float part[256];
  reduce_rows_body<COLS, SLICES>(part, (int)blockIdx.x, partial, R, pitch, C, out, split, out1, accumulate);
}

// Many such reductions in one launch (the per-block partial sums that the LayerNorm backward calls of a whole
// backward pass left behind): workgroup b serves one column block of the entry whose block range holds b, with
// the slicing launch_reduce_rows would pick for that entry, so the sums are the same bit for bit.
constexpr int kRowsGroupMax = 48;

struct RowsEntry {
  const float* partial;
  float* out;
  float* out1;
  int64_t pitch;
  int32_t R, C, split, accumulate;
};

struct RowsGroup {
  RowsEntry e[kRowsGroupMax];
  int32_t first_block[kRowsGroupMax + 1];
  int32_t count;
};

__global__ __launch_bounds__(256) void reduce_rows_group_kernel(const RowsGroup g) {
  __shared__ float part[256];
  int i = 0;
  while (i + 1 < g.count && g.first_block[i + 1] <= (int)blockIdx.x) ++i;
  const RowsEntry& e = g.e[i];
  const int cb = (int)blockIdx.x - g.first_block[i];
  if (e.R >= 64)
    reduce_rows_body<16, 16>(part, cb, e.partial, e.R, e.pitch, e.C, e.out, e.split, e.out1, e.accumulate);
  else
    reduce_rows_body<64, 4>(part, cb, e.partial, e.R, e.pitch, e.C, e.out, e.split, e.out1, e.accumulate);
}

// Many partial rows: sixteen slices of sixteen-column blocks shorten each thread's serial chain.
static void launch_reduce_rows(hipStream_t s, const float* partial, int R, int64_t pitch, int C, float* out,
                               int split = 0x7fffffff, float* out1 = nullptr, int accumulate = 0) {
  if (R >= 64)
    hipLaunchKernelGGL((reduce_rows_kernel<16, 16>), dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, partial, R, pitch,
                       C, out, split, out1, accumulate);
  else
    hipLaunchKernelGGL((reduce_rows_kernel<64, 4>), dim3((unsigned)((C + 63) / 64)), dim3(256), 0, s, partial, R, pitch,
                       C, out, split, out1, accumulate);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm (+ReLU) backward for y = [relu](gamma * (z - mean) * rstd + beta)
//   dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy' * gamma,  dy' = dy * (y > 0)
// One wave per row; per-block partial sums of dgamma = sum dy' * xhat and dbeta = sum dy'.
// ---------------------------------------------------------------------------------------------
template <int NPL>  // contiguous columns per lane (N <= 64 * NPL); lane owns columns lane*NPL .. +NPL-1
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, int64_t dy_ld,
                                                     const float* __restrict__ z, int64_t z_ld,
                                                     const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int relu, int64_t M, int N, int vec, float* __restrict__ dz,
                                                     int64_t dz_ld, float* __restrict__ partial /*[grid][2][N]*/,
                                                     const float* __restrict__ add, int64_t add_ld, int dz_half) {
  // dz_half: dz points at bfloat16 elements (dz_ld counts them); values are rounded to nearest even
  __shared__ float red[4][2][64 * NPL];
  constexpr int U = 2;  // rows in flight per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * NPL;
  float g_[NPL], b_[NPL], dg[NPL], db[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    g_[j] = c0 + j < N ? gamma[c0 + j] : 0.0f;
    b_[j] = c0 + j < N ? beta[c0 + j] : 0.0f;
    dg[j] = 0.0f;
    db[j] = 0.0f;
  }
  const float inv_n = 1.0f / (float)N;
  const bool full = vec && (c0 + NPL <= N);
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t r0 = (int64_t)blockIdx.x * 4 + wave; r0 < M; r0 += U * stride) {
    float zz[U][NPL], dd[U][NPL], mean[U], rstd[U];
    bool okr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * stride;
      okr[u] = r < M;
      const int64_t rc = okr[u] ? r : r0;
      mean[u] = stats[2 * rc];
      rstd[u] = stats[2 * rc + 1];
      if (NPL == 2 && full) {
        const float2 a = *reinterpret_cast<const float2*>(z + rc * z_ld + c0);
        const float2 b = *reinterpret_cast<const float2*>(dy + rc * dy_ld + c0);
        zz[u][0] = a.x; zz[u][NPL - 1] = a.y; dd[u][0] = b.x; dd[u][NPL - 1] = b.y;
      } else {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
          zz[u][j] = c0 + j < N ? z[rc * z_ld + c0 + j] : 0.0f;
          dd[u][j] = c0 + j < N ? dy[rc * dy_ld + c0 + j] : 0.0f;
        }
      }
    }
    float xh[U][NPL], gg[U][NPL], s1[U], s2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s1[u] = 0.0f;
      s2[u] = 0.0f;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const bool in = okr[u] && c0 + j < N;
        const float x = in ? (zz[u][j] - mean[u]) * rstd[u] : 0.0f;
        float d = in ? dd[u][j] : 0.0f;
        if (relu && !(x * g_[j] + b_[j] > 0.0f)) d = 0.0f;
        xh[u][j] = x;
        dg[j] += d * x;
        db[j] += d;
        gg[u][j] = d * g_[j];
        s1[u] += gg[u][j];
        s2[u] += gg[u][j] * x;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s1[u] = wave_sum(s1[u]) * inv_n;
      s2[u] = wave_sum(s2[u]) * inv_n;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!okr[u]) continue;
      const int64_t r = r0 + u * stride;
      float o[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) o[j] = rstd[u] * (gg[u][j] - s1[u] - xh[u][j] * s2[u]);
      if (add != nullptr) {  // gradient arriving at the un-normalised input by another route (a residual branch)
#pragma unroll
        for (int j = 0; j < NPL; ++j)
          if (c0 + j < N) o[j] += add[r * add_ld + c0 + j];
      }
      if (dz_half) {
        unsigned short* hz = reinterpret_cast<unsigned short*>(dz) + r * dz_ld + c0;
        if (NPL == 2 && full) {
          *reinterpret_cast<unsigned*>(hz) = (unsigned)to_bf16_bits(o[0]) | ((unsigned)to_bf16_bits(o[NPL - 1]) << 16);
        } else {
#pragma unroll
          for (int j = 0; j < NPL; ++j)
            if (c0 + j < N) hz[j] = to_bf16_bits(o[j]);
        }
      } else if (NPL == 2 && full) {
        *reinterpret_cast<float2*>(dz + r * dz_ld + c0) = make_float2(o[0], o[NPL - 1]);
      } else {
#pragma unroll
        for (int j = 0; j < NPL; ++j)
          if (c0 + j < N) dz[r * dz_ld + c0 + j] = o[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    red[wave][0][c0 + j] = dg[j];
    red[wave][1][c0 + j] = db[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += 256) {
    const int which = i / N, c = i % N;
    partial[((int64_t)blockIdx.x * 2 + which) * N + c] =
        ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
  }
}

// The same for N = 128 exactly and 16-byte aligned rows: HALF a wave per row (a lane owns four consecutive columns,
// one 16-byte load per operand), two rows per wave at a time and two such pairs in flight -- half the load
// instructions per row of the kernel above and one exchange step less in each of the two reductions over a row.
// Partial sums leave in the same [grid][2][N] layout.
__global__ __launch_bounds__(256) void ln_bwd128_kernel(const float* __restrict__ dy, int64_t dy_ld,
                                                        const float* __restrict__ z, int64_t z_ld,
                                                        const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int relu, int64_t M, float* __restrict__ dz, int64_t dz_ld,
                                                        float* __restrict__ partial /*[grid][2][128]*/,
                                                        const float* __restrict__ add, int64_t add_ld, int dz_half) {
  constexpr int N = 128, U = 2;
  __shared__ float red[8][2][N];  // [wave * 2 + row half][dgamma | dbeta][column]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 5, c0 = (lane & 31) * 4;
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + c0), b4 = *reinterpret_cast<const float4*>(beta + c0);
  const float g_[4] = {g4.x, g4.y, g4.z, g4.w}, b_[4] = {b4.x, b4.y, b4.z, b4.w};
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  const float inv_n = 1.0f / (float)N;
  const int64_t stride = (int64_t)gridDim.x * 8;  // rows a pass of the grid covers
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 2; base < M; base += U * stride) {
    float zz[U][4], dd[U][4], mean[U], rstd[U];
    bool okr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + sub + u * stride;
      okr[u] = r < M;
      const int64_t rc = okr[u] ? r : base;  // (base < M: a row that exists)
      mean[u] = stats[2 * rc];
      rstd[u] = stats[2 * rc + 1];
      const float4 a = *reinterpret_cast<const float4*>(z + rc * z_ld + c0);
      const float4 b = *reinterpret_cast<const float4*>(dy + rc * dy_ld + c0);
      zz[u][0] = a.x; zz[u][1] = a.y; zz[u][2] = a.z; zz[u][3] = a.w;
      dd[u][0] = b.x; dd[u][1] = b.y; dd[u][2] = b.z; dd[u][3] = b.w;
    }
    float xh[U][4], gg[U][4], s1[U], s2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s1[u] = 0.0f;
      s2[u] = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = okr[u] ? (zz[u][j] - mean[u]) * rstd[u] : 0.0f;
        float d = okr[u] ? dd[u][j] : 0.0f;
        if (relu && !(x * g_[j] + b_[j] > 0.0f)) d = 0.0f;
        xh[u][j] = x;
        dg[j] += d * x;
        db[j] += d;
        gg[u][j] = d * g_[j];
        s1[u] += gg[u][j];
        s2[u] += gg[u][j] * x;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s1[u] = group_sum<32>(s1[u]) * inv_n;
      s2[u] = group_sum<32>(s2[u]) * inv_n;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!okr[u]) continue;
      const int64_t r = base + sub + u * stride;
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rstd[u] * (gg[u][j] - s1[u] - xh[u][j] * s2[u]);
      if (add != nullptr) {  // gradient arriving at the un-normalised input by another route (a residual branch)
        const float4 e = *reinterpret_cast<const float4*>(add + r * add_ld + c0);
        o[0] += e.x; o[1] += e.y; o[2] += e.z; o[3] += e.w;
      }
      if (dz_half) {
        uint2 h;
        h.x = (unsigned)to_bf16_bits(o[0]) | ((unsigned)to_bf16_bits(o[1]) << 16);
        h.y = (unsigned)to_bf16_bits(o[2]) | ((unsigned)to_bf16_bits(o[3]) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dz) + r * dz_ld + c0) = h;
      } else {
        *reinterpret_cast<float4*>(dz + r * dz_ld + c0) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[wave * 2 + sub][0][c0 + j] = dg[j];
    red[wave * 2 + sub][1][c0 + j] = db[j];
  }
  __syncthreads();
  {  // 256 threads = 2 x 128 sums, the eight row groups added in a fixed order
    const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
    float t = red[0][which][c];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += red[q][which][c];
    partial[((int64_t)blockIdx.x * 2 + which) * N + c] = t;
  }
}

// Stand-alone LayerNorm forward: y = gamma * (x - mean) * rstd + beta, one wave per row, two rows in
// flight; mean / rstd are kept for the backward pass (two-pass variance, as in the fused epilogue).
template <int NPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t x_ld,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, int64_t M, int N, int vec, float* __restrict__ y,
                                                     int64_t y_ld, float* __restrict__ stats) {
  constexpr int U = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * NPL;
  float g_[NPL], b_[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    g_[j] = c0 + j < N ? gamma[c0 + j] : 0.0f;
    b_[j] = c0 + j < N ? beta[c0 + j] : 0.0f;
  }
  const float inv_n = 1.0f / (float)N;
  const bool full = vec && (c0 + NPL <= N);
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t r0 = (int64_t)blockIdx.x * 4 + wave; r0 < M; r0 += U * stride) {
    float v[U][NPL], s[U], mean[U];
    bool okr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * stride;
      okr[u] = r < M;
      const int64_t rc = okr[u] ? r : r0;
      if (NPL == 2 && full) {
        const float2 a = *reinterpret_cast<const float2*>(x + rc * x_ld + c0);
        v[u][0] = a.x; v[u][NPL - 1] = a.y;
      } else {
#pragma unroll
        for (int j = 0; j < NPL; ++j) v[u][j] = c0 + j < N ? x[rc * x_ld + c0 + j] : 0.0f;
      }
      s[u] = 0.0f;
#pragma unroll
      for (int j = 0; j < NPL; ++j) s[u] += v[u][j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mean[u] = wave_sum(s[u]) * inv_n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float sq = 0.0f;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const float d = c0 + j < N ? v[u][j] - mean[u] : 0.0f;
        sq += d * d;
      }
      s[u] = sq;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = rsqrtf(wave_sum(s[u]) * inv_n + eps);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!okr[u]) continue;
      const int64_t r = r0 + u * stride;
      if (lane == 0) {
        stats[2 * r] = mean[u];
        stats[2 * r + 1] = s[u];
      }
      float o[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) o[j] = (v[u][j] - mean[u]) * s[u] * g_[j] + b_[j];
      if (NPL == 2 && full) {
        *reinterpret_cast<float2*>(y + r * y_ld + c0) = make_float2(o[0], o[NPL - 1]);
      } else {
#pragma unroll
        for (int j = 0; j < NPL; ++j)
          if (c0 + j < N) y[r * y_ld + c0 + j] = o[j];
      }
    }
  }
}

// N = 128 exactly, 16-byte aligned rows: half a wave per row, four columns per lane (see ln_bwd128_kernel).
__global__ __launch_bounds__(256) void ln_fwd128_kernel(const float* __restrict__ x, int64_t x_ld,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int64_t M, float* __restrict__ y, int64_t y_ld,
                                                        float* __restrict__ stats) {
  constexpr int N = 128, U = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 5, c0 = (lane & 31) * 4;
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + c0), b4 = *reinterpret_cast<const float4*>(beta + c0);
  const float inv_n = 1.0f / (float)N;
  const int64_t stride = (int64_t)gridDim.x * 8;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 2; base < M; base += U * stride) {
    float4 v[U];
    float s[U], mean[U];
    bool okr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + sub + u * stride;
      okr[u] = r < M;
      v[u] = *reinterpret_cast<const float4*>(x + (okr[u] ? r : base) * x_ld + c0);
      s[u] = (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mean[u] = group_sum<32>(s[u]) * inv_n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float d0 = v[u].x - mean[u], d1 = v[u].y - mean[u], d2 = v[u].z - mean[u], d3 = v[u].w - mean[u];
      s[u] = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = rsqrtf(group_sum<32>(s[u]) * inv_n + eps);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!okr[u]) continue;
      const int64_t r = base + sub + u * stride;
      if ((lane & 31) == 0) {
        stats[2 * r] = mean[u];
        stats[2 * r + 1] = s[u];
      }
      *reinterpret_cast<float4*>(y + r * y_ld + c0) =
          make_float4((v[u].x - mean[u]) * s[u] * g4.x + b4.x, (v[u].y - mean[u]) * s[u] * g4.y + b4.y,
                      (v[u].z - mean[u]) * s[u] * g4.z + b4.z, (v[u].w - mean[u]) * s[u] * g4.w + b4.w);
    }
  }
}

// out[code][c] = sum over rows m with index[m] == code of x[m][c], for a handful of codes (<= 32): the gradient
// of a small embedding table whose rows were added per row of a big tensor.  Every wave owns a contiguous run of
// rows and walks it in order with ONE register accumulator for the code it is currently collecting; when the code
// changes the accumulator is added to the wave's own [code][column] table in LDS.  Rows without a code (index
// outside [0, R)) leave the current run as it is, so the common inputs -- a step's token on the padded agents of
// that step, a category per edge -- cost two additions per element or one LDS update per change of code, where one
// selected accumulator per code cost 2 R vector instructions per element.  Sums are formed in a fixed order.
constexpr int kMaxCodes = 32;

template <int NPL>
__global__ __launch_bounds__(256) void rowsum_by_code_kernel(const float* __restrict__ x, int64_t x_ld,
                                                             const int32_t* __restrict__ index, int64_t M, int N,
                                                             int R, float* __restrict__ partial /*[grid][R][N]*/) {
  extern __shared__ float table[];  // [4 waves][R][64 * NPL]
  constexpr int W = 64 * NPL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * NPL;
  float* mine = table + (size_t)wave * R * W;
  for (int i = lane; i < R * W; i += 64) mine[i] = 0.0f;
  const int64_t per_wave = (M + (int64_t)gridDim.x * 4 - 1) / ((int64_t)gridDim.x * 4);
  const int64_t first = ((int64_t)blockIdx.x * 4 + wave) * per_wave;
  const int64_t last = first + per_wave < M ? first + per_wave : M;
  float acc[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) acc[j] = 0.0f;
  int current = -1;
  auto flush = [&]() {
    if (current >= 0) {
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        mine[current * W + c0 + j] += acc[j];
        acc[j] = 0.0f;
      }
    }
  };
  constexpr int U = 4;  // rows requested together
  for (int64_t m0 = first; m0 < last; m0 += U) {
    float v[U][NPL];
    int code[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = m0 + u;
      const bool ok = m < last;
      const int64_t mc = ok ? m : m0;
      const int c = index[mc];
      code[u] = (ok && c >= 0 && c < R) ? c : -1;
#pragma unroll
      for (int j = 0; j < NPL; ++j) v[u][j] = c0 + j < N ? x[mc * x_ld + c0 + j] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = __builtin_amdgcn_readfirstlane(code[u]);  // (the same for every lane: one row, one code)
      if (c < 0) continue;
      if (c != current) {
        flush();
        current = c;
      }
#pragma unroll
      for (int j = 0; j < NPL; ++j) acc[j] += v[u][j];
    }
  }
  flush();
  __syncthreads();
  // the four waves' tables in wave order: ((w0 + w1) + w2) + w3
  for (int i = threadIdx.x; i < R * N; i += 256) {
    const int r = i / N, c = i % N;
    const float* t = table + r * W + c;
    partial[((int64_t)blockIdx.x * R + r) * N + c] =
        ((t[0] + t[(size_t)R * W]) + t[(size_t)2 * R * W]) + t[(size_t)3 * R * W];
  }
}

// Gradient through the direct epilogue's dropout (and ReLU): dz = dy * keep * 1/(1-p) [* (y > 0)].
// With a ReLU in front of the dropout, y > 0 already says "active and kept", so the mask is not
// regenerated; without one it is recomputed from (seed, row * N + col) exactly as the forward did.
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, int64_t dy_ld,
                                                          const float* __restrict__ y, int64_t y_ld, int relu,
                                                          float p, uint64_t seed_host,
                                                          const uint64_t* __restrict__ seed_ptr, int64_t M, int N,
                                                          float* __restrict__ dz, int64_t dz_ld) {
  const float scale = 1.0f / (1.0f - p);
  const uint64_t seed = seed_host + (seed_ptr != nullptr ? *seed_ptr : 0ull);
  const int64_t total = M * (int64_t)N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N;
    const int c = (int)(i - m * N);
    const float g = dy[m * dy_ld + c];
    bool keep;
    if (relu)
      keep = y[m * y_ld + c] > 0.0f;
    else
      keep = p <= 0.0f || dropout_uniform(seed, (uint64_t)i) >= p;
    dz[m * dz_ld + c] = keep ? g * scale : 0.0f;
  }
}

// The same, four consecutive columns per thread (N, the leading dimensions and the addresses allow 16-byte
// accesses); HALF_OUT: dz is bfloat16 in memory (8-byte stores), values rounded to nearest even.
template <bool HALF_OUT>
__global__ __launch_bounds__(256) void dropout_bwd_vec4_kernel(const float* __restrict__ dy, int64_t dy_ld,
                                                               const float* __restrict__ y, int64_t y_ld, int relu,
                                                               float p, uint64_t seed_host,
                                                               const uint64_t* __restrict__ seed_ptr, int64_t M, int N,
                                                               void* __restrict__ dz, int64_t dz_ld) {
  const float scale = 1.0f / (1.0f - p);
  const uint64_t seed = seed_host + (seed_ptr != nullptr ? *seed_ptr : 0ull);
  const int n4 = N / 4;
  const int64_t total = M * (int64_t)n4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / n4;
    const int c = (int)(i - m * n4) * 4;
    const float4 g = *reinterpret_cast<const float4*>(dy + m * dy_ld + c);
    bool k0, k1, k2, k3;
    if (relu) {
      const float4 t = *reinterpret_cast<const float4*>(y + m * y_ld + c);
      k0 = t.x > 0.0f; k1 = t.y > 0.0f; k2 = t.z > 0.0f; k3 = t.w > 0.0f;
    } else {
      const uint64_t e = (uint64_t)(m * N + c);
      k0 = p <= 0.0f || dropout_uniform(seed, e + 0) >= p;
      k1 = p <= 0.0f || dropout_uniform(seed, e + 1) >= p;
      k2 = p <= 0.0f || dropout_uniform(seed, e + 2) >= p;
      k3 = p <= 0.0f || dropout_uniform(seed, e + 3) >= p;
    }
    float4 r;
    r.x = k0 ? g.x * scale : 0.0f;
    r.y = k1 ? g.y * scale : 0.0f;
    r.z = k2 ? g.z * scale : 0.0f;
    r.w = k3 ? g.w * scale : 0.0f;
    if constexpr (HALF_OUT) {
      uint2 h;
      h.x = (unsigned)to_bf16_bits(r.x) | ((unsigned)to_bf16_bits(r.y) << 16);
      h.y = (unsigned)to_bf16_bits(r.z) | ((unsigned)to_bf16_bits(r.w) << 16);
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dz) + m * dz_ld + c) = h;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(dz) + m * dz_ld + c) = r;
    }
  }
}

// Operand precision of the three products: 0 = fp32 (exact), 1 = bfloat16 operands with fp32 accumulation.
// Process-wide, set through hivt_set_matmul_bf16(); read once per call on the host.
std::atomic<int> g_matmul_bf16{0};

int pick_chunks(int64_t M, int tiles) {
  // aim at ~2 blocks per CU overall, at least 256 rows per chunk
  int64_t want = (512 + tiles - 1) / tiles;
  int64_t max_by_rows = (M + 255) / 256;
  int64_t c = want < max_by_rows ? want : max_by_rows;
  if (c < 1) c = 1;
  if (c > 1024) c = 1024;
  return (int)c;
}

}  // namespace

int matmul_bf16_mask() { return g_matmul_bf16.load(std::memory_order_relaxed); }

void reduce_partials(const float* partial, int rows, int cols, float* out, hipStream_t stream, bool accumulate) {
  launch_reduce_rows(stream, partial, rows, (int64_t)cols, cols, out, 0x7fffffff, nullptr, accumulate ? 1 : 0);
}

int reduce_column_sums_group(const ColumnSums* entries, int count, hipStream_t stream) {
  static_assert(kJoinMax == 4, "ColumnSums carries four terms");
  for (int first = 0; first < count; first += kGroupMax) {
    ReduceGroup g{};
    const int n = count - first < kGroupMax ? count - first : kGroupMax;
    g.count = n;
    g.first_block[0] = 0;
    for (int i = 0; i < n; ++i) {
      const ColumnSums& c = entries[first + i];
      if (c.cols <= 0 || c.split <= 0 || c.out == nullptr || c.rows[0] <= 0 || c.term[0] == nullptr) {
        set_error("reduce_column_sums_group: entry %d: bad sizes or null pointers", first + i);
        return HIVT_ERR_ARG;
      }
      ReduceEntry& e = g.e[i];
      for (int t = 0; t < kJoinMax; ++t) {
        e.term_base[t] = c.term[t];
        e.term_chunks[t] = c.term[t] != nullptr ? c.rows[t] : 0;
      }
      e.pitch = c.pitch;
      e.out = c.out; e.out1 = c.out1;
      e.cols = c.cols; e.split = c.split;
      e.accumulate = c.accumulate;
      e.K = c.split; e.ld = c.split;
      g.first_block[i + 1] = g.first_block[i] + reduce_entry_blocks(e);
    }
    hipLaunchKernelGGL(reduce_group_kernel, dim3((unsigned)g.first_block[n]), dim3(256), 0, stream, g);
  }
  return HIVT_OK;
}

}  // namespace hivt

using namespace hivt;

extern "C" int hivt_set_matmul_bf16(int mask) {
  return g_matmul_bf16.exchange(mask & (HIVT_BF16_FWD | HIVT_BF16_DGRAD | HIVT_BF16_WGRAD | HIVT_BF16_FWD_SPLIT),
                                std::memory_order_relaxed);
}

#ifdef HIVT_PROBE
extern "C" int hivt_probe_set(long long* buffer) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(d_probe), &buffer, sizeof(buffer));
}
#endif

// gate_pos / gate_neg: with gate_pos != 0 the tensor in `residual` gates the result instead of being added
// Split precision: reductions shorter than this run at three workgroups per compute unit (13-14 % faster at
// K = 128 than at two); longer ones keep the full register budget (10 % faster at K = 512).
constexpr int kSplitLongK = 512;

static int linear_fwd_impl(const float* x, int64_t x_ld, const float* w, int64_t w_ld, const float* bias,
                           const float* add, int64_t add_ld, const int32_t* add_index, int32_t add_rows,
                           const float* gamma, const float* beta, float eps,
                           int epilogue, float neg_slope, float dropout_p, uint64_t seed,
                           const uint64_t* seed_ptr, const float* residual, int64_t res_ld, int64_t M, int32_t N,
                           int32_t K, float* y,
                           int64_t y_ld, float* z, int64_t z_ld, float* stats, hivt_stream_t stream,
                           float gate_pos, float gate_neg) {
  const char* what = "hivt_linear_fwd";
  HIVT_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && N > 0 && K > 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  const bool sign_bits = (epilogue & HIVT_EPI_SIGN_BITS) != 0;
  epilogue &= ~HIVT_EPI_SIGN_BITS;
  HIVT_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_LN_RELU, HIVT_ERR_ARG, "%s: unknown epilogue %d", what,
               epilogue);
  HIVT_REQUIRE(!sign_bits || (epilogue == EPI_RELU && stats != nullptr), HIVT_ERR_ARG,
               "%s: sign bits go with the ReLU epilogue and arrive where `stats` points", what);
  if (M == 0) return HIVT_OK;
  HIVT_REQUIRE(x && w && y, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(add_index == nullptr || (add != nullptr && add_rows >= 1), HIVT_ERR_ARG,
               "%s: add_index needs a table in add with add_rows >= 1", what);
  const bool ln = epilogue == EPI_LN || epilogue == EPI_LN_RELU;
  HIVT_REQUIRE(!ln || (gamma && beta && N <= 128), HIVT_ERR_ARG,
               "%s: LayerNorm epilogue needs gamma, beta and N <= 128 (N=%d)", what, N);
  FwdArgs a{};
  a.x = make_operand(x, x_ld, (int32_t)M, K, true);
  a.w = make_operand(w, w_ld, N, K, true);
  HIVT_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, HIVT_ERR_ARG, "%s: dropout_p must be in [0,1)", what);
  HIVT_REQUIRE(!ln || (dropout_p == 0.0f && residual == nullptr), HIVT_ERR_UNSUPPORTED,
               "%s: dropout / residual are not built for the LayerNorm epilogues", what);
  a.t.bias = bias; a.t.add = add; a.t.add_ld = add_ld; a.t.add_index = add_index; a.t.add_rows = add_rows;
  HIVT_REQUIRE(neg_slope == 1.0f || epilogue == EPI_NONE, HIVT_ERR_UNSUPPORTED,
               "%s: a LeakyReLU slope goes with the plain epilogue only", what);
  a.t.neg_slope = neg_slope;
  a.t.drop_p = dropout_p; a.t.seed = seed; a.t.seed_ptr = seed_ptr; a.t.residual = residual; a.t.res_ld = res_ld;
  if (gate_pos != 0.0f && residual != nullptr) {
    a.t.residual_is_mask = true;
    a.t.mask_scale = gate_pos;
    a.t.mask_neg_scale = gate_neg;
  }
  a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.y = y; a.y_ld = y_ld; a.z = z; a.z_ld = z_ld; a.stats = stats; a.M = (int32_t)M; a.N = N;
  if (sign_bits) {
    a.t.sign_bits = reinterpret_cast<uint32_t*>(stats);
    a.t.bits_ld = (N + 31) / 32;
    a.stats = nullptr;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool vec = a.x.vec_ok && a.w.vec_ok;
  const int mask = g_matmul_bf16.load(std::memory_order_relaxed);
  const bool split = (mask & HIVT_BF16_FWD_SPLIT) != 0;
  const bool bf16 = !split && (mask & HIVT_BF16_FWD) != 0;
  // a reduction of 128 or more: four steps in flight (the whole reduction at 128)
  const bool deep4 = HIVT_DEPTH4 != 0 && K >= 128;
  // grids that leave compute units with a single workgroup get the deep-prefetch variant
#define HIVT_FWD_CASE(BM_, BN_, E_)                                                                          \
  if (vec && split && deep4)                                                                                 \
    hipLaunchKernelGGL((linear_fwd_kernel<BM_, BN_, E_, true, 2, PREC_SPLIT>), grid, dim3(kThreads), 0, s, a); \
  else if (vec && split && K < kSplitLongK)                                                                  \
    hipLaunchKernelGGL((linear_fwd_kernel<BM_, BN_, E_, true, true, PREC_SPLIT, true>), grid, dim3(kThreads), 0, s, a); \
  else if (vec && bf16 && deep4)                                                                             \
    hipLaunchKernelGGL((linear_fwd_kernel<BM_, BN_, E_, true, 2, PREC_BF16>), grid, dim3(kThreads), 0, s, a); \
  else if (vec && split)                                                                                     \
    hipLaunchKernelGGL((linear_fwd_kernel<BM_, BN_, E_, true, true, PREC_SPLIT>), grid, dim3(kThreads), 0, s, a); \
  else if (vec && bf16) /* bf16 forward: the deep-prefetch variant wins at every grid size (13-20 % on large ones) */ \
    hipLaunchKernelGGL((linear_fwd_kernel<BM_, BN_, E_, true, true, PREC_BF16>), grid, dim3(kThreads), 0, s, a);  \
  else if (vec && grid.x * grid.y <= kSmallGrid)                                                             \
    hipLaunchKernelGGL((linear_fwd_kernel<BM_, BN_, E_, true, true>), grid, dim3(kThreads), 0, s, a);        \
  else if (vec)                                                                                              \
    hipLaunchKernelGGL((linear_fwd_kernel<BM_, BN_, E_, true>), grid, dim3(kThreads), 0, s, a);              \
  else                                                                                                       \
    hipLaunchKernelGGL((linear_fwd_kernel<BM_, BN_, E_, false>), grid, dim3(kThreads), 0, s, a);             \
  break;
  if (N <= 64) {
    dim3 grid((unsigned)((M + 127) / 128), 1);
    switch (epilogue) {
      case EPI_NONE: HIVT_FWD_CASE(128, 64, EPI_NONE)
      case EPI_RELU: HIVT_FWD_CASE(128, 64, EPI_RELU)
      case EPI_LN: HIVT_FWD_CASE(128, 64, EPI_LN)
      default: HIVT_FWD_CASE(128, 64, EPI_LN_RELU)
    }
  } else {
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 127) / 128));
    switch (epilogue) {
      case EPI_NONE: HIVT_FWD_CASE(64, 128, EPI_NONE)
      case EPI_RELU: HIVT_FWD_CASE(64, 128, EPI_RELU)
      case EPI_LN: HIVT_FWD_CASE(64, 128, EPI_LN)
      default: HIVT_FWD_CASE(64, 128, EPI_LN_RELU)
    }
  }
#undef HIVT_FWD_CASE
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_linear_fwd(const float* x, int64_t x_ld, const float* w, int64_t w_ld, const float* bias,
                               const float* add, int64_t add_ld, const int32_t* add_index, int32_t add_rows,
                               const float* gamma, const float* beta, float eps,
                               int epilogue, float neg_slope, float dropout_p, uint64_t seed,
                               const uint64_t* seed_ptr, const float* residual, int64_t res_ld, int64_t M, int32_t N,
                               int32_t K, float* y,
                               int64_t y_ld, float* z, int64_t z_ld, float* stats, hivt_stream_t stream) {
  return linear_fwd_impl(x, x_ld, w, w_ld, bias, add, add_ld, add_index, add_rows, gamma, beta, eps, epilogue, neg_slope,
                         dropout_p, seed, seed_ptr, residual, res_ld, M, N, K, y, y_ld, z, z_ld, stats, stream, 0.0f,
                         0.0f);
}

static int linear_bwd_data_impl(const float* dy, int64_t dy_ld, const float* w, int64_t w_ld, int64_t M,
                                int32_t N, int32_t K, const float* mask, int64_t mask_ld, float mask_scale,
                                float mask_neg_scale, const float* add, int64_t add_ld, float* dx, int64_t dx_ld,
                                hivt_stream_t stream, int half_flags = 0) {
  const char* what = "hivt_linear_bwd_data";
  HIVT_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && N > 0 && K > 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  if (M == 0) return HIVT_OK;
  HIVT_REQUIRE(dy && w && dx, HIVT_ERR_ARG, "%s: null pointer", what);
  const bool dy_half = (half_flags & HIVT_HALF_DY) != 0, dx_half = (half_flags & HIVT_HALF_DX) != 0;
  BwdDataArgs a{};
  a.dy = make_operand(dy, dy_ld, (int32_t)M, N, true, dy_half);
  a.w = make_operand(w, w_ld, K, N, false);  // W[n][k]: reduction index n is the row of the memory image
  a.dx = dx; a.dx_ld = dx_ld; a.M = (int32_t)M; a.K = K; a.dx_half = dx_half ? 1 : 0;
  HIVT_REQUIRE(mask == nullptr || add == nullptr, HIVT_ERR_UNSUPPORTED, "%s: mask and add are exclusive", what);
  a.mask = mask; a.mask_ld = mask_ld; a.mask_scale = mask_scale; a.mask_neg_scale = mask_neg_scale;
  a.add = add; a.add_ld = add_ld;
  a.mask_bits = (half_flags & HIVT_MASK_BITS) ? 1 : 0;
  HIVT_REQUIRE(!a.mask_bits || (mask != nullptr && mask_ld >= (K + 31) / 32), HIVT_ERR_ARG,
               "%s: a bit mask needs at least ceil(K / 32) words per row", what);
  half_flags &= ~HIVT_MASK_BITS;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = a.dy.vec_ok && a.w.vec_ok;
  const bool bf16 = (g_matmul_bf16.load(std::memory_order_relaxed) & HIVT_BF16_DGRAD) != 0;
  // bfloat16 gradients exist only between products that round their operands to bfloat16 anyway
  HIVT_REQUIRE(half_flags == 0 || (bf16 && vec), HIVT_ERR_UNSUPPORTED,
               "%s: bfloat16 dy / dx need the bf16 data-gradient mode and 8-byte aligned rows", what);
  if (dy_half) {
    if (K <= 64) {
      dim3 grid((unsigned)((M + 127) / 128), 1);
      if (grid.x <= kSmallGrid)
        hipLaunchKernelGGL((linear_bwd_data_kernel<128, 64, true, true, PREC_BF16, true>), grid, dim3(kThreads), 0, s, a);
      else
        hipLaunchKernelGGL((linear_bwd_data_kernel<128, 64, true, false, PREC_BF16, true>), grid, dim3(kThreads), 0, s, a);
    } else {
      dim3 grid((unsigned)((M + 63) / 64), (unsigned)((K + 127) / 128));
      if (grid.x * grid.y <= kSmallGrid)
        hipLaunchKernelGGL((linear_bwd_data_kernel<64, 128, true, true, PREC_BF16, true>), grid, dim3(kThreads), 0, s, a);
      else
        hipLaunchKernelGGL((linear_bwd_data_kernel<64, 128, true, false, PREC_BF16, true>), grid, dim3(kThreads), 0, s, a);
    }
    HIVT_CHECK_LAUNCH(what);
    return HIVT_OK;
  }
  if (K <= 64) {
    dim3 grid((unsigned)((M + 127) / 128), 1);
    if (vec && bf16 && grid.x <= kSmallGrid)
      hipLaunchKernelGGL((linear_bwd_data_kernel<128, 64, true, true, PREC_BF16>), grid, dim3(kThreads), 0, s, a);
    else if (vec && bf16)
      hipLaunchKernelGGL((linear_bwd_data_kernel<128, 64, true, false, PREC_BF16>), grid, dim3(kThreads), 0, s, a);
    else if (vec && grid.x <= kSmallGrid)
      hipLaunchKernelGGL((linear_bwd_data_kernel<128, 64, true, true>), grid, dim3(kThreads), 0, s, a);
    else if (vec)
      hipLaunchKernelGGL((linear_bwd_data_kernel<128, 64, true>), grid, dim3(kThreads), 0, s, a);
    else
      hipLaunchKernelGGL((linear_bwd_data_kernel<128, 64, false>), grid, dim3(kThreads), 0, s, a);
  } else {
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((K + 127) / 128));
    if (vec && bf16 && grid.x * grid.y <= kSmallGrid)
      hipLaunchKernelGGL((linear_bwd_data_kernel<64, 128, true, true, PREC_BF16>), grid, dim3(kThreads), 0, s, a);
    else if (vec && bf16)
      hipLaunchKernelGGL((linear_bwd_data_kernel<64, 128, true, false, PREC_BF16>), grid, dim3(kThreads), 0, s, a);
    else if (vec && grid.x * grid.y <= kSmallGrid)
      hipLaunchKernelGGL((linear_bwd_data_kernel<64, 128, true, true>), grid, dim3(kThreads), 0, s, a);
    else if (vec)
      hipLaunchKernelGGL((linear_bwd_data_kernel<64, 128, true>), grid, dim3(kThreads), 0, s, a);
    else
      hipLaunchKernelGGL((linear_bwd_data_kernel<64, 128, false>), grid, dim3(kThreads), 0, s, a);
  }
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_linear_bwd_data(const float* dy, int64_t dy_ld, const float* w, int64_t w_ld, int64_t M,
                                    int32_t N, int32_t K, const float* mask, int64_t mask_ld, float mask_scale,
                                    const float* add, int64_t add_ld, float* dx, int64_t dx_ld,
                                    hivt_stream_t stream) {
  return linear_bwd_data_impl(dy, dy_ld, w, w_ld, M, N, K, mask, mask_ld, mask_scale, 0.0f, add, add_ld, dx, dx_ld,
                              stream);
}

namespace hivt {

int linear_fwd_gated(const float* x, int64_t x_ld, const float* w, int64_t w_ld, int64_t M, int32_t N, int32_t K,
                     const float* gate, int64_t gate_ld, float pos_scale, float neg_scale, float* y, int64_t y_ld,
                     hivt_stream_t stream) {
  return linear_fwd_impl(x, x_ld, w, w_ld, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.0f, EPI_NONE, 1.0f,
                         0.0f, 0, nullptr, gate, gate_ld, M, N, K, y, y_ld, nullptr, 0, nullptr, stream, pos_scale,
                         neg_scale);
}

int linear_bwd_data_gated(const float* dy, int64_t dy_ld, const float* w, int64_t w_ld, int64_t M, int32_t N, int32_t K,
                          const float* gate, int64_t gate_ld, float pos_scale, float neg_scale, float* dx,
                          int64_t dx_ld, hivt_stream_t stream) {
  return linear_bwd_data_impl(dy, dy_ld, w, w_ld, M, N, K, gate, gate_ld, pos_scale, neg_scale, nullptr, 0, dx, dx_ld,
                              stream);
}

}  // namespace hivt

extern "C" int hivt_linear_bwd_data_half(const void* dy, int64_t dy_ld, const float* w, int64_t w_ld, int64_t M,
                                         int32_t N, int32_t K, const float* mask, int64_t mask_ld, float mask_scale,
                                         const float* add, int64_t add_ld, void* dx, int64_t dx_ld, int32_t half_flags,
                                         hivt_stream_t stream) {
  return linear_bwd_data_impl(reinterpret_cast<const float*>(dy), dy_ld, w, w_ld, M, N, K, mask, mask_ld, mask_scale, 0.0f,
                              add, add_ld, reinterpret_cast<float*>(dx), dx_ld, stream, half_flags);
}

extern "C" size_t hivt_linear_wgrad_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  const int tiles = ((N + 63) / 64) * ((K + 127) / 128);
  const int chunks = pick_chunks(M, tiles);
  return align_up((size_t)chunks * ((size_t)N * K + (size_t)N) * sizeof(float), 256);
}

static int linear_wgrad_impl(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, int64_t M, int32_t N,
                             int32_t K, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                             hivt_stream_t stream, bool accumulate, bool dy_half = false) {
  const char* what = "hivt_linear_wgrad";
  HIVT_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && N > 0 && K > 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  HIVT_REQUIRE(dw != nullptr, HIVT_ERR_ARG, "%s: dw is null", what);
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) {
    if (accumulate) return HIVT_OK;
    zero_words(dw, (int64_t)N * K, s);
    if (dbias) zero_words(dbias, N, s);
    HIVT_CHECK_LAUNCH(what);
    return HIVT_OK;
  }
  HIVT_REQUIRE(dy && x, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(workspace && workspace_bytes >= hivt_linear_wgrad_workspace_bytes(M, N, K), HIVT_ERR_WORKSPACE,
               "%s: workspace too small", what);
  const int tiles_n = (N + 63) / 64, tiles_k = (K + 127) / 128;
  const int chunks = pick_chunks(M, tiles_n * tiles_k);
  const int chunk_rows = (int)(((M + chunks - 1) / chunks + kBK - 1) / kBK * kBK);
  const int real_chunks = (int)((M + chunk_rows - 1) / chunk_rows);
  WgradArgs a{};
  a.dy = make_operand(dy, dy_ld, N, (int32_t)M, false, dy_half);
  a.x = make_operand(x, x_ld, K, (int32_t)M, false);
  a.dy_half = dy_half ? 1 : 0;
  a.partial = reinterpret_cast<float*>(workspace);
  a.pitch = (int64_t)N * K + (dbias ? N : 0);
  a.bias_partial = dbias ? a.partial + (size_t)N * K : nullptr;
  a.N = N; a.K = K; a.M = (int32_t)M; a.chunk_rows = chunk_rows;
  dim3 grid((unsigned)tiles_n, (unsigned)tiles_k, (unsigned)real_chunks);
  const bool bf16 = (g_matmul_bf16.load(std::memory_order_relaxed) & HIVT_BF16_WGRAD) != 0;
  HIVT_REQUIRE(!dy_half || (bf16 && a.dy.vec_ok && a.x.vec_ok), HIVT_ERR_ARG,
               "%s: a bfloat16 dy needs the bf16 weight-gradient mode and operands that allow wide loads", what);
  if (dy_half && grid.x * grid.y * grid.z <= kSmallGrid)
    hipLaunchKernelGGL((linear_wgrad_kernel<64, 128, true, true, PREC_BF16, true>), grid, dim3(kThreads), 0, s, a);
  else if (dy_half)
    hipLaunchKernelGGL((linear_wgrad_kernel<64, 128, true, false, PREC_BF16, true>), grid, dim3(kThreads), 0, s, a);
  else if (a.dy.vec_ok && a.x.vec_ok && bf16 && grid.x * grid.y * grid.z <= kSmallGrid)
    hipLaunchKernelGGL((linear_wgrad_kernel<64, 128, true, true, PREC_BF16>), grid, dim3(kThreads), 0, s, a);
  else if (a.dy.vec_ok && a.x.vec_ok && bf16)
    hipLaunchKernelGGL((linear_wgrad_kernel<64, 128, true, false, PREC_BF16>), grid, dim3(kThreads), 0, s, a);
  else if (a.dy.vec_ok && a.x.vec_ok && grid.x * grid.y * grid.z <= kSmallGrid)
    hipLaunchKernelGGL((linear_wgrad_kernel<64, 128, true, true>), grid, dim3(kThreads), 0, s, a);
  else if (a.dy.vec_ok && a.x.vec_ok)
    hipLaunchKernelGGL((linear_wgrad_kernel<64, 128, true>), grid, dim3(kThreads), 0, s, a);
  else
    hipLaunchKernelGGL((linear_wgrad_kernel<64, 128, false>), grid, dim3(kThreads), 0, s, a);
  HIVT_CHECK_LAUNCH(what);
  // one reduction over the chunks for the weight sums and (columns beyond N*K) the bias sums
  const int cnt = N * K, cols = (int)a.pitch;
  launch_reduce_rows(s, a.partial, real_chunks, a.pitch, cols, dw, cnt, dbias, accumulate ? 1 : 0);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_linear_wgrad(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, int64_t M, int32_t N,
                                 int32_t K, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                 hivt_stream_t stream) {
  return linear_wgrad_impl(dy, dy_ld, x, x_ld, M, N, K, dw, dbias, workspace, workspace_bytes, stream, false);
}

extern "C" int hivt_linear_wgrad_half(const void* dy, int64_t dy_ld, const float* x, int64_t x_ld, int64_t M, int32_t N,
                                      int32_t K, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                      int32_t half_flags, hivt_stream_t stream) {
  HIVT_REQUIRE((half_flags & ~HIVT_HALF_DY) == 0, HIVT_ERR_ARG, "hivt_linear_wgrad_half: unknown flag bits");
  return linear_wgrad_impl(reinterpret_cast<const float*>(dy), dy_ld, x, x_ld, M, N, K, dw, dbias, workspace,
                           workspace_bytes, stream, false, (half_flags & HIVT_HALF_DY) != 0);
}

namespace hivt {
int linear_wgrad_accumulate(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, int64_t M, int32_t N,
                            int32_t K, float* dw, void* workspace, size_t workspace_bytes, hivt_stream_t stream) {
  return linear_wgrad_impl(dy, dy_ld, x, x_ld, M, N, K, dw, nullptr, workspace, workspace_bytes, stream, true);
}
}  // namespace hivt

// ---------------------------------------------------------------------------------------------
// Several weight gradients per call.  items: count x 10 int64 -- dy, dy_ld, x, x_ld, M, N, K, dw, dbias (0: none),
// flags (bit 0: add to what dw / dbias hold; bit 1: this term joins the item before it: same dw, its sum is
// added to that item's, no bias term of its own; bit 2: dy is bfloat16 in memory, dy_ld in bfloat16 elements,
// bf16 weight-gradient mode only; bit 3, on a joining term: its column sums of dy count towards the item's dbias;
// bit 4: dw is one half of a [N, 2K] matrix, its rows 2K floats apart).  Terms share launches, sixteen per launch, long chunks
// first.  A term that hivt_linear_wgrad would cut into fewer than 64 row chunks gets the same chunks and the
// same order of sums here (bit-identical results); a longer one is summed in chunks of about 2048 rows (equal
// to rounding).  Terms whose operands do not allow 16-byte loads go through the single-product path one by one.
// ---------------------------------------------------------------------------------------------
namespace {

struct GroupItem {
  const float* dy; int64_t dy_ld; const float* x; int64_t x_ld; int64_t M; int32_t N, K; float* dw; float* dbias;
  bool accumulate, joins, dy_half, join_bias;
  bool wide_rows;                 // dw is one half of a [N, 2K] matrix: its rows are 2K apart
  int head;                       // index of the item a joining term adds to (its own index otherwise)
  int chunks, chunk_rows, tiles;  // of the single-product path
  bool groupable;
  size_t ws_floats;               // its share of the workspace
};

int parse_group(const char* what, const int64_t* items, int32_t count, std::vector<GroupItem>& out) {
  HIVT_REQUIRE(count >= 0 && (count == 0 || items != nullptr), HIVT_ERR_ARG, "%s: bad item table", what);
  out.resize((size_t)count);
  const bool bf16_wgrad = (g_matmul_bf16.load(std::memory_order_relaxed) & HIVT_BF16_WGRAD) != 0;
  for (int32_t i = 0; i < count; ++i) {
    const int64_t* t = items + (size_t)i * 10;
    GroupItem& g = out[(size_t)i];
    g.dy = reinterpret_cast<const float*>((uintptr_t)t[0]); g.dy_ld = t[1];
    g.x = reinterpret_cast<const float*>((uintptr_t)t[2]); g.x_ld = t[3];
    g.M = t[4]; g.N = (int32_t)t[5]; g.K = (int32_t)t[6];
    g.dw = reinterpret_cast<float*>((uintptr_t)t[7]); g.dbias = reinterpret_cast<float*>((uintptr_t)t[8]);
    g.accumulate = (t[9] & 1) != 0; g.joins = (t[9] & 2) != 0; g.dy_half = (t[9] & 4) != 0;
    g.join_bias = g.joins && (t[9] & 8) != 0;
    g.wide_rows = (t[9] & 16) != 0;
    g.head = i;
    HIVT_REQUIRE(g.M >= 0 && g.M < ((int64_t)1 << 31) && t[5] > 0 && t[6] > 0 && t[5] < ((int64_t)1 << 24) &&
                 t[6] < ((int64_t)1 << 24), HIVT_ERR_ARG, "%s: item %d: bad sizes", what, (int)i);
    HIVT_REQUIRE(g.dw != nullptr, HIVT_ERR_ARG, "%s: item %d: dw is null", what, (int)i);
    HIVT_REQUIRE(g.M == 0 || (g.dy && g.x), HIVT_ERR_ARG, "%s: item %d: null operand", what, (int)i);
    if (g.joins) {
      HIVT_REQUIRE(i > 0 && out[(size_t)i - 1].dw == g.dw && out[(size_t)i - 1].N == g.N && out[(size_t)i - 1].K == g.K,
                   HIVT_ERR_ARG, "%s: item %d joins an item with another dw or other sizes", what, (int)i);
      g.head = out[(size_t)i - 1].head;
      HIVT_REQUIRE(i - g.head < kJoinMax, HIVT_ERR_ARG, "%s: item %d: at most %d terms join one item", what, (int)i,
                   kJoinMax - 1);
      g.dbias = nullptr;
    }
    g.tiles = ((g.N + 63) / 64) * ((g.K + 127) / 128);
    g.chunks = g.M > 0 ? pick_chunks(g.M, g.tiles) : 0;
    g.chunk_rows = g.M > 0 ? (int)(((g.M + g.chunks - 1) / g.chunks + kBK - 1) / kBK * kBK) : 0;
    if (g.M > 0) g.chunks = (int)((g.M + g.chunk_rows - 1) / g.chunk_rows);
    g.ws_floats = align_up(hivt_linear_wgrad_workspace_bytes(g.M, g.N, g.K), 256) / sizeof(float);
    if (g.M > 0 && (g.chunks >= 64 || (unsigned)(g.chunks * g.tiles) > kSmallGrid)) {
      // a long term: on its own it is cut into hundreds of chunks to fill the device; inside a group the other
      // terms do that, so it takes chunks of kGroupChunkRows rows (fewer partial sums to write and add up)
      g.chunk_rows = kGroupChunkRows;
      g.chunks = (int)((g.M + g.chunk_rows - 1) / g.chunk_rows);
      if (g.chunks > 63) {
        g.chunk_rows = (int)(((g.M + 62) / 63 + kBK - 1) / kBK * kBK);
        g.chunks = (int)((g.M + g.chunk_rows - 1) / g.chunk_rows);
      }
    }
    const Operand a = make_operand(g.dy, g.dy_ld, g.N, (int32_t)g.M, false, g.dy_half);
    const Operand b = make_operand(g.x, g.x_ld, g.K, (int32_t)g.M, false);
    g.groupable = g.M > 0 && a.vec_ok && b.vec_ok;
    HIVT_REQUIRE(!g.dy_half || g.M == 0 || (g.groupable && bf16_wgrad), HIVT_ERR_ARG,
                 "%s: item %d: a bfloat16 dy needs the bf16 weight-gradient mode and operands that allow wide loads",
                 what, (int)i);
  }
  // an entry (an item and the terms that join it) goes through the grouped kernels only as a whole
  for (int32_t i = 0; i < count; ++i)
    if (out[(size_t)i].joins && !out[(size_t)i].groupable) out[(size_t)out[(size_t)i].head].groupable = false;
  for (int32_t i = 0; i < count; ++i)
    if (out[(size_t)i].joins && !out[(size_t)out[(size_t)i].head].groupable) out[(size_t)i].groupable = false;
  for (int32_t i = 0; i < count; ++i)
    HIVT_REQUIRE(!(out[(size_t)i].wide_rows && !out[(size_t)i].groupable), HIVT_ERR_UNSUPPORTED,
                 "%s: item %d: a gradient with wide rows needs a non-empty term whose operands allow 16-byte loads", what,
                 (int)i);
  for (int32_t i = 0; i < count; ++i)
    HIVT_REQUIRE(!(out[(size_t)i].join_bias && !out[(size_t)i].groupable && out[(size_t)i].M > 0), HIVT_ERR_UNSUPPORTED,
                 "%s: item %d: a joining term with bias sums needs operands that allow 16-byte loads", what, (int)i);
  return HIVT_OK;
}

}  // namespace

extern "C" size_t hivt_linear_wgrad_group_workspace_bytes(const int64_t* items, int32_t count) {
  std::vector<GroupItem> g;
  if (parse_group("hivt_linear_wgrad_group_workspace_bytes", items, count, g) != HIVT_OK) return 0;
  size_t floats = 64;
  for (const GroupItem& it : g) floats += it.ws_floats;
  return floats * sizeof(float);
}

extern "C" int hivt_linear_wgrad_group(const int64_t* items, int32_t count, void* workspace, size_t workspace_bytes,
                                       hivt_stream_t stream) {
  const char* what = "hivt_linear_wgrad_group";
  std::vector<GroupItem> g;
  { const int rc = parse_group(what, items, count, g); if (rc != HIVT_OK) return rc; }
  if (count == 0) return HIVT_OK;
  size_t need = 64;
  for (const GroupItem& it : g) need += it.ws_floats;
  HIVT_REQUIRE(workspace && workspace_bytes >= need * sizeof(float), HIVT_ERR_WORKSPACE, "%s: workspace too small", what);
  hipStream_t s = (hipStream_t)stream;
  const bool bf16 = (g_matmul_bf16.load(std::memory_order_relaxed) & HIVT_BF16_WGRAD) != 0;
  float* ws = reinterpret_cast<float*>(workspace);
  std::vector<float*> region((size_t)count);
  {
    size_t off = 0;
    for (int32_t i = 0; i < count; ++i) { region[(size_t)i] = ws + off; off += g[(size_t)i].ws_floats; }
  }
  // terms outside the grouped kernels' range: the single-product path, in item order
  for (int32_t i = 0; i < count; ++i) {
    const GroupItem& it = g[(size_t)i];
    if (it.groupable) continue;
    const int rc = linear_wgrad_impl(it.dy, it.dy_ld, it.x, it.x_ld, it.M, it.N, it.K, it.dw, it.dbias, region[(size_t)i],
                                     it.ws_floats * sizeof(float), stream, it.accumulate || it.joins, it.dy_half);
    if (rc != HIVT_OK) return rc;
  }
  // grouped: the products, sixteen terms per launch ...
  WgradGroup wg{};
  auto flush_products = [&]() {
    if (wg.count == 0) return;
    const unsigned blocks = (unsigned)wg.first_block[wg.count];
    if (bf16)
      hipLaunchKernelGGL((linear_wgrad_group_kernel<64, 128, PREC_BF16>), dim3(blocks), dim3(kThreads), 0, s, wg);
    else
      hipLaunchKernelGGL((linear_wgrad_group_kernel<64, 128, PREC_F32>), dim3(blocks), dim3(kThreads), 0, s, wg);
    wg = WgradGroup{};
  };
  // launch order: long chunks first, so that workgroups of similar duration share a launch (a launch lasts as
  // long as its longest workgroup); a joining term stays behind its item
  std::vector<int32_t> order;
  for (int32_t i = 0; i < count; ++i)
    if (g[(size_t)i].groupable && !g[(size_t)i].joins) order.push_back(i);
  std::stable_sort(order.begin(), order.end(),
                   [&](int32_t a, int32_t b) { return g[(size_t)a].chunk_rows > g[(size_t)b].chunk_rows; });
  auto add_product = [&](int32_t i) {
    const GroupItem& it = g[(size_t)i];
    // a joining term lays its chunks out like the item it joins (bias slots included, cleared)
    const GroupItem& head = g[(size_t)it.head];
    const int64_t pitch = (int64_t)head.N * head.K + (head.dbias ? head.N : 0);
    float* base = region[(size_t)i];
    WgradArgs a{};
    a.dy = make_operand(it.dy, it.dy_ld, it.N, (int32_t)it.M, false, it.dy_half);
    a.x = make_operand(it.x, it.x_ld, it.K, (int32_t)it.M, false);
    a.dy_half = it.dy_half ? 1 : 0;
    a.partial = base;
    a.pitch = pitch;
    a.bias_partial = head.dbias ? base + (size_t)it.N * it.K : nullptr;
    a.N = it.N; a.K = it.K; a.M = (int32_t)it.M; a.chunk_rows = it.chunk_rows;
    wg.item[wg.count] = a;
    wg.zero_bias[wg.count] = (it.joins && !it.join_bias) ? 1 : 0;
    wg.first_block[wg.count + 1] = wg.first_block[wg.count] + it.tiles * it.chunks;
    if (++wg.count == kGroupMax) flush_products();
  };
  for (int32_t i : order) {
    add_product(i);
    for (int32_t j = i + 1; j < count && g[(size_t)j].joins; ++j) add_product(j);
  }
  flush_products();
  HIVT_CHECK_LAUNCH(what);
  // ... and the sums over their chunks
  ReduceGroup rg{};
  auto flush_sums = [&]() {
    if (rg.count == 0) return;
    hipLaunchKernelGGL(reduce_group_kernel, dim3((unsigned)rg.first_block[rg.count]), dim3(256), 0, s, rg);
    rg = ReduceGroup{};
  };
  for (int32_t i = 0; i < count; ++i) {
    const GroupItem& it = g[(size_t)i];
    if (!it.groupable || it.joins) continue;
    ReduceEntry e{};
    e.term_base[0] = region[(size_t)i];
    e.term_chunks[0] = it.chunks;
    for (int32_t j = i + 1, t = 1; j < count && g[(size_t)j].joins; ++j, ++t) {
      e.term_base[t] = region[(size_t)j];
      e.term_chunks[t] = g[(size_t)j].chunks;
    }
    e.pitch = (int64_t)it.N * it.K + (it.dbias ? it.N : 0);
    e.out = it.dw; e.out1 = it.dbias;
    e.cols = (int32_t)e.pitch; e.split = it.N * it.K;
    e.K = it.K; e.ld = it.wide_rows ? 2 * it.K : it.K;
    e.accumulate = it.accumulate ? 1 : 0;
    rg.first_block[rg.count + 1] = rg.first_block[rg.count] + reduce_entry_blocks(e);
    rg.e[rg.count] = e;
    if (++rg.count == kGroupMax) flush_sums();
  }
  flush_sums();
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

// Workgroups of the LayerNorm-backward kernel = rows of its partial-sum buffer: enough waves to keep
// HBM busy on long inputs, few enough that the reduction over the partial rows stays short.
static int64_t ln_bwd_blocks(int64_t M) {
  const int64_t cap = M >= 40000 ? 1024 : 512;
  int64_t blocks = (M + 3) / 4;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return blocks;
}

extern "C" size_t hivt_ln_bwd_workspace_bytes(int64_t M, int32_t N) {
  const int64_t blocks = ln_bwd_blocks(M);
  return align_up((size_t)blocks * 2 * (size_t)N * sizeof(float), 256);
}

static int ln_bwd_impl(const float* dy, int64_t dy_ld, const float* z, int64_t z_ld, const float* stats,
                       const float* gamma, const float* beta, int relu, int64_t M, int32_t N, const float* add,
                       int64_t add_ld, float* dz, int64_t dz_ld, float* dgamma, float* dbeta, void* workspace,
                       size_t workspace_bytes, hivt_stream_t stream, bool dz_half) {
  const char* what = "hivt_ln_bwd";
  HIVT_REQUIRE(M >= 0 && N > 0 && N <= 128, HIVT_ERR_ARG, "%s: N must be in [1,128] (N=%d)", what, N);
  HIVT_REQUIRE((dgamma != nullptr) == (dbeta != nullptr), HIVT_ERR_ARG, "%s: dgamma and dbeta go together", what);
  const bool leave_partials = dgamma == nullptr;  // the caller sums workspace[hivt_ln_bwd_partial_rows(M)][2][N] itself
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) {
    if (!leave_partials) {
      zero_words(dgamma, N, s);
      zero_words(dbeta, N, s);
      HIVT_CHECK_LAUNCH(what);
    }
    return HIVT_OK;
  }
  HIVT_REQUIRE(dy && z && stats && gamma && beta && dz, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(workspace && workspace_bytes >= hivt_ln_bwd_workspace_bytes(M, N), HIVT_ERR_WORKSPACE,
               "%s: workspace too small", what);
  const int64_t blocks = ln_bwd_blocks(M);
  float* partial = reinterpret_cast<float*>(workspace);
  auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  // (a bfloat16 dz: a lane's two columns are one 4-byte store, which the requirement below keeps aligned)
  HIVT_REQUIRE(!dz_half || (al8(dz) && dz_ld % 4 == 0 && N % 4 == 0), HIVT_ERR_ARG,
               "%s: a bfloat16 dz needs N and dz_ld that are multiples of 4 and rows on 8-byte boundaries", what);
  const int vec = (al8(dy) && al8(z) && al8(dz) && dy_ld % 2 == 0 && z_ld % 2 == 0 && dz_ld % 2 == 0) ? 1 : 0;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool wide = N == 128 && al16(dy) && al16(z) && al16(gamma) && al16(beta) && dy_ld % 4 == 0 && z_ld % 4 == 0 &&
                    dz_ld % 4 == 0 && (dz_half ? al8(dz) : al16(dz)) && (add == nullptr || (al16(add) && add_ld % 4 == 0));
  if (wide)
    hipLaunchKernelGGL(ln_bwd128_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, dy_ld, z, z_ld, stats, gamma, beta,
                       relu, M, dz, dz_ld, partial, add, add_ld, dz_half ? 1 : 0);
  else if (N <= 64)
    hipLaunchKernelGGL((ln_bwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, dy, dy_ld, z, z_ld, stats, gamma,
                       beta, relu, M, N, vec, dz, dz_ld, partial, add, add_ld, dz_half ? 1 : 0);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, dy, dy_ld, z, z_ld, stats, gamma,
                       beta, relu, M, N, vec, dz, dz_ld, partial, add, add_ld, dz_half ? 1 : 0);
  HIVT_CHECK_LAUNCH(what);
  if (leave_partials) return HIVT_OK;
  // partial is [blocks][2][N]: one reduction over the 2N columns, first half -> dgamma, second -> dbeta
  launch_reduce_rows(s, partial, (int)blocks, (int64_t)2 * N, 2 * N, dgamma, N, dbeta);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_ln_bwd(const float* dy, int64_t dy_ld, const float* z, int64_t z_ld, const float* stats,
                           const float* gamma, const float* beta, int relu, int64_t M, int32_t N, const float* add,
                           int64_t add_ld, float* dz, int64_t dz_ld, float* dgamma, float* dbeta, void* workspace,
                           size_t workspace_bytes, hivt_stream_t stream) {
  return ln_bwd_impl(dy, dy_ld, z, z_ld, stats, gamma, beta, relu, M, N, add, add_ld, dz, dz_ld, dgamma, dbeta, workspace,
                     workspace_bytes, stream, false);
}

extern "C" int hivt_ln_bwd_half(const float* dy, int64_t dy_ld, const float* z, int64_t z_ld, const float* stats,
                                const float* gamma, const float* beta, int relu, int64_t M, int32_t N, const float* add,
                                int64_t add_ld, void* dz, int64_t dz_ld, float* dgamma, float* dbeta, void* workspace,
                                size_t workspace_bytes, hivt_stream_t stream) {
  return ln_bwd_impl(dy, dy_ld, z, z_ld, stats, gamma, beta, relu, M, N, add, add_ld, reinterpret_cast<float*>(dz), dz_ld,
                     dgamma, dbeta, workspace, workspace_bytes, stream, true);
}

extern "C" size_t hivt_ln_bwd_partial_rows(int64_t M) { return M > 0 ? (size_t)ln_bwd_blocks(M) : 0; }

extern "C" int hivt_reduce_rows_group(const int64_t* items, int32_t count, hivt_stream_t stream) {
  const char* what = "hivt_reduce_rows_group";
  HIVT_REQUIRE(count >= 0 && (count == 0 || items != nullptr), HIVT_ERR_ARG, "%s: bad item list", what);
  hipStream_t s = (hipStream_t)stream;
  for (int32_t base = 0; base < count; base += kRowsGroupMax) {
    RowsGroup g{};
    g.count = std::min<int32_t>(kRowsGroupMax, count - base);
    int32_t blocks = 0;
    for (int32_t i = 0; i < g.count; ++i) {
      const int64_t* it = items + (int64_t)(base + i) * 8;
      RowsEntry& e = g.e[i];
      e.partial = reinterpret_cast<const float*>(it[0]);
      e.R = (int32_t)it[1];
      e.pitch = it[2];
      e.C = (int32_t)it[3];
      e.out = reinterpret_cast<float*>(it[4]);
      e.split = (int32_t)it[5];
      e.out1 = reinterpret_cast<float*>(it[6]);
      e.accumulate = (int32_t)it[7];
      HIVT_REQUIRE(e.R >= 0 && e.C > 0 && e.pitch >= e.C && e.out != nullptr && (e.R == 0 || e.partial != nullptr) &&
                       e.split >= 0 && (e.split >= e.C || e.out1 != nullptr),
                   HIVT_ERR_ARG, "%s: item %d is malformed", what, base + i);
      g.first_block[i] = blocks;
      blocks += e.R >= 64 ? (e.C + 15) / 16 : (e.C + 63) / 64;
    }
    g.first_block[g.count] = blocks;
    hipLaunchKernelGGL(reduce_rows_group_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g);
    HIVT_CHECK_LAUNCH(what);
  }
  return HIVT_OK;
}

extern "C" int hivt_ln_fwd(const float* x, int64_t x_ld, const float* gamma, const float* beta, float eps, int64_t M,
                           int32_t N, float* y, int64_t y_ld, float* stats, hivt_stream_t stream) {
  const char* what = "hivt_ln_fwd";
  HIVT_REQUIRE(M >= 0 && N > 0 && N <= 128, HIVT_ERR_ARG, "%s: N must be in [1,128] (N=%d)", what, N);
  if (M == 0) return HIVT_OK;
  HIVT_REQUIRE(x && gamma && beta && y && stats, HIVT_ERR_ARG, "%s: null pointer", what);
  int64_t blocks = (M + 7) / 8;
  if (blocks > 2048) blocks = 2048;
  auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  const int vec = (al8(x) && al8(y) && x_ld % 2 == 0 && y_ld % 2 == 0) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (N == 128 && al16(x) && al16(y) && al16(gamma) && al16(beta) && x_ld % 4 == 0 && y_ld % 4 == 0)
    hipLaunchKernelGGL(ln_fwd128_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, x_ld, gamma, beta, eps, M, y, y_ld,
                       stats);
  else if (N <= 64)
    hipLaunchKernelGGL((ln_fwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, x, x_ld, gamma, beta, eps, M, N, vec,
                       y, y_ld, stats);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, x, x_ld, gamma, beta, eps, M, N, vec,
                       y, y_ld, stats);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" size_t hivt_rowsum_by_code_workspace_bytes(int64_t M, int32_t N, int32_t R) {
  int64_t blocks = (M + 7) / 8;
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  return align_up((size_t)blocks * (size_t)R * (size_t)N * sizeof(float), 256);
}

extern "C" int hivt_rowsum_by_code(const float* x, int64_t x_ld, const int32_t* index, int64_t M, int32_t N,
                                   int32_t R, float* out, void* workspace, size_t workspace_bytes,
                                   hivt_stream_t stream) {
  const char* what = "hivt_rowsum_by_code";
  HIVT_REQUIRE(M >= 0 && N > 0 && N <= 128 && R > 0 && R <= kMaxCodes, HIVT_ERR_ARG,
               "%s: need N in [1,128] and at most %d codes (N=%d, R=%d)", what, kMaxCodes, N, R);
  HIVT_REQUIRE(out != nullptr, HIVT_ERR_ARG, "%s: out is null", what);
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) {
    zero_words(out, (int64_t)R * N, s);
    HIVT_CHECK_LAUNCH(what);
    return HIVT_OK;
  }
  HIVT_REQUIRE(x && index, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(workspace && workspace_bytes >= hivt_rowsum_by_code_workspace_bytes(M, N, R), HIVT_ERR_WORKSPACE,
               "%s: workspace too small", what);
  int64_t blocks = (M + 7) / 8;
  if (blocks > 512) blocks = 512;
  float* partial = reinterpret_cast<float*>(workspace);
  if (N <= 64)
    hipLaunchKernelGGL((rowsum_by_code_kernel<1>), dim3((unsigned)blocks), dim3(256), (size_t)4 * R * 64 * sizeof(float), s,
                       x, x_ld, index, M, N, R, partial);
  else
    hipLaunchKernelGGL((rowsum_by_code_kernel<2>), dim3((unsigned)blocks), dim3(256), (size_t)4 * R * 128 * sizeof(float),
                       s, x, x_ld, index, M, N, R, partial);
  HIVT_CHECK_LAUNCH(what);
  reduce_partials(partial, (int)blocks, R * N, out, s);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

static int dropout_bwd_impl(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, int relu, float dropout_p,
                            uint64_t seed, const uint64_t* seed_ptr, int64_t M, int32_t N, void* dz, int64_t dz_ld,
                            bool half_out, hivt_stream_t stream) {
  const char* what = "hivt_dropout_bwd";
  HIVT_REQUIRE(M >= 0 && N > 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  HIVT_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, HIVT_ERR_ARG, "%s: dropout_p must be in [0,1)", what);
  if (M == 0) return HIVT_OK;
  HIVT_REQUIRE(dy && dz && (!relu || y), HIVT_ERR_ARG, "%s: null pointer", what);
  const bool wide = N % 4 == 0 && dy_ld % 4 == 0 && dz_ld % 4 == 0 && aligned16(dy) &&
                    (half_out ? (reinterpret_cast<uintptr_t>(dz) & 7) == 0 : aligned16(dz)) &&
                    (!relu || (y_ld % 4 == 0 && aligned16(y)));
  HIVT_REQUIRE(wide || !half_out, HIVT_ERR_ARG,
               "%s: a bfloat16 dz needs a column count and leading dimensions that are multiples of 4 and aligned rows",
               what);
  int64_t blocks = (M * (int64_t)N / (wide ? 4 : 1) + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (wide && half_out)
    hipLaunchKernelGGL(dropout_bwd_vec4_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, dy_ld,
                       y, y_ld, relu, dropout_p, seed, seed_ptr, M, N, dz, dz_ld);
  else if (wide)
    hipLaunchKernelGGL(dropout_bwd_vec4_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, dy_ld,
                       y, y_ld, relu, dropout_p, seed, seed_ptr, M, N, dz, dz_ld);
  else
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, dy_ld, y, y_ld,
                       relu, dropout_p, seed, seed_ptr, M, N, reinterpret_cast<float*>(dz), dz_ld);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_dropout_bwd(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, int relu, float dropout_p,
                                uint64_t seed, const uint64_t* seed_ptr, int64_t M, int32_t N, float* dz,
                                int64_t dz_ld, hivt_stream_t stream) {
  return dropout_bwd_impl(dy, dy_ld, y, y_ld, relu, dropout_p, seed, seed_ptr, M, N, dz, dz_ld, false, stream);
}

extern "C" int hivt_dropout_bwd_half(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, int relu,
                                     float dropout_p, uint64_t seed, const uint64_t* seed_ptr, int64_t M, int32_t N,
                                     void* dz, int64_t dz_ld, hivt_stream_t stream) {
  return dropout_bwd_impl(dy, dy_ld, y, y_ld, relu, dropout_p, seed, seed_ptr, M, N, dz, dz_ld, true, stream);
}
