// Edge construction for the HiVT edge-attention path: order-preserving edge filters (validity
// sub-graph + radius), stable CSR build, and the atomics-free segment sum of rows.
//
// All of these are integer / byte work bound by HBM and launch latency; nothing here is shaped
// into a GEMM.  The filters are a three-launch stream compaction (count per 1024-item tile, one
// single-block scan of the tile counts, ordered write) so the output keeps the input order, which
// fixes the CSR order and therefore the floating-point accumulation order downstream.
#include "common.h"

#include <mutex>

namespace hivt {
namespace {

constexpr int kFilterThreads = 256;
constexpr int kFilterItemsPerThread = 4;
constexpr int kFilterTile = kFilterThreads * kFilterItemsPerThread;

struct EdgeRec {
  int32_t src, dst;
  float ax, ay;
};

// Predicate for per-timestep agent-agent edges (models/local_encoder.py:58-61 + utils.py:69-78).
struct TemporalPred {
  const int64_t* edge_index;
  int64_t E0;
  const uint8_t* pm;
  int64_t pm_stride;
  const float* pos;
  int64_t pos_sn, pos_st;
  int32_t N, t0;
  float radius;

  __device__ __forceinline__ bool operator()(int64_t item, EdgeRec& r) const {
    const int32_t tt = (int32_t)(item / E0);
    const int64_t e = item - (int64_t)tt * E0;
    const int32_t t = t0 + tt;
    const int64_t s = edge_index[e];
    const int64_t d = edge_index[E0 + e];
    if (pm[s * pm_stride + t] | pm[d * pm_stride + t]) return false;
    const float2 ps = *reinterpret_cast<const float2*>(pos + s * pos_sn + (int64_t)t * pos_st);
    const float2 pd = *reinterpret_cast<const float2*>(pos + d * pos_sn + (int64_t)t * pos_st);
    r.ax = ps.x - pd.x;
    r.ay = ps.y - pd.y;
    r.src = (int32_t)s + tt * N;
    r.dst = (int32_t)d + tt * N;
    if (radius > 0.0f) {
      // same rounding as the host framework's 2-norm over a pair: sqrt(fma(y, y, fl(x*x)))
      const float d2 = __fmaf_rn(r.ay, r.ay, __fmul_rn(r.ax, r.ax));
      return __fsqrt_rn(d2) < radius;
    }
    return true;
  }
};

// Predicate for a given edge list with given attributes (utils.py:69-78 at local_encoder.py:80).
struct RadiusPred {
  const int64_t* edge_index;
  int64_t E0;
  const float* attr;
  float radius;

  __device__ __forceinline__ bool operator()(int64_t item, EdgeRec& r) const {
    const float2 a = *reinterpret_cast<const float2*>(attr + 2 * item);
    r.ax = a.x;
    r.ay = a.y;
    r.src = (int32_t)edge_index[item];
    r.dst = (int32_t)edge_index[E0 + item];
    if (radius > 0.0f) {
      const float d2 = __fmaf_rn(a.y, a.y, __fmul_rn(a.x, a.x));
      return __fsqrt_rn(d2) < radius;
    }
    return true;
  }
};

template <class Pred>
__global__ __launch_bounds__(kFilterThreads) void filter_count_kernel(Pred pred, int64_t items,
                                                                      int32_t* __restrict__ tile_counts) {
  __shared__ int wave_tot[kFilterThreads / kWave];
  const int64_t base = (int64_t)blockIdx.x * kFilterTile;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < kFilterItemsPerThread; ++j) {
    const int64_t item = base + j * kFilterThreads + threadIdx.x;
    EdgeRec r;
    if (item < items && pred(item, r)) ++cnt;
  }
  cnt = wave_sum_i(cnt);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
#pragma unroll
    for (int w = 0; w < kFilterThreads / kWave; ++w) s += wave_tot[w];
    tile_counts[blockIdx.x] = s;
  }
}

// Exclusive scan of `n` int32 values by ONE block; also writes the grand total.
__global__ __launch_bounds__(1024) void scan_single_block_kernel(const int32_t* __restrict__ in, int64_t n,
                                                                 int32_t* __restrict__ out,
                                                                 int32_t* __restrict__ total_out) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t lo = (int64_t)tid * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  int s = 0;
  for (int64_t i = lo; i < hi; ++i) s += in[i];
  part[tid] = s;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 1024 partials
  for (int o = 1; o < 1024; o <<= 1) {
    int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = tid == 0 ? 0 : part[tid - 1];
  for (int64_t i = lo; i < hi; ++i) {
    const int v = in[i];
    out[i] = run;
    run += v;
  }
  if (tid == 1023 && total_out != nullptr) *total_out = part[1023];
}

// The same scan for the row lengths of a CSR build, followed by a ranking of the rows by length, longest first
// (`order` [n]: a permutation of the rows; rows of one length class in no particular order).  A kernel that gives
// every row a workgroup and walks the rows in this order starts its long rows first, so that the launch does not
// end on a few of them (the order changes nothing but the schedule).  Classes of two lengths, 512 of them.
constexpr int kLengthClasses = 512;
__global__ __launch_bounds__(1024) void scan_rank_single_block_kernel(const int32_t* __restrict__ in, int64_t n,
                                                                      int32_t* __restrict__ out,
                                                                      int32_t* __restrict__ total_out,
                                                                      int32_t* __restrict__ order) {
  __shared__ int part[1024];
  __shared__ int cls[kLengthClasses];
  const int tid = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t lo = (int64_t)tid * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  auto class_of = [](int len) { return kLengthClasses - 1 - min(len >> 1, kLengthClasses - 1); };  // longest = class 0
  for (int c = tid; c < kLengthClasses; c += 1024) cls[c] = 0;
  int s = 0;
  for (int64_t i = lo; i < hi; ++i) s += in[i];
  part[tid] = s;
  __syncthreads();
  for (int64_t i = lo; i < hi; ++i) atomicAdd(&cls[class_of(in[i])], 1);
  for (int o = 1; o < 1024; o <<= 1) {
    int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = tid == 0 ? 0 : part[tid - 1];
  for (int64_t i = lo; i < hi; ++i) {
    const int v = in[i];
    out[i] = run;
    run += v;
  }
  if (tid == 1023 && total_out != nullptr) *total_out = part[1023];
  __syncthreads();
  // exclusive scan of the class counts by one wave (eight classes per lane), then every row takes the next slot of
  // its class
  if (tid < kWave) {
    constexpr int PER = kLengthClasses / kWave;
    int c[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      c[j] = cls[tid * PER + j];
      sum += c[j];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int v = __shfl_up(incl, o, kWave);
      if (tid >= o) incl += v;
    }
    int first = incl - sum;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      cls[tid * PER + j] = first;  // first slot of the class
      first += c[j];
    }
  }
  __syncthreads();
  for (int64_t i = lo; i < hi; ++i) order[atomicAdd(&cls[class_of(in[i])], 1)] = (int32_t)i;
}

template <class Pred>
__global__ __launch_bounds__(kFilterThreads) void filter_write_kernel(Pred pred, int64_t items,
                                                                      const int32_t* __restrict__ tile_offsets,
                                                                      int32_t* __restrict__ src_out,
                                                                      int32_t* __restrict__ dst_out,
                                                                      float* __restrict__ attr_out) {
  __shared__ int wave_tot[kFilterThreads / kWave];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kFilterTile;
  int running = tile_offsets[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kFilterItemsPerThread; ++j) {
    const int64_t item = base + j * kFilterThreads + threadIdx.x;
    EdgeRec r;
    const bool keep = item < items && pred(item, r);
    const unsigned long long ballot = __ballot(keep);
    const int lane_prefix = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(ballot);
    __syncthreads();
    int wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kFilterThreads / kWave; ++w) {
      const int c = wave_tot[w];
      if (w < wave) wave_off += c;
      total += c;
    }
    if (keep) {
      const int64_t p = (int64_t)running + wave_off + lane_prefix;
      src_out[p] = r.src;
      dst_out[p] = r.dst;
      *reinterpret_cast<float2*>(attr_out + 2 * p) = make_float2(r.ax, r.ay);
    }
    running += total;
    __syncthreads();
  }
}

template <class Pred>
int run_filter(const Pred& pred, int64_t items, int32_t* src_out, int32_t* dst_out, float* attr_out,
               int32_t* count_out, void* ws, size_t ws_bytes, hipStream_t stream, const char* what) {
  if (items == 0) {
    zero_words(count_out, 1, stream);
    HIVT_CHECK_LAUNCH(what);
    return HIVT_OK;
  }
  const int64_t tiles = (items + kFilterTile - 1) / kFilterTile;
  HIVT_REQUIRE(ws != nullptr && ws_bytes >= hivt_edges_workspace_bytes(items), HIVT_ERR_WORKSPACE,
               "%s: workspace too small (%zu < %zu)", what, ws_bytes, hivt_edges_workspace_bytes(items));
  int32_t* tile_counts = reinterpret_cast<int32_t*>(ws);
  int32_t* tile_offsets = tile_counts + align_up(tiles, 64);
  hipLaunchKernelGGL(filter_count_kernel<Pred>, dim3((unsigned)tiles), dim3(kFilterThreads), 0, stream, pred, items,
                     tile_counts);
  HIVT_CHECK_LAUNCH(what);
  hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(1024), 0, stream, tile_counts, tiles, tile_offsets,
                     count_out);
  HIVT_CHECK_LAUNCH(what);
  hipLaunchKernelGGL(filter_write_kernel<Pred>, dim3((unsigned)tiles), dim3(kFilterThreads), 0, stream, pred, items,
                     tile_offsets, src_out, dst_out, attr_out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

// ---- all-pairs edges straight into CSR -------------------------------------------------------
// The reference's candidate list is every ordered pair of agents of a scene (nuscenes_dataset.py:139-141),
// so the filtered, target-grouped edge set of a timestep can be produced from the agents themselves:
// row (t, i) holds the sources j of i's scene, ascending, with j != i, both valid at t and (optionally)
// within the radius -- exactly the stable target-sort of the reference's filtered COO list.  No candidate
// list is read, nothing is sorted: degrees, a two-level scan, an ordered fill.
constexpr int kPairThreads = 256;

struct PairArgs {
  const int32_t* scene_range;  // [N][2]: first agent of i's scene, one past its last
  const uint8_t* pm;
  int64_t pm_stride;
  const float* pos;
  int64_t pos_sn, pos_st;
  int32_t N, t0, T;
  float radius;
};

__device__ __forceinline__ bool pair_kept(const PairArgs& a, int32_t t, int32_t j, const float2& pi, float2& attr) {
  if (a.pm[(int64_t)j * a.pm_stride + t]) return false;
  const float2 pj = *reinterpret_cast<const float2*>(a.pos + (int64_t)j * a.pos_sn + (int64_t)t * a.pos_st);
  attr.x = pj.x - pi.x;
  attr.y = pj.y - pi.y;
  if (a.radius > 0.0f) {
    // same rounding as the host framework's 2-norm over a pair: sqrt(fma(y, y, fl(x*x)))
    const float d2 = __fmaf_rn(attr.y, attr.y, __fmul_rn(attr.x, attr.x));
    return __fsqrt_rn(d2) < a.radius;
  }
  return true;
}

// deg[r] for row r = tt * N + i, and the sum of every block's 256 rows.
__global__ __launch_bounds__(kPairThreads) void pairs_degree_kernel(const PairArgs a, int64_t rows,
                                                                    int32_t* __restrict__ deg,
                                                                    int32_t* __restrict__ block_totals) {
  __shared__ int wave_tot[kPairThreads / kWave];
  const int64_t r = (int64_t)blockIdx.x * kPairThreads + threadIdx.x;
  int cnt = 0;
  if (r < rows) {
    const int32_t tt = (int32_t)(r / a.N), i = (int32_t)(r - (int64_t)tt * a.N), t = a.t0 + tt;
    if (!a.pm[(int64_t)i * a.pm_stride + t]) {
      const int32_t lo = a.scene_range[2 * i], hi = a.scene_range[2 * i + 1];
      const float2 pi = *reinterpret_cast<const float2*>(a.pos + (int64_t)i * a.pos_sn + (int64_t)t * a.pos_st);
      for (int32_t j = lo; j < hi; ++j) {
        float2 attr;
        if (j != i && pair_kept(a, t, j, pi, attr)) ++cnt;
      }
    }
    deg[r] = cnt;
  }
  const int w = wave_sum_i(cnt);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < kPairThreads / kWave; ++k) s += wave_tot[k];
    block_totals[blockIdx.x] = s;
  }
}

// rowptr[r] holds deg[r] on entry and the (capacity-clamped) exclusive prefix on exit; every row then
// writes its edges in ascending source order.  Entries that would land at or beyond `capacity` are dropped
// (the caller sees total > capacity in count_out and discards the batch).
__global__ __launch_bounds__(kPairThreads) void pairs_fill_kernel(const PairArgs a, int64_t rows, int64_t capacity,
                                                                  const int32_t* __restrict__ block_offsets,
                                                                  const int32_t* __restrict__ total,
                                                                  int32_t* __restrict__ rowptr,
                                                                  int32_t* __restrict__ src_out,
                                                                  int32_t* __restrict__ dst_out,
                                                                  float* __restrict__ attr_out) {
  __shared__ int wave_tot[kPairThreads / kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * kPairThreads + threadIdx.x;
  const int d = r < rows ? rowptr[r] : 0;
  // inclusive scan inside the wave, then across the block's four waves
  int incl = d;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int wave_off = 0;
#pragma unroll
  for (int k = 0; k < kPairThreads / kWave; ++k)
    if (k < wave) wave_off += wave_tot[k];
  int64_t base = (int64_t)block_offsets[blockIdx.x] + wave_off + incl - d;
  if (r < rows) rowptr[r] = (int32_t)(base < capacity ? base : capacity);
  if (r == rows - 1) {
    const int64_t tot = *total;
    rowptr[rows] = (int32_t)(tot < capacity ? tot : capacity);
  }
  if (r >= rows || d == 0) return;
  const int32_t tt = (int32_t)(r / a.N), i = (int32_t)(r - (int64_t)tt * a.N), t = a.t0 + tt;
  const int32_t lo = a.scene_range[2 * i], hi = a.scene_range[2 * i + 1];
  const float2 pi = *reinterpret_cast<const float2*>(a.pos + (int64_t)i * a.pos_sn + (int64_t)t * a.pos_st);
  for (int32_t j = lo; j < hi; ++j) {
    float2 attr;
    if (j != i && pair_kept(a, t, j, pi, attr)) {
      if (base < capacity) {
        src_out[base] = j + tt * a.N;
        dst_out[base] = i + tt * a.N;
        *reinterpret_cast<float2*>(attr_out + 2 * base) = attr;
      }
      ++base;
    }
  }
}

// ---- CSR build ------------------------------------------------------------------------------
// `count` (optional, device memory) holds the number of meaningful entries when the list is padded
// to a fixed capacity E; entries at or beyond it are ignored.
__device__ __forceinline__ int64_t live_entries(int64_t E, const int32_t* count) {
  if (count == nullptr) return E;
  const int64_t c = *count;
  return c < E ? (c < 0 ? 0 : c) : E;
}

__global__ void csr_hist_kernel(const int32_t* __restrict__ key, int64_t E, const int32_t* __restrict__ count,
                                int32_t* __restrict__ deg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < live_entries(E, count)) atomicAdd(&deg[key[i]], 1);
}

// Unordered placement: slot = rowptr[key] + arrival ticket, the tickets counted DOWN from the row's length that the
// histogram left in `cursor` (no second clearing of the cursors between the two passes).  The order inside a row is
// fixed afterwards by csr_row_sort_kernel, so the result does not depend on atomic arrival order.
__global__ void csr_place_kernel(const int32_t* __restrict__ key, int64_t E, const int32_t* __restrict__ count,
                                 const int32_t* __restrict__ rowptr, int32_t* __restrict__ cursor,
                                 int32_t* __restrict__ tmp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < live_entries(E, count)) {
    const int32_t k = key[i];
    const int32_t slot = rowptr[k] + atomicSub(&cursor[k], 1) - 1;
    tmp[slot] = (int32_t)i;
  }
}

constexpr int kSortChunks = 8;  // rows of up to kSortChunks * 64 ids are ranked out of registers

// One wave per row: rank every edge id among the ids of its row (ids are distinct) and write it to
// its rank, giving ascending original positions = the stable order.
__global__ __launch_bounds__(256) void csr_row_sort_kernel(const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ tmp, int32_t num_rows,
                                                           int32_t* __restrict__ perm) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= num_rows) return;
  const int lane = threadIdx.x & 63;
  const int beg = rowptr[row], end = rowptr[row + 1];
  const int len = end - beg;
  if (len <= kWave) {
    const int mine = lane < len ? tmp[beg + lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < len; ++j) rank += (__shfl(mine, j, 64) < mine) ? 1 : 0;
    if (lane < len) perm[beg + rank] = mine;
  } else if (len <= kSortChunks * kWave) {
    // up to eight ids per lane, held in registers: every id of the row is broadcast once and compared with all of
    // them (len broadcasts instead of len memory reads per id)
    int v[kSortChunks], rank[kSortChunks];
#pragma unroll
    for (int c = 0; c < kSortChunks; ++c) {
      const int i = c * kWave + lane;
      v[c] = i < len ? tmp[beg + i] : 0x7fffffff;
      rank[c] = 0;
    }
#pragma unroll
    for (int cs = 0; cs < kSortChunks; ++cs) {
      const int n = min(kWave, len - cs * kWave);  // ids in chunk cs (uniform over the wave; <= 0: none)
      for (int j = 0; j < n; ++j) {
        const int s = __shfl(v[cs], j, 64);
#pragma unroll
        for (int c = 0; c < kSortChunks; ++c) rank[c] += (s < v[c]) ? 1 : 0;
      }
    }
#pragma unroll
    for (int c = 0; c < kSortChunks; ++c)
      if (c * kWave + lane < len) perm[beg + rank[c]] = v[c];
  } else {
    for (int i = lane; i < len; i += kWave) {
      const int mine = tmp[beg + i];
      int rank = 0;
      for (int j = 0; j < len; ++j) rank += (tmp[beg + j] < mine) ? 1 : 0;
      perm[beg + rank] = mine;
    }
  }
}

// ---- segment sum of rows --------------------------------------------------------------------
// One wave per output row; lanes cover the row in groups of four columns, four input rows requested together and
// added in ascending order.  HALF: the input rows hold bfloat16 values (row_stride counts them), widened exactly.
template <bool HALF>
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const float* __restrict__ rows, int64_t row_stride,
                                                               int32_t width, const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ perm, int32_t num_rows,
                                                               float* __restrict__ out, int64_t out_stride) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= num_rows) return;
  const int lane = threadIdx.x & 63;
  const int beg = rowptr[row], end = rowptr[row + 1];
  auto load = [&](int64_t r, int c) {
    if constexpr (HALF) {
      const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(rows) + r * row_stride + c);
      return make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                         __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u));
    } else {
      return *reinterpret_cast<const float4*>(rows + r * row_stride + c);
    }
  };
  for (int c = lane * 4; c < width; c += kWave * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = beg;
    for (; i + 4 <= end; i += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = load(perm ? perm[i + u] : i + u, c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc.x += v[u].x;
        acc.y += v[u].y;
        acc.z += v[u].z;
        acc.w += v[u].w;
      }
    }
    for (; i < end; ++i) {
      const float4 v = load(perm ? perm[i] : i, c);
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (int64_t)row * out_stride + c) = acc;
  }
}

}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" size_t hivt_edges_workspace_bytes(int64_t items) {
  const int64_t tiles = (items + kFilterTile - 1) / kFilterTile;
  return align_up((size_t)(2 * align_up((size_t)(tiles > 0 ? tiles : 1), 64)) * sizeof(int32_t), 256);
}

extern "C" int hivt_edges_temporal(const int64_t* edge_index, int64_t E0, const uint8_t* padding_mask,
                                   int64_t pm_stride, const float* positions, int64_t pos_stride_n,
                                   int64_t pos_stride_t, int32_t N, int32_t t0, int32_t T, float radius,
                                   int32_t* src_out, int32_t* dst_out, float* attr_out, int32_t* count_out,
                                   void* workspace, size_t workspace_bytes, hivt_stream_t stream) {
  const char* what = "hivt_edges_temporal";
  HIVT_REQUIRE(E0 >= 0 && N >= 0 && T >= 0 && t0 >= 0, HIVT_ERR_ARG, "%s: negative size", what);
  HIVT_REQUIRE(count_out != nullptr, HIVT_ERR_ARG, "%s: count_out is null", what);
  const int64_t items = E0 * (int64_t)T;
  HIVT_REQUIRE(items < (int64_t)1 << 31 && (int64_t)N * T < (int64_t)1 << 31, HIVT_ERR_ARG,
               "%s: %lld candidate edges / %lld nodes exceed int32 indexing", what, (long long)items,
               (long long)N * T);
  HIVT_REQUIRE(items == 0 || (edge_index && padding_mask && positions && src_out && dst_out && attr_out),
               HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(pos_stride_t % 2 == 0 && pos_stride_n % 2 == 0, HIVT_ERR_ARG,
               "%s: position strides must be even (8-byte aligned pairs)", what);
  TemporalPred pred{edge_index, E0, padding_mask, pm_stride, positions, pos_stride_n, pos_stride_t, N, t0, radius};
  return run_filter(pred, items, src_out, dst_out, attr_out, count_out, workspace, workspace_bytes,
                    (hipStream_t)stream, what);
}

extern "C" int hivt_edges_radius(const int64_t* edge_index, int64_t E0, const float* edge_attr, float radius,
                                 int32_t* src_out, int32_t* dst_out, float* attr_out, int32_t* count_out,
                                 void* workspace, size_t workspace_bytes, hivt_stream_t stream) {
  const char* what = "hivt_edges_radius";
  HIVT_REQUIRE(E0 >= 0 && E0 < (int64_t)1 << 31, HIVT_ERR_ARG, "%s: bad edge count %lld", what, (long long)E0);
  HIVT_REQUIRE(count_out != nullptr, HIVT_ERR_ARG, "%s: count_out is null", what);
  HIVT_REQUIRE(E0 == 0 || (edge_index && edge_attr && src_out && dst_out && attr_out), HIVT_ERR_ARG,
               "%s: null pointer", what);
  RadiusPred pred{edge_index, E0, edge_attr, radius};
  return run_filter(pred, E0, src_out, dst_out, attr_out, count_out, workspace, workspace_bytes,
                    (hipStream_t)stream, what);
}

extern "C" size_t hivt_edges_all_pairs_workspace_bytes(int64_t rows) {
  const int64_t blocks = (rows + kPairThreads - 1) / kPairThreads;
  return align_up((size_t)(2 * align_up((size_t)(blocks > 0 ? blocks : 1), 64)) * sizeof(int32_t), 256);
}

extern "C" int hivt_edges_all_pairs_csr(const int32_t* scene_range, const uint8_t* padding_mask, int64_t pm_stride,
                                        const float* positions, int64_t pos_stride_n, int64_t pos_stride_t,
                                        int32_t N, int32_t t0, int32_t T, float radius, int64_t capacity,
                                        int32_t* rowptr, int32_t* src_out, int32_t* dst_out, float* attr_out,
                                        int32_t* count_out, void* workspace, size_t workspace_bytes,
                                        hivt_stream_t stream_) {
  const char* what = "hivt_edges_all_pairs_csr";
  hipStream_t stream = (hipStream_t)stream_;
  HIVT_REQUIRE(N >= 0 && T >= 0 && t0 >= 0 && capacity >= 0, HIVT_ERR_ARG, "%s: negative size", what);
  const int64_t rows = (int64_t)N * T;
  HIVT_REQUIRE(rows < (int64_t)1 << 31 && capacity < (int64_t)1 << 31, HIVT_ERR_ARG,
               "%s: %lld rows / capacity %lld exceed int32 indexing", what, (long long)rows, (long long)capacity);
  HIVT_REQUIRE(rowptr != nullptr && count_out != nullptr, HIVT_ERR_ARG, "%s: rowptr / count_out is null", what);
  if (rows == 0) {
    zero_words(rowptr, 1, stream);
    zero_words(count_out, 1, stream);
    HIVT_CHECK_LAUNCH(what);
    return HIVT_OK;
  }
  HIVT_REQUIRE(scene_range && padding_mask && positions, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(capacity == 0 || (src_out && dst_out && attr_out), HIVT_ERR_ARG, "%s: null output", what);
  HIVT_REQUIRE(pos_stride_t % 2 == 0 && pos_stride_n % 2 == 0, HIVT_ERR_ARG,
               "%s: position strides must be even (8-byte aligned pairs)", what);
  HIVT_REQUIRE(workspace != nullptr && workspace_bytes >= hivt_edges_all_pairs_workspace_bytes(rows),
               HIVT_ERR_WORKSPACE, "%s: workspace too small", what);
  const int64_t blocks = (rows + kPairThreads - 1) / kPairThreads;
  int32_t* block_totals = reinterpret_cast<int32_t*>(workspace);
  int32_t* block_offsets = block_totals + align_up((size_t)blocks, 64);
  PairArgs a{scene_range, padding_mask, pm_stride, positions, pos_stride_n, pos_stride_t, N, t0, T, radius};
  hipLaunchKernelGGL(pairs_degree_kernel, dim3((unsigned)blocks), dim3(kPairThreads), 0, stream, a, rows, rowptr,
                     block_totals);
  HIVT_CHECK_LAUNCH(what);
  hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(1024), 0, stream, block_totals, blocks, block_offsets,
                     count_out);
  HIVT_CHECK_LAUNCH(what);
  hipLaunchKernelGGL(pairs_fill_kernel, dim3((unsigned)blocks), dim3(kPairThreads), 0, stream, a, rows, capacity,
                     block_offsets, count_out, rowptr, src_out, dst_out, attr_out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" size_t hivt_csr_workspace_bytes(int64_t E, int32_t num_rows) {
  // cursor[num_rows] + tmp[E]
  return align_up((size_t)(num_rows > 0 ? num_rows : 1) * sizeof(int32_t), 256) +
         align_up((size_t)(E > 0 ? E : 1) * sizeof(int32_t), 256);
}

// Launch sequence shared by hivt_csr_build and hivt_csr_build_static: `cursor` (num_rows words) and `tmp`
// (E words) are the scratch the kernels need; both are fully (re)initialised here before use.
static int csr_build_launch(const char* what, const int32_t* key, int64_t E, const int32_t* count, int32_t num_rows,
                            int32_t* rowptr, int32_t* perm, int32_t* cursor, int32_t* tmp, hipStream_t stream,
                            int32_t* row_order = nullptr) {
  // degree histogram goes into `cursor`, the scan of it into rowptr, then the counts are the placement's tickets
  zero_words(cursor, num_rows > 0 ? num_rows : 1, stream);
  HIVT_CHECK_LAUNCH(what);
  if (E > 0) {
    hipLaunchKernelGGL(csr_hist_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, key, E, count,
                       cursor);
    HIVT_CHECK_LAUNCH(what);
  }
  if (row_order != nullptr && num_rows > 0)
    hipLaunchKernelGGL(scan_rank_single_block_kernel, dim3(1), dim3(1024), 0, stream, cursor, (int64_t)num_rows, rowptr,
                       rowptr + num_rows, row_order);
  else
    hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(1024), 0, stream, cursor, (int64_t)num_rows, rowptr,
                       rowptr + num_rows);
  HIVT_CHECK_LAUNCH(what);
  if (E > 0) {
    hipLaunchKernelGGL(csr_place_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, key, E, count,
                       rowptr, cursor, tmp);
    HIVT_CHECK_LAUNCH(what);
    hipLaunchKernelGGL(csr_row_sort_kernel, dim3((unsigned)((num_rows + 3) / 4)), dim3(256), 0, stream, rowptr, tmp,
                       num_rows, perm);
    HIVT_CHECK_LAUNCH(what);
  }
  return HIVT_OK;
}

extern "C" int hivt_csr_build(const int32_t* key, int64_t E, const int32_t* count, int32_t num_rows,
                              int32_t* rowptr, int32_t* perm, int32_t* row_order, void* workspace,
                              size_t workspace_bytes, hivt_stream_t stream_) {
  const char* what = "hivt_csr_build";
  hipStream_t stream = (hipStream_t)stream_;
  HIVT_REQUIRE(E >= 0 && E < (int64_t)1 << 31 && num_rows >= 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  HIVT_REQUIRE(rowptr != nullptr, HIVT_ERR_ARG, "%s: rowptr is null", what);
  HIVT_REQUIRE(E == 0 || (key && perm), HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(workspace != nullptr && workspace_bytes >= hivt_csr_workspace_bytes(E, num_rows), HIVT_ERR_WORKSPACE,
               "%s: workspace too small (%zu < %zu)", what, workspace_bytes, hivt_csr_workspace_bytes(E, num_rows));
  int32_t* cursor = reinterpret_cast<int32_t*>(workspace);
  int32_t* tmp = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) +
                                            align_up((size_t)(num_rows > 0 ? num_rows : 1) * sizeof(int32_t), 256));
  return csr_build_launch(what, key, E, count, num_rows, rowptr, perm, cursor, tmp, stream, row_order);
}

// ---- the entry point that allocates nothing ---------------------------------------------------
// The CSR build's scratch (row cursors + unordered placement) lives in a file-scope device array, so
// a caller that may not allocate (graph capture, a serving loop) needs no workspace.  Host code never
// uses the array's name as an address: each device's copy is looked up once with
// hipGetSymbolAddress and the kernels receive that pointer.
static __device__ __align__(256) int32_t scratch[HIVT_CSR_STATIC_MAX_ROWS + HIVT_CSR_STATIC_MAX_EDGES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
int32_t* g_scratch_ptr[kMaxDevices] = {};  // per device: address of that device's `scratch`
}  // namespace

extern "C" int hivt_csr_build_static(const int32_t* key, int64_t E, const int32_t* count, int32_t num_rows,
                                     int32_t* rowptr, int32_t* perm, hivt_stream_t stream_) {
  const char* what = "hivt_csr_build_static";
  hipStream_t stream = (hipStream_t)stream_;
  HIVT_REQUIRE(E >= 0 && num_rows >= 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  HIVT_REQUIRE(E <= HIVT_CSR_STATIC_MAX_EDGES && num_rows <= HIVT_CSR_STATIC_MAX_ROWS, HIVT_ERR_ARG,
               "%s: E=%lld / num_rows=%d exceed the static scratch (%d edges, %d rows); use hivt_csr_build with a "
               "workspace", what, (long long)E, num_rows, HIVT_CSR_STATIC_MAX_EDGES, HIVT_CSR_STATIC_MAX_ROWS);
  HIVT_REQUIRE(rowptr != nullptr, HIVT_ERR_ARG, "%s: rowptr is null", what);
  HIVT_REQUIRE(E == 0 || (key && perm), HIVT_ERR_ARG, "%s: null pointer", what);
  int dev = -1;
  HIVT_CHECK_HIP(hipGetDevice(&dev), what);
  HIVT_REQUIRE(dev >= 0 && dev < kMaxDevices, HIVT_ERR_ARG, "%s: device ordinal %d out of range", what, dev);
  int32_t* base = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
      void* p = nullptr;
      HIVT_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), what);
      HIVT_REQUIRE(p != nullptr, HIVT_ERR_LAUNCH, "%s: the scratch symbol has no address on device %d", what, dev);
      g_scratch_ptr[dev] = reinterpret_cast<int32_t*>(p);
    }
    base = g_scratch_ptr[dev];
  }
  return csr_build_launch(what, key, E, count, num_rows, rowptr, perm, base, base + HIVT_CSR_STATIC_MAX_ROWS, stream);
}

static int segment_sum_rows_impl(const char* what, const void* rows, int64_t row_stride, int32_t width,
                                 const int32_t* rowptr, const int32_t* perm, int32_t num_rows, float* out,
                                 int64_t out_stride, bool half, hivt_stream_t stream) {
  HIVT_REQUIRE(width > 0 && width % 4 == 0 && row_stride % 4 == 0 && out_stride % 4 == 0, HIVT_ERR_ARG,
               "%s: width and strides must be multiples of 4 elements", what);
  HIVT_REQUIRE(num_rows >= 0, HIVT_ERR_ARG, "%s: negative row count", what);
  if (num_rows == 0) return HIVT_OK;
  HIVT_REQUIRE(rows && rowptr && out, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE((reinterpret_cast<uintptr_t>(rows) & (half ? 7 : 15)) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               HIVT_ERR_ARG, "%s: rows and out must be aligned for four-element accesses", what);
  const dim3 grid((unsigned)((num_rows + 3) / 4));
  const float* r = reinterpret_cast<const float*>(rows);
  if (half)
    hipLaunchKernelGGL(segment_sum_rows_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, r, row_stride, width,
                       rowptr, perm, num_rows, out, out_stride);
  else
    hipLaunchKernelGGL(segment_sum_rows_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, r, row_stride, width,
                       rowptr, perm, num_rows, out, out_stride);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_segment_sum_rows(const float* rows, int64_t row_stride, int32_t width, const int32_t* rowptr,
                                     const int32_t* perm, int32_t num_rows, float* out, int64_t out_stride,
                                     hivt_stream_t stream) {
  return segment_sum_rows_impl("hivt_segment_sum_rows", rows, row_stride, width, rowptr, perm, num_rows, out, out_stride,
                               false, stream);
}

extern "C" int hivt_segment_sum_rows_half(const void* rows, int64_t row_stride, int32_t width, const int32_t* rowptr,
                                          const int32_t* perm, int32_t num_rows, float* out, int64_t out_stride,
                                          hivt_stream_t stream) {
  return segment_sum_rows_impl("hivt_segment_sum_rows_half", rows, row_stride, width, rowptr, perm, num_rows, out,
                               out_stride, true, stream);
}
