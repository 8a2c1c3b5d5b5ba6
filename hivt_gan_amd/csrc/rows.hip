// Row bookkeeping around the edge-attention stages: everything the step used to do with a few dozen tiny
// framework launches each (index arithmetic, masks, gathers of token rows, assembling the decoder's output) as
// one launch per job.  All of it is HBM / latency bound integer and copy work; nothing here is shaped into a
// product.
//
//   batch_prepare      rotation matrices, rotated targets and histories, lane codes, per-agent loss weights
//   row_subset         packed (step, agent) nodes of the agent-agent stage + the plan's overflow flag
//   center_rows        rotated displacement and begin-of-sequence code of every packed node
//   rows_put_tokens    rows replaced by token rows (begin-of-sequence tokens), and the gradient's mask
//   temporal_input     packed rows -> [T + 1, N, D] with padding / class tokens and the positional rows
//   decoder_output     (loc, raw scale) -> [.., H, 4] trajectories with the scale's ELU + 1 and floor
//   select_mode        best-mode trajectories out of [F, N, H, 4], and their gradient added back in place
//   sum_padded_rows    sum of row blocks of different widths (the critics' input gradients)
#include "common.h"

#include <math.h>

namespace hivt {
namespace {

// ---------------------------------------------------------------------------------------------
// batch_prepare (models/hivt.py:153-162, models/local_encoder.py:135-149,332-334, models/hivt.py:227-231)
// ---------------------------------------------------------------------------------------------
struct PrepareArgs {
  const float* angles;      // [N]
  const float* y;           // [N, Hf, 2] or null
  const float* x;           // [N, *, 2] with element strides x_sn, x_st
  int64_t x_sn, x_st;
  const uint8_t* pm;        // [N, pm_stride]
  int64_t pm_stride;
  const uint8_t* is_int;    // [L]
  const uint8_t* turn;
  const uint8_t* ctrl;
  int32_t N, T, Hf, L;
  float* rot;               // [N, 2, 2]
  float* y_rot;             // [N, Hf, 2]
  float* x_rot;             // [T * N, 2], row t * N + n
  float* x_steps;           // [T * N, 2]: x itself in that row order
  int32_t* lane_code;       // [L]
  float* keep_w;            // [N]
};

// row vector (a, b) times [[c, -s], [s, c]], products and sum rounded separately (as the framework's
// elementwise operators do)
__device__ __forceinline__ float2 rotate_row(float a, float b, float c, float s) {
#pragma clang fp contract(off)  // no fused multiply-add: three roundings per component, as three operators give
  const float ac = a * c, bs = b * s, as = a * -s, bc = b * c;
  return make_float2(ac + bs, as + bc);
}

__device__ __forceinline__ bool any_future(const PrepareArgs& a, int n) {
  const uint8_t* row = a.pm + (int64_t)n * a.pm_stride + a.T;
  bool keep = false;
  for (int h = 0; h < a.Hf; ++h) keep |= row[h] == 0;
  return keep;
}

__global__ __launch_bounds__(256) void batch_prepare_kernel(const PrepareArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < a.N) {
    const float ang = a.angles[i];
    const float s = sinf(ang), c = cosf(ang);
    *reinterpret_cast<float4*>(a.rot + 4 * i) = make_float4(c, -s, s, c);
    a.keep_w[i] = any_future(a, (int)i) ? 1.0f : 0.0f;
  }
  if (i < (int64_t)a.N * a.T) {  // i = n * T + t: reads of x are contiguous
    const int n = (int)(i / a.T), t = (int)(i - (int64_t)n * a.T);
    const float ang = a.angles[n];
    const float s = sinf(ang), c = cosf(ang);
    const float2 v = *reinterpret_cast<const float2*>(a.x + (int64_t)n * a.x_sn + (int64_t)t * a.x_st);
    *reinterpret_cast<float2*>(a.x_rot + 2 * ((int64_t)t * a.N + n)) = rotate_row(v.x, v.y, c, s);
    *reinterpret_cast<float2*>(a.x_steps + 2 * ((int64_t)t * a.N + n)) = v;
  }
  if (a.y != nullptr && i < (int64_t)a.N * a.Hf) {
    const int n = (int)(i / a.Hf);
    const float ang = a.angles[n];
    const float s = sinf(ang), c = cosf(ang);
    const float2 v = *reinterpret_cast<const float2*>(a.y + 2 * i);
    *reinterpret_cast<float2*>(a.y_rot + 2 * i) = rotate_row(v.x, v.y, c, s);
  }
  if (i < a.L) a.lane_code[i] = ((int)a.is_int[i] * 3 + (int)a.turn[i]) * 2 + (int)a.ctrl[i];
}

// ---------------------------------------------------------------------------------------------
// row_subset: the nodes (t, n) with padding_mask[n][t] == 0, packed in ascending node id t * N + n
// ---------------------------------------------------------------------------------------------
struct SubsetArgs {
  const uint8_t* pm;
  int64_t pm_stride;
  int32_t N, T, capacity;
  const int32_t* rowptr;         // [T * N + 1] or null
  const int32_t* edge_count[3];  // optional device counters checked against edge_cap
  int64_t edge_cap[3];
  int32_t* rows;                 // [capacity]
  int32_t* new_id;               // [T * N]
  uint8_t* live;                 // [capacity]
  int32_t* count;                // [2]: live rows, id of the first live row
  int32_t* packed_rowptr;        // [capacity + 1] or null
  uint8_t* overflow;             // [1]
};

__device__ __forceinline__ int node_valid(const SubsetArgs& a, int64_t m) {
  const int64_t t = m / a.N, n = m - t * a.N;
  return a.pm[n * a.pm_stride + t] == 0 ? 1 : 0;
}

// Two launches.  (1) every workgroup counts the valid nodes among its 256; (2) every workgroup adds up the counts of
// the workgroups before it (a few hundred values), ranks its own nodes with wave ballots and fills its share of the
// outputs; the last workgroup also writes the total, the tail of the packed arrays and the overflow flag.
constexpr int kSubsetThreads = 256;

__global__ __launch_bounds__(kSubsetThreads) void subset_count_kernel(const SubsetArgs a, int32_t* __restrict__ block_counts) {
  __shared__ int wave_tot[kSubsetThreads / kWave];
  const int64_t M = (int64_t)a.N * a.T;
  const int64_t m = (int64_t)blockIdx.x * kSubsetThreads + threadIdx.x;
  const int v = m < M ? node_valid(a, m) : 0;
  const int w = __popcll(__ballot(v != 0));
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < kSubsetThreads / kWave; ++k) s += wave_tot[k];
    block_counts[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kSubsetThreads) void subset_fill_kernel(const SubsetArgs a, const int32_t* __restrict__ block_counts,
                                                                     int32_t blocks) {
  __shared__ int red[kSubsetThreads / kWave];
  __shared__ int wave_tot[kSubsetThreads / kWave];
  __shared__ int s_first;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t M = (int64_t)a.N * a.T;
  // nodes before this workgroup, and (last workgroup) all nodes
  int before = 0, total = 0;
  for (int b = threadIdx.x; b < blocks; b += kSubsetThreads) {
    const int c = block_counts[b];
    total += c;
    if (b < (int)blockIdx.x) before += c;
  }
  before = wave_sum_i(before);
  total = wave_sum_i(total);
  if (lane == 0) { red[wave] = before; wave_tot[wave] = total; }
  __syncthreads();
  before = 0; total = 0;
#pragma unroll
  for (int k = 0; k < kSubsetThreads / kWave; ++k) { before += red[k]; total += wave_tot[k]; }
  __syncthreads();
  const int64_t m = (int64_t)blockIdx.x * kSubsetThreads + threadIdx.x;
  const int v = m < M ? node_valid(a, m) : 0;
  const unsigned long long ballot = __ballot(v != 0);
  const int lane_rank = __popcll(ballot & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wave] = __popcll(ballot);
  __syncthreads();
  int wave_off = 0;
#pragma unroll
  for (int k = 0; k < kSubsetThreads / kWave; ++k)
    if (k < wave) wave_off += wave_tot[k];
  if (m < M) {
    const int32_t p = before + wave_off + lane_rank - (v ? 0 : 1);  // position of the last valid node up to m
    if (v && p < a.capacity) {
      a.rows[p] = (int32_t)m;
      if (a.packed_rowptr != nullptr) a.packed_rowptr[p] = a.rowptr[m];
    }
    if (v && p == 0) a.count[1] = (int32_t)m;  // the first valid node
    a.new_id[m] = p < 0 ? 0 : (p < a.capacity ? p : a.capacity - 1);
  }
  if ((int)blockIdx.x == blocks - 1 && threadIdx.x == 0) {  // totals and flags
    a.count[0] = total;
    if (total == 0) a.count[1] = (int32_t)(M > 0 ? M - 1 : 0);
    bool over = total > a.capacity;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (a.edge_count[k] != nullptr) over |= (int64_t)*a.edge_count[k] > a.edge_cap[k];
    a.overflow[0] = over ? 1 : 0;
    if (a.packed_rowptr != nullptr) a.packed_rowptr[a.capacity] = a.rowptr[M];  // the end of the last packed row
  }
  // this workgroup's slice of the packed arrays: the live flags, and behind the live rows a valid node id (gathers
  // through it stay finite) and the end of the edge list
  const int packed = total < a.capacity ? total : a.capacity;
  const int64_t i = m;  // capacity <= M, so the slices of all workgroups cover [0, capacity)
  const bool tail_here = (int64_t)blockIdx.x * kSubsetThreads + kSubsetThreads > packed &&
                         (int64_t)blockIdx.x * kSubsetThreads <= a.capacity;
  if (!tail_here) {
    if (i < a.capacity) a.live[i] = 1;
    return;
  }
  if (threadIdx.x == 0) {  // id of the first valid node: its workgroup from the counts, then its nodes one by one
    int first = (int)(M > 0 ? M - 1 : 0);
    for (int b = 0; b < blocks; ++b) {
      if (block_counts[b] > 0) {
        for (int64_t q = (int64_t)b * kSubsetThreads; q < M; ++q)
          if (node_valid(a, q)) { first = (int)q; break; }
        break;
      }
    }
    s_first = first;
  }
  __syncthreads();
  if (i < a.capacity) {
    a.live[i] = i < total ? 1 : 0;
    if (i >= packed) a.rows[i] = s_first;
  }
  if (a.packed_rowptr != nullptr && i >= packed && i < a.capacity) a.packed_rowptr[i] = a.rowptr[M];
}

// ---------------------------------------------------------------------------------------------
// center_rows / rows_put_tokens (models/local_encoder.py:149-150)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void center_rows_kernel(const float* __restrict__ x_rot, const uint8_t* __restrict__ bos,
                                                          int64_t bos_stride, const int32_t* __restrict__ rows, int32_t N,
                                                          int32_t capacity, float* __restrict__ xc,
                                                          int32_t* __restrict__ code) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= capacity) return;
  const int32_t m = rows[i];
  const int32_t t = m / N, n = m - t * N;
  *reinterpret_cast<float2*>(xc + 2 * (int64_t)i) = *reinterpret_cast<const float2*>(x_rot + 2 * (int64_t)m);
  code[i] = bos[(int64_t)n * bos_stride + t] ? t : -1;
}

// out[m] = code[m] >= 0 ? token[code[m]] : x[m]   (x == null: rows without a code become zero; that is the
// gradient's form: pass the negated selection)
__global__ __launch_bounds__(256) void rows_put_tokens_kernel(const float* __restrict__ x, int64_t x_ld,
                                                              const int32_t* __restrict__ code,
                                                              const float* __restrict__ token, int64_t M, int32_t D4,
                                                              int32_t keep_coded, float* __restrict__ out) {
  const int64_t total = M * D4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / D4;
    const int c = (int)(i - m * D4) * 4;
    const int32_t k = code[m];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (token != nullptr) {
      v = k >= 0 ? *reinterpret_cast<const float4*>(token + (int64_t)k * D4 * 4 + c)
                 : *reinterpret_cast<const float4*>(x + m * x_ld + c);
    } else if ((k >= 0) == (keep_coded != 0)) {
      v = *reinterpret_cast<const float4*>(x + m * x_ld + c);
    }
    *reinterpret_cast<float4*>(out + m * (int64_t)D4 * 4 + c) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// temporal_input (models/local_encoder.py:230-232)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void temporal_input_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ new_id,
                                                                 const uint8_t* __restrict__ pm, int64_t pm_stride,
                                                                 const float* __restrict__ pad_token,
                                                                 const float* __restrict__ cls,
                                                                 const float* __restrict__ pos, int32_t T, int32_t N,
                                                                 int32_t D4, float* __restrict__ out,
                                                                 int32_t* __restrict__ pad_code) {
  const int64_t total = (int64_t)(T + 1) * N * D4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / D4;
    const int c = (int)(i - m * D4) * 4;
    const int s = (int)(m / N), n = (int)(m - (int64_t)s * N);
    float4 v;
    int code = -1;
    if (s == T) {
      v = *reinterpret_cast<const float4*>(cls + c);
    } else if (pm[(int64_t)n * pm_stride + s]) {
      v = *reinterpret_cast<const float4*>(pad_token + (int64_t)s * D4 * 4 + c);
      code = s;
    } else {
      v = *reinterpret_cast<const float4*>(x + (int64_t)(new_id != nullptr ? new_id[m] : (int32_t)m) * D4 * 4 + c);
    }
    const float4 p = *reinterpret_cast<const float4*>(pos + (int64_t)s * D4 * 4 + c);
    *reinterpret_cast<float4*>(out + m * (int64_t)D4 * 4 + c) = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
    if (c == 0) pad_code[m] = code;
  }
}

// dx[r] = live row r ? dy[rows[r]] : 0
__global__ __launch_bounds__(256) void gather_live_rows_kernel(const float* __restrict__ dy, const int32_t* __restrict__ rows,
                                                               const int32_t* __restrict__ count, int32_t capacity,
                                                               int32_t D4, float* __restrict__ dx) {
  const int64_t total = (int64_t)capacity * D4;
  const int32_t live = *count;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / D4;
    const int c = (int)(i - r * D4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < live) v = *reinterpret_cast<const float4*>(dy + (int64_t)rows[r] * D4 * 4 + c);
    *reinterpret_cast<float4*>(dx + r * (int64_t)D4 * 4 + c) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// decoder_output (models/decoder.py:131-139)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float scale_value(float v, float floor_) {
  const float e = v > 0.0f ? v : expm1f(v);
  return fmaxf(e + 1.0f, floor_);
}

// input row m of the mode-major output row o: the inputs are either in the same order (F <= 1) or agent-major
// (row n * F + f for output row f * N + n)
__device__ __forceinline__ int64_t input_row(int64_t o, int64_t M, int32_t F) {
  if (F <= 1) return o;
  const int64_t N = M / F, f = o / N, n = o - f * N;
  return n * F + f;
}

__global__ __launch_bounds__(256) void decoder_output_fwd_kernel(const float* __restrict__ loc, int64_t loc_ld,
                                                                 const float* __restrict__ raw, int64_t raw_ld,
                                                                 int64_t M, int32_t H, int32_t F, float floor_,
                                                                 float* __restrict__ out) {
  const int64_t total = M * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t o = i / H;
    const int h = (int)(i - o * H);
    const int64_t m = input_row(o, M, F);
    const float2 l = *reinterpret_cast<const float2*>(loc + m * loc_ld + 2 * h);
    const float2 r = *reinterpret_cast<const float2*>(raw + m * raw_ld + 2 * h);
    *reinterpret_cast<float4*>(out + 4 * i) = make_float4(l.x, l.y, scale_value(r.x, floor_), scale_value(r.y, floor_));
  }
}

__device__ __forceinline__ float scale_grad(float v, float g, float floor_) {
  const float e = v > 0.0f ? v : expm1f(v);
  if (!(e + 1.0f >= floor_)) return 0.0f;
  return v > 0.0f ? g : g * expf(v);
}

__global__ __launch_bounds__(256) void decoder_output_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ raw,
                                                                 int64_t raw_ld, int64_t M, int32_t H, int32_t F,
                                                                 float floor_, float* __restrict__ dloc,
                                                                 float* __restrict__ draw) {
  const int64_t total = M * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t o = i / H;
    const int h = (int)(i - o * H);
    const int64_t m = input_row(o, M, F);
    const float4 g = *reinterpret_cast<const float4*>(dout + 4 * i);
    const float2 r = *reinterpret_cast<const float2*>(raw + m * raw_ld + 2 * h);
    *reinterpret_cast<float2*>(dloc + m * 2 * H + 2 * h) = make_float2(g.x, g.y);
    *reinterpret_cast<float2*>(draw + m * 2 * H + 2 * h) =
        make_float2(scale_grad(r.x, g.z, floor_), scale_grad(r.y, g.w, floor_));
  }
}

// ---------------------------------------------------------------------------------------------
// select_mode (models/hivt.py:221-224)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_mode_fwd_kernel(const float* __restrict__ y_hat, const int64_t* __restrict__ best,
                                                              int32_t F, int32_t N, int32_t H, int32_t C,
                                                              float* __restrict__ out) {
  const int64_t total = (int64_t)N * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / H;
    int64_t f = best[n];
    f = f < 0 ? 0 : (f >= F ? F - 1 : f);
    *reinterpret_cast<float2*>(out + 2 * i) = *reinterpret_cast<const float2*>(y_hat + ((f * N) * H + i) * C);
  }
}

// d_y_hat[best[n], n, h, 0:2] += g[n, h, :]
__global__ __launch_bounds__(256) void select_mode_bwd_kernel(const float* __restrict__ g, const int64_t* __restrict__ best,
                                                              int32_t F, int32_t N, int32_t H, int32_t C,
                                                              float* __restrict__ d_y_hat) {
  const int64_t total = (int64_t)N * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / H;
    int64_t f = best[n];
    f = f < 0 ? 0 : (f >= F ? F - 1 : f);
    float2* dst = reinterpret_cast<float2*>(d_y_hat + ((f * N) * H + i) * C);
    const float2 a = *dst, b = *reinterpret_cast<const float2*>(g + 2 * i);
    *dst = make_float2(a.x + b.x, a.y + b.y);
  }
}

// ---------------------------------------------------------------------------------------------
// sum_padded_rows: out[b][c] = sum over parts p with c < width[p] of part[p][b][c], parts added in order
// ---------------------------------------------------------------------------------------------
struct PaddedParts {
  const float* part[4];
  int32_t width[4];
  int32_t count;
};

__global__ __launch_bounds__(256) void sum_padded_rows_kernel(const PaddedParts p, int64_t B, int32_t W,
                                                              float* __restrict__ out) {
  const int64_t total = B * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / W;
    const int c = (int)(i - b * W);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < p.count && c < p.width[k]) s += p.part[k][b * p.width[k] + c];
    out[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// category_table: the 12 sums a[i] + b[j] + c[k] of three small embedding tables (models/local_encoder.py:361-363)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void category_table_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 const float* __restrict__ c, int32_t na, int32_t nb,
                                                                 int32_t nc, int32_t D, float* __restrict__ out) {
  const int total = na * nb * nc * D;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int col = i % D, row = i / D;
    const int k = row % nc, j = (row / nc) % nb, ia = row / (nc * nb);
    out[i] = (a[ia * D + col] + b[j * D + col]) + c[k * D + col];
  }
}

// da[i] = sum over (j, k) of g[(i nb + j) nc + k], likewise db, dc: rows added in ascending order
__global__ __launch_bounds__(256) void category_table_bwd_kernel(const float* __restrict__ g, int32_t na, int32_t nb,
                                                                 int32_t nc, int32_t D, float* __restrict__ da,
                                                                 float* __restrict__ db, float* __restrict__ dc) {
  const int total = (na + nb + nc) * D;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int col = i % D, r = i / D;
    float s = 0.0f;
    for (int row = 0; row < na * nb * nc; ++row) {
      const int k = row % nc, j = (row / nc) % nb, ia = row / (nc * nb);
      const bool mine = r < na ? ia == r : (r < na + nb ? j == r - na : k == r - na - nb);
      if (mine) s += g[row * D + col];
    }
    if (r < na) da[r * D + col] = s;
    else if (r < na + nb) db[(r - na) * D + col] = s;
    else dc[(r - na - nb) * D + col] = s;
  }
}

inline unsigned grid_for(int64_t items) {
  int64_t b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace
}  // namespace hivt

using namespace hivt;

extern "C" int hivt_batch_prepare(const float* rotate_angles, const float* y, const float* x, int64_t x_stride_n,
                                  int64_t x_stride_t, const uint8_t* padding_mask, int64_t pm_stride,
                                  const uint8_t* is_intersections, const uint8_t* turn_directions,
                                  const uint8_t* traffic_controls, int32_t N, int32_t T, int32_t Hf, int32_t L,
                                  float* rotate_mat, float* y_rot, float* x_rot, float* x_steps, int32_t* lane_code,
                                  float* keep_w, hivt_stream_t stream) {
  const char* what = "hivt_batch_prepare";
  HIVT_REQUIRE(N >= 0 && T > 0 && Hf >= 0 && L >= 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  HIVT_REQUIRE(N == 0 || (rotate_angles && x && padding_mask && rotate_mat && x_rot && x_steps && keep_w), HIVT_ERR_ARG,
               "%s: null pointer", what);
  HIVT_REQUIRE(y == nullptr || y_rot != nullptr, HIVT_ERR_ARG, "%s: y without y_rot", what);
  HIVT_REQUIRE(L == 0 || (is_intersections && turn_directions && traffic_controls && lane_code), HIVT_ERR_ARG,
               "%s: null lane attribute", what);
  HIVT_REQUIRE(pm_stride >= T + Hf, HIVT_ERR_ARG, "%s: padding_mask rows shorter than T + Hf", what);
  HIVT_REQUIRE(N == 0 || (al8(x) && al8(x_rot) && al8(x_steps) && al16(rotate_mat) && x_stride_n % 2 == 0 && x_stride_t % 2 == 0 &&
                          (y == nullptr || (al8(y) && al8(y_rot)))),
               HIVT_ERR_ARG, "%s: rows of x / y must start on 8-byte boundaries (rotate_mat on 16)", what);
  PrepareArgs a{};
  a.angles = rotate_angles; a.y = y; a.x = x; a.x_sn = x_stride_n; a.x_st = x_stride_t;
  a.pm = padding_mask; a.pm_stride = pm_stride;
  a.is_int = is_intersections; a.turn = turn_directions; a.ctrl = traffic_controls;
  a.N = N; a.T = T; a.Hf = Hf; a.L = L;
  a.rot = rotate_mat; a.y_rot = y_rot; a.x_rot = x_rot; a.x_steps = x_steps; a.lane_code = lane_code; a.keep_w = keep_w;
  int64_t items = (int64_t)N * (T > Hf ? T : Hf);
  if (items < L) items = L;
  if (items < N) items = N;
  hipLaunchKernelGGL(batch_prepare_kernel, dim3((unsigned)((items + 255) / 256 > 0 ? (items + 255) / 256 : 1)), dim3(256),
                     0, (hipStream_t)stream, a);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" size_t hivt_row_subset_workspace_bytes(int32_t N, int32_t T) {
  const int64_t blocks = ((int64_t)(N > 0 ? N : 0) * (T > 0 ? T : 0) + kSubsetThreads - 1) / kSubsetThreads;
  return align_up((size_t)(blocks > 0 ? blocks : 1) * sizeof(int32_t), 256);
}

extern "C" int hivt_row_subset(const uint8_t* padding_mask, int64_t pm_stride, int32_t N, int32_t T, int32_t capacity,
                               const int32_t* rowptr, const int32_t* edge_count0, int64_t edge_cap0,
                               const int32_t* edge_count1, int64_t edge_cap1, const int32_t* edge_count2,
                               int64_t edge_cap2, int32_t* rows, int32_t* new_id, uint8_t* live, int32_t* count,
                               int32_t* packed_rowptr, uint8_t* overflow, void* workspace, size_t workspace_bytes,
                               hivt_stream_t stream) {
  const char* what = "hivt_row_subset";
  HIVT_REQUIRE(N > 0 && T > 0 && capacity > 0 && (int64_t)N * T < ((int64_t)1 << 31) && capacity <= (int64_t)N * T,
               HIVT_ERR_ARG, "%s: bad sizes (the capacity is at most N * T)", what);
  HIVT_REQUIRE(padding_mask && rows && new_id && live && count && overflow, HIVT_ERR_ARG, "%s: null pointer", what);
  HIVT_REQUIRE(pm_stride >= T, HIVT_ERR_ARG, "%s: padding_mask rows shorter than T", what);
  HIVT_REQUIRE((rowptr == nullptr) == (packed_rowptr == nullptr), HIVT_ERR_ARG,
               "%s: rowptr and packed_rowptr go together", what);
  SubsetArgs a{};
  a.pm = padding_mask; a.pm_stride = pm_stride; a.N = N; a.T = T; a.capacity = capacity; a.rowptr = rowptr;
  a.edge_count[0] = edge_count0; a.edge_count[1] = edge_count1; a.edge_count[2] = edge_count2;
  a.edge_cap[0] = edge_cap0; a.edge_cap[1] = edge_cap1; a.edge_cap[2] = edge_cap2;
  a.rows = rows; a.new_id = new_id; a.live = live; a.count = count; a.packed_rowptr = packed_rowptr;
  a.overflow = overflow;
  hipStream_t s = (hipStream_t)stream;
  const int32_t blocks = (int32_t)(((int64_t)N * T + kSubsetThreads - 1) / kSubsetThreads);
  HIVT_REQUIRE(workspace != nullptr && workspace_bytes >= hivt_row_subset_workspace_bytes(N, T), HIVT_ERR_WORKSPACE,
               "%s: workspace too small", what);
  int32_t* block_counts = reinterpret_cast<int32_t*>(workspace);
  hipLaunchKernelGGL(subset_count_kernel, dim3((unsigned)blocks), dim3(kSubsetThreads), 0, s, a, block_counts);
  HIVT_CHECK_LAUNCH(what);
  hipLaunchKernelGGL(subset_fill_kernel, dim3((unsigned)blocks), dim3(kSubsetThreads), 0, s, a, block_counts, blocks);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_center_rows(const float* x_rot, const uint8_t* bos_mask, int64_t bos_stride, const int32_t* rows,
                                int32_t N, int32_t T, int32_t capacity, float* xc, int32_t* code,
                                hivt_stream_t stream) {
  const char* what = "hivt_center_rows";
  HIVT_REQUIRE(N > 0 && T > 0 && capacity >= 0 && bos_stride >= T, HIVT_ERR_ARG, "%s: bad sizes", what);
  if (capacity == 0) return HIVT_OK;
  HIVT_REQUIRE(x_rot && bos_mask && rows && xc && code && al8(x_rot) && al8(xc), HIVT_ERR_ARG, "%s: null / unaligned pointer",
               what);
  hipLaunchKernelGGL(center_rows_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x_rot, bos_mask, bos_stride, rows, N, capacity, xc, code);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_rows_put_tokens(const float* x, int64_t x_ld, const int32_t* code, const float* token, int64_t M,
                                    int32_t D, int32_t keep_coded, float* out, hivt_stream_t stream) {
  const char* what = "hivt_rows_put_tokens";
  HIVT_REQUIRE(M >= 0 && D > 0 && D % 4 == 0, HIVT_ERR_ARG, "%s: the width must be a multiple of 4", what);
  if (M == 0) return HIVT_OK;
  HIVT_REQUIRE(x && code && out && al16(x) && al16(out) && x_ld % 4 == 0 && (token == nullptr || al16(token)),
               HIVT_ERR_ARG, "%s: null / unaligned pointer", what);
  hipLaunchKernelGGL(rows_put_tokens_kernel, dim3(grid_for(M * (D / 4))), dim3(256), 0, (hipStream_t)stream, x, x_ld,
                     code, token, M, D / 4, keep_coded, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_temporal_input_fwd(const float* x, const int32_t* new_id, const uint8_t* padding_mask,
                                       int64_t pm_stride, const float* padding_token, const float* cls_token,
                                       const float* pos_embed, int32_t T, int32_t N, int32_t D, float* out,
                                       int32_t* pad_code, hivt_stream_t stream) {
  const char* what = "hivt_temporal_input_fwd";
  HIVT_REQUIRE(T > 0 && N >= 0 && D > 0 && D % 4 == 0 && pm_stride >= T, HIVT_ERR_ARG, "%s: bad sizes", what);
  if (N == 0) return HIVT_OK;
  HIVT_REQUIRE(x && padding_mask && padding_token && cls_token && pos_embed && out && pad_code, HIVT_ERR_ARG,
               "%s: null pointer", what);
  HIVT_REQUIRE(al16(x) && al16(padding_token) && al16(cls_token) && al16(pos_embed) && al16(out), HIVT_ERR_ARG,
               "%s: rows must start on 16-byte boundaries", what);
  hipLaunchKernelGGL(temporal_input_fwd_kernel, dim3(grid_for((int64_t)(T + 1) * N * (D / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, new_id, padding_mask, pm_stride, padding_token, cls_token, pos_embed, T, N,
                     D / 4, out, pad_code);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_gather_live_rows(const float* dy, const int32_t* rows, const int32_t* count, int32_t capacity,
                                     int32_t D, float* dx, hivt_stream_t stream) {
  const char* what = "hivt_gather_live_rows";
  HIVT_REQUIRE(capacity >= 0 && D > 0 && D % 4 == 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  if (capacity == 0) return HIVT_OK;
  HIVT_REQUIRE(dy && rows && count && dx && al16(dy) && al16(dx), HIVT_ERR_ARG, "%s: null / unaligned pointer", what);
  hipLaunchKernelGGL(gather_live_rows_kernel, dim3(grid_for((int64_t)capacity * (D / 4))), dim3(256), 0,
                     (hipStream_t)stream, dy, rows, count, capacity, D / 4, dx);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_decoder_output_fwd(const float* loc, int64_t loc_ld, const float* scale_raw, int64_t raw_ld, int64_t M,
                                       int32_t H, int32_t F, float min_scale, float* out, hivt_stream_t stream) {
  const char* what = "hivt_decoder_output_fwd";
  HIVT_REQUIRE(M >= 0 && H > 0 && (F <= 1 || M % F == 0), HIVT_ERR_ARG, "%s: bad sizes", what);
  if (M == 0) return HIVT_OK;
  HIVT_REQUIRE(loc && scale_raw && out && al8(loc) && al8(scale_raw) && al16(out) && loc_ld % 2 == 0 && raw_ld % 2 == 0,
               HIVT_ERR_ARG, "%s: null / unaligned pointer", what);
  hipLaunchKernelGGL(decoder_output_fwd_kernel, dim3(grid_for(M * H)), dim3(256), 0, (hipStream_t)stream, loc, loc_ld,
                     scale_raw, raw_ld, M, H, F, min_scale, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_decoder_output_bwd(const float* d_out, const float* scale_raw, int64_t raw_ld, int64_t M, int32_t H,
                                       int32_t F, float min_scale, float* d_loc, float* d_scale_raw,
                                       hivt_stream_t stream) {
  const char* what = "hivt_decoder_output_bwd";
  HIVT_REQUIRE(M >= 0 && H > 0 && (F <= 1 || M % F == 0), HIVT_ERR_ARG, "%s: bad sizes", what);
  if (M == 0) return HIVT_OK;
  HIVT_REQUIRE(d_out && scale_raw && d_loc && d_scale_raw && al16(d_out) && al8(scale_raw) && al8(d_loc) &&
                   al8(d_scale_raw) && raw_ld % 2 == 0,
               HIVT_ERR_ARG, "%s: null / unaligned pointer", what);
  hipLaunchKernelGGL(decoder_output_bwd_kernel, dim3(grid_for(M * H)), dim3(256), 0, (hipStream_t)stream, d_out,
                     scale_raw, raw_ld, M, H, F, min_scale, d_loc, d_scale_raw);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_select_mode_fwd(const float* y_hat, const int64_t* best_mode, int32_t F, int32_t N, int32_t H,
                                    int32_t C, float* out, hivt_stream_t stream) {
  const char* what = "hivt_select_mode_fwd";
  HIVT_REQUIRE(F > 0 && N >= 0 && H > 0 && C >= 2 && C % 2 == 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  if (N == 0) return HIVT_OK;
  HIVT_REQUIRE(y_hat && best_mode && out && al8(y_hat) && al8(out), HIVT_ERR_ARG, "%s: null / unaligned pointer", what);
  hipLaunchKernelGGL(select_mode_fwd_kernel, dim3(grid_for((int64_t)N * H)), dim3(256), 0, (hipStream_t)stream, y_hat,
                     best_mode, F, N, H, C, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_select_mode_bwd(const float* g, const int64_t* best_mode, int32_t F, int32_t N, int32_t H, int32_t C,
                                    float* d_y_hat, hivt_stream_t stream) {
  const char* what = "hivt_select_mode_bwd";
  HIVT_REQUIRE(F > 0 && N >= 0 && H > 0 && C >= 2 && C % 2 == 0, HIVT_ERR_ARG, "%s: bad sizes", what);
  if (N == 0) return HIVT_OK;
  HIVT_REQUIRE(g && best_mode && d_y_hat && al8(g) && al8(d_y_hat), HIVT_ERR_ARG, "%s: null / unaligned pointer", what);
  hipLaunchKernelGGL(select_mode_bwd_kernel, dim3(grid_for((int64_t)N * H)), dim3(256), 0, (hipStream_t)stream, g,
                     best_mode, F, N, H, C, d_y_hat);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_sum_padded_rows(const float* const* parts, const int32_t* widths, int32_t count, int64_t B,
                                    int32_t W, float* out, hivt_stream_t stream) {
  const char* what = "hivt_sum_padded_rows";
  HIVT_REQUIRE(count >= 1 && count <= 4 && B >= 0 && W > 0 && parts && widths, HIVT_ERR_ARG, "%s: 1..4 parts", what);
  if (B == 0) return HIVT_OK;
  HIVT_REQUIRE(out != nullptr, HIVT_ERR_ARG, "%s: out is null", what);
  PaddedParts p{};
  p.count = count;
  for (int k = 0; k < count; ++k) {
    HIVT_REQUIRE(parts[k] != nullptr && widths[k] >= 1 && widths[k] <= W, HIVT_ERR_ARG, "%s: part %d malformed", what, k);
    p.part[k] = parts[k];
    p.width[k] = widths[k];
  }
  hipLaunchKernelGGL(sum_padded_rows_kernel, dim3(grid_for(B * W)), dim3(256), 0, (hipStream_t)stream, p, B, W, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

namespace hivt {
namespace {
// out[r] = [a[r] | b[r]] and bias_out = bias_a + bias_b for up to eight pairs of layers in one launch
constexpr int kStackMax = 8;
struct StackPair {
  const float* a;
  const float* b;
  float* out;
  const float* bias_a;
  const float* bias_b;
  float* bias_out;
  int32_t rows, ka, kb;
};
struct StackArgs {
  StackPair pair[kStackMax];
  int32_t first_block[kStackMax + 1];
  int32_t count;
};

__global__ __launch_bounds__(256) void stack_pairs_kernel(const StackArgs g) {
  int i = 0;
  while (i + 1 < g.count && g.first_block[i + 1] <= (int)blockIdx.x) ++i;
  const StackPair& p = g.pair[i];
  const int width = p.ka + p.kb;
  const int64_t total = (int64_t)p.rows * width;
  const int blocks = g.first_block[i + 1] - g.first_block[i];
  const int64_t start = (int64_t)((int)blockIdx.x - g.first_block[i]) * 256 + threadIdx.x;
  for (int64_t e = start; e < total; e += (int64_t)blocks * 256) {
    const int r = (int)(e / width), c = (int)(e % width);
    p.out[e] = c < p.ka ? p.a[(int64_t)r * p.ka + c] : p.b[(int64_t)r * p.kb + (c - p.ka)];
  }
  if (p.bias_out != nullptr)
    for (int64_t r = start; r < p.rows; r += (int64_t)blocks * 256) p.bias_out[r] = p.bias_a[r] + p.bias_b[r];
}
}  // namespace
}  // namespace hivt

extern "C" int hivt_stack_pairs(const int64_t* items, int32_t count, hivt_stream_t stream) {
  const char* what = "hivt_stack_pairs";
  HIVT_REQUIRE(count >= 0 && (count == 0 || items != nullptr), HIVT_ERR_ARG, "%s: bad arguments", what);
  for (int32_t first = 0; first < count; first += kStackMax) {
    StackArgs g{};
    const int n = count - first < kStackMax ? count - first : kStackMax;
    g.count = n;
    for (int i = 0; i < n; ++i) {
      const int64_t* t = items + (size_t)(first + i) * 9;
      StackPair& p = g.pair[i];
      p.a = reinterpret_cast<const float*>(t[0]);
      p.b = reinterpret_cast<const float*>(t[1]);
      p.out = reinterpret_cast<float*>(t[2]);
      p.rows = (int32_t)t[3]; p.ka = (int32_t)t[4]; p.kb = (int32_t)t[5];
      p.bias_a = reinterpret_cast<const float*>(t[6]);
      p.bias_b = reinterpret_cast<const float*>(t[7]);
      p.bias_out = reinterpret_cast<float*>(t[8]);
      HIVT_REQUIRE(p.a && p.b && p.out && p.rows > 0 && p.ka > 0 && p.kb > 0, HIVT_ERR_ARG,
                   "%s: item %d: null pointer or empty shape", what, first + i);
      HIVT_REQUIRE((p.bias_out == nullptr) == (p.bias_a == nullptr) && (p.bias_a == nullptr) == (p.bias_b == nullptr),
                   HIVT_ERR_ARG, "%s: item %d: the three bias pointers go together", what, first + i);
      const int64_t total = (int64_t)p.rows * (p.ka + p.kb);
      int64_t blocks = (total + 1023) / 1024;
      if (blocks > 64) blocks = 64;
      g.first_block[i + 1] = g.first_block[i] + (int32_t)blocks;
    }
    hipLaunchKernelGGL(stack_pairs_kernel, dim3((unsigned)g.first_block[n]), dim3(256), 0, (hipStream_t)stream, g);
    HIVT_CHECK_LAUNCH(what);
  }
  return HIVT_OK;
}

extern "C" int hivt_category_table_fwd(const float* a, const float* b, const float* c, int32_t na, int32_t nb, int32_t nc,
                                       int32_t D, float* out, hivt_stream_t stream) {
  const char* what = "hivt_category_table_fwd";
  HIVT_REQUIRE(na > 0 && nb > 0 && nc > 0 && D > 0 && (int64_t)na * nb * nc * D < (1 << 24), HIVT_ERR_ARG, "%s: bad sizes",
               what);
  HIVT_REQUIRE(a && b && c && out, HIVT_ERR_ARG, "%s: null pointer", what);
  hipLaunchKernelGGL(category_table_fwd_kernel, dim3(grid_for((int64_t)na * nb * nc * D)), dim3(256), 0,
                     (hipStream_t)stream, a, b, c, na, nb, nc, D, out);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}

extern "C" int hivt_category_table_bwd(const float* g, int32_t na, int32_t nb, int32_t nc, int32_t D, float* da, float* db,
                                       float* dc, hivt_stream_t stream) {
  const char* what = "hivt_category_table_bwd";
  HIVT_REQUIRE(na > 0 && nb > 0 && nc > 0 && D > 0 && (int64_t)na * nb * nc * D < (1 << 24), HIVT_ERR_ARG, "%s: bad sizes",
               what);
  HIVT_REQUIRE(g && da && db && dc, HIVT_ERR_ARG, "%s: null pointer", what);
  hipLaunchKernelGGL(category_table_bwd_kernel, dim3(grid_for((int64_t)(na + nb + nc) * D)), dim3(256), 0,
                     (hipStream_t)stream, g, na, nb, nc, D, da, db, dc);
  HIVT_CHECK_LAUNCH(what);
  return HIVT_OK;
}
