// The tile core of the dense products (operand staging, the MFMA reduction loop, the direct epilogue): shared by
// the three products of linear.hip and by the critics' layer chains of critic.hip.  Everything lives in an
// unnamed namespace: each translation unit that includes this header gets its own copy.
#pragma once

#include "common.h"

#include <math.h>

namespace hivt {
namespace {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kBK = 32;
constexpr int kThreads = 256;
constexpr unsigned kSmallGrid = 384;  // workgroups: up to 1.5 per compute unit
#ifndef HIVT_DEPTH4
#define HIVT_DEPTH4 0
#endif
#ifndef HIVT_INTERLEAVE
#define HIVT_INTERLEAVE 1
#endif

enum Epilogue : int { EPI_NONE = 0, EPI_RELU = 1, EPI_LN = 2, EPI_LN_RELU = 3 };

// Timeline probe for kernel tuning (tools/probe_linear.py builds a private copy of this file with
// -DHIVT_PROBE); the shipped library compiles the stamps away.
#ifdef HIVT_PROBE
__device__ long long* d_probe = nullptr;
#define HIVT_STAMP(i)                                                                                   \
  do {                                                                                                  \
    if (threadIdx.x == 0 && d_probe != nullptr)                                                         \
      d_probe[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = (long long)wall_clock64();     \
  } while (0)
#else
#define HIVT_STAMP(i) do { } while (0)
#endif

// Workgroups that read the same rows of an operand (the column tiles of one row tile; the tiles of one row chunk
// of a weight gradient) should share an L2.  Workgroups are dealt round-robin over the eight XCDs in the order of
// their linear id, so the `inner` siblings of eight consecutive `outer` units are walked together, one unit per
// XCD: sibling s of unit u gets a linear id that agrees modulo 8 with its other siblings.  Placement is an
// observation, not a contract -- only speed depends on it.  linear: the workgroup's id in dispatch order.
__device__ __forceinline__ void xcd_walk(int linear, int outer_count, int inner_count, int& outer, int& inner) {
  const int grouped = outer_count / 8 * 8 * inner_count;
  if (linear < grouped) {
    const int rem = linear % (8 * inner_count);
    outer = linear / (8 * inner_count) * 8 + rem % 8;
    inner = rem / 8;
  } else {
    outer = outer_count / 8 * 8 + (linear - grouped) / inner_count;
    inner = (linear - grouped) % inner_count;
  }
}

// Operand descriptor: logical matrix Op[row][red] (row = output index, red = reduction index).
// TRANS = true : memory is [row][red] (contiguous along the reduction)  -> transposing stage
// TRANS = false: memory is [red][row] (contiguous along the output row) -> straight copy
struct Operand {
  const float* p;
  int64_t ld;     // leading dimension in floats
  int32_t rows;   // valid output rows
  int32_t red;    // valid reduction length
  int32_t vec_ok; // 16-byte vector loads are legal (alignment + multiples of 4)
};

// Operand precision of a product.
//   PREC_F32    fp32 operands on v_mfma_f32_32x32x2_f32 (exact)
//   PREC_BF16   operands rounded to bfloat16 (nearest even) on v_mfma_f32_32x32x16_bf16
//   PREC_SPLIT  every operand value x enters as the pair (hi, lo) = (bf16(x), bf16(x - hi)) and a product is
//               the three bf16 products hi*hi + hi*lo + lo*hi: 16 significant bits per operand, results
//               within about 1e-5 of the fp32 ones at three matrix instructions instead of one
enum : int { PREC_F32 = 0, PREC_BF16 = 1, PREC_SPLIT = 2 };

// bf16 modes: the LDS image holds bfloat16 [row][k] (k contiguous, row pitch kBK + 8 halves = 80 bytes: the
// 16-byte operand reads of 16 consecutive rows then start on 16 distinct groups of four banks); values are
// rounded on their way from the fp32 registers to LDS, sums that leave the stage stay fp32.  PREC_SPLIT
// keeps a second image of the same shape (the lo parts) PLANE_H halves behind the first.
// MASKED = false: the caller guarantees that every element the stage loads lies inside the operand (a tile in
// the interior, whole reduction steps), so the validity bits are neither built nor applied.
// MEMH: the operand's memory holds bfloat16 values (gradients that only bf16 products consume are kept that way):
// a piece is an 8-byte load, widened to fp32 in registers (exact) so that everything after the load is shared.
template <int ROWS, bool TRANS, bool VEC, bool ACCUM = false, int PREC = PREC_F32, bool MASKED = true,
          bool MEMH = false>
struct Stage {
  static_assert(!MEMH || (VEC && PREC == PREC_BF16), "bfloat16 memory: vector loads, bf16 products");
  static constexpr int ELEM = MEMH ? 2 : 4;  // bytes per element in memory
  static constexpr bool BF16 = PREC != PREC_F32;
  static constexpr bool SPLIT3 = PREC == PREC_SPLIT;
  static_assert(!SPLIT3 || TRANS, "split operands are built for [row][k] memory images only (the forward product)");
  static constexpr int PITCH = TRANS ? ROWS + 1 : ROWS + 4;
  // halves per row of the bf16 image: 80-byte rows for images written by the paired staging, bare 64-byte rows
  // (chunks exchanged by the row's bits 2-3, see image_half) for the others
  static constexpr int PITCH_H = (BF16 && !TRANS) ? kBK + 8 : kBK;
  static constexpr int PLANE_H = ROWS * PITCH_H;  // halves of one image
  static constexpr int FLOATS = BF16 ? (SPLIT3 ? 2 : 1) * PLANE_H / 2 : kBK * PITCH;  // LDS floats of one staged step
  static constexpr int NVEC = ROWS * kBK / 4 / kThreads;  // float4 per thread per step
  // bf16 image of a row-major operand (reduction index strided): a thread owns pieces (2q, 2q + 1) = two
  // consecutive reduction indices of the same four rows, so the rounded values leave as four 32-bit
  // stores per pair instead of eight 16-bit ones
  static constexpr bool PAIRED = BF16 && !TRANS;
  static_assert(!PAIRED || NVEC % 2 == 0, "paired staging needs an even number of pieces");
  static_assert(!PAIRED || ROWS % 64 == 0, "paired staging covers 64-row segments");
  // (the running column sums rely on a thread staging the same four rows in every piece)
  static_assert(!(PAIRED && ACCUM) || ROWS == 64, "column sums of paired staging: one 64-row segment");

  // bf16 image: the four 16-byte chunks of a row's 32 k trade places by the row's bits 4-5.  Sixteen
  // consecutive rows (an aligned group: one exchange for all of them) still start on sixteen distinct groups
  // of four of the 64 banks, which is what the operand reads (one chunk per lane) need; the 32-bit stores of
  // the paired staging gain: sixteen lanes hold rows 4 apart, which an 80-byte row pitch maps to four banks --
  // with the exchange they spread over sixteen, and the four quarter-waves (four different words of the
  // chunk) over all sixty-four.
  // Images of [row][k] operands (not written by the paired staging) have rows of exactly 64 bytes: a quarter of
  // the image's LDS goes, and with it a workgroup per compute unit more fits.  Their chunks trade places by the
  // row's bits 2-3: the sixteen lanes of a ds_read_b128 group read sixteen rows whose (row & 3, row >> 2 & 3)
  // pairs are all different, i.e. sixteen distinct groups of four banks; the 8-byte stores of a row fill its 64
  // bytes whatever the order.
  __device__ static __forceinline__ int image_half(int row, int k) {
    if constexpr (!PAIRED) return row * PITCH_H + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7);
    return row * PITCH_H + ((((k >> 3) ^ (row >> 4)) & 3) << 3) + (k & 7);
  }

  // paired staging: pair p of a step = rows 4 (p % 16) .. + 3 of 64-row segment p / 256, reduction indices
  // 2 ((p / 16) % 16) and + 1: a quarter-wave reads 256 contiguous bytes and owns one reduction pair
  __device__ static __forceinline__ void pair_place(int pair, int& row, int& red) {
    row = (pair >> 8) * 64 + (pair & 15) * 4;
    red = 2 * ((pair >> 4) & 15);
  }
  float4 v[NVEC];
  unsigned ok;  // 4 validity bits per float4; applied when the values are written to LDS
  float4 colsum = {0.f, 0.f, 0.f, 0.f};  // ACCUM: running sum of everything this thread staged

  // Branch-free: every lane loads from a clamped in-bounds address; out-of-range values are zeroed
  // later, in store(), so the loads of a step issue back to back and stay in flight under the MFMAs.
  __device__ __forceinline__ void load(const Operand& op, int row0, int red0, int tid) {
    ok = 0u;
#pragma unroll
    for (int r = 0; r < NVEC; ++r) {
      const int idx = tid + r * kThreads;
      int row, red;
      if (TRANS) {
        row = idx / (kBK / 4);
        red = (idx % (kBK / 4)) * 4;
      } else if (PAIRED) {
        pair_place(tid + (r >> 1) * kThreads, row, red);
        red += r & 1;
      } else {
        red = idx / (ROWS / 4);
        row = (idx % (ROWS / 4)) * 4;
      }
      const int gr = row0 + row, gk = red0 + red;
      // index along the contiguous dimension (4 consecutive elements) and along the strided one
      const int fast = TRANS ? gk : gr, fast_n = TRANS ? op.red : op.rows;
      const int slow = TRANS ? gr : gk, slow_n = TRANS ? op.rows : op.red;
      const bool slow_ok = slow < slow_n;
      const int slow_c = slow_ok ? slow : slow_n - 1;
      if (VEC) {
        // fast_n % 4 == 0 here: all four elements are valid or none is
        const int fast_c = fast < fast_n ? fast : fast_n - 4;
        if constexpr (MEMH)
          v[r] = widen(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(op.p) +
                                                       (int64_t)slow_c * op.ld + fast_c));
        else
          v[r] = *reinterpret_cast<const float4*>(op.p + (int64_t)slow_c * op.ld + fast_c);
        ok |= (slow_ok && fast < fast_n) ? (0xFu << (4 * r)) : 0u;
      } else {
        const float* base = op.p + (int64_t)slow_c * op.ld;
        const int last = fast_n - 1;
        v[r].x = base[min(fast + 0, last)];
        v[r].y = base[min(fast + 1, last)];
        v[r].z = base[min(fast + 2, last)];
        v[r].w = base[min(fast + 3, last)];
        unsigned m = 0u;
        m |= (fast + 0 < fast_n) ? 1u : 0u;
        m |= (fast + 1 < fast_n) ? 2u : 0u;
        m |= (fast + 2 < fast_n) ? 4u : 0u;
        m |= (fast + 3 < fast_n) ? 8u : 0u;
        ok |= slow_ok ? (m << (4 * r)) : 0u;
      }
    }
  }

  // Interior tiles (MASKED = false, VEC): a thread's piece r of every step lies at a fixed byte offset from the
  // step's origin inside the operand, so the offsets are computed once (offsets()), the origin is a wave-uniform
  // pointer that the caller moves from step to step with scalar arithmetic (origin() + step * step_bytes()),
  // and a fetch is NVEC loads without any per-lane address work.  `live` false: the step lies past the end of
  // the reduction range (the deep routes fetch ahead); the caller then passes the last step's origin, and the
  // values take no part in the running column sums.
  bool live = true;

  __device__ static __forceinline__ void offsets(const Operand& op, int tid, unsigned (&off)[NVEC]) {
#pragma unroll
    for (int r = 0; r < NVEC; ++r) {
      const int idx = tid + r * kThreads;
      int row, red;
      if (TRANS) {
        row = idx / (kBK / 4);
        red = (idx % (kBK / 4)) * 4;
      } else if (PAIRED) {
        pair_place(tid + (r >> 1) * kThreads, row, red);
        red += r & 1;
      } else {
        red = idx / (ROWS / 4);
        row = (idx % (ROWS / 4)) * 4;
      }
      off[r] = (unsigned)(TRANS ? row * (int)op.ld + red : red * (int)op.ld + row) * (unsigned)ELEM;
    }
  }

  __device__ static __forceinline__ const char* origin(const Operand& op, int row0, int red0) {
    return reinterpret_cast<const char*>(op.p) +
           (TRANS ? (int64_t)row0 * op.ld + red0 : (int64_t)red0 * op.ld + row0) * ELEM;
  }

  __device__ static __forceinline__ int64_t step_bytes(const Operand& op) {
    return (TRANS ? (int64_t)kBK : (int64_t)kBK * op.ld) * ELEM;
  }

  // four bfloat16 values -> fp32 (exact)
  __device__ static __forceinline__ float4 widen(uint2 h) {
    return make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16),
                       __uint_as_float(h.y & 0xffff0000u));
  }

  // every offset of a step fits 32 bits (and the int arithmetic of offsets())
  __device__ static __forceinline__ bool offsets_fit(const Operand& op) {
    return op.ld * (int64_t)(TRANS ? ROWS : kBK) < (1ll << 28);
  }

  __device__ __forceinline__ void fetch(const char* step_origin, const unsigned (&off)[NVEC], bool is_live) {
    live = is_live;
#pragma unroll
    for (int r = 0; r < NVEC; ++r) {
      if constexpr (MEMH)
        v[r] = widen(*reinterpret_cast<const uint2*>(step_origin + off[r]));
      else
        v[r] = *reinterpret_cast<const float4*>(step_origin + off[r]);
    }
  }

  // One float4 of the staged step (r < NVEC) goes to LDS; store() sends all of them.
  __device__ __forceinline__ float4 masked(int r) const {
    if constexpr (!MASKED) return v[r];
    return zero_invalid(r);
  }

  // The running column sums (ACCUM) always go by the validity bits: the deep routes stage steps past the end
  // of the reduction range, which are never multiplied but would be summed.
  __device__ __forceinline__ float4 zero_invalid(int r) const {
    const unsigned m = ok >> (4 * r);
    float4 val;
    val.x = (m & 1u) ? v[r].x : 0.0f;
    val.y = (m & 2u) ? v[r].y : 0.0f;
    val.z = (m & 4u) ? v[r].z : 0.0f;
    val.w = (m & 8u) ? v[r].w : 0.0f;
    return val;
  }

  __device__ __forceinline__ void store_piece(float* lds, int tid, int r) {
    if constexpr (PAIRED) {
      if ((r & 1) == 0) return;  // leaves together with its odd neighbour
      const float4 lo = masked(r - 1), hi = masked(r);
      if (ACCUM) {
        if (MASKED || live) {  // `live` is uniform over the wave
          colsum.x += lo.x + hi.x;
          colsum.y += lo.y + hi.y;
          colsum.z += lo.z + hi.z;
          colsum.w += lo.w + hi.w;
        }
      }
      // row i of the pair: (lo[i], hi[i]) are its two consecutive reduction indices, one packed conversion each
      const float l4[4] = {lo.x, lo.y, lo.z, lo.w}, h4[4] = {hi.x, hi.y, hi.z, hi.w};
      int row, red;
      pair_place(tid + (r >> 1) * kThreads, row, red);
      unsigned* image = reinterpret_cast<unsigned*>(lds);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 two = {l4[i], h4[i]};
        image[image_half(row + i, red) / 2] = __builtin_bit_cast(unsigned, __builtin_convertvector(two, bf16x2));
      }
      return;
    }
    const int idx = tid + r * kThreads;
    const float4 val = masked(r);
    if (ACCUM) {
      if (MASKED || live) {
        colsum.x += val.x;
        colsum.y += val.y;
        colsum.z += val.z;
        colsum.w += val.w;
      }
    }
    if (BF16) {
      const f32x4 f = {val.x, val.y, val.z, val.w};
      const bf16x4 h = __builtin_convertvector(f, bf16x4);
      // TRANS (the other layout took the paired path above): four consecutive k of one row, one 8-byte store
      unsigned short* image = reinterpret_cast<unsigned short*>(lds);
      const int row = idx / (kBK / 4);
      const int red = (idx % (kBK / 4)) * 4;
      *reinterpret_cast<bf16x4*>(image + image_half(row, red)) = h;
      if constexpr (SPLIT3) {
        const f32x4 rest = f - __builtin_convertvector(h, f32x4);  // exact: hi shares f's leading bits
        *reinterpret_cast<bf16x4*>(image + PLANE_H + image_half(row, red)) = __builtin_convertvector(rest, bf16x4);
      }
    } else if (TRANS) {
      const int row = idx / (kBK / 4);
      const int red = (idx % (kBK / 4)) * 4;
      lds[(red + 0) * PITCH + row] = val.x;
      lds[(red + 1) * PITCH + row] = val.y;
      lds[(red + 2) * PITCH + row] = val.z;
      lds[(red + 3) * PITCH + row] = val.w;
    } else {
      const int red = idx / (ROWS / 4);
      const int row = (idx % (ROWS / 4)) * 4;
      *reinterpret_cast<float4*>(lds + red * PITCH + row) = val;
    }
  }

  __device__ __forceinline__ void store(float* lds, int tid) {
#pragma unroll
    for (int r = 0; r < NVEC; ++r) store_piece(lds, tid, r);
  }
};

// What the direct epilogue applies to an accumulator element before it is stored:
//   out = residual + dropout([relu | leaky_relu](acc + bias[col] + add[row or add_index[row]][col]))
struct Tail {
  const float* bias = nullptr;
  const float* add = nullptr;
  int64_t add_ld = 0;
  const int32_t* add_index = nullptr;  // optional: row gr adds add[add_index[gr]] (rows of a small table)
  int32_t add_rows = 0;                // rows of that table; indices are clamped into it
  float neg_slope = 1.0f;              // LeakyReLU slope for negative values (1 = identity), applied before dropout
  float drop_p = 0.0f;                 // dropout probability; the mask is dropout_uniform(seed, row * cols + col) >= p
  uint64_t seed = 0;
  const uint64_t* seed_ptr = nullptr;  // optional device-side value added to seed
  const float* residual = nullptr;
  int64_t res_ld = 0;
  // residual_is_mask: the tensor in `residual` is not added; it gates the value instead:
  // out = value * (residual > 0 ? mask_scale : residual < 0 ? mask_neg_scale : 0): the gradient through a
  // ReLU or LeakyReLU [+ dropout] whose output `residual` is (a dropped element is exactly 0)
  bool residual_is_mask = false;
  float mask_scale = 1.0f;
  float mask_neg_scale = 0.0f;
  // the result is stored as bfloat16 (nearest even): `out` points at 2-byte elements, out_ld counts elements
  bool out_half = false;
  // One bit per element instead of a whole tensor as the gate of a ReLU [+ dropout]: bit (col & 31) of word
  // [row][col >> 5] (bits_ld words per row).  sign_bits: the forward writes value > 0 of what it stores (before a
  // residual is added).  gate_bits: the gradient is multiplied by mask_scale where the bit is set and is zero elsewhere.
  uint32_t* sign_bits = nullptr;
  const uint32_t* gate_bits = nullptr;
  int32_t bits_ld = 0;
};

__device__ __forceinline__ unsigned short to_bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

// DEEP = 1 / 2: two / four register sets per operand, so that the global loads of steps s + 2 (.. s + 4) are
// in flight while step s is multiplied (for grids where no other workgroup shares the compute unit to hide
// the load latency behind; it costs registers, hence occupancy, on large grids).  With four sets a reduction
// of 128 has all of its loads in flight at once: one memory latency per tile instead of two.
// MEMH_A: operand A's memory holds bfloat16 values.
template <int BM, int BN, bool TA, bool TB, bool VEC, bool ACC_A = false, int DEEP = 0, int PREC = PREC_F32,
          bool MEMH_A = false>
struct Core {
  static constexpr bool BF16 = PREC != PREC_F32;
  static constexpr bool SPLIT3 = PREC == PREC_SPLIT;
  template <bool MASKED> using StageA = Stage<BM, TA, VEC, ACC_A, PREC, MASKED, MEMH_A>;
  template <bool MASKED> using StageB = Stage<BN, TB, VEC, false, PREC, MASKED>;
  using SA = StageA<true>;   // sizes and pitches do not depend on the masking
  using SB = StageB<true>;
  static constexpr int A_FLOATS = SA::FLOATS;
  static constexpr int B_FLOATS = SB::FLOATS;
  static constexpr int C_PITCH = BN + 4;
  static constexpr int STAGE_FLOATS = 2 * (A_FLOATS + B_FLOATS);
  static constexpr int C_FLOATS = BM * C_PITCH;
  static constexpr int LDS_FLOATS = STAGE_FLOATS > C_FLOATS ? STAGE_FLOATS : C_FLOATS;
  static constexpr int WN = BN / 64;  // waves along the columns; each wave owns 32 x 64
  static_assert(SA::NVEC + SB::NVEC <= kBK / 4, "one staged float4 per second k-pair must cover a step");

  // Accumulates the product over [red_begin, red_end) into the calling wave's two 32x32 accumulators
  // (rows wr*32.., columns wc*64.. and wc*64+32..).  All threads of the block must call it.
  // With ACC_A, *a_colsum receives this thread's sum of the operand-A values it staged.
  __device__ static void accumulate(const Operand& a, const Operand& b, int row0, int col0, int red_begin,
                                    int red_end, float* lds, f32x16& acc0, f32x16& acc1,
                                    float4* a_colsum = nullptr) {
    // a tile in the interior of both operands whose reduction range is whole steps (whole rounds of the
    // register sets on the deep routes, which run the padded steps too) needs no validity masks
    constexpr int ROUND = kBK * (DEEP > 0 ? 2 * DEEP : 1);
    const bool whole = VEC && row0 + BM <= a.rows && col0 + BN <= b.rows && red_end <= a.red && red_end <= b.red &&
                       (red_end - red_begin) % ROUND == 0 && red_end > red_begin && SA::offsets_fit(a) &&
                       SB::offsets_fit(b);
    if (whole)
      accumulate_impl<false>(a, b, row0, col0, red_begin, red_end, lds, acc0, acc1, a_colsum);
    else
      accumulate_impl<true>(a, b, row0, col0, red_begin, red_end, lds, acc0, acc1, a_colsum);
  }

  template <bool MASKED>
  __device__ static void accumulate_impl(const Operand& a, const Operand& b, int row0, int col0, int red_begin,
                                         int red_end, float* lds, f32x16& acc0, f32x16& acc1, float4* a_colsum) {
    using SA = StageA<MASKED>;
    using SB = StageB<MASKED>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.0f;
      acc1[r] = 0.0f;
    }
    // buffer b of operand A starts at b * BUF, of operand B at b * BUF + A_FLOATS (plain integer
    // offsets into the one __shared__ array, so every access stays an LDS instruction)
    constexpr int BUF = A_FLOATS + B_FLOATS;
    const int steps = (red_end - red_begin + kBK - 1) / kBK;
    // All operand reads of a step are issued before its first MFMA (two halves of 8 k-pairs), so the LDS
    // latency is paid once per half instead of once per k-pair when a wave has its SIMD to itself.
    // between(kp) runs after the MFMAs of k-pair kp (0..15): work placed there is issued while the
    // matrix unit is busy.
    auto multiply = [&](int cur, auto&& between) {
      if constexpr (BF16) {
        // v_mfma_f32_32x32x16_bf16: lane (row lo, half hi) supplies k = 8*hi .. 8*hi + 7 of its row in each
        // group of 16 k; two groups cover the 32 k of a step, all operand reads (16 bytes each) go first
        const unsigned short* image = reinterpret_cast<const unsigned short*>(lds + cur);
        // (rows lo and lo + 64 of a tile exchange their chunks the same way; lo + 32 is looked up on its own)
        const unsigned short* pa = image;
        const unsigned short* pb = image + 2 * A_FLOATS;
        constexpr int G = kBK / 16;
        bf16x8 ah[G], b0h[G], b1h[G], al[G], b0l[G], b1l[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int ka = SA::image_half(wr * 32 + lo, 16 * g + 8 * hi);
          const int kb = SB::image_half(wc * 64 + lo, 16 * g + 8 * hi);
          ah[g] = *reinterpret_cast<const bf16x8*>(pa + ka);
          b0h[g] = *reinterpret_cast<const bf16x8*>(pb + kb);
          const int kb1 = SB::image_half(wc * 64 + 32 + lo, 16 * g + 8 * hi);
          b1h[g] = *reinterpret_cast<const bf16x8*>(pb + kb1);
          if constexpr (SPLIT3) {
            al[g] = *reinterpret_cast<const bf16x8*>(pa + SA::PLANE_H + ka);
            b0l[g] = *reinterpret_cast<const bf16x8*>(pb + SB::PLANE_H + kb);
            b1l[g] = *reinterpret_cast<const bf16x8*>(pb + SB::PLANE_H + kb1);
          }
        }
        // HIVT_FENCE: nothing moves across -- the conversions of the operand set that is being published
        // (vector ALU work inside between()) stay between the matrix instructions they are written between
        // instead of queueing up behind the last one
#if HIVT_INTERLEAVE
#define HIVT_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define HIVT_FENCE() do { } while (0)
#endif
        HIVT_FENCE();
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if constexpr (SPLIT3) {  // the small terms first, the leading one last
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g], b0h[g], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g], b1h[g], acc1, 0, 0, 0);
            HIVT_FENCE();
            between(8 * g + 1);
            HIVT_FENCE();
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], b0l[g], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], b1l[g], acc1, 0, 0, 0);
            HIVT_FENCE();
            between(8 * g + 3);
            HIVT_FENCE();
          }
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], b0h[g], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], b1h[g], acc1, 0, 0, 0);
          HIVT_FENCE();
          if constexpr (!SPLIT3) {
            between(8 * g + 1);
            between(8 * g + 3);
          }
          between(8 * g + 5);
          between(8 * g + 7);
          HIVT_FENCE();
        }
#undef HIVT_FENCE
        return;
      }
      const int ao = cur + wr * 32 + lo;
      const int bo = cur + A_FLOATS + wc * 64 + lo;
      constexpr int HALF = kBK / 4;  // k-pairs per half
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float av[HALF], b0[HALF], b1[HALF];
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
          const int kk = (half * HALF + i) * 2;
          av[i] = lds[ao + (kk + hi) * SA::PITCH];
          b0[i] = lds[bo + (kk + hi) * SB::PITCH];
          b1[i] = lds[bo + (kk + hi) * SB::PITCH + 32];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs: the scheduler would sink them
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b0[i], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b1[i], acc1, 0, 0, 0);
          between(half * HALF + i);
        }
      }
    };
    auto nothing = [](int) {};
    // Publishes a staged register set in LDS piecewise from inside multiply(): one float4 after every
    // second k-pair, all of them out by k-pair 11 of 16.
    auto publish = [&](SA& pa, SB& pb, int dst) {
      return [&pa, &pb, dst, lds, tid](int kp) {
        const int piece = kp >> 1;
        if ((kp & 1) == 0 || piece >= SA::NVEC + SB::NVEC) return;
        if (piece < SA::NVEC)
          pa.store_piece(lds + dst, tid, piece);
        else
          pb.store_piece(lds + dst + A_FLOATS, tid, piece - SA::NVEC);
      };
    };
    // MASKED = false: fixed per-thread offsets and wave-uniform step origins (Stage::fetch); the step that a
    // fetch names is clamped to the last one, so fetching ahead of the range stays inside the operand
    unsigned aoff[SA::NVEC], boff[SB::NVEC];
    const char* a0 = nullptr;
    const char* b0 = nullptr;
    int64_t a_step = 0, b_step = 0;
    if constexpr (!MASKED) {
      SA::offsets(a, tid, aoff);
      SB::offsets(b, tid, boff);
      a0 = SA::origin(a, row0, red_begin);
      b0 = SB::origin(b, col0, red_begin);
      a_step = SA::step_bytes(a);
      b_step = SB::step_bytes(b);
    }
    auto fetch = [&](SA& pa, SB& pb, int step) {
      if constexpr (MASKED) {
        pa.load(a, row0, red_begin + step * kBK, tid);
        pb.load(b, col0, red_begin + step * kBK, tid);
      } else {
        const int t = step < steps ? step : steps - 1;
        pa.fetch(a0 + (int64_t)t * a_step, aoff, step < steps);
        pb.fetch(b0 + (int64_t)t * b_step, boff, step < steps);
      }
    };
    if constexpr (DEEP == 0) {
      SA sa;
      SB sb;
      if (steps > 0) {
        fetch(sa, sb, 0);
        sa.store(lds, tid);
        sb.store(lds + A_FLOATS, tid);
      }
      __syncthreads();
      HIVT_STAMP(1);
      for (int s = 0; s < steps; ++s) {
        const int cur = (s & 1) * BUF;
        const int nxt = BUF - cur;
        if (s + 1 < steps) fetch(sa, sb, s + 1);
        multiply(cur, nothing);
        if (s + 1 < steps) {
          sa.store(lds + nxt, tid);
          sb.store(lds + nxt + A_FLOATS, tid);
        }
        __syncthreads();
      }
      if (ACC_A) *a_colsum = sa.colsum;
    } else {
      // Every load and every LDS store below is unconditional: loads past the end of the reduction
      // range are clamped in bounds and masked to zero (Stage::load), and the step count is padded to a
      // multiple of the set count by all-zero steps.  Straight-line code lets the compiler count the loads
      // in flight, so waiting for one register set leaves the other sets' loads outstanding.
      constexpr int D = 2 * DEEP;  // register sets = steps in flight; set j holds steps j, j + D, ...
      SA sa[D];
      SB sb[D];
#pragma unroll
      for (int j = 0; j < D; ++j) fetch(sa[j], sb[j], j);
      sa[0].store(lds, tid);
      sb[0].store(lds + A_FLOATS, tid);
      __syncthreads();
      HIVT_STAMP(1);
      for (int s = 0; s < steps; s += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          // step s + j: multiply buffer j & 1; set j (its step is in LDS already) fetches step s + j + D; set
          // j + 1 (step s + j + 1) is published in the other buffer while the matrix unit is busy
          fetch(sa[j], sb[j], s + j + D);
          multiply((j & 1) * BUF, publish(sa[(j + 1) % D], sb[(j + 1) % D], ((j + 1) & 1) * BUF));
          __syncthreads();
        }
      }
      if (ACC_A) {
        float4 c = sa[0].colsum;
#pragma unroll
        for (int j = 1; j < D; ++j) {
          c.x += sa[j].colsum.x;
          c.y += sa[j].colsum.y;
          c.z += sa[j].colsum.z;
          c.w += sa[j].colsum.w;
        }
        *a_colsum = c;
      }
    }
  }

  // accumulators -> LDS tile C[row][col] (pitch C_PITCH); the staging buffers are dead after the last
  // barrier of accumulate().  Ends with a barrier.
  __device__ static void spill(float* lds, const f32x16& acc0, const f32x16& acc1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      lds[row * C_PITCH + wc * 64 + lo] = acc0[r];
      lds[row * C_PITCH + wc * 64 + 32 + lo] = acc1[r];
    }
    __syncthreads();
  }

  // Bias values of the two columns a lane owns; loaded before the reduction loop so that the
  // latency is not paid in the epilogue.
  __device__ static float2 bias_pair(const Tail& t, int col0, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = col0 + (wave % WN) * 64 + (lane & 31), c1 = c0 + 32;
    float2 b;
    b.x = (t.bias != nullptr && c0 < cols) ? t.bias[c0] : 0.0f;
    b.y = (t.bias != nullptr && c1 < cols) ? t.bias[c1] : 0.0f;
    return b;
  }

  // accumulators -> global memory without the LDS round trip: for a fixed register the 32 low lanes
  // hold 32 consecutive columns of one row (a 128-byte segment) and the high lanes the row 4 below.
  // The added / residual values of all 16 registers are requested first and consumed afterwards, so
  // their latencies overlap instead of adding up.
  // The same for a tile that lies inside the output (no bounds to check): every feature of the tail is one
  // pass over the sixteen register pairs behind ONE uniform branch, addresses are 32-bit offsets from the
  // tile's corner, and the dropout draws come from the 32-bit mix.
  template <bool RELU>
  __device__ static void store_interior(const f32x16& acc0, const f32x16& acc1, int row0, int col0, int cols,
                                        float* __restrict__ out, int64_t out_ld, const Tail& t, float2 bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int lo = lane & 31, hi = lane >> 5;
    const int lrow = wr * 32 + 4 * hi, lcol = wc * 64 + lo;
    float v0[16], v1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v0[r] = acc0[r] + bias.x;
      v1[r] = acc1[r] + bias.y;
    }
    if (t.add != nullptr) {
      const float* corner = t.add + (int64_t)row0 * t.add_ld + col0;
      const int ld = (int)t.add_ld, off = lrow * ld + lcol;
      float a0[16], a1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = off + ((r & 3) + 8 * (r >> 2)) * ld;
        a0[r] = corner[o];
        a1[r] = corner[o + 32];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v0[r] += a0[r];
        v1[r] += a1[r];
      }
    }
    if (RELU) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v0[r] = fmaxf(v0[r], 0.0f);
        v1[r] = fmaxf(v1[r], 0.0f);
      }
    } else if (t.neg_slope != 1.0f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v0[r] = v0[r] > 0.0f ? v0[r] : v0[r] * t.neg_slope;
        v1[r] = v1[r] > 0.0f ? v1[r] : v1[r] * t.neg_slope;
      }
    }
    if (t.drop_p > 0.0f) {
      const float keep_scale = 1.0f / (1.0f - t.drop_p);
      const uint64_t seed = t.seed + (t.seed_ptr != nullptr ? *t.seed_ptr : 0ull);
      const uint64_t first = (uint64_t)(row0 + lrow) * (uint64_t)cols + (uint64_t)(col0 + lcol);
      if (((uint64_t)(row0 + BM) * (uint64_t)cols >> 32) == 0) {  // the tile's indices fit 32 bits (uniform)
        const uint32_t f32 = (uint32_t)first, c32 = (uint32_t)cols;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t i0 = f32 + (uint32_t)((r & 3) + 8 * (r >> 2)) * c32;
          v0[r] = uniform01_u32(seed, i0) >= t.drop_p ? v0[r] * keep_scale : 0.0f;
          v1[r] = uniform01_u32(seed, i0 + 32u) >= t.drop_p ? v1[r] * keep_scale : 0.0f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint64_t i0 = first + (uint64_t)((r & 3) + 8 * (r >> 2)) * (uint64_t)cols;
          v0[r] = dropout_uniform(seed, i0) >= t.drop_p ? v0[r] * keep_scale : 0.0f;
          v1[r] = dropout_uniform(seed, i0 + 32) >= t.drop_p ? v1[r] * keep_scale : 0.0f;
        }
      }
    }
    if (t.sign_bits != nullptr) {  // (interior tile: every lane is active and every word is whole)
      uint32_t* corner = t.sign_bits + (int64_t)(row0 + lrow) * t.bits_ld + ((col0 + wc * 64) >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned long long b0 = __ballot(v0[r] > 0.0f), b1 = __ballot(v1[r] > 0.0f);
        if (lo == 0) {
          uint32_t* word = corner + ((r & 3) + 8 * (r >> 2)) * t.bits_ld;
          word[0] = hi ? (uint32_t)(b0 >> 32) : (uint32_t)b0;
          word[1] = hi ? (uint32_t)(b1 >> 32) : (uint32_t)b1;
        }
      }
    }
    if (t.gate_bits != nullptr) {
      const uint32_t* corner = t.gate_bits + (int64_t)(row0 + lrow) * t.bits_ld + ((col0 + wc * 64) >> 5);
      uint32_t g0[16], g1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t* word = corner + ((r & 3) + 8 * (r >> 2)) * t.bits_ld;
        g0[r] = word[0];
        g1[r] = word[1];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v0[r] = ((g0[r] >> lo) & 1u) ? v0[r] * t.mask_scale : 0.0f;
        v1[r] = ((g1[r] >> lo) & 1u) ? v1[r] * t.mask_scale : 0.0f;
      }
    }
    if (t.residual != nullptr) {
      const float* corner = t.residual + (int64_t)row0 * t.res_ld + col0;
      const int ld = (int)t.res_ld, off = lrow * ld + lcol;
      float q0[16], q1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = off + ((r & 3) + 8 * (r >> 2)) * ld;
        q0[r] = corner[o];
        q1[r] = corner[o + 32];
      }
      if (t.residual_is_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v0[r] = q0[r] > 0.0f ? v0[r] * t.mask_scale : (q0[r] < 0.0f ? v0[r] * t.mask_neg_scale : 0.0f);
          v1[r] = q1[r] > 0.0f ? v1[r] * t.mask_scale : (q1[r] < 0.0f ? v1[r] * t.mask_neg_scale : 0.0f);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v0[r] += q0[r];
          v1[r] += q1[r];
        }
      }
    }
    const int ld = (int)out_ld, off = lrow * ld + lcol;
    if (t.out_half) {
      unsigned short* corner = reinterpret_cast<unsigned short*>(out) + (int64_t)row0 * out_ld + col0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = off + ((r & 3) + 8 * (r >> 2)) * ld;
        corner[o] = to_bf16_bits(v0[r]);
        corner[o + 32] = to_bf16_bits(v1[r]);
      }
      return;
    }
    float* corner = out + (int64_t)row0 * out_ld + col0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = off + ((r & 3) + 8 * (r >> 2)) * ld;
      corner[o] = v0[r];
      corner[o + 32] = v1[r];
    }
  }

  template <bool RELU>
  __device__ static void store_direct(const f32x16& acc0, const f32x16& acc1, int row0, int col0, int rows, int cols,
                                      float* __restrict__ out, int64_t out_ld, const Tail& t, float2 bias) {
    // tiles inside the output whose row offsets fit 32 bits take the short route
    const bool interior = row0 + BM <= rows && col0 + BN <= cols && t.add_index == nullptr &&
                          (int64_t)BM * out_ld < (1ll << 30) && (int64_t)BM * t.add_ld < (1ll << 30) &&
                          (int64_t)BM * t.res_ld < (1ll << 30);
    if (interior)
      store_interior<RELU>(acc0, acc1, row0, col0, cols, out, out_ld, t, bias);
    else
      store_general<RELU>(acc0, acc1, row0, col0, rows, cols, out, out_ld, t, bias);
  }

  template <bool RELU>
  __device__ static void store_general(const f32x16& acc0, const f32x16& acc1, int row0, int col0, int rows, int cols,
                                       float* __restrict__ out, int64_t out_ld, const Tail& t, float2 bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int lo = lane & 31, hi = lane >> 5;
    const int c0 = col0 + wc * 64 + lo, c1 = c0 + 32;
    const bool ok0 = c0 < cols, ok1 = c1 < cols;
    const int c0c = ok0 ? c0 : 0, c1c = ok1 ? c1 : 0;  // in-bounds columns for the unconditional loads
    const bool drop = t.drop_p > 0.0f;
    const float keep_scale = drop ? 1.0f / (1.0f - t.drop_p) : 1.0f;
    const uint64_t seed = t.seed + ((drop && t.seed_ptr != nullptr) ? *t.seed_ptr : 0ull);
    const int row_base = row0 + wr * 32 + 4 * hi;
    float add0[16], add1[16], res0[16], res1[16];
    if (t.add != nullptr) {
      int ar[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ar[r] = min(row_base + (r & 3) + 8 * (r >> 2), rows - 1);
      if (t.add_index != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ar[r] = t.add_index[ar[r]];
#pragma unroll
        for (int r = 0; r < 16; ++r) ar[r] = min(max(ar[r], 0), t.add_rows - 1);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        add0[r] = t.add[(int64_t)ar[r] * t.add_ld + c0c];
        add1[r] = t.add[(int64_t)ar[r] * t.add_ld + c1c];
      }
    }
    if (t.residual != nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = min(row_base + (r & 3) + 8 * (r >> 2), rows - 1);
        res0[r] = t.residual[(int64_t)gr * t.res_ld + c0c];
        res1[r] = t.residual[(int64_t)gr * t.res_ld + c1c];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gr = row_base + (r & 3) + 8 * (r >> 2);
      if (gr < rows) {
        float v0 = acc0[r] + bias.x, v1 = acc1[r] + bias.y;
        if (t.add != nullptr) {
          v0 += add0[r];
          v1 += add1[r];
        }
        if (RELU) {
          v0 = fmaxf(v0, 0.0f);
          v1 = fmaxf(v1, 0.0f);
        } else if (t.neg_slope != 1.0f) {
          v0 = v0 > 0.0f ? v0 : v0 * t.neg_slope;
          v1 = v1 > 0.0f ? v1 : v1 * t.neg_slope;
        }
        if (drop) {
          const uint64_t base = (uint64_t)gr * (uint64_t)cols;
          v0 = dropout_uniform(seed, base + (uint64_t)c0) >= t.drop_p ? v0 * keep_scale : 0.0f;
          v1 = dropout_uniform(seed, base + (uint64_t)c1) >= t.drop_p ? v1 * keep_scale : 0.0f;
        }
        const int w0 = c0 >> 5, w1 = c1 >> 5;  // (uniform over the half wave: its lanes hold 32 aligned columns)
        if (t.sign_bits != nullptr) {
          const unsigned long long b0 = __ballot(ok0 && v0 > 0.0f), b1 = __ballot(ok1 && v1 > 0.0f);
          if (lo == 0) {
            uint32_t* word = t.sign_bits + (int64_t)gr * t.bits_ld;
            if (w0 < t.bits_ld) word[w0] = hi ? (uint32_t)(b0 >> 32) : (uint32_t)b0;
            if (w1 < t.bits_ld) word[w1] = hi ? (uint32_t)(b1 >> 32) : (uint32_t)b1;
          }
        }
        if (t.gate_bits != nullptr) {
          const uint32_t* word = t.gate_bits + (int64_t)gr * t.bits_ld;
          const uint32_t g0 = w0 < t.bits_ld ? word[w0] : 0u, g1 = w1 < t.bits_ld ? word[w1] : 0u;
          v0 = ((g0 >> lo) & 1u) ? v0 * t.mask_scale : 0.0f;
          v1 = ((g1 >> lo) & 1u) ? v1 * t.mask_scale : 0.0f;
        }
        if (t.residual != nullptr) {
          if (t.residual_is_mask) {
            v0 = res0[r] > 0.0f ? v0 * t.mask_scale : (res0[r] < 0.0f ? v0 * t.mask_neg_scale : 0.0f);
            v1 = res1[r] > 0.0f ? v1 * t.mask_scale : (res1[r] < 0.0f ? v1 * t.mask_neg_scale : 0.0f);
          } else {
            v0 += res0[r];
            v1 += res1[r];
          }
        }
        if (t.out_half) {
          unsigned short* o16 = reinterpret_cast<unsigned short*>(out);
          if (ok0) o16[(int64_t)gr * out_ld + c0] = to_bf16_bits(v0);
          if (ok1) o16[(int64_t)gr * out_ld + c1] = to_bf16_bits(v1);
        } else {
          if (ok0) out[(int64_t)gr * out_ld + c0] = v0;
          if (ok1) out[(int64_t)gr * out_ld + c1] = v1;
        }
      }
    }
  }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// half: the memory holds bfloat16 values (p is then a pointer to 2-byte elements in disguise, ld counts elements;
// four of them are one 8-byte vector load)
inline Operand make_operand(const float* p, int64_t ld, int32_t rows, int32_t red, bool trans, bool half = false) {
  Operand o;
  o.p = p;
  o.ld = ld;
  o.rows = rows;
  o.red = red;
  // vector loads run along the contiguous dimension: the reduction if trans, else the row index
  const int32_t fast_n = trans ? red : rows;
  const bool aligned = half ? (reinterpret_cast<uintptr_t>(p) & 7) == 0 : aligned16(p);
  o.vec_ok = (aligned && ld % 4 == 0 && fast_n % 4 == 0 && fast_n >= 4) ? 1 : 0;
  return o;
}

}  // namespace
}  // namespace hivt
