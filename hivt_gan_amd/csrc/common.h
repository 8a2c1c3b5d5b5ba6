// Shared helpers for the gfx950 (CDNA4, wave64) kernels of libhivt_mi355.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/hivt_mi355.h"

namespace hivt {

constexpr int kWave = 64;

// ---- error reporting ------------------------------------------------------------------------
// One message per host thread; forward runs on the Python thread and backward on the autograd
// worker thread, so the text must not be shared between threads.
void set_error(const char* fmt, ...);

#define HIVT_REQUIRE(cond, code, ...)          \
  do {                                         \
    if (!(cond)) {                             \
      ::hivt::set_error(__VA_ARGS__);          \
      return (code);                           \
    }                                          \
  } while (0)

#define HIVT_CHECK_LAUNCH(what)                                                         \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) {                                                            \
      ::hivt::set_error("%s: launch failed: %s", (what), hipGetErrorString(e__));       \
      return HIVT_ERR_LAUNCH;                                                           \
    }                                                                                   \
  } while (0)

#define HIVT_CHECK_HIP(expr, what)                                                      \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      ::hivt::set_error("%s: %s", (what), hipGetErrorString(e__));                      \
      return HIVT_ERR_LAUNCH;                                                           \
    }                                                                                   \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Zero `words` 32-bit words with a kernel.  The library never uses the runtime's memset: a memset
// recorded into a HIP graph is a separate node type, and every launch here must stay a plain
// kernel node so that a recorded step replays safely.
__global__ void zero_words_kernel(uint32_t* __restrict__ p, int64_t words);
inline void zero_words(void* p, int64_t words, hipStream_t stream) {
  if (words <= 0) return;
  int64_t blocks = (words + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(p),
                     words);
}

// out[c] = sum over r < rows of partial[r * cols + c], summed in a fixed order (deterministic).
// the operand-precision mask set through hivt_set_matmul_bf16()
int matmul_bf16_mask();

// out[c] (+)= sum over rows of partial[row][c], rows summed in a fixed order
void reduce_partials(const float* partial, int rows, int cols, float* out, hipStream_t stream,
                     bool accumulate = false);

// Column sums of partial rows, several outputs per launch: out[c] (c < split) and out1[c - split] (c >= split,
// out1 may be null) (+)= sum over the terms t and their rows r < rows[t] of term[t][r * pitch + c], c < cols; per
// term four row slices summed in ascending order and combined in slice order, terms added in term order.
struct ColumnSums {
  const float* term[4];
  int32_t rows[4];   // 0: no such term
  int64_t pitch;
  float* out;
  float* out1;
  int32_t cols, split;
  int32_t accumulate;
};
int reduce_column_sums_group(const ColumnSums* entries, int count, hipStream_t stream);

// dw[N,K] += dy[M,N]^T x[M,K] (hivt_linear_wgrad without the bias sums, added to what dw holds)
int linear_wgrad_accumulate(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, int64_t M, int32_t N,
                            int32_t K, float* dw, void* workspace, size_t workspace_bytes, hivt_stream_t stream);

// Dense products whose result passes through the derivative of Dropout(LeakyReLU(.)) recovered from the
// activation's saved output `gate` ([M, columns of the result]):
//   out *= gate > 0 ? pos_scale : gate < 0 ? neg_scale : 0
// y[M,N] = x[M,K] w[N,K]^T gated, and dx[M,K] = dy[M,N] w[N,K] gated.
int linear_fwd_gated(const float* x, int64_t x_ld, const float* w, int64_t w_ld, int64_t M, int32_t N, int32_t K,
                     const float* gate, int64_t gate_ld, float pos_scale, float neg_scale, float* y, int64_t y_ld,
                     hivt_stream_t stream);
int linear_bwd_data_gated(const float* dy, int64_t dy_ld, const float* w, int64_t w_ld, int64_t M, int32_t N, int32_t K,
                          const float* gate, int64_t gate_ld, float pos_scale, float neg_scale, float* dx,
                          int64_t dx_ld, hivt_stream_t stream);

// ---- device helpers -------------------------------------------------------------------------
// The value of lane (lane ^ O).  Offsets 1 and 2 stay inside a quad and offset 8 inside a row of 16 lanes, where the
// data-parallel-primitive operand modifiers move them at full rate inside the vector unit; every other offset goes
// through the LDS crossbar.  ROTATE4: the caller guarantees that lanes l and l ^ 8 hold the same value (the state
// right after the offset-8 step of a butterfly), which makes a rotation by 4 inside the row deliver the value of
// lane l ^ 4.  All lanes of a group must be active together, as with the crossbar version.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int O, bool ROTATE4 = false>
__device__ __forceinline__ float lane_xor(float v) {
  if constexpr (O == 1) return dpp_move<0xB1>(v);        // quad_perm:[1,0,3,2]
  else if constexpr (O == 2) return dpp_move<0x4E>(v);   // quad_perm:[2,3,0,1]
  else if constexpr (O == 8) return dpp_move<0x128>(v);  // row_ror:8
  else if constexpr (O == 4 && ROTATE4) return dpp_move<0x124>(v);  // row_ror:4
  else return __shfl_xor(v, O, 64);
}

// Sum over the aligned group of GROUP consecutive lanes (GROUP a power of two <= 64): the butterfly
// v += v[lane ^ o] for o = GROUP/2 ... 1, every lane ending with the same bits.
template <int GROUP>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (GROUP >= 64) v += lane_xor<32>(v);
  if constexpr (GROUP >= 32) v += lane_xor<16>(v);
  if constexpr (GROUP >= 16) v += lane_xor<8>(v);
  if constexpr (GROUP >= 8) v += lane_xor<4, GROUP >= 16>(v);
  if constexpr (GROUP >= 4) v += lane_xor<2>(v);
  if constexpr (GROUP >= 2) v += lane_xor<1>(v);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Counter-based uniform in [0,1): one 64-bit mix (splitmix64 finaliser) of (seed, index).
// Forward and backward evaluate it at the same (seed, index) and so agree on the mask.
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t index) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (index + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// The same for a 32-bit index: three 32-bit multiplies instead of three 64-bit ones, for kernels that
// draw many numbers per loaded byte (the temporal attention draws one per score).  uniform_bits24 is the
// draw as an integer in [0, 2^24), uniform01_u32 the same draw scaled to [0, 1).
__device__ __forceinline__ uint32_t uniform_bits24(uint64_t seed, uint32_t index) {
  uint32_t x = index ^ (uint32_t)seed;
  x *= 0x9E3779B1u;
  x ^= x >> 15;
  x += (uint32_t)(seed >> 32);
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x >> 8;
}
__device__ __forceinline__ float uniform01_u32(uint64_t seed, uint32_t index) {
  return (float)uniform_bits24(seed, index) * (1.0f / 16777216.0f);
}

// Draw of the dropout masks of the dense layers' epilogues (and of the kernel that applies the same mask to the
// gradient): the 32-bit mix on the low half of the element index, the high half folded into the seed.
__device__ __forceinline__ float dropout_uniform(uint64_t seed, uint64_t index) {
  return uniform01_u32(seed + (index >> 32) * 0x9E3779B97F4A7C15ull, (uint32_t)index);
}

}  // namespace hivt
