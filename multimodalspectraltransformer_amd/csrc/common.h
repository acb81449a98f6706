// Shared device helpers for the gfx950 kernels of the spectral-transformer path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDS_AS __attribute__((address_space(3)))

#define MMST_OK 0
#define MMST_ERR_ARG (-1)
#define MMST_ERR_LAUNCH (-2)

// A completion event for the NEXT kernel launched through MMST_LAUNCH on this thread.  It is bound to that kernel's own
// dispatch (hipExtLaunchKernelGGL), so another stream can wait for the kernel without a marker packet on the
// launching stream: a separate hipEventRecord there holds the following kernel up for ~6 us.
extern thread_local hipEvent_t mmst_armed_stop_event;   // defined in gemm.hip
#define MMST_LAUNCH(kernel, grid, block, shmem, stream, ...)                                            \
  do {                                                                                                   \
    hipEvent_t ev__ = mmst_armed_stop_event;                                                             \
    mmst_armed_stop_event = nullptr;                                                                     \
    if (ev__) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, ev__, 0, __VA_ARGS__);  \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                           \
  } while (0)

#define MMST_CHECK_LAUNCH()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return MMST_ERR_LAUNCH;  \
  } while (0)

// Host-side helpers shared between the entry-point files.
// `to` (if not null) waits for everything enqueued on `from` so far; events come from a per-device ring (layers.hip).
int mmst_stream_fork(hipStream_t from, hipStream_t to);
// target-embedding launchers with explicit token strides (misc.hip); `clear`: optional fp32 slot set to zero
int launch_trg_embed_fwd(const int64_t* tok, int tok_ld, int tok_bs, const float* embed, const float* pos, void* out,
                         int B, int T, float drop_p, uint32_t seed, hipStream_t stream, float* clear);
// One stream of the library's own per device (created on first use, never destroyed): work that is neither on the
// caller's critical path nor ordered with its second stream (layers.hip)
hipStream_t mmst_side_stream();
int launch_trg_embed_bwd(const int64_t* tok, int tok_ld, int tok_bs, const void* dout, float* d_embed, float* d_pos,
                         int B, int T, int V, float drop_p, uint32_t seed, hipStream_t stream);

// Counter-based dropout: one 32-bit mix per PAIR of elements, 16 bits each.
// keep(idx) == true with probability 1 - thr/65536.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_bits(uint32_t seed, uint32_t pair_idx) {
  return mix32(pair_idx * 0x9E3779B9U + seed);
}
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint32_t idx, uint32_t thr) {
  uint32_t h = drop_bits(seed, idx >> 1);
  uint32_t b = (idx & 1) ? (h >> 16) : (h & 0xFFFFu);
  return b >= thr;
}
// Row-seeded dropout (attention probabilities; hidden activation of the fused feed-forward block): the mask
// bit of (row, column) is a function of a per-row seed (one full mix per row) and the column PAIR (one cheap
// mix per pair, 16 bits per column), and can be regenerated from any register layout.  The pair mix is built
// from full-rate instructions only: a 24-bit multiply (v_mul_u32_u24; the 32-bit v_mul_lo_u32 issues at quarter
// rate) between two xor-folds by 16 bits (one sub-dword-select instruction each); the 16-bit halves of the result
// are compared with sub-dword selects as well.
#define ATTN_PAIR_MUL 0x85EBCA6BU
#define ATTN_PAIR_K24 0x9E3779U
__device__ __forceinline__ uint32_t attn_row_seed(uint32_t seed, uint32_t rowid) {
  return mix32(seed + rowid * 0x9E3779B9U);
}
__device__ __forceinline__ uint32_t attn_pair_bits(uint32_t row_seed, uint32_t pair_term) {
  uint32_t t = row_seed + pair_term;   // pair_term = (key >> 1) * ATTN_PAIR_MUL
  t ^= t >> 16;
  t = __umul24(t, ATTN_PAIR_K24);      // low 24 bits of t times a 24-bit constant, low 32 bits of the product
  t ^= t >> 16;
  return t;
}

__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
  return (uint32_t)(p * 65536.0f + 0.5f);
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// Sum over the 16 lanes of a row (the lanes with equal lane >> 4), left in every one of them.  Data-parallel-primitive
// operands (neighbour exchange inside the add instruction) instead of __shfl_xor, which goes through the LDS crossbar
// (ds_bpermute_b32) with an address register and a wait per step: quads by two permutes, then the mirrored half row
// (lane i with 7 - i: the other quad's sum) and the mirrored row (i with 15 - i: the other half's sum).
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_f32<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_f32<0x141>(v);   // row_half_mirror
  v += dpp_f32<0x140>(v);   // row_mirror
  return v;
}

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return v;
}
// over 8 consecutive lanes (lanes with equal lane >> 3)
__device__ __forceinline__ float row8_sum(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  return v;
}
// Exchanges between the rows of 16 lanes, in the vector ALU as well (the lane-swap instructions of gfx950): both
// operands start as v; v_permlane32_swap exchanges the upper half of the first with the lower half of the second, which
// leaves (lower, lower) and (upper, upper); v_permlane16_swap does the same with the odd rows of the first and the even
// rows of the second.  (Two wait states between a vector write of an operand and the swap.)
__device__ __forceinline__ void lanes_swap32(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lanes_swap16(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// v + (v of lane ^ 32), v + (v of lane ^ 16), and the maxima
__device__ __forceinline__ float xor32_sum(float v) { float a = v, b = v; lanes_swap32(a, b); return a + b; }
__device__ __forceinline__ float xor16_sum(float v) { float a = v, b = v; lanes_swap16(a, b); return a + b; }
__device__ __forceinline__ float xor32_max(float v) { float a = v, b = v; lanes_swap32(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor16_max(float v) { float a = v, b = v; lanes_swap16(a, b); return fmaxf(a, b); }
// whole wave, no trip through the LDS crossbar
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }

// Token choice of one decode step by one wave: lane i < V holds x = logit_i / temperature, the other lanes -infinity.
// Greedy (sample == false): the first index of the maximum (torch.argmax's tie rule).  Sampling: the inverse CDF of
// softmax(x) at a uniform number that is a pure function of (seed, counter, row) -- 24 bits of two rounds of the
// counter hash -- so a draw can be repeated, and needs no generator state on the device: the first index whose
// cumulative probability exceeds u, never an index of probability zero.  Returns the index on every lane; *prob =
// its softmax probability.
__device__ __forceinline__ int wave_select_token(float x, int V, bool sample, uint32_t seed, uint32_t counter,
                                                 uint32_t row, float* prob) {
  const int lane = threadIdx.x & 63;
  const float mx = wave_max(x);
  const float e = lane < V ? __expf(x - mx) : 0.f;
  const float se = wave_sum(e);
  int arg;
  if (!sample) {
    arg = __ffsll((long long)__ballot(x == mx)) - 1;
  } else {
    float cum = e;   // inclusive prefix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float t = __shfl_up(cum, d);
      if (lane >= d) cum += t;
    }
    const uint32_t h = mix32(mix32(seed + counter * 0x9E3779B9U) + row * 0x85EBCA6BU);
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f) * se;   // [0, se)
    const unsigned long long hit = __ballot(cum > u && e > 0.f);
    const unsigned long long any = __ballot(e > 0.f);
    arg = hit ? __ffsll((long long)hit) - 1 : 63 - __clzll((long long)any);   // rounding at the top end: the last index
  }
  *prob = __shfl(e, arg) / se;
  return arg;
}
