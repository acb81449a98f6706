// One new target position through the WHOLE decoder stack in a single launch (inference).
//
// The decode loops of the reference (validate_generate_MMT_v15_4.py:744-766, :1023-1034,
// models_MMT_v15_4.py:1238-1262) produce one token per decoder pass.  With cached keys/values a pass is one row
// per sequence: ~0.6 M multiply-adds per layer, far too little for a grid of GEMM tiles, and as a chain of ~50
// small launches a token costs ~0.4 ms of launch latency.  Here one workgroup owns one sequence and walks all
// layers: every projection is a matrix-vector product read straight from the bf16 weights (L2-resident: the
// stack is 4 MB and every workgroup reads the same rows at about the same time), the row lives in LDS, both
// attentions are a lane per (key, head) with an online softmax and a merge through LDS.  Optionally the kernel
// also embeds the input token, applies the vocabulary projection and selects the next token (greedy), so that a
// token is ONE launch.
//
// Arithmetic and rounding points are those of mmst_decoder_layer_step (bf16 rows between the stages, fp32
// accumulation, fp32 softmax and LayerNorm statistics); only the summation order differs.
#include "../../include/mmst.h"
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

#ifndef MMST_DEC_SKIP   // measurement builds only: bit 0 self-attention, bit 1 cross-attention, bit 2 feed-forward left out
#define MMST_DEC_SKIP 0
#endif
#ifndef MMST_DEC_STREAM_KV   // measurement builds: 0 = default cache policy for the memory keys/values
#define MMST_DEC_STREAM_KV 1
#endif
#ifndef MMST_DEC_CROSS_U     // measurement builds: keys in flight per lane in the cross-attention of the split form's
#define MMST_DEC_CROSS_U 0   // launches (0 = as everywhere: 2 for three or more queries; 3 measured equal, 4 slower)
#endif
#ifndef MMST_DECODE_SHARED_CROSS
#define MMST_DECODE_SHARED_CROSS 1
#endif
#ifndef MMST_DEC_HEAD_MINOR_MAXR   // measurement builds: the lane layout of the attention passes, see head_minor()
#define MMST_DEC_HEAD_MINOR_MAXR 1
#endif
constexpr int D = MMST_D, FF = MMST_FF, NT = 1024, NWAVE = NT / 64, NH = 16, HD = 8;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

struct StackArgs {
  mmst_dec_params p[MMST_MAX_DEC_LAYERS];
  mmst_dec_stack_step s;
  int n_layers;
  // Split token step (mmst_dec_stack_step.ffn_ws given): a launch runs layers [l_begin, l_end).  MODE_PRO: its input rows
  // are the output of layer l_begin - 1, whose feed-forward block ran as a launch of its own (decode_ffn_kernel): sum of
  // its hidden-slice partials + bias + residual, then norm3.  MODE_SPLIT: the launch stops behind norm2 of layer
  // l_begin and hands the rows to that kernel.
  int l_begin, l_end, mode;
};
enum { MODE_PRO = 1, MODE_SPLIT = 2 };
constexpr int FFN_HS = 16;   // hidden slices of the token-batched feed-forward kernel (128 hidden columns each)
// workspace of the split token step: bf16 rows [B][128] (the feed-forward input), then fp32 partials [FFN_HS][B][128]
__device__ __forceinline__ bf16* ws_rows(void* ws) { return (bf16*)ws; }
__device__ __forceinline__ float* ws_slabs(void* ws, int B) { return (float*)((char*)ws + (((size_t)B * D * 2 + 255) & ~(size_t)255)); }

// R = sequences per workgroup.  R > 1: the workgroup owns the R sequences (beams) of ONE molecule; every weight row
// it streams serves R rows, and the molecule's projected memory keys/values are read once for all of them.
// Every row that feeds a projection is bf16-exact, so it is kept in LDS as bf16: a lane fetches its slice of a row
// with one or two 16-byte LDS reads where it needs it instead of holding all R rows in registers across the
// weight loop (registers are what limits the number of weight loads in flight).
template <int R>
struct Smem {
  float x[R][D];          // input rows of the current layer (fp32 copy for the residual)
  float qkv[R][3 * D];    // Q|K|V of the new position
  float x1[R][D];
  float qc[R][D];
  float x2[R][D];
  __attribute__((aligned(16))) bf16 vx[R][D];      // packed copies of the rows that enter the next projection
  __attribute__((aligned(16))) bf16 vattn[R][D];
  __attribute__((aligned(16))) bf16 vh[R][FF];
  float w2part[2][R][D];           // the two contraction halves of the second feed-forward projection
  float part[R][NWAVE][NH][HD + 2];   // per-query, per-wave softmax partials: m, l, acc[8]
  float red[2][NWAVE];
};

__device__ __forceinline__ float rbf(float v) { return (float)(bf16)v; }

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, not for its global
// loads (__syncthreads() also drains those, which would serialise the loads issued ahead for later stages).
// Nothing written to global memory here is read back inside the kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Two multiply-adds per instruction on packed bf16 pairs (v_dot2c_f32_bf16), fp32 accumulation.  Every vector that
// enters a projection here is already bf16-exact, so packing it loses nothing; converting weights to fp32 and
// multiplying them one by one costs four instructions per pair and makes a one-row projection VALU-bound.
__device__ __forceinline__ bf16x2 pack2(float a, float b) { return bf16x2{(bf16)a, (bf16)b}; }
__device__ __forceinline__ float dot8p(const bf16x8& w, const bf16x2 (&v)[4], float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 0, 1), v[0], acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 2, 3), v[1], acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 4, 5), v[2], acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 6, 7), v[3], acc, false);
  return acc;
}

__device__ __forceinline__ float dot8b(const bf16x8& w, const bf16x8& v, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 0, 1), __builtin_shufflevector(v, v, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 2, 3), __builtin_shufflevector(v, v, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 4, 5), __builtin_shufflevector(v, v, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(w, w, 6, 7), __builtin_shufflevector(v, v, 6, 7), acc, false);
  return acc;
}

// out[r][n] = bias[n] + sum_k W[n][k] v[r][k] for K = 128, n < rows <= N (N a multiple of NT/8 or smaller than it), R
// input rows v + r*vstride (bf16 in LDS): eight lanes per output row (16 contiguous weights each, loaded ONCE for the
// R rows), NT/8 output rows per pass of the workgroup, the loads of U passes in flight before the first one is
// consumed (a dependent L2 round trip per pass would dominate otherwise).  `emit(r, n, value)` is called by one lane
// per output row.
template <int R, int N, int U, typename F>
__device__ __forceinline__ void gemv_k128(const bf16* __restrict__ W, const float* __restrict__ bias, const bf16* v,
                                          int vstride, int tid, F emit, int rows = N) {
  constexpr int RP = NT / 8, PASSES = (N + RP - 1) / RP;
  static_assert(PASSES % U == 0, "U must divide the number of passes");
  const int sub = tid & 7, r0 = tid >> 3;
#pragma unroll 1
  for (int p0 = 0; p0 < PASSES; p0 += U) {
    bf16x8 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = min((p0 + u) * RP + r0, rows - 1);
      a[u] = *reinterpret_cast<const bf16x8*>(W + (size_t)n * D + sub * 16);
      b[u] = *reinterpret_cast<const bf16x8*>(W + (size_t)n * D + sub * 16 + 8);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bf16x8 va = *reinterpret_cast<const bf16x8*>(v + r * vstride + sub * 16);
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(v + r * vstride + sub * 16 + 8);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = (p0 + u) * RP + r0;
        float s = dot8b(a[u], va, 0.f) + dot8b(b[u], vb, 0.f);
        s = row8_sum(s);
        if (sub == 0 && n < rows) emit(r, n, s + (bias ? bias[n] : 0.f));
      }
    }
  }
}

// out[r][n] = bias[n] + sum_k W[n][k] v[r][k] for K = 2048, N = 128: a wave per output row, U rows (4 U loads) in
// flight, every weight row used for the R input rows (bf16 in LDS)
template <int R, int U, typename F>
__device__ __forceinline__ void gemv_k2048(const bf16* __restrict__ W, const float* __restrict__ bias, const bf16* v,
                                           int vstride, int tid, F emit) {
  constexpr int ROWS = D / NWAVE;   // rows per wave
  static_assert(ROWS % U == 0, "U must divide the rows per wave");
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
  for (int i0 = 0; i0 < ROWS; i0 += U) {
    bf16x8 a[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        a[u][c] = *reinterpret_cast<const bf16x8*>(W + (size_t)(wave * ROWS + i0 + u) * FF + c * 512 + lane * 8);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      bf16x8 vin[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) vin[c] = *reinterpret_cast<const bf16x8*>(v + r * vstride + c * 512 + lane * 8);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = wave * ROWS + i0 + u;
        float s = (dot8b(a[u][0], vin[0], 0.f) + dot8b(a[u][1], vin[1], 0.f)) +
                  (dot8b(a[u][2], vin[2], 0.f) + dot8b(a[u][3], vin[3], 0.f));
        s = wave_sum(s);
        if (lane == 0) emit(r, n, s + (bias ? bias[n] : 0.f));
      }
    }
  }
}

// Projections on the matrix cores.  One v_mfma_f32_16x16x32_bf16 multiplies a 16 x 32 tile of a weight matrix (operand
// A: lane l supplies W[n0 + (l & 15)][k0 + 8 (l >> 4) .. + 8], sixteen contiguous bytes of a weight row) by the
// matching 32 x 16 slice of up to 16 input rows (operand B: lane l supplies v[row l & 15][k0 + 8 (l >> 4) .. + 8] from
// LDS, zero for rows >= R); lane l then holds out[n0 + 4 (l >> 4) + i][row l & 15], i = 0..3.  The matrix pipe sums
// over k, so there are no cross-lane reductions, and the cost does not depend on R: a weight row streamed from L2
// serves every sequence of the workgroup.
//
// out[r][n] = sum_k W[n][k] v[r][k] for K = 128 and NTILES * 16 output columns (n < rows): wave w takes the column
// tiles w, w + 16, ...; the weight fragments of U tiles (4 U loads per lane) are in flight before the first MFMA.
// `emit(r, n, value)` is called by the lane that holds the value (bias etc. are the caller's business).
template <int R, int NTILES, int U, typename F>
__device__ __forceinline__ void proj_k128(const bf16* __restrict__ W, const float* __restrict__ bias, const bf16* v,
                                          int vstride, int tid, F emit, int rows = NTILES * 16) {
  const int lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  constexpr int PER = (NTILES + NWAVE - 1) / NWAVE;
  static_assert(PER % U == 0, "U must divide the tiles per wave");
  bf16x8 bx[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bx[ks] = bf16x8{};
    if (j < R) bx[ks] = *reinterpret_cast<const bf16x8*>(v + j * vstride + ks * 32 + q * 8);
  }
#pragma unroll 1
  for (int t0 = 0; t0 < PER; t0 += U) {
    bf16x8 a[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tile = min(wave + (t0 + u) * NWAVE, NTILES - 1);          // a surplus tile re-reads a valid one
      const int n = min(tile * 16 + j, rows - 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[u][ks] = *reinterpret_cast<const bf16x8*>(W + (size_t)n * D + ks * 32 + q * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tile = wave + (t0 + u) * NWAVE;
      if (tile < NTILES) {   // uniform across the wave
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][ks], bx[ks], acc, 0, 0, 0);
        if (j < R) {   // outputs n .. n + 3 (n + i < rows is the caller's check); requesting the bias together with
                       // the weights was measured slower than fetching it here
          const int n = tile * 16 + q * 4;
          if (bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += bias[min(n + i, rows - 1)];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (n + i < rows) emit(j, n + i, acc[i]);
        }
      }
    }
  }
}

// partial[kh][r][n] = sum over half kh of k of W[n][k] v[r][k] for K = 2048, N = 128: wave w takes column tile w & 7
// and contraction half w >> 3 (1024 k = 32 MFMA steps, U of them in flight); the caller adds the two halves.
template <int R, int U>
__device__ __forceinline__ void proj_k2048(const bf16* __restrict__ W, const bf16* v, int tid, float (*partial)[R][D]) {
  const int lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int nt = wave & 7, kh = wave >> 3;
  const bf16* wrow = W + (size_t)(nt * 16 + j) * FF + kh * (FF / 2) + q * 8;
  const bf16* vrow = v + (size_t)(j < R ? j : 0) * FF + kh * (FF / 2) + q * 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int STEPS = FF / 2 / 32;
  static_assert(STEPS % U == 0, "U must divide the steps");
#pragma unroll 1
  for (int k0 = 0; k0 < STEPS; k0 += U) {
    bf16x8 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const bf16x8*>(wrow + (k0 + u) * 32);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bf16x8 b = bf16x8{};
      if (j < R) b = *reinterpret_cast<const bf16x8*>(vrow + (k0 + u) * 32);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b, acc, 0, 0, 0);
    }
  }
  if (j < R) {
#pragma unroll
    for (int i = 0; i < 4; ++i) partial[kh][j][nt * 16 + q * 4 + i] = acc[i];
  }
}

// One row per workgroup: dot products on the vector ALU (measured faster than a 1/16-filled MFMA tile at R = 1);
// several rows: the matrix-core projections above.  N outputs, K = 128; `emit(r, n, value)`, bias included.
template <int R, int N, int U1, typename F>
__device__ __forceinline__ void project_k128(const bf16* W, const float* bias, const bf16* v, int tid, F emit, int rows = N) {
  if constexpr (R == 1) {
    gemv_k128<1, N, U1>(W, bias, v, D, tid, emit, rows);
  } else {
    constexpr int NT16 = N / 16, PER = (NT16 + NWAVE - 1) / NWAVE;
    proj_k128<R, NT16, (PER % 4 == 0 ? 4 : (PER % 2 == 0 ? 2 : 1))>(W, bias, v, D, tid, emit, rows);
  }
}

// R rows z[r][0..128) (thread tid < R*128 holds element tid & 127 of row tid >> 7, i.e. waves 2r and 2r+1 hold row r)
// -> LayerNorm -> out[r][] as bf16-rounded floats
template <int R>
__device__ __forceinline__ void layernorm_rows(Smem<R>& sm, float z, const float* gamma, const float* beta, float (*out)[D],
                                               bf16 (*packed)[D], int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const bool own = tid < R * D;
  const int r = own ? (tid >> 7) : 0, c = tid & (D - 1);
  float s = wave_sum(own ? z : 0.f);
  if (lane == 0) sm.red[0][wave] = s;
  lds_barrier();
  const float mean = (sm.red[0][2 * r] + sm.red[0][2 * r + 1]) * (1.0f / D);
  const float d = own ? z - mean : 0.f;
  float vs = wave_sum(d * d);
  if (lane == 0) sm.red[1][wave] = vs;
  lds_barrier();
  const float rstd = rsqrtf((sm.red[1][2 * r] + sm.red[1][2 * r + 1]) * (1.0f / D) + 1e-5f);
  if (own) {
    const bf16 y = (bf16)(d * rstd * gamma[c] + beta[c]);
    out[r][c] = (float)y;
    packed[r][c] = y;
  }
  lds_barrier();
}

// Lane layout of the attention passes: a lane serves one head (8 of a row's 128 columns, 16 bytes) of one of the four
// keys a wave covers per step.  Head-minor (lane = 16 key slot + head): the 16 lanes of a key slot read that key's
// 256-byte row, so every four neighbouring lanes fetch 64 contiguous bytes; head-major (lane = 4 head + key slot): the
// four lanes of a quad touch four rows, and the address unit takes a cycle per row a quad touches, but the merge of the
// four key slots is free (quad permutes).  One sequence per workgroup is bound by the pass over the memory keys/values
// and takes the first; measured with 3 sequences per workgroup (beam search) the second is ahead.
template <int R> __device__ __forceinline__ constexpr bool head_minor() { return R <= MMST_DEC_HEAD_MINOR_MAXR; }
template <int R> __device__ __forceinline__ int head_of(int lane) { return head_minor<R>() ? (lane & 15) : (lane >> 2); }
template <int R> __device__ __forceinline__ int key_slot_of(int lane) { return head_minor<R>() ? (lane >> 4) : (lane & 3); }

// merge of the per-lane online-softmax states (m, l, acc[8]) of Q query rows at once: the 4 key slots of a wave (the
// lanes of a quad), then the waves through LDS (two barriers for all the rows); results -> out + r*D (bf16-rounded)
template <int R, int Q>
__device__ __forceinline__ void merge_softmax_all(Smem<R>& sm, float (&m)[Q], float (&l)[Q], float (&acc)[Q][HD], bf16* out,
                                                  int tid) {
  static_assert(Q <= R, "one partial array per sequence of the workgroup");
  // the four key slots of a head are merged without a trip through the LDS crossbar: 16 and 32 lanes apart, by the two
  // lane-swap instructions; or the lanes of a quad, by quad permutes inside the add / max instructions
  const int lane = tid & 63, wave = tid >> 6, hh = head_of<R>(lane);
#pragma unroll
  for (int r = 0; r < Q; ++r) {
    float M, lr;
    if constexpr (head_minor<R>()) {
      M = xor32_max(xor16_max(m[r]));
      const float f = __builtin_amdgcn_exp2f(m[r] - M);
      lr = xor32_sum(xor16_sum(l[r] * f));
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[r][d] = xor32_sum(xor16_sum(acc[r][d] * f));
    } else {
      M = fmaxf(m[r], dpp_f32<0xB1>(m[r]));
      M = fmaxf(M, dpp_f32<0x4E>(M));
      const float f = __builtin_amdgcn_exp2f(m[r] - M);
      lr = l[r] * f;
      lr += dpp_f32<0xB1>(lr);
      lr += dpp_f32<0x4E>(lr);
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        acc[r][d] *= f;
        acc[r][d] += dpp_f32<0xB1>(acc[r][d]);
        acc[r][d] += dpp_f32<0x4E>(acc[r][d]);
      }
    }
    if (key_slot_of<R>(lane) == 0) {
      sm.part[r][wave][hh][0] = M;
      sm.part[r][wave][hh][1] = lr;
#pragma unroll
      for (int d = 0; d < HD; ++d) sm.part[r][wave][hh][2 + d] = acc[r][d];
    }
  }
  lds_barrier();
  if (tid < Q * D) {
    const int r = tid >> 7, c = tid & (D - 1), h2 = c >> 3, d = c & 7;
    float MM = NEG_BIG;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) MM = fmaxf(MM, sm.part[r][w][h2][0]);
    float L = 0.f, A = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) {
      const float g = __builtin_amdgcn_exp2f(sm.part[r][w][h2][0] - MM);
      L += sm.part[r][w][h2][1] * g;
      A += sm.part[r][w][h2][2 + d] * g;
    }
    out[r * D + c] = (bf16)(A * (L > 0.f ? 1.0f / L : 0.f));
  }
  lds_barrier();
}

// Attention of Q query rows (q + r*qstride, 16 heads x 8) over the SAME `nkeys` keys: K row of key j at kbase + j*ld, V
// at vbase + j*ld (bf16, 128 wide) -- unless `slot_of` is given: then key j lives in the cache of sequence slot_of[j] of
// the molecule, kbase + (slot_of[j]*cap + j)*ld (a beam's history points into the caches of the beams it descends
// from).  A wave covers 4 keys per step, the workgroup 64 (lane layout: head_minor()).  If last_k != nullptr (Q == 1 only), key nkeys-1 is read from LDS instead (the row this workgroup
// has just produced).  Results -> out + r*D (bf16-rounded).
// STREAM: the keys/values are read once per launch by this workgroup only (the projected encoder memory): they are
// loaded with the non-temporal hint, so that they do not push the decoder weights, which every workgroup of the XCD
// reads, out of its L2.
template <int R, int Q, bool STREAM = false, int UU = 0>
__device__ __forceinline__ void attend_scores(const float* q, int qstride, const bf16* kbase, const bf16* vbase, size_t ld,
                                              int nkeys, const float* kbias, const float* last_k, const float* last_v,
                                              const int32_t* slot_of, int cap, int tid, float (&m)[Q], float (&l)[Q],
                                              float (&acc)[Q][HD]) {
  const int lane = tid & 63, wave = tid >> 6;
  const int hh = head_of<R>(lane), slot = wave * 4 + key_slot_of<R>(lane);
  const float scale2 = rsqrtf((float)HD) * LOG2E;
  bf16x2 qv[Q][4];
#pragma unroll
  for (int r = 0; r < Q; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) qv[r][j] = pack2(q[r * qstride + hh * HD + 2 * j], q[r * qstride + hh * HD + 2 * j + 1]);
#pragma unroll
  for (int r = 0; r < Q; ++r) {
    m[r] = NEG_BIG;
    l[r] = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[r][d] = 0.f;
  }
  constexpr int U = UU > 0 ? UU : (Q > 2 ? 2 : 4), STEP = 4 * NWAVE;   // keys in flight per lane
#pragma unroll 1
  for (int k0 = slot; k0 < nkeys; k0 += U * STEP) {
    bf16x8 kt[U], vt[U];
    float kb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = min(k0 + u * STEP, nkeys - 1);
      const size_t row = slot_of ? (size_t)slot_of[key] * cap + key : (size_t)key;
      if (STREAM) {
        kt[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kbase + row * ld + hh * HD));
        vt[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(vbase + row * ld + hh * HD));
      } else {
        kt[u] = *reinterpret_cast<const bf16x8*>(kbase + row * ld + hh * HD);
        vt[u] = *reinterpret_cast<const bf16x8*>(vbase + row * ld + hh * HD);
      }
      kb[u] = kbias ? kbias[key] * LOG2E : 0.f;
    }
    if (Q == 1 && last_k) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (k0 + u * STEP == nkeys - 1) {   // the row this workgroup has just produced: not yet visible in memory
#pragma unroll
          for (int d = 0; d < HD; ++d) { kt[u][d] = (bf16)last_k[hh * HD + d]; vt[u][d] = (bf16)last_v[hh * HD + d]; }
        }
    }
#pragma unroll
    for (int r = 0; r < Q; ++r) {
      // scores of the batch, one rescale of the running sums per batch
      float sc[U];
      float mb = m[r];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        sc[u] = (k0 + u * STEP) < nkeys ? dot8p(kt[u], qv[r], 0.f) * scale2 + kb[u] : NEG_BIG;
        mb = fmaxf(mb, sc[u]);
      }
      const float alpha = __builtin_amdgcn_exp2f(m[r] - mb);
      m[r] = mb;
      l[r] *= alpha;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[r][d] *= alpha;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float p = (k0 + u * STEP) < nkeys ? __builtin_amdgcn_exp2f(sc[u] - mb) : 0.f;
        l[r] += p;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[r][d] += p * (float)vt[u][d];
      }
    }
  }
}
template <int R, int Q, bool STREAM = false, int UU = 0>
__device__ __forceinline__ void attend(Smem<R>& sm, const float* q, int qstride, const bf16* kbase, const bf16* vbase, size_t ld,
                                       int nkeys, const float* kbias, const float* last_k, const float* last_v,
                                       const int32_t* slot_of, int cap, bf16* out, int tid) {
  float m[Q], l[Q], acc[Q][HD];
  attend_scores<R, Q, STREAM, UU>(q, qstride, kbase, vbase, ld, nkeys, kbias, last_k, last_v, slot_of, cap, tid, m, l, acc);
  merge_softmax_all<R, Q>(sm, m, l, acc, out, tid);
}

// Timing builds (-DMMST_DEC_STAMP): the first lane of workgroup 0 of the launch of layer 2 leaves shader-clock stamps of
// its phases behind the used part of the split step's workspace (tools/bench_gen.py allocates the room)
#ifdef MMST_DEC_STAMP
#define DSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && args.l_begin == 2) \
    ((unsigned long long*)((char*)args.s.ffn_ws + MMST_DEC_STACK_WS_BYTES(args.s.B)))[i] = __builtin_readcyclecounter(); } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif

// SPLIT: an instance for the launches of the split token step that stop behind norm2 (no feed-forward code in it)
template <int R, bool SPLIT = false>
__global__ __launch_bounds__(NT) void decode_stack_kernel(const StackArgs args) {
  __shared__ Smem<R> sm;
  const mmst_dec_stack_step& s = args.s;
  const int tid = threadIdx.x, g = blockIdx.x;
  const int b0 = g * R;               // first sequence of this workgroup
  const int pos = s.pos;
  DSTAMP(0);
  if (args.mode & MODE_PRO) {
    // input rows = norm3(x2 + b2 + sum of the feed-forward partials) of the layer in front
    const mmst_dec_params& pp = args.p[args.l_begin - 1];
    float z = 0.f;
    if (tid < R * D) {
      const int r = tid >> 7, c = tid & (D - 1);
      const size_t row = (size_t)(b0 + r);
      const float* slab = ws_slabs(s.ffn_ws, s.B) + row * D + c;
      float part[FFN_HS];
#pragma unroll
      for (int hs = 0; hs < FFN_HS; ++hs) part[hs] = slab[(size_t)hs * s.B * D];
      z = pp.b_2[c] + (float)ws_rows(s.ffn_ws)[row * D + c];
#pragma unroll
      for (int hs = 0; hs < FFN_HS; ++hs) z += part[hs];
    }
    layernorm_rows<R>(sm, z, pp.g3, pp.be3, sm.x, sm.vx, tid);
  } else if (tid < R * D) {
    // input rows: given, or embedding of the token + positional row
    const int r = tid >> 7, c = tid & (D - 1);
    bf16 v;
    if (s.tok) {
      const int64_t tk = s.tok[b0 + r];
      v = (bf16)(s.embed[(size_t)tk * D + c] + s.pos_row[c]);
    } else {
      v = ((const bf16*)s.x)[(size_t)(b0 + r) * D + c];
    }
    sm.x[r][c] = (float)v;
    sm.vx[r][c] = v;
  }
  lds_barrier();
  DSTAMP(1);
  const int bm = R > 1 ? g : b0 / s.kv_group;   // row block of the encoder memory these sequences read
  const float* mbias = s.mem_bias ? s.mem_bias + (size_t)bm * s.L : nullptr;
  for (int l = args.l_begin; l < args.l_end; ++l) {
    const mmst_dec_params& p = args.p[l];
    bf16* cache = (bf16*)s.qkv_cache[l] + (size_t)b0 * s.cap * 3 * D;   // caches of this workgroup's sequences
    // 1. Q|K|V of the new position, appended to each sequence's own cache
    project_k128<R, 3 * D, 3>((const bf16*)p.w_qkv, p.b_qkv, &sm.vx[0][0], tid, [&](int r, int n, float v) {
      const bf16 q = (bf16)v;
      sm.qkv[r][n] = (float)q;
      cache[((size_t)r * s.cap + pos) * 3 * D + n] = q;
    });
    lds_barrier();
    DSTAMP(2);
    // 2. self-attention over positions 0..pos (every cached key is visible).  With a history table the key of position
    //    j is read from the cache of the sequence that produced it (slot hist[b][j] of this molecule); position `pos`
    //    itself is the row just computed.
    if (!(MMST_DEC_SKIP & 1)) {   // the R sequences one after the other (their keys differ), their merges at once
      float am[R], al[R], aacc[R][HD];
      if (R > 1 && pos + 1 <= 4 * NWAVE) {
        // at most one key per lane: the key/value rows of ALL R sequences are requested before the first one is used
        // (one round trip to the caches instead of R), the arithmetic is that of the general loop for a single key
        const int lane = tid & 63, hh = head_of<R>(lane), key = (tid >> 6) * 4 + key_slot_of<R>(lane);
        const bool valid = key <= pos;
        const float scale2 = rsqrtf((float)HD) * LOG2E;
        bf16x8 kt[R], vt[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int32_t* hist = s.hist ? s.hist + (size_t)(b0 + r) * s.cap : nullptr;
          const bf16* base = hist ? cache : cache + (size_t)r * s.cap * 3 * D;
          const int kk = valid ? key : pos;
          const size_t row = hist ? (size_t)hist[kk] * s.cap + kk : (size_t)kk;
          kt[r] = *reinterpret_cast<const bf16x8*>(base + D + row * 3 * D + hh * HD);
          vt[r] = *reinterpret_cast<const bf16x8*>(base + 2 * D + row * 3 * D + hh * HD);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (key == pos) {   // the row this workgroup has just produced: not yet visible in memory
#pragma unroll
            for (int d = 0; d < HD; ++d) { kt[r][d] = (bf16)sm.qkv[r][D + hh * HD + d]; vt[r][d] = (bf16)sm.qkv[r][2 * D + hh * HD + d]; }
          }
          bf16x2 qv[4];
#pragma unroll
          for (int jq = 0; jq < 4; ++jq) qv[jq] = pack2(sm.qkv[r][hh * HD + 2 * jq], sm.qkv[r][hh * HD + 2 * jq + 1]);
          am[r] = valid ? dot8p(kt[r], qv, 0.f) * scale2 : NEG_BIG;
          al[r] = valid ? 1.f : 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) aacc[r][d] = valid ? (float)vt[r][d] : 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int32_t* hist = s.hist ? s.hist + (size_t)(b0 + r) * s.cap : nullptr;
          const bf16* base = hist ? cache : cache + (size_t)r * s.cap * 3 * D;
          float m1[1], l1[1], a1[1][HD];
          attend_scores<R, 1, (MMST_DEC_STREAM_KV & 2) != 0>(sm.qkv[r], 0, base + D, base + 2 * D, 3 * D, pos + 1, nullptr,
                                                             sm.qkv[r] + D, sm.qkv[r] + 2 * D, hist, s.cap, tid, m1, l1, a1);
          am[r] = m1[0];
          al[r] = l1[0];
#pragma unroll
          for (int d = 0; d < HD; ++d) aacc[r][d] = a1[0][d];
        }
      }
      DSTAMP(3);
      merge_softmax_all<R, R>(sm, am, al, aacc, &sm.vattn[0][0], tid);
    }
    DSTAMP(4);
    // 3. out_proj + residual + norm1
    project_k128<R, D, 1>((const bf16*)p.w_o, p.b_o, &sm.vattn[0][0], tid,
                          [&](int r, int n, float v) { sm.qc[r][n] = v + sm.x[r][n]; });
    lds_barrier();
    float z = tid < R * D ? sm.qc[tid >> 7][tid & (D - 1)] : 0.f;
    DSTAMP(5);
    layernorm_rows<R>(sm, z, p.g1, p.be1, sm.x1, sm.vx, tid);
    DSTAMP(6);
    // 4. cross-attention queries
    project_k128<R, D, 1>((const bf16*)p.w_cqkv, p.b_cqkv, &sm.vx[0][0], tid,
                          [&](int r, int n, float v) { sm.qc[r][n] = rbf(v); });
    lds_barrier();
    DSTAMP(7);
    // 5. cross-attention over the projected memory: one pass over the molecule's keys/values for all R queries
    const bf16* kv = (const bf16*)s.kvc[l] + (size_t)bm * s.L * 2 * D;
#if MMST_DEC_SKIP & 2
#elif MMST_DECODE_SHARED_CROSS
    attend<R, R, (MMST_DEC_STREAM_KV & 1) != 0, (SPLIT ? MMST_DEC_CROSS_U : 0)>(sm, &sm.qc[0][0], D, kv, kv + D, 2 * D, s.L, mbias,
                                                                          nullptr, nullptr, nullptr, 0, &sm.vattn[0][0], tid);
#else
#pragma unroll
    for (int r = 0; r < R; ++r)
      attend<R, 1>(sm, sm.qc[r], 0, kv, kv + D, 2 * D, s.L, mbias, nullptr, nullptr, nullptr, 0, sm.vattn[r], tid);
#endif
    DSTAMP(8);
    // 6. out_proj + residual + norm2
    project_k128<R, D, 1>((const bf16*)p.w_co, p.b_co, &sm.vattn[0][0], tid,
                          [&](int r, int n, float v) { sm.qc[r][n] = v + sm.x1[r][n]; });
    lds_barrier();
    z = tid < R * D ? sm.qc[tid >> 7][tid & (D - 1)] : 0.f;
    layernorm_rows<R>(sm, z, p.g2, p.be2, sm.x2, sm.vx, tid);
    DSTAMP(9);
    if constexpr (SPLIT) {   // the feed-forward block of this layer is the next launch
      if (tid < R * D) ws_rows(s.ffn_ws)[(size_t)b0 * D + tid] = sm.vx[tid >> 7][tid & (D - 1)];
      return;
    }
    // 7. feed-forward block + norm3
    if constexpr (!SPLIT) {
    if (!(MMST_DEC_SKIP & 4)) {
      project_k128<R, FF, 8>((const bf16*)p.w_1, p.b_1, &sm.vx[0][0], tid,
                             [&](int r, int n, float v) { sm.vh[r][n] = (bf16)fmaxf(v, 0.f); });
      lds_barrier();
      if constexpr (R == 1) {
        gemv_k2048<1, 4>((const bf16*)p.w_2, nullptr, &sm.vh[0][0], FF, tid,
                         [&](int r, int n, float v) { sm.w2part[0][r][n] = v; sm.w2part[1][r][n] = 0.f; });
      } else {
        proj_k2048<R, 16>((const bf16*)p.w_2, &sm.vh[0][0], tid, sm.w2part);
      }
      lds_barrier();
    }
    z = 0.f;
    if (tid < R * D) {
      const int r = tid >> 7, c = tid & (D - 1);
      z = sm.w2part[0][r][c] + sm.w2part[1][r][c] + p.b_2[c] + sm.x2[r][c];
    }
    layernorm_rows<R>(sm, z, p.g3, p.be3, sm.x, sm.vx, tid);
    }
  }
  if constexpr (SPLIT) return;
  if (s.y && tid < R * D) ((bf16*)s.y)[(size_t)b0 * D + tid] = sm.vx[tid >> 7][tid & (D - 1)];
  if (!s.logits) return;
  // vocabulary projection (weight rows beyond the vocabulary are zero padding of the caller's matrix)
  project_k128<R, 64, 1>((const bf16*)s.w_out, s.b_out, &sm.vx[0][0], tid, [&](int r, int n, float v) {
    sm.qc[r][n] = v;
    s.logits[(size_t)(b0 + r) * s.ld_logits + n] = v;
  }, s.n_out);
  if (!s.next_tok) return;
  lds_barrier();
  if (tid < 64 * R) {   // wave r: softmax(logits / temperature) of row r; first index of the maximum, or a draw
    const int lane = tid & 63, r = tid >> 6;
    const float x = lane < s.V ? sm.qc[r][lane] / s.temperature : -INFINITY;
    float pr;
    const int arg = wave_select_token(x, s.V, s.sample != 0, s.sample_seed, (uint32_t)pos, (uint32_t)(b0 + r), &pr);
    if (lane == 0) {
      s.next_tok[b0 + r] = arg;
      if (s.next_prob) s.next_prob[b0 + r] = pr;
    }
  }
}

// Feed-forward block of ONE decoder layer for all sequences of a token step (without bias2, residual and norm3, which
// the next launch applies): partial[hs][row][n] = sum over hidden slice hs of relu(x W1^T + b1)[row][j] W2[n][j].
// Grid = (64-row tiles, FFN_HS hidden slices of 128 columns); a wave owns 16 rows.  The slice's weights (128 rows of
// linear1, 128 columns of linear2: 64 KiB of the 1 MiB a layer's block weighs; the per-sequence kernel streams the whole
// MiB through every workgroup) come in by LDS-DMA, all 64 requests of a lane in flight at once, as [128][16 chunks of
// 16 B] images whose chunk index is XORed with the row (conflict-free fragment reads; the swizzle is applied to the
// source address, the DMA destination being lane-linear).  Both GEMMs are v_mfma_f32_16x16x32_bf16 with the WEIGHTS
// as the A operand and the rows as the B operand.  The first GEMM leaves lane (row, q) with hidden columns
// 16 t + 4 q + i of tile t; two tiles side by side are the 8 contraction entries of the second GEMM's B operand, and
// its weight fragments are read in that order (two 8-byte pieces per lane).
typedef const __attribute__((address_space(1))) void* dec_gptr_t;
typedef __attribute__((address_space(3))) void* dec_lptr_t;
__global__ __launch_bounds__(256) void decode_ffn_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W1,
                                                         const float* __restrict__ b1, const bf16* __restrict__ W2,
                                                         float* __restrict__ slabs, int rows, int Bcap) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 32768];
  unsigned char* w1img = smem;            // [128 hidden rows][256 B]: row r, logical chunk c at position c ^ (r & 15)
  unsigned char* w2img = smem + 32768;    // [128 output rows][256 B of this slice's hidden columns], same swizzle
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int hs = blockIdx.y, m0 = ((int)blockIdx.x * 4 + wave) * 16;
  // 32 DMA instructions per image over the 4 waves: instruction i of a wave fills rows 4 (8 wave + i) .. + 3
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 4 * (8 * wave + i) + (lane >> 4), pos = lane & 15, c = pos ^ (r & 15);
    __builtin_amdgcn_global_load_lds((dec_gptr_t)(W1 + (size_t)(hs * 128 + r) * D + c * 8),
                                     (dec_lptr_t)(w1img + (8 * wave + i) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((dec_gptr_t)(W2 + (size_t)r * FF + hs * 128 + c * 8),
                                     (dec_lptr_t)(w2img + (8 * wave + i) * 1024), 16, 0, 0);
  }
  const int m = m0 + j;
  bf16x8 xf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    xf[ks] = bf16x8{};
    if (m < rows) xf[ks] = *reinterpret_cast<const bf16x8*>(X + (size_t)m * D + ks * 32 + q * 8);
  }
  f32x4 bb[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) bb[t] = *reinterpret_cast<const f32x4*>(b1 + hs * 128 + t * 16 + 4 * q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (m0 >= rows) return;
  bf16x8 hf[4];   // the hidden slice as B fragments: hf[p] = columns 32 p + 4 q + {0..3} and 32 p + 16 + 4 q + {0..3}
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    f32x4 a[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * pp + u, r = 16 * t + j;
      a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w1img + r * 256 + (((ks * 4 + q) ^ (r & 15)) << 4));
        a[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[ks], a[u], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) a[u][i] = fmaxf(a[u][i] + bb[t][i], 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { hf[pp][i] = (bf16)a[0][i]; hf[pp][4 + i] = (bf16)a[1][i]; }
  }
  float* out = slabs + ((size_t)hs * Bcap + m) * D + 4 * q;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int r = 16 * t + j;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      // hidden columns 32 pp + 4 q + {0..3} and 32 pp + 16 + 4 q + {0..3} of this slice: halves of chunks 4 pp + (q >> 1)
      // and 4 pp + 2 + (q >> 1)
      const int c0 = 4 * pp + (q >> 1), c1 = c0 + 2, hb = (q & 1) * 8;
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(w2img + r * 256 + ((c0 ^ (r & 15)) << 4) + hb);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(w2img + r * 256 + ((c1 ^ (r & 15)) << 4) + hb);
      const bf16x8 wf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hf[pp], acc, 0, 0, 0);
    }
    if (m < rows) *reinterpret_cast<f32x4*>(out + t * 16) = acc;   // columns 16 t + 4 q + i of row m
  }
}

template <int R>
void launch_stack(const StackArgs& a, int B, hipStream_t stream) {
  if (a.mode & MODE_SPLIT) hipLaunchKernelGGL((decode_stack_kernel<R, true>), dim3(B / R), dim3(NT), 0, stream, a);
  else hipLaunchKernelGGL((decode_stack_kernel<R, false>), dim3(B / R), dim3(NT), 0, stream, a);
}

}  // namespace

namespace {
int g_split_min = -1;
int decode_split_min() {
  if (g_split_min < 0) {
    const char* e = getenv("MMST_DECODE_SPLIT_FFN");
    g_split_min = e ? atoi(e) : 1;
    if (g_split_min < 0) g_split_min = 1;
  }
  return g_split_min;
}
}  // namespace

extern "C" int mmst_sizeof_dec_stack_step() { return (int)sizeof(mmst_dec_stack_step); }
// least number of sequences from which mmst_decoder_stack_step takes the split form (0 = never, negative = default)
extern "C" int mmst_decoder_stack_config(int split_min_sequences) {
  g_split_min = split_min_sequences;
  return MMST_OK;
}

extern "C" int mmst_decoder_stack_step(int n_layers, const mmst_dec_params* p, const mmst_dec_stack_step* s, int nhead,
                                       void* stream) {
  if (!p || !s || nhead != NH || n_layers < 1 || n_layers > MMST_MAX_DEC_LAYERS) return MMST_ERR_ARG;
  if (s->B < 1 || s->pos < 0 || s->pos >= s->cap || s->L < 1 || s->kv_group < 0) return MMST_ERR_ARG;
  if (s->kv_group > 1 && s->B % s->kv_group != 0) return MMST_ERR_ARG;
  if (s->hist && (s->kv_group < 1 || s->kv_group > 4)) return MMST_ERR_ARG;   // history tables need the grouped kernel
  if (!s->tok && !s->x) return MMST_ERR_ARG;
  if (s->tok && (!s->embed || !s->pos_row)) return MMST_ERR_ARG;
  if (!s->y && !s->logits) return MMST_ERR_ARG;
  if (s->logits && (!s->w_out || s->n_out < 1 || s->n_out > 64 || s->ld_logits < s->n_out)) return MMST_ERR_ARG;
  if (s->next_tok && (!s->logits || s->V < 1 || s->V > 64 || s->V > s->n_out || !(s->temperature > 0.f))) return MMST_ERR_ARG;
  StackArgs a;
  memset(&a, 0, sizeof(a));
  for (int l = 0; l < n_layers; ++l) {
    if (!s->qkv_cache[l] || !s->kvc[l]) return MMST_ERR_ARG;
    a.p[l] = p[l];
  }
  a.s = *s;
  if (a.s.kv_group < 1) a.s.kv_group = 1;
  a.n_layers = n_layers;
  a.l_begin = 0; a.l_end = n_layers; a.mode = 0;
  // the beams of a molecule (kv_group of 2..4 sequences) share one workgroup; otherwise one workgroup per sequence
  const int R = (a.s.kv_group >= 2 && a.s.kv_group <= 4 && a.s.rows_per_group != 1) ? a.s.kv_group : 1;
  if (a.s.hist && R != a.s.kv_group) return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  auto launch = [&](const StackArgs& x) {
    switch (R) {
      case 1: launch_stack<1>(x, s->B, st); break;
      case 2: launch_stack<2>(x, s->B, st); break;
      case 3: launch_stack<3>(x, s->B, st); break;
      default: launch_stack<4>(x, s->B, st); break;
    }
  };
  // Split token step: per layer one launch of the per-sequence kernel up to norm2 and one of the token-batched
  // feed-forward kernel, which reads a layer's feed-forward weights once per token step instead of once per workgroup.
  // Taken when the caller provides the workspace (mmst_decoder_stack_config, or MMST_DECODE_SPLIT_FFN: the least number
  // of sequences, default 1; 0 = never).  The one-launch form makes every workgroup stream all feed-forward weights
  // (1 MiB per layer) at the rate one compute unit fetches from L2: measured on an MI355X, 6 layers, 127 positions,
  // 8 / 64 / 128 sequences: 222 / 209 / 207 us per position against 154 / 168 / 174 us split.
  const int split_min = decode_split_min();
  if (s->ffn_ws && split_min > 0 && s->B >= split_min) {
    for (int l = 0; l < n_layers; ++l) {
      a.l_begin = l; a.l_end = l + 1; a.mode = MODE_SPLIT | (l > 0 ? MODE_PRO : 0);
      launch(a);
      hipLaunchKernelGGL(decode_ffn_kernel, dim3((s->B + 63) / 64, FFN_HS), dim3(256), 0, st,
                         (const bf16*)s->ffn_ws, (const bf16*)p[l].w_1, p[l].b_1, (const bf16*)p[l].w_2,
                         (float*)((char*)s->ffn_ws + (((size_t)s->B * D * 2 + 255) & ~(size_t)255)), s->B, s->B);
    }
    a.l_begin = a.l_end = n_layers; a.mode = MODE_PRO;
    launch(a);
  } else {
    launch(a);
  }
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
