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
#include <string.h>

namespace {

constexpr int D = MMST_D, FF = MMST_FF, NT = 1024, NWAVE = NT / 64, NH = 16, HD = 8;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

struct StackArgs {
  mmst_dec_params p[MMST_MAX_DEC_LAYERS];
  mmst_dec_stack_step s;
  int n_layers;
};

struct Smem {
  float x[D];          // input row of the current layer
  float qkv[3 * D];    // Q|K|V of the new position
  float attn[D];
  float x1[D];
  float qc[D];
  float x2[D];
  float h[FF];
  float part[NWAVE][NH][HD + 2];   // per-wave softmax partials: m, l, acc[8]
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

// out[n] = bias[n] + sum_k W[n][k] v[k] for K = 128, n < rows <= N (N a multiple of NT/8 or smaller than it): eight lanes
// per output row (16 contiguous weights each), NT/8 rows per pass of the workgroup, the loads of U passes in
// flight before the first one is consumed (a dependent L2 round trip per pass would dominate otherwise).
// `emit(n, value)` is called by one lane per row.
template <int N, int U, typename F>
__device__ __forceinline__ void gemv_k128(const bf16* __restrict__ W, const float* __restrict__ bias, const float* v, int tid,
                                          F emit, int rows = N) {
  constexpr int RP = NT / 8, PASSES = (N + RP - 1) / RP;
  static_assert(PASSES % U == 0, "U must divide the number of passes");
  const int sub = tid & 7, r0 = tid >> 3;
  bf16x2 va[4], vb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    va[j] = pack2(v[sub * 16 + 2 * j], v[sub * 16 + 2 * j + 1]);
    vb[j] = pack2(v[sub * 16 + 8 + 2 * j], v[sub * 16 + 8 + 2 * j + 1]);
  }
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
    for (int u = 0; u < U; ++u) {
      const int n = (p0 + u) * RP + r0;
      float s = dot8p(a[u], va, 0.f) + dot8p(b[u], vb, 0.f);
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      if (sub == 0 && n < rows) emit(n, s + (bias ? bias[n] : 0.f));
    }
  }
}

// out[n] = bias[n] + sum_k W[n][k] v[k] for K = 2048, N = 128: a wave per output row, U rows (4 U loads) in flight
template <int U, typename F>
__device__ __forceinline__ void gemv_k2048(const bf16* __restrict__ W, const float* __restrict__ bias, const float* v, int tid,
                                           F emit) {
  constexpr int ROWS = D / NWAVE;   // rows per wave
  static_assert(ROWS % U == 0, "U must divide the rows per wave");
  const int lane = tid & 63, wave = tid >> 6;
  bf16x2 vin[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) vin[c][j] = pack2(v[c * 512 + lane * 8 + 2 * j], v[c * 512 + lane * 8 + 2 * j + 1]);
#pragma unroll 1
  for (int i0 = 0; i0 < ROWS; i0 += U) {
    bf16x8 a[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        a[u][c] = *reinterpret_cast<const bf16x8*>(W + (size_t)(wave * ROWS + i0 + u) * FF + c * 512 + lane * 8);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = (dot8p(a[u][0], vin[0], 0.f) + dot8p(a[u][1], vin[1], 0.f)) +
                (dot8p(a[u][2], vin[2], 0.f) + dot8p(a[u][3], vin[3], 0.f));
      s = wave_sum(s);
      const int n = wave * ROWS + i0 + u;
      if (lane == 0) emit(n, s + (bias ? bias[n] : 0.f));
    }
  }
}

// z[0..128) (one value per thread tid < 128) -> LayerNorm -> out[] as bf16-rounded floats
__device__ __forceinline__ void layernorm_row(Smem& sm, float z, const float* gamma, const float* beta, float* out, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const bool own = tid < D;
  float s = wave_sum(own ? z : 0.f);
  if (lane == 0) sm.red[0][wave] = s;
  lds_barrier();
  const float mean = (sm.red[0][0] + sm.red[0][1]) * (1.0f / D);
  const float d = own ? z - mean : 0.f;
  float vs = wave_sum(d * d);
  if (lane == 0) sm.red[1][wave] = vs;
  lds_barrier();
  const float rstd = rsqrtf((sm.red[1][0] + sm.red[1][1]) * (1.0f / D) + 1e-5f);
  if (own) out[tid] = rbf(d * rstd * gamma[tid] + beta[tid]);
  lds_barrier();
}

// Attention of one query row (sm q[128], 16 heads x 8) over `nkeys` keys: K row of key j at kbase + j*ld, V at
// vbase + j*ld (bf16, 128 wide).  Lane = (key slot, head): 16 lanes read one key's 256-byte row, a wave covers
// 4 keys per step, the workgroup 32.  If last_k != nullptr, key nkeys-1 is read from LDS instead (the row this
// workgroup has just produced).  Result -> sm.attn (bf16-rounded).
__device__ __forceinline__ void attend(Smem& sm, const float* q, const bf16* kbase, const bf16* vbase, size_t ld, int nkeys,
                                       const float* kbias, const float* last_k, const float* last_v, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int hh = lane & 15, slot = wave * 4 + (lane >> 4);
  const float scale2 = rsqrtf((float)HD) * LOG2E;
  bf16x2 qv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) qv[j] = pack2(q[hh * HD + 2 * j], q[hh * HD + 2 * j + 1]);
  float m = NEG_BIG, l = 0.f, acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  constexpr int U = 4, STEP = 4 * NWAVE;
#pragma unroll 1
  for (int k0 = slot; k0 < nkeys; k0 += U * STEP) {
    bf16x8 kt[U], vt[U];
    float sc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = min(k0 + u * STEP, nkeys - 1);
      kt[u] = *reinterpret_cast<const bf16x8*>(kbase + (size_t)key * ld + hh * HD);
      vt[u] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)key * ld + hh * HD);
      sc[u] = kbias ? kbias[key] * LOG2E : 0.f;
    }
    // scores of the batch, one rescale of the running sums per batch
    float mb = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = k0 + u * STEP;
      if (last_k && key == nkeys - 1) {   // the row this workgroup has just produced: not yet visible in memory
#pragma unroll
        for (int d = 0; d < HD; ++d) { kt[u][d] = (bf16)last_k[hh * HD + d]; vt[u][d] = (bf16)last_v[hh * HD + d]; }
      }
      sc[u] = key < nkeys ? dot8p(kt[u], qv, 0.f) * scale2 + sc[u] : NEG_BIG;
      mb = fmaxf(mb, sc[u]);
    }
    const float alpha = __builtin_amdgcn_exp2f(m - mb);
    m = mb;
    l *= alpha;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] *= alpha;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float p = (k0 + u * STEP) < nkeys ? __builtin_amdgcn_exp2f(sc[u] - mb) : 0.f;
      l += p;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] += p * (float)vt[u][d];
    }
  }
  // merge the 4 key slots of the wave (lanes with equal head), then the waves through LDS
  float M = fmaxf(m, __shfl_xor(m, 16));
  M = fmaxf(M, __shfl_xor(M, 32));
  const float f = __builtin_amdgcn_exp2f(m - M);
  l *= f;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    acc[d] *= f;
    acc[d] += __shfl_xor(acc[d], 16);
    acc[d] += __shfl_xor(acc[d], 32);
  }
  if (lane < NH) {
    sm.part[wave][hh][0] = M;
    sm.part[wave][hh][1] = l;
#pragma unroll
    for (int d = 0; d < HD; ++d) sm.part[wave][hh][2 + d] = acc[d];
  }
  lds_barrier();
  if (tid < D) {
    const int h2 = tid >> 3, d = tid & 7;
    float MM = NEG_BIG;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) MM = fmaxf(MM, sm.part[w][h2][0]);
    float L = 0.f, A = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) {
      const float g = __builtin_amdgcn_exp2f(sm.part[w][h2][0] - MM);
      L += sm.part[w][h2][1] * g;
      A += sm.part[w][h2][2 + d] * g;
    }
    sm.attn[tid] = rbf(A * (L > 0.f ? 1.0f / L : 0.f));
  }
  lds_barrier();
}

__global__ __launch_bounds__(NT) void decode_stack_kernel(const StackArgs args) {
  __shared__ Smem sm;
  const mmst_dec_stack_step& s = args.s;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int pos = s.pos;
  // input row: given, or embedding of the token + positional row
  if (tid < D) {
    float v;
    if (s.tok) {
      const int64_t tk = s.tok[b];
      v = rbf(s.embed[(size_t)tk * D + tid] + s.pos_row[tid]);
    } else {
      v = (float)((const bf16*)s.x)[(size_t)b * D + tid];
    }
    sm.x[tid] = v;
  }
  lds_barrier();
  const int bm = b / s.kv_group;   // row block of the encoder memory this sequence reads
  const float* mbias = s.mem_bias ? s.mem_bias + (size_t)bm * s.L : nullptr;
  for (int l = 0; l < args.n_layers; ++l) {
    const mmst_dec_params& p = args.p[l];
    bf16* cache = (bf16*)s.qkv_cache[l] + (size_t)b * s.cap * 3 * D;
    // 1. Q|K|V of the new position, appended to the cache
    bf16* crow = cache + (size_t)pos * 3 * D;
    gemv_k128<3 * D, 3>((const bf16*)p.w_qkv, p.b_qkv, sm.x, tid, [&](int n, float v) {
      const bf16 r = (bf16)v;
      sm.qkv[n] = (float)r;
      crow[n] = r;
    });
    lds_barrier();
    // 2. self-attention over positions 0..pos (every cached key is visible)
    attend(sm, sm.qkv, cache + D, cache + 2 * D, 3 * D, pos + 1, nullptr, sm.qkv + D, sm.qkv + 2 * D, tid);
    // 3. out_proj + residual + norm1
    float z = 0.f;
    gemv_k128<D, 1>((const bf16*)p.w_o, p.b_o, sm.attn, tid, [&](int n, float v) { sm.qc[n] = v + sm.x[n]; });
    lds_barrier();
    z = tid < D ? sm.qc[tid] : 0.f;
    layernorm_row(sm, z, p.g1, p.be1, sm.x1, tid);
    // 4. cross-attention query
    gemv_k128<D, 1>((const bf16*)p.w_cqkv, p.b_cqkv, sm.x1, tid, [&](int n, float v) { sm.qc[n] = rbf(v); });
    lds_barrier();
    // 5. cross-attention over the projected memory
    const bf16* kv = (const bf16*)s.kvc[l] + (size_t)bm * s.L * 2 * D;
    attend(sm, sm.qc, kv, kv + D, 2 * D, s.L, mbias, nullptr, nullptr, tid);
    // 6. out_proj + residual + norm2
    gemv_k128<D, 1>((const bf16*)p.w_co, p.b_co, sm.attn, tid, [&](int n, float v) { sm.qc[n] = v + sm.x1[n]; });
    lds_barrier();
    z = tid < D ? sm.qc[tid] : 0.f;
    layernorm_row(sm, z, p.g2, p.be2, sm.x2, tid);
    // 7. feed-forward block + norm3
    gemv_k128<FF, 8>((const bf16*)p.w_1, p.b_1, sm.x2, tid, [&](int n, float v) { sm.h[n] = rbf(fmaxf(v, 0.f)); });
    lds_barrier();
    gemv_k2048<4>((const bf16*)p.w_2, p.b_2, sm.h, tid, [&](int n, float v) { sm.qc[n] = v + sm.x2[n]; });
    lds_barrier();
    z = tid < D ? sm.qc[tid] : 0.f;
    layernorm_row(sm, z, p.g3, p.be3, sm.x, tid);
  }
  if (s.y && tid < D) ((bf16*)s.y)[(size_t)b * D + tid] = (bf16)sm.x[tid];
  if (!s.logits) return;
  // vocabulary projection (weight rows beyond the vocabulary are zero padding of the caller's matrix)
  float* lg = s.logits + (size_t)b * s.ld_logits;
  gemv_k128<64, 1>((const bf16*)s.w_out, s.b_out, sm.x, tid, [&](int n, float v) {
    sm.qc[n] = v;
    lg[n] = v;
  }, s.n_out);
  if (!s.next_tok) return;
  lds_barrier();
  if (tid < 64) {   // softmax(logits / temperature), first index of the maximum
    const int lane = tid;
    const float x = lane < s.V ? sm.qc[lane] / s.temperature : -INFINITY;
    const float mx = wave_max(x);
    const float e = lane < s.V ? __expf(x - mx) : 0.f;
    const float se = wave_sum(e);
    const unsigned long long mk = __ballot(x == mx);
    const int arg = __ffsll((long long)mk) - 1;
    if (lane == arg) {
      s.next_tok[b] = arg;
      if (s.next_prob) s.next_prob[b] = e / se;
    }
  }
}

}  // namespace

extern "C" int mmst_sizeof_dec_stack_step() { return (int)sizeof(mmst_dec_stack_step); }

extern "C" int mmst_decoder_stack_step(int n_layers, const mmst_dec_params* p, const mmst_dec_stack_step* s, int nhead,
                                       void* stream) {
  if (!p || !s || nhead != NH || n_layers < 1 || n_layers > MMST_MAX_DEC_LAYERS) return MMST_ERR_ARG;
  if (s->B < 1 || s->pos < 0 || s->pos >= s->cap || s->L < 1 || s->kv_group < 0) return MMST_ERR_ARG;
  if (s->kv_group > 1 && s->B % s->kv_group != 0) return MMST_ERR_ARG;
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
  hipLaunchKernelGGL(decode_stack_kernel, dim3(s->B), dim3(NT), 0, (hipStream_t)stream, a);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
