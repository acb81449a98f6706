/* libmmst.so -- the building-block launchers under the layer-level entry points of mmst.h.
 *
 * The layer calls of mmst.h are fixed sequences of three launchers: a grouped bf16 GEMM with fused row
 * epilogues, multi-head attention, and the fused feed-forward block.  They are exported for three callers: the
 * kernel-level parity tests (each kernel against fp32 torch arithmetic), the one-off projections of the host layer
 * (IR embedding models_MMT_v15_4.py:440-444, fingerprint head :946-948, vocabulary projection :974 and their
 * gradients), and measurement (the event tap).  Same conventions as mmst.h: raw device pointers, a hipStream_t passed
 * as void*, 0 or a negative error code, nothing allocated on the device, kernels only enqueued.
 *
 * Process-wide state lives here and only here: mmst_ffn_config / mmst_ffn_wgrad_config select kernel paths for
 * every later call of the process, and the event tap (mmst_tap_*) is a single global recorder.  Neither is
 * re-entrant: configure before the threads that launch, not beside them.
 *
 * bf16 pointers are declared `const void*` / `void*`; fields marked "launcher" are filled in by the library.
 */
#ifndef MMST_LOWLEVEL_H
#define MMST_LOWLEVEL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- grouped GEMM: OUT[Pn x Qn] = sum_c Pop[p][c] * Qop[q][c], fp32 accumulation.  "P" is the token side, "Q" the
 * weight side (nn.Linear: P = x [M][K], Q = weight [N][K]); either may be stored contraction-major. ---- */
enum mmst_gemm_epilogue {
  MMST_EPI_BIAS = 0,   /* C = act(acc + bias) (+dropout) (+R), bf16 or fp32 out: nn.Linear forward, dX = dY W */
  MMST_EPI_DRELU = 1,  /* C = acc * (R > 0 ? scale : 0): gradient through ReLU (+dropout) */
  MMST_EPI_LN = 2,     /* z = R + drop(acc + bias); C = LayerNorm(z); Z = z; stats = (mean, rstd): out_proj + norm */
  MMST_EPI_LNBWD = 3,  /* dy = acc + R; LayerNorm backward -> C (dz), C2 (dz through the dropout mask), dgamma, dbeta */
  MMST_EPI_ATOMIC = 4  /* C(fp32)[p][q] += acc: split-K weight gradients dW += dY^T X (+ dbias column sums) */
};
enum mmst_gemm_flags { MMST_GF_RELU = 1, MMST_GF_OUT_F32 = 2 };

typedef struct {
  const void* P; const void* Q;       /* bf16 operands */
  void* C; const float* bias; const void* R;
  const float* gamma; const float* beta;
  void* Z; float* stats; void* C2;
  float* dgamma; float* dbeta; float* dbias;
  int ldp, ldq, ldc, ldr;
  int Pn, Qn, K;
  int flags;
  float drop_p; uint32_t seed;
  int cS, cL, coff, rS, rL, roff;     /* optional row remap of C / R: row b*S + s lives at row b*L + off + s */
  int ksplit;                         /* K slices (MMST_EPI_ATOMIC), >= 1 */
  int tile_begin, ptiles, qtiles;     /* launcher */
} mmst_gemm_prob;

#define MMST_GEMM_MAX_GROUPS 16
typedef struct {
  int n;
  int lds_pad;                        /* extra dynamic LDS per workgroup in bytes (occupancy control), normally 0 */
  mmst_gemm_prob p[MMST_GEMM_MAX_GROUPS];
} mmst_gemm_batch;

int mmst_gemm(const mmst_gemm_batch* batch, int epilogue, int p_contraction_major, int q_contraction_major,
              void* stream);

/* ---- multi-head attention (head_dim 8 or 32): nn.MultiheadAttention's softmax(Q K^T / sqrt(hd) + bias) V and its
 * backward.  Query row (b, t) at Q + (b*Tq + t)*ldq + h*hd, key/value row (b, s) at K + (b*Sk + s)*ldk + h*hd. ---- */
typedef struct {
  const void* Q; const void* K; const void* V;
  const float* kbias;                 /* additive [B][Sk], or NULL */
  void* O;
  float* lse;                         /* [B][H][Tq], log2 domain */
  const void* dO;
  float* delta;                       /* [B][H][Tq] rowsum(dO * O), written by the dQ pass */
  void* dQ; void* dK; void* dV;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int B, H, Tq, Sk;
  uint32_t seed;
  int block_begin, blocks_per_bh;     /* launcher */
  int kv_rows;                        /* forward: rows per batch element of the K/V buffers (0 = Sk) */
  int tail_o, tail_s;                 /* launcher */
} mmst_attn_prob;

typedef struct {
  int n;                              /* <= MMST_MAX_GROUPS problems */
  int hd;                             /* 8 or 32 */
  int causal;
  float drop_p;
  mmst_attn_prob p[6];
} mmst_attn_batch;

int mmst_attn_fwd(const mmst_attn_batch* batch, void* stream);
int mmst_attn_bwd(const mmst_attn_batch* batch, void* stream);   /* dQ (+delta), then dK / dV */

/* ---- fused feed-forward block, models_MMT_v15_4.py:510-541 (linear1, ReLU, dropout, linear2, dropout, residual,
 * norm): backward == 0: y = LN(x + drop(W2 drop(relu(W1 x + b1)) + b2)); else the activation-gradient pass ---- */
typedef struct {
  mmst_gemm_prob epi;                 /* final row epilogue (MMST_EPI_LN forward / MMST_EPI_LNBWD backward) */
  const void* X; const void* W1; const void* W2; const float* b1;
  void* H;                            /* [M][2048] hidden activation: written (forward), read (backward) */
  void* dH;                           /* optional [M][2048] hidden gradient (backward) */
  float* ws;                          /* optional fp32 [8][M][128]: hidden split for few row tiles */
  const void* Wn; void* Cn;           /* optional (backward): Cn[M][128] = (epilogue output after its dropout) x Wn[128][128],
                                       * the gradient through the linear layer in front of the block (out_proj); written
                                       * only by the fused kernel without hidden split, see mmst_ffn */
  uint32_t* Hs;                       /* optional [M][64] sign words of H (bit set = entry > 0; word 16 q + j of a row  <->
                                       * hidden columns 64 c + 32 t + 8 q + e, c = 2 j or 2 j + 1, t = 0..1, e = 0..7; the flag
                                       * of (c, t, e) is bit (e odd ? 31 : 15) - 4 t - (e >> 1) - 8 (c & 1)):
                                       * written by a forward launch; a backward launch that gets them for every problem,
                                       * no dH and no hidden split reads these 256 B per row in place of the 4 KiB of H */
  float drop_p; uint32_t seed;
  int tile_begin, hsplit;             /* launcher */
  const void* A; const void* Wp;      /* optional (forward): X = LN(pre.R + drop(A Wp^T + pre.bias)), the out_proj +
                                       * dropout + residual + norm1 in front of the block (models_MMT_v15_4.py:497-509 via
                                       * nn.TransformerEncoderLayer), computed by the same launch; pre = that GEMM's row
                                       * epilogue (MMST_EPI_LN fields; pre.C must be X); Wp NULL = X is read */
  mmst_gemm_prob pre;
  const void* Wq; const float* bq;    /* optional (forward): Cq[M][nq] = Y Wq^T + bq on the rows the block has produced (the
                                       * in_proj of the next layer); computed only by the fused kernel without hidden split
                                       * and row remap: mmst_ffn then returns 2 */
  void* Cq; int nq;                   /* nq: a multiple of 64, at most 384 */
} mmst_ffn_prob;
typedef struct { int n; mmst_ffn_prob p[6]; } mmst_ffn_batch;
/* returns 0, a negative error code, (backward only) 1 when the launch also wrote every problem's Cn, or (forward only)
 * 2 when it also wrote every problem's Cq */
int mmst_ffn(const mmst_ffn_batch* batch, int backward, void* stream);

/* weight gradients of the block with the hidden gradient recomputed in the kernel (accumulating) */
typedef struct {
  const void* dZ; const void* H; const void* X; const void* W2T;
  float* dW1; float* db1; float* dW2; float* db2;
  int M; float drop_p;
  int tile_begin, rs; float skew;     /* launcher */
} mmst_ffn_wgrad_prob;
typedef struct { int n; mmst_ffn_wgrad_prob p[6]; } mmst_ffn_wgrad_batch;
int mmst_ffn_wgrad(const mmst_ffn_wgrad_batch* batch, void* stream);

/* ---- process-wide configuration (not re-entrant, see the head of this file) ---- */
/* launches with fewer 128-row tiles than this run the feed-forward block as two grouped GEMMs; negative = default */
int mmst_ffn_config(int fuse_min_tiles);
/* 1 (default) or 2: feed-forward weight gradients of fused launches recompute the hidden gradient, with the 4-wave
 * or the 8-wave shape of the kernel; 3: as 1, and the encoder layer entry points hand the sign words of the hidden
 * activation (mmst_enc_acts.hsign) from the forward to the backward launch, which then does not read the hidden
 * activation (2.4 GB less HBM traffic per training step at 64 molecules, 0.03 ms more time); 0: the backward kernel
 * stores the hidden gradient and a split-K GEMM reads it; negative: back to the default */
int mmst_ffn_wgrad_config(int recompute);
int mmst_ffn_wgrad_mode(void);   /* the setting in force: 0, 1, 2 or 3 */
/* mmst_attn_bwd on self-attention (the encoder layers and the decoder's causal self-attention; torch call site
 * nn.MultiheadAttention backward): launches whose rows fit (head_dim 32: 608 rows; head_dim 8: 128 tiled rows) can
 * run ONE kernel that evaluates the probabilities once for dQ, dK and dV instead of two passes (dQ, then dK/dV).
 * Bits of `mode`: 0 = head_dim 32, 1 = head_dim 8 without a mask, 2 = head_dim 8 with the causal mask; bit 3 (8):
 * head_dim 8 without a mask and at most 128 tiled rows runs both passes in one workgroup per (batch, 8 heads) that
 * stages Q, K, V and dO once.  Default 0 (inside a training step the two passes share the GPU better with the second
 * stream's kernels); negative: back to the default.  All forms give the same results to bf16 rounding. */
int mmst_attn_bwd_config(int mode);
int mmst_attn_bwd_mode(void);    /* the setting in force: 0..15 */
/* mmst_decoder_stack_fwd and the decoder part of mmst_train_step on target sequences of at most 64 positions: 1 (default)
 * = a workgroup per molecule runs everything of a layer but the cross-attention in two launches per layer; 0 = the
 * launch sequences of the per-layer entry points; negative: back to the default.  Same saved activations and dropout
 * masks either way; results agree to bf16 rounding. */
int mmst_decoder_fused_config(int on);
int mmst_decoder_fused_mode(void);

/* ---- diagnostic: HIP events around every launch of ONE GEMM variant, recorded on the launch stream.
 * mmst_tap_config allocates host memory and events (the only entry point of the library that does). ---- */
#define MMST_TAP_FFN_WGRAD 100   /* as `epilogue`: the launches of mmst_ffn_wgrad's kernel (the other two arguments unused) */
int mmst_tap_config(int epilogue, int p_contraction_major, int q_contraction_major, int capacity);
int mmst_tap_reset(void);
/* sum of the durations, number of timed launches, their algorithmic flops / bytes, the kernel's name */
int mmst_tap_read(float* total_ms, int* count, double* flops, double* bytes, char* name, int cap);

/* struct sizes as compiled into the library: a binding checks its own layout against them */
int mmst_sizeof_gemm_batch(void);
int mmst_sizeof_attn_batch(void);
int mmst_sizeof_ffn_batch(void);
int mmst_sizeof_ffn_wgrad_batch(void);
int mmst_sizeof_dec_stack_step(void);

#ifdef __cplusplus
}
#endif
#endif
