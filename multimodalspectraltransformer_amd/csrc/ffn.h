// Fused feed-forward block of a post-LN transformer layer (gfx950): the two chained GEMMs of
//   forward   y  = LN(x + drop(relu-drop(x W1^T + b1) W2^T + b2))
//   backward  dx = LNbwd(((dz W2) * mask(h)) W1 + r)
// run in ONE kernel per 128-row tile.  The 2048-wide hidden activation never returns from HBM between the
// two GEMMs: it is produced 128 columns at a time in MFMA accumulators, re-packed in registers as the
// token-side operand of the second GEMM and (forward only) written out once for the backward pass.
#pragma once
#include "gemm.h"

#define FFN_MAX_GROUPS 6
#define FFN_MAX_HSPLIT 8

struct FfnProb {
  GemmProb epi;        // final row epilogue (EPI_LN forward / EPI_LNBWD backward); Pn = rows, Qn = 128
  const bf16* X;       // [M][128] token-side input: layer input (forward) or gradient of the FFN output (backward)
  const bf16* W1;      // first-GEMM weight  [2048][128]: linear1 (forward), transpose of linear2 (backward)
  const bf16* W2;      // second-GEMM weight [128][2048]: linear2 (forward), transpose of linear1 (backward)
  const float* b1;     // [2048] (forward)
  bf16* H;             // [M][2048] hidden activation after ReLU and dropout: written (forward), read (backward)
  bf16* dH;            // optional [M][2048] output of the backward kernel (gradient at the hidden pre-activation)
  float* ws;           // optional fp32 [FFN_MAX_HSPLIT][M][128]: enables the hidden-dimension split for few row tiles
  const bf16* Wn;      // optional (backward, fused launches without hidden split): [128][128] weight of the linear layer
  bf16* Cn;            //   that precedes the block; Cn[M][128] = E Wn, E = the epilogue's output after its dropout mask
  uint32_t* Hs;        // optional [M][64] sign words of H (bit set = entry > 0), 256 B per row in place of 4 KiB: written
                       //   by the forward launch; a backward launch that gets them (and no dH) never reads H.  Word
                       //   16 q + j of a row  <->  hidden columns 64 c + 32 t + 8 q + e, c = 2 j, 2 j + 1; the flag of
                       //   (c, t, e) is bit (e odd ? 31 : 15) - 4 t - (e >> 1) - 8 (c & 1)
  float drop_p;        // dropout on the hidden activation
  uint32_t seed;
  int tile_begin;      // filled by the launcher
  int hsplit;          // filled by the launcher: workgroups per row tile (each walks 2048/hsplit hidden columns)
  // Optional (forward): the linear layer and LayerNorm in FRONT of the block (out_proj + dropout + residual + norm1 of a
  // transformer layer), X = LN(pre.R + drop(A Wp^T + pre.bias)).  pre is that GEMM's row epilogue (EPI_LN: C = X, Z,
  // stats, R, bias, gamma, beta, drop_p, seed; C must be X).  The fused kernel without hidden split computes it as a
  // prologue on its row tile (the rows go from the MFMA accumulators into the block without another launch); the other
  // paths run it as a GEMM launch of its own first.
  const bf16* A;       // [M][128] input of that linear layer (attention output)
  const bf16* Wp;      // [128][128] its weight; null = no prologue, X is read
  GemmProb pre;
  // Optional (forward): a linear layer BEHIND the block on the rows it has just produced, Cq[M][nq] = Y Wq^T + bq (the
  // in_proj of the next transformer layer, nq = 384).  Only the fused kernel without hidden split and without a row
  // remap of its output computes it (launch_ffn_fwd reports whether it did); otherwise the caller launches that GEMM.
  const bf16* Wq;      // [nq][128]; null = nothing chained
  const float* bq;     // [nq] or null
  bf16* Cq;
  int nq;              // a multiple of 64, at most 384
};

struct FfnBatch {
  int n;
  FfnProb p[FFN_MAX_GROUPS];
};

// Weight gradients of the block, recomputing the hidden gradient (ffn_wgrad.hip).  Accumulates (fp32 atomics).
struct FfnWgradProb {
  const bf16* dZ;      // [M][128] gradient of the block output (after the output dropout mask)
  const bf16* H;       // [M][2048] saved hidden activation
  const bf16* X;       // [M][128] block input
  const bf16* W2T;     // [2048][128] transpose of linear2
  float* dW1; float* db1;   // [2048][128], [2048]
  float* dW2; float* db2;   // [128][2048], [128]
  int M;
  float drop_p;
  int tile_begin;      // filled by the launcher: first workgroup of this problem
  int rs;              // filled by the launcher: row splits (at most 16), 2000 to 3100 rows each on average
  float skew;          // filled by the launcher: the splits' sizes grow from (1 - skew) to (1 + skew) of the average
};
struct FfnWgradBatch {
  int n;
  FfnWgradProb p[FFN_MAX_GROUPS];
};
int launch_ffn_wgrad(FfnWgradBatch& b, hipStream_t stream);

// post_done (optional): set to whether the launch also produced Cq of every problem
int launch_ffn_fwd(FfnBatch& b, hipStream_t stream, bool* post_done = nullptr);
// next_done (optional): set to whether the launch also produced Cn = E Wn (only the fused kernel without hidden split does)
int launch_ffn_bwd(FfnBatch& b, hipStream_t stream, bool* next_done = nullptr);
// Launches with fewer 128-row tiles than this run as two grouped GEMM launches instead of the fused kernel
// (default 96, or the environment variable MMST_FFN_FUSE_MIN_TILES; negative restores the default).
void ffn_set_fuse_min_tiles(int tiles);
int ffn_fuse_min_tiles();
int ffn_wgrad_recompute();          // see ffn.hip
bool ffn_bwd_signs();               // see ffn.hip: setting 3 of ffn_wgrad_recompute()
void ffn_set_wgrad_recompute(int on);
