// Decoder layers of a SHORT target sequence (T <= 64 rows per molecule) with one workgroup per molecule (gfx950).
//
// The per-layer entry points run a decoder layer as eight to eleven launches of 7-27 us each over B*T rows (6 % of a
// training step's rows): every launch is a dependent round trip through memory and most CUs idle.  A molecule's target
// rows are independent of every other molecule's, so here one workgroup keeps the molecule's 64 x 128 rows in LDS and
// walks through everything of a layer that is row-local or local to the molecule (projections, causal self-attention,
// LayerNorms, the feed-forward block); only the cross-attention over the 582 memory rows, which is exp/VALU-bound and
// needs the whole chip, stays a launch of its own (attn.hip).  Saved activations, dropout masks and statistics are the
// ones the per-layer entry points write, so either path's backward reads either path's forward.
#pragma once
#include "../../include/mmst.h"
#include "common.h"

// One call = [tail of layer l-1] followed by [head of layer l] (either part optional), two launches:
//   tail: x2 = LN2(x1 + drop(attn2 Wco^T + b)), y = LN3(x2 + drop(W2 drop(relu(W1 x2 + b1)) + b2))
//         (row-local: four workgroups per molecule share the hidden dimension, the launch that follows adds them up)
//   head: qkv = x Wqkv^T + b, causal self-attention, x1 = LN1(x + drop(attn1 Wo^T + b)), qc = x1 Wcq^T + b
struct DecFusedFwd {
  mmst_dec_params pt; mmst_dec_acts at; uint32_t seed_t; int do_tail;
  mmst_dec_params ph; mmst_dec_acts ah; uint32_t seed_h; int do_head;
  int B, T;
  float drop_p;
  float* ws;   // fp32 [DEC_FFN_PARTS][B*T][128]: partial sums of the feed-forward block (required with do_tail)
};
#define DEC_FFN_PARTS 4   // workgroups per molecule that share the hidden dimension of the feed-forward block
int launch_dec_fused_fwd(const DecFusedFwd& a, hipStream_t stream);

// Backward, activation gradients (the parameter gradients stay with the grouped split-K launches that read what these
// kernels store).  Per layer, from its incoming gradient (dz3, dzd3):
//   mid : [optionally first: (dz3, dzd3) of THIS layer = LayerNorm-3 backward of (dqkv' Wqkv' + dz1') of the layer above]
//         dh = ((dzd3 W2) * mask(h)) / (1 - p), partial sums of dh W1 per hidden quarter -> workspace
//   ln  : (dz2, dzd2) = LayerNorm-2 backward of (sum of the partial sums + dz3); da2 = dzd2 Wco
//         ... cross-attention backward (attn.hip) gives dqc ...
//   ln  : (dz1, dzd1) = LayerNorm-1 backward of (dqc Wcq + dz2); da = dzd1 Wo
//         ... self-attention backward (attn.hip) gives dqkv; the layer below (or a GEMM launch) turns it into its dz3
struct DecFusedBwd {
  mmst_dec_params p; mmst_dec_grads g; mmst_dec_acts a; mmst_dec_bwd w;   // this layer
  uint32_t seed;            // this layer's dropout seed
  const void* up_dqkv;      // layer above: dqkv' [B*T][384], or NULL (then w.dz3 / w.dzd3 are read)
  const void* up_dz1;       //              dz1' (the residual)
  const void* up_ft_qkv;    //              fragment-major image of Wqkv'^T
  int B, T;
  float drop_p;
  float* ws;                // fp32 [DEC_FFN_PARTS][B*T][128]
};
struct DecFusedLn {
  const float* ws;          // partial sums to add up (then `in` / ft_in are unused), or NULL
  const void* in; const void* ft_in;     // input rows [B*T][128] and the image of the transposed weight they go through
  const void* res;          // residual gradient rows added to it
  const void* z; const float* stats; const float* gamma; float* dgamma; float* dbeta;   // the LayerNorm
  void* dz; void* dzd;      // out: gradient at the pre-norm sum, the same through the dropout mask (seed)
  const void* ft_out; void* out;         // out = dzd Wout (image of Wout^T)
  uint32_t seed;
  int B, T;
  float drop_p;
};
// self : (dz1, dzd1) = LayerNorm-1 backward of (dqc Wcq + dz2); da = dzd1 Wo; causal self-attention backward -> dqkv;
//        dx = dqkv Wqkv + dz1, which is either the output (bottom layer: z3 == NULL) or goes through the LayerNorm-3
//        backward of the layer below (its dz3 / dzd3 and gamma / beta gradients).  One launch in place of the second
//        `ln` launch, the attention launch and the first part of the next `mid` launch.
struct DecFusedSelf {
  const void* dqc; const void* ft_cq; const void* dz2;              // input rows, image of Wcq^T, residual
  const void* z1; const float* st1; const float* g1; float* dg1; float* dbe1;
  void* dz1; void* dzd1;
  const void* ft_o;                                                 // image of Wo^T
  const void* qkv; const void* attn1; const float* lse1; void* dqkv;
  const void* ft_qkv;                                               // image of Wqkv^T
  const void* z3; const float* st3; const float* g3; float* dg3; float* dbe3; void* dz3; void* dzd3;   // layer below, or NULL
  void* dx;                                                         // bottom layer: dx rows [B*T][128]
  uint32_t seed, seed_below;                                        // layer seeds (sites are added to them)
  int B, T;
  float drop_p;
};
int launch_dec_bwd_self(const DecFusedSelf& a, hipStream_t stream);
int launch_dec_bwd_mid(const DecFusedBwd& a, hipStream_t stream);
int launch_dec_bwd_ln(const DecFusedLn& a, hipStream_t stream);
bool dec_fused_enabled();   // environment MMST_DEC_FUSED (default on), or what dec_fused_set() said last
void dec_fused_set(int on);  // 1 / 0; negative: back to the default
