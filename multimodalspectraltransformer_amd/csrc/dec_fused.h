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
bool dec_fused_enabled();   // environment MMST_DEC_FUSED (default on), or what dec_fused_set() said last
void dec_fused_set(int on);  // 1 / 0; negative: back to the default
