// Multi-head attention forward / backward for gfx950 (head_dim 8 or 32).
//
// Layout: batch-first token rows.  Query row (b, t) lives at Q + (b*Tq + t)*ldq + h*HD,
// key/value row (b, s) at K + (b*Sk + s)*ldk + h*HD; the same pointer/stride scheme
// addresses slices of a fused QKV buffer.  kbias is an ADDITIVE fp32 key bias [B][Sk]
// (0/1 reference-style padding bias, -inf, or anything else), or null.
#pragma once
#include "common.h"

struct AttnProb {
  const bf16* Q; const bf16* K; const bf16* V;
  const float* kbias;
  bf16* O;
  float* lse;    // [B][H][Tq], log2 domain
  const bf16* dO;
  float* delta;  // [B][H][Tq] rowsum(dO * O)
  bf16* dQ; bf16* dK; bf16* dV;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int B, H, Tq, Sk;
  uint32_t seed;
  int block_begin;     // filled by the launcher
  int blocks_per_bh;   // filled by the launcher
  int kv_rows;         // forward only: rows per batch element of the K/V buffers (0 = Sk); > Sk for a K/V
                       // cache that has room for more positions than are attended to
  int tail_o, tail_s;  // filled by the launcher: owner / streamed rows handled outside the 32-row tiling
};

#define ATTN_MAX_GROUPS 6
struct AttnBatch {
  int n;
  int hd;        // 8 or 32
  int causal;
  float drop_p;
  AttnProb p[ATTN_MAX_GROUPS];
};

int launch_attn_fwd(AttnBatch& b, hipStream_t stream);
// dq pass also writes delta; dkv pass reads it: call dq first.
int launch_attn_bwd(AttnBatch& b, hipStream_t stream);      // = _q followed by _kv
int launch_attn_bwd_q(AttnBatch& b, hipStream_t stream);    // dQ, delta
int launch_attn_bwd_kv(AttnBatch& b, hipStream_t stream);   // dK, dV (reads delta)
// Single-pass backward (dQ, dK, dV from one evaluation of the probabilities) for self-attention without a causal mask:
// bit 0 = head_dim 32, bit 1 = head_dim 8, bit 2 = head_dim 8 with a causal mask; other values restore the default (attn.hip)
void attn_set_bwd_merged(int mode);
int attn_bwd_merged();
