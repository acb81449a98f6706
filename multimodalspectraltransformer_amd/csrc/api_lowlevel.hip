// Low-level C-ABI: the grouped GEMM and attention launchers, exported for the parity tests
// and for the one-off projections of the host layer (IR embedding, fingerprint head, vocabulary
// projection and their gradients).  Structs are plain C (see gemm.h / attn.h).
#include "../../include/mmst_lowlevel.h"
#include "attn.h"
#include "dec_fused.h"
#include "ffn.h"
#include "gemm.h"
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

// The public structs of include/mmst_lowlevel.h are the kernels' own structs with the bf16 pointer types erased.
static_assert(sizeof(mmst_gemm_prob) == sizeof(GemmProb) && sizeof(mmst_gemm_batch) == sizeof(GemmBatch), "gemm ABI");
static_assert(offsetof(mmst_gemm_prob, ldp) == offsetof(GemmProb, ldp) && offsetof(mmst_gemm_prob, seed) == offsetof(GemmProb, seed) &&
              offsetof(mmst_gemm_prob, ksplit) == offsetof(GemmProb, ksplit) && offsetof(mmst_gemm_prob, qtiles) == offsetof(GemmProb, qtiles) &&
              offsetof(mmst_gemm_batch, p) == offsetof(GemmBatch, p), "gemm ABI");
static_assert(sizeof(mmst_attn_prob) == sizeof(AttnProb) && sizeof(mmst_attn_batch) == sizeof(AttnBatch) &&
              offsetof(mmst_attn_prob, ldq) == offsetof(AttnProb, ldq) && offsetof(mmst_attn_prob, kv_rows) == offsetof(AttnProb, kv_rows) &&
              offsetof(mmst_attn_batch, p) == offsetof(AttnBatch, p), "attention ABI");
static_assert(sizeof(mmst_ffn_prob) == sizeof(FfnProb) && sizeof(mmst_ffn_batch) == sizeof(FfnBatch) &&
              offsetof(mmst_ffn_prob, X) == offsetof(FfnProb, X) && offsetof(mmst_ffn_prob, Cn) == offsetof(FfnProb, Cn) && offsetof(mmst_ffn_prob, Hs) == offsetof(FfnProb, Hs) &&
              offsetof(mmst_ffn_prob, hsplit) == offsetof(FfnProb, hsplit) && offsetof(mmst_ffn_prob, Wp) == offsetof(FfnProb, Wp) &&
              offsetof(mmst_ffn_prob, pre) == offsetof(FfnProb, pre) && offsetof(mmst_ffn_prob, nq) == offsetof(FfnProb, nq), "ffn ABI");
static_assert(sizeof(mmst_ffn_wgrad_prob) == sizeof(FfnWgradProb) && sizeof(mmst_ffn_wgrad_batch) == sizeof(FfnWgradBatch) &&
              offsetof(mmst_ffn_wgrad_prob, M) == offsetof(FfnWgradProb, M) && offsetof(mmst_ffn_wgrad_prob, rs) == offsetof(FfnWgradProb, rs) &&
              offsetof(mmst_ffn_wgrad_prob, skew) == offsetof(FfnWgradProb, skew), "ffn weight-gradient ABI");
static_assert(MMST_GEMM_MAX_GROUPS == GEMM_MAX_GROUPS && MMST_EPI_ATOMIC == EPI_ATOMIC && MMST_GF_OUT_F32 == GF_OUT_F32, "gemm ABI");

extern "C" {

int mmst_gemm(const mmst_gemm_batch* batch, int epi, int p_contraction_major, int q_contraction_major, void* stream) {
  if (!batch) return MMST_ERR_ARG;
  GemmBatch b = *reinterpret_cast<const GemmBatch*>(batch);
  return launch_gemm(b, epi, p_contraction_major != 0, q_contraction_major != 0, (hipStream_t)stream);
}

int mmst_attn_fwd(const mmst_attn_batch* batch, void* stream) {
  if (!batch) return MMST_ERR_ARG;
  AttnBatch b = *reinterpret_cast<const AttnBatch*>(batch);
  return launch_attn_fwd(b, (hipStream_t)stream);
}

int mmst_attn_bwd(const mmst_attn_batch* batch, void* stream) {
  if (!batch) return MMST_ERR_ARG;
  AttnBatch b = *reinterpret_cast<const AttnBatch*>(batch);
  return launch_attn_bwd(b, (hipStream_t)stream);
}

// fused feed-forward block: backward == 0 -> y = LN(x + drop(FFN(x))), else the activation-gradient pass
int mmst_ffn(const mmst_ffn_batch* batch, int backward, void* stream) {
  if (!batch) return MMST_ERR_ARG;
  FfnBatch b = *reinterpret_cast<const FfnBatch*>(batch);
  if (!backward) {
    bool post_done = false;
    const int rc = launch_ffn_fwd(b, (hipStream_t)stream, &post_done);
    return rc != MMST_OK ? rc : (post_done ? 2 : 0);
  }
  bool next_done = false;
  const int rc = launch_ffn_bwd(b, (hipStream_t)stream, &next_done);
  return rc != MMST_OK ? rc : (next_done ? 1 : 0);
}

// feed-forward weight gradients with the hidden gradient recomputed in the kernel (accumulating)
int mmst_ffn_wgrad(const mmst_ffn_wgrad_batch* batch, void* stream) {
  if (!batch) return MMST_ERR_ARG;
  FfnWgradBatch b = *reinterpret_cast<const FfnWgradBatch*>(batch);
  return launch_ffn_wgrad(b, (hipStream_t)stream);
}

// threshold (in 128-row tiles) below which mmst_ffn and the layer entry points use the two-launch path
int mmst_ffn_config(int fuse_min_tiles) {
  ffn_set_fuse_min_tiles(fuse_min_tiles);
  return MMST_OK;
}
// 1 / 2: feed-forward weight gradients of fused launches recompute the hidden gradient (4- / 8-wave kernel shape);
// 3: as 1 with the sign words of the hidden activation handed from the forward to the backward launch of a layer;
// 0: they read a stored one; negative: the default (0, or what MMST_FFN_WGRAD_RECOMPUTE says)
int mmst_ffn_wgrad_config(int recompute) {
  ffn_set_wgrad_recompute(recompute);
  return MMST_OK;
}

int mmst_ffn_wgrad_mode(void) { return ffn_wgrad_recompute(); }

// attention backward of self-attention in one pass: bit 0 = head_dim 32, bit 1 = head_dim 8, bit 2 = head_dim 8 with a
// causal mask, bit 3 = both passes of a short head_dim 8 sequence in one workgroup; 0 = the two passes (dQ, then dK/dV)
// everywhere; negative: the default (0, or what MMST_ATTN_BWD_MERGED says)
int mmst_attn_bwd_config(int mode) {
  attn_set_bwd_merged(mode);
  return MMST_OK;
}
int mmst_attn_bwd_mode(void) { return attn_bwd_merged(); }

// decoder layers of short target sequences with a workgroup per molecule (mmst_decoder_stack_fwd and the decoder part of
// mmst_train_step): 1 = on, 0 = the per-layer launch sequences, negative = the default (on, or MMST_DEC_FUSED)
int mmst_decoder_fused_config(int on) {
  dec_fused_set(on);
  return MMST_OK;
}
int mmst_decoder_fused_mode(void) { return dec_fused_enabled() ? 1 : 0; }

// ---- diagnostic tap: times every launch of one GEMM variant with HIP events on its own stream ----
int mmst_tap_config(int epi, int p_contraction_major, int q_contraction_major, int capacity) {
  GemmTap& t = g_gemm_tap;
  if (capacity > t.capacity) {
    hipEvent_t* s = (hipEvent_t*)malloc(sizeof(hipEvent_t) * capacity);
    hipEvent_t* e = (hipEvent_t*)malloc(sizeof(hipEvent_t) * capacity);
    if (!s || !e) return MMST_ERR_ARG;
    for (int i = 0; i < capacity; ++i) {
      if (i < t.capacity) { s[i] = t.start[i]; e[i] = t.stop[i]; }
      else if (hipEventCreate(&s[i]) != hipSuccess || hipEventCreate(&e[i]) != hipSuccess) return MMST_ERR_LAUNCH;
    }
    free(t.start); free(t.stop);
    t.start = s; t.stop = e; t.capacity = capacity;
  }
  t.epi = epi; t.pt = p_contraction_major; t.qt = q_contraction_major;
  t.count = 0; t.flops = 0.0; t.bytes = 0.0;
  return MMST_OK;
}

int mmst_tap_reset(void) {
  g_gemm_tap.count = 0; g_gemm_tap.flops = 0.0; g_gemm_tap.bytes = 0.0;
  return MMST_OK;
}

// Sum of the timed launches' durations.  Synchronises on the last recorded event.
int mmst_tap_read(float* total_ms, int* count, double* flops, double* bytes, char* name, int cap) {
  GemmTap& t = g_gemm_tap;
  float tot = 0.f;
  if (t.count > 0 && hipEventSynchronize(t.stop[t.count - 1]) != hipSuccess) return MMST_ERR_LAUNCH;
  for (int i = 0; i < t.count; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.start[i], t.stop[i]) != hipSuccess) return MMST_ERR_LAUNCH;
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (count) *count = t.count;
  if (flops) *flops = t.flops;
  if (bytes) *bytes = t.bytes;
  if (name && cap > 0) gemm_kernel_name(t.epi, t.pt != 0, t.qt != 0, name, cap);
  return MMST_OK;
}

int mmst_sizeof_gemm_batch(void) { return (int)sizeof(GemmBatch); }
int mmst_sizeof_attn_batch(void) { return (int)sizeof(AttnBatch); }
int mmst_sizeof_ffn_batch(void) { return (int)sizeof(FfnBatch); }
int mmst_sizeof_ffn_wgrad_batch(void) { return (int)sizeof(FfnWgradBatch); }
int mmst_version(void) { return 2; }
}
