// One whole training step behind one entry point (mmst_train_step): the launch sequence a caller obtains by
// driving the layer entry points of mmst.h one by one, issued from C++ so that the host cost per launch is the HIP
// runtime's (a few microseconds) instead of a foreign-function call's.  Nothing is allocated here.
#include "../../include/mmst.h"
#include "gemm.h"
#include <math.h>
#include <string.h>

namespace {

constexpr int D = MMST_D, VPAD = 64, IR_BINS = 1000;

#define RUN(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__) return rc__;   \
  } while (0)

inline GemmProb zero_prob() {
  GemmProb p;
  memset(&p, 0, sizeof(p));
  p.ksplit = 1;
  return p;
}

inline int one_gemm(const GemmProb& p, int epi, bool pt, bool qt, hipStream_t st) {
  GemmBatch gb;
  gb.n = 1;
  gb.p[0] = p;
  return launch_gemm(gb, epi, pt, qt, st);
}

// dropout seed of launch group `uid` in the step with seed `step_seed` (layers use seed .. seed + 7 per group)
inline uint32_t layer_seed(uint32_t step_seed, uint32_t uid) { return step_seed * 2654435761u + uid * 64u; }
enum { UID_MODAL = 0, UID_CROSS = 32, UID_DEC = 64, UID_TRG = 100 };

bool plan_ok(const mmst_step_plan* pl) {
  if (!pl || pl->B < 1 || pl->T < 1 || pl->L < 1) return false;
  if (pl->n_groups < 1 || pl->n_groups > MMST_MAX_GROUPS) return false;
  if (pl->n_enc_layers < 1 || pl->n_enc_layers > MMST_MAX_LAYERS || pl->n_dec_layers < 1 || pl->n_dec_layers > MMST_MAX_LAYERS)
    return false;
  if (pl->vocab < 1 || pl->vocab > VPAD || pl->n_ranges < 0 || pl->n_ranges > MMST_MAX_RANGES || pl->n_ffn < 0 ||
      pl->n_ffn > MMST_MAX_FFN || pl->ir_group >= pl->n_groups || pl->n_frag < 0 || pl->n_frag > MMST_MAX_FRAG)
    return false;
  for (int l = 0; l < pl->n_enc_layers; ++l)
    if (!pl->modal_p[l] || !pl->modal_g[l] || !pl->modal_a[l] || !pl->modal_w[l] || !pl->cross_p[l] || !pl->cross_g[l] ||
        !pl->cross_a[l] || !pl->cross_w[l])
      return false;
  for (int l = 0; l < pl->n_dec_layers; ++l)
    if (!pl->dec_p[l] || !pl->dec_g[l] || !pl->dec_a[l] || !pl->dec_w[l]) return false;
  // the sequence geometry the kernels will index with must be the plan's
  int rows = 0;
  for (int i = 0; i < pl->n_groups; ++i) {
    const mmst_enc_acts& a = pl->modal_a[pl->n_enc_layers - 1][i];
    if (a.B != pl->B || a.S != pl->embed_g[i].S || a.y_L != pl->L || a.y_off != pl->embed_g[i].off_glob) return false;
    rows += a.S;
  }
  if (rows > pl->L || pl->cross_a[0][0].B != pl->B || pl->cross_a[0][0].S != pl->L) return false;
  if (pl->dec_a[0]->B != pl->B || pl->dec_a[0]->L != pl->L) return false;
  return pl->memory && pl->mean && pl->fp1_w && pl->fingerprint && pl->trg_emb && pl->embed_trg && pl->pe_trg &&
         pl->fc_out_w && pl->fc_out_b && pl->logits && pl->dlogits && pl->loss && pl->dmem && pl->dmem_in && pl->dtrg &&
         pl->P && pl->G && pl->Pb && pl->PbT;
}

int forward(const mmst_step_plan* pl, const mmst_step_args* ar, const mmst_embed_group* eg, const mmst_embed_shared* es,
            hipStream_t st, hipStream_t wst) {
  const int B = pl->B, T = pl->T, L = pl->L, n = pl->n_groups;
  const int nl = pl->n_enc_layers, ndl = pl->n_dec_layers;
  RUN(mmst_embed_fwd(n, eg, es, st));
  if (pl->ir_group >= 0) {
    if (!ar->src_ir || !pl->ir_bf16 || !pl->ir_w) return MMST_ERR_ARG;
    RUN(mmst_cast_bf16(ar->src_ir, pl->ir_bf16, B, IR_BINS, IR_BINS, st));
    GemmProb q = zero_prob();   // relu(Linear(1000 -> 128)) into row 0 of the IR sequence of every molecule
    q.P = (const bf16*)pl->ir_bf16; q.ldp = IR_BINS; q.Q = (const bf16*)pl->ir_w; q.ldq = IR_BINS;
    q.C = eg[pl->ir_group].xseq; q.ldc = eg[pl->ir_group].S * D; q.bias = pl->ir_b;
    q.Pn = B; q.Qn = D; q.K = IR_BINS; q.flags = GF_RELU;
    RUN(one_gemm(q, EPI_BIAS, false, false, st));
  }
  uint32_t seeds[MMST_MAX_LAYERS];
  for (int l = 0; l < nl; ++l) seeds[l] = layer_seed(ar->step_seed, UID_MODAL + l);
  RUN(mmst_encoder_stack_fwd(n, nl, pl->modal_p, pl->modal_a, 16, ar->drop_layers, seeds, st));
  for (int l = 0; l < nl; ++l) seeds[l] = layer_seed(ar->step_seed, UID_CROSS + l);
  RUN(mmst_encoder_stack_fwd(1, nl, pl->cross_p, pl->cross_a, pl->cross_heads, ar->drop_layers, seeds, st));
  RUN(mmst_seq_mean(pl->memory, pl->mean, B, L, st));
  {
    GemmProb q = zero_prob();
    q.P = (const bf16*)pl->mean; q.ldp = D; q.Q = (const bf16*)pl->fp1_w; q.ldq = D;
    q.C = pl->fingerprint; q.ldc = pl->fingerprint_size; q.bias = pl->fp1_b;
    q.Pn = B; q.Qn = pl->fingerprint_size; q.K = D; q.flags = GF_OUT_F32;
    RUN(one_gemm(q, EPI_BIAS, false, false, st));
  }
  // decoder on the batch-first token matrix: input token of (t, b) = trg[b][t]
  RUN(launch_trg_embed_fwd(ar->trg, 1, ar->trg_ld, pl->embed_trg, pl->pe_trg, pl->trg_emb, B, T, ar->drop_embed,
                           layer_seed(ar->step_seed, UID_TRG), st, pl->loss));
  // the decoder reads the fragment-major weight images, which the previous step's optimizer refreshed on wst
  if (pl->n_frag > 0) RUN(mmst_stream_fork(wst, st));
  {
    mmst_dec_acts acts[MMST_MAX_LAYERS];
    const mmst_dec_acts* ap[MMST_MAX_LAYERS];
    for (int l = 0; l < ndl; ++l) {
      acts[l] = *pl->dec_a[l];
      acts[l].T = T; acts[l].kv_ready = 0; acts[l].no_causal = 0;
      ap[l] = &acts[l];
      seeds[l] = layer_seed(ar->step_seed, UID_DEC + l);
    }
    RUN(mmst_decoder_stack_fwd(ndl, pl->dec_p, ap, 16, ar->drop_layers, seeds, st, wst));
  }
  {
    GemmProb q = zero_prob();
    q.P = (const bf16*)pl->dec_a[ndl - 1]->y; q.ldp = D; q.Q = (const bf16*)pl->fc_out_w; q.ldq = D;
    q.C = pl->logits; q.ldc = VPAD; q.bias = pl->fc_out_b;
    q.Pn = B * T; q.Qn = VPAD; q.K = D; q.flags = GF_OUT_F32;
    RUN(one_gemm(q, EPI_BIAS, false, false, st));
  }
  return mmst_ce_loss(pl->logits, VPAD, pl->vocab, ar->trg, ar->trg_ld, B, T, pl->loss, pl->dlogits,
                      ar->loss_grad_scale, st);
}

// AdamW over the active ranges inside [lo, hi) (it also leaves that part of the gradient zeroed for the next step)
int adam_between(const mmst_step_plan* pl, const mmst_step_args* ar, long lo, long hi, hipStream_t s) {
  long rlo[MMST_MAX_RANGES], rhi[MMST_MAX_RANGES];
  int n = 0;
  for (int i = 0; i < pl->n_ranges; ++i) {
    const long a = pl->range_lo[i] > lo ? pl->range_lo[i] : lo, b = pl->range_hi[i] < hi ? pl->range_hi[i] : hi;
    if (a < b) { rlo[n] = a; rhi[n] = b; ++n; }
  }
  if (n == 0) return MMST_OK;
  return mmst_adamw_ranges(pl->P, pl->G, pl->M, pl->V, pl->Pb, n, rlo, rhi, ar->lr, ar->beta1, ar->beta2, ar->eps,
                           ar->weight_decay, ar->adam_step, ar->grad_scale, 1, s);
}
// Optimizer beside backward: once every launch that adds to the gradient above offset `from` has been enqueued (on st,
// wst or the library's side stream), that part [from, *done_from) is updated on the side stream, which waits for the
// other two up to here; nothing in the rest of the step reads those weights again.  Off: MMST_ADAM_EARLY=0, a plan
// without segment offsets, a step without optimizer.
bool adam_early(const mmst_step_plan* pl, const mmst_step_args* ar) {
  static const bool on = [] { const char* e = getenv("MMST_ADAM_EARLY"); return e ? atoi(e) != 0 : true; }();
  return on && ar->adam_step > 0 && pl->seg_decoder > 0 && pl->M && pl->V && mmst_side_stream() != nullptr;
}
int adam_segment(const mmst_step_plan* pl, const mmst_step_args* ar, long from, long* done_from, hipStream_t st, hipStream_t wst) {
  if (!adam_early(pl, ar) || from >= *done_from) return MMST_OK;
  hipStream_t xst = mmst_side_stream();
  RUN(mmst_stream_fork(st, xst));
  RUN(mmst_stream_fork(wst, xst));
  RUN(adam_between(pl, ar, from, *done_from, xst));
  *done_from = from;
  return MMST_OK;
}

int backward(const mmst_step_plan* pl, const mmst_step_args* ar, const mmst_embed_group* eg, const mmst_embed_shared* es,
             hipStream_t st, hipStream_t wst, long* adam_done_from) {
  const int B = pl->B, T = pl->T, L = pl->L, n = pl->n_groups, M = B * T;
  const int nl = pl->n_enc_layers, ndl = pl->n_dec_layers;
  const float dp = ar->drop_layers;
  const bool dr = dp > 0.f;
  // the transposed feed-forward weights were refreshed on wst after the previous optimizer step: the critical path
  // waits for that (nothing else is pending on wst), then the second stream waits for the forward pass
  RUN(mmst_stream_fork(wst, st));
  RUN(mmst_stream_fork(st, wst));
  // vocabulary projection backward fused with the last decoder layer's norm3 backward
  {
    const mmst_dec_acts* last = pl->dec_a[ndl - 1];
    const mmst_dec_params* lp = pl->dec_p[ndl - 1];
    const mmst_dec_grads* lg = pl->dec_g[ndl - 1];
    GemmProb q = zero_prob();
    q.P = (const bf16*)pl->dlogits; q.ldp = VPAD; q.Q = (const bf16*)pl->fc_out_w; q.ldq = D;
    q.C = (void*)pl->dec_w[ndl - 1]->dz3; q.ldc = D; q.C2 = dr ? (bf16*)pl->dec_dzd[ndl - 1] : nullptr;
    q.Z = (bf16*)last->z3; q.stats = last->st3; q.gamma = lp->g3; q.dgamma = lg->g3; q.dbeta = lg->be3;
    q.Pn = M; q.Qn = D; q.K = VPAD; q.drop_p = dp; q.seed = layer_seed(ar->step_seed, UID_DEC + ndl - 1) + 5;
    RUN(one_gemm(q, EPI_LNBWD, false, true, st));
    GemmProb g = zero_prob();   // d fc_out.weight / bias
    g.P = (const bf16*)pl->dlogits; g.ldp = VPAD; g.Q = (const bf16*)last->y; g.ldq = D;
    g.C = pl->d_fc_out_w; g.ldc = D; g.dbias = pl->d_fc_out_b;
    g.Pn = VPAD; g.Qn = D; g.K = M;
    g.ksplit = M / 256 < 1 ? 1 : (M / 256 > 16 ? 16 : M / 256);
    RUN(one_gemm(g, EPI_ATOMIC, true, true, wst));
  }
  {
    mmst_dec_acts acts[MMST_MAX_LAYERS];
    mmst_dec_bwd bw[MMST_MAX_LAYERS];
    const mmst_dec_acts* ap[MMST_MAX_LAYERS];
    const mmst_dec_bwd* wp[MMST_MAX_LAYERS];
    uint32_t seeds[MMST_MAX_LAYERS];
    for (int l = 0; l < ndl; ++l) {
      acts[l] = *pl->dec_a[l];
      acts[l].T = T; acts[l].kv_ready = 0; acts[l].no_causal = 0;
      bw[l] = *pl->dec_w[l];
      bw[l].dzd3 = dr ? pl->dec_dzd[l] : bw[l].dz3;
      seeds[l] = layer_seed(ar->step_seed, UID_DEC + l);
      bw[l].prev_seed = l > 0 ? layer_seed(ar->step_seed, UID_DEC + l - 1) : 0;
      ap[l] = &acts[l];
      wp[l] = &bw[l];
    }
    RUN(mmst_decoder_stack_bwd_streams(ndl, pl->dec_p, pl->dec_g, ap, wp, 16, dp, seeds, st, wst));
  }
  // the target-embedding gradients feed parameters only: second stream
  RUN(mmst_stream_fork(st, wst));
  RUN(launch_trg_embed_bwd(ar->trg, 1, ar->trg_ld, pl->dtrg, pl->d_embed_trg, pl->d_pe_trg, B, T, pl->in_size,
                           ar->drop_embed, layer_seed(ar->step_seed, UID_TRG), wst));
  if (ar->on_segment) ar->on_segment(MMST_SEG_DECODER, ar->user);
  RUN(adam_segment(pl, ar, pl->seg_decoder, adam_done_from, st, wst));
  // cross encoder
  {
    const mmst_enc_acts& la = pl->cross_a[nl - 1][0];
    const mmst_enc_bwd& lw = pl->cross_w[nl - 1][0];
    RUN(mmst_layernorm_bwd(B * L, pl->dmem, 0, 0, 0, la.z2, la.st2, pl->cross_p[nl - 1][0].g2, (void*)lw.dz2,
                           pl->cross_dzd[nl - 1], pl->cross_g[nl - 1][0].g2, pl->cross_g[nl - 1][0].be2, dp,
                           layer_seed(ar->step_seed, UID_CROSS + nl - 1) + 3, st));
  }
  for (int l = nl - 1; l >= 0; --l) {
    mmst_enc_bwd w = pl->cross_w[l][0];
    w.dzd2 = dr ? pl->cross_dzd[l] : w.dz2;
    w.prev_seed = l > 0 ? layer_seed(ar->step_seed, UID_CROSS + l - 1) : 0;
    RUN(mmst_encoder_layer_bwd_streams(1, pl->cross_p[l], pl->cross_g[l], pl->cross_a[l], &w, pl->cross_heads, dp,
                                       layer_seed(ar->step_seed, UID_CROSS + l), l < nl - 1 ? 2 : 0, st, wst));
  }
  if (ar->on_segment) ar->on_segment(MMST_SEG_CROSS, ar->user);
  RUN(adam_segment(pl, ar, pl->seg_cross, adam_done_from, st, wst));
  // modality encoders: the top LayerNorm backward of every group reads its slice of the memory gradient
  {
    GemmBatch gb;
    gb.n = n;
    for (int i = 0; i < n; ++i) {
      const mmst_enc_acts& a = pl->modal_a[nl - 1][i];
      const mmst_enc_bwd& w = pl->modal_w[nl - 1][i];
      GemmProb q = zero_prob();
      q.C = (void*)w.dz2; q.ldc = D; q.C2 = dr ? (bf16*)pl->modal_dzd[nl - 1][i] : nullptr;
      q.R = (const bf16*)pl->dmem_in; q.ldr = D; q.rS = a.S; q.rL = L; q.roff = a.y_off;
      q.Z = (bf16*)a.z2; q.stats = a.st2; q.gamma = pl->modal_p[nl - 1][i].g2;
      q.dgamma = pl->modal_g[nl - 1][i].g2; q.dbeta = pl->modal_g[nl - 1][i].be2;
      q.ldp = 8; q.ldq = 8; q.Pn = B * a.S; q.Qn = D; q.K = 0;
      q.drop_p = dp; q.seed = layer_seed(ar->step_seed, UID_MODAL + nl - 1) + 8 * i + 3;
      gb.p[i] = q;
    }
    RUN(launch_gemm(gb, EPI_LNBWD, false, false, st));
  }
  for (int l = nl - 1; l >= 0; --l) {
    mmst_enc_bwd w[MMST_MAX_GROUPS];
    for (int i = 0; i < n; ++i) {
      w[i] = pl->modal_w[l][i];
      w[i].dzd2 = dr ? pl->modal_dzd[l][i] : w[i].dz2;
      w[i].prev_seed = l > 0 ? layer_seed(ar->step_seed, UID_MODAL + l - 1) : 0;
    }
    RUN(mmst_encoder_layer_bwd_streams(n, pl->modal_p[l], pl->modal_g[l], pl->modal_a[l], w, 16, dp,
                                       layer_seed(ar->step_seed, UID_MODAL + l), l < nl - 1 ? 2 : 0, st, wst));
    if (ar->on_segment) ar->on_segment(MMST_SEG_MODAL0 + l, ar->user);
    if (l > 0) RUN(adam_segment(pl, ar, pl->seg_modal[l], adam_done_from, st, wst));   // depth 0 goes with the embeddings
  }
  // embeddings
  RUN(mmst_embed_bwd(n, eg, es, (const void* const*)pl->dxseq, st));
  if (pl->ir_group >= 0) {
    GemmProb g = zero_prob();
    g.P = (const bf16*)eg[pl->ir_group].ir_dpre; g.ldp = D; g.Q = (const bf16*)pl->ir_bf16; g.ldq = IR_BINS;
    g.C = pl->ir_dw; g.ldc = IR_BINS; g.dbias = pl->ir_db;
    g.Pn = D; g.Qn = IR_BINS; g.K = B; g.ksplit = 1;
    RUN(one_gemm(g, EPI_ATOMIC, true, true, st));
  }
  if (ar->on_segment) ar->on_segment(MMST_SEG_EMBED, ar->user);
  return mmst_stream_fork(wst, st);   // what follows on `stream` sees every gradient
}

int optimizer(const mmst_step_plan* pl, const mmst_step_args* ar, hipStream_t st, hipStream_t wst, long done_from) {
  if (!pl->M || !pl->V) return MMST_ERR_ARG;
  // one launch over every active range that has not been updated beside backward; it also leaves the gradient zeroed
  // for the next step.  `st` then waits for the side stream's updates.
  RUN(adam_between(pl, ar, 0, done_from, st));
  if (done_from < pl->range_hi[pl->n_ranges > 0 ? pl->n_ranges - 1 : 0]) RUN(mmst_stream_fork(mmst_side_stream(), st));
  // the transposed feed-forward weights are read by the backward pass only: refresh them on the second stream,
  // beside the next forward pass (the next backward waits for that stream first)
  RUN(mmst_stream_fork(st, wst));
  if (pl->n_ffn > 0) {
    RUN(mmst_transpose_bf16(pl->Pb, pl->PbT, pl->n_ffn, pl->ffn1_off, MMST_FF, D, wst));
    RUN(mmst_transpose_bf16(pl->Pb, pl->PbT, pl->n_ffn, pl->ffn2_off, D, MMST_FF, wst));
  }
  if (pl->n_frag > 0) RUN(mmst_fragment_bf16(pl->n_frag, pl->frag, wst));
  return MMST_OK;
}

}  // namespace

extern "C" {

int mmst_train_step(const mmst_step_plan* pl, const mmst_step_args* ar, void* stream_, void* wstream_) {
  if (!plan_ok(pl) || !ar || !ar->trg || ar->trg_ld < pl->T + 1 || !wstream_ || stream_ == wstream_) return MMST_ERR_ARG;
  if (ar->adam_step < 0 || !(ar->drop_layers >= 0.f && ar->drop_layers < 1.f) ||
      !(ar->drop_embed >= 0.f && ar->drop_embed < 1.f))
    return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_, wst = (hipStream_t)wstream_;
  // this batch's input pointers and mask semantics go into copies of the plan's embedding descriptors
  mmst_embed_group eg[MMST_MAX_GROUPS];
  mmst_embed_shared es = pl->embed_s;
  for (int i = 0; i < pl->n_groups; ++i) {
    eg[i] = pl->embed_g[i];
    if (i != pl->ir_group) {
      if (!ar->src[i] || !ar->mask[i]) return MMST_ERR_ARG;
      eg[i].src = ar->src[i];
      eg[i].mask = ar->mask[i];
    }
    eg[i].mod_bool = ar->mod_bool;
  }
  es.glob_bool = ar->glob_bool;
  if (es.has_mf) { if (!ar->mf_tok || !ar->mf_mask) return MMST_ERR_ARG; es.mf_tok = ar->mf_tok; es.mf_mask = ar->mf_mask; es.mf_bool = ar->mf_bool; }
  if (es.has_ms) { if (!ar->ms_tok || !ar->ms_mask) return MMST_ERR_ARG; es.ms_tok = ar->ms_tok; es.ms_mask = ar->ms_mask; es.ms_bool = ar->ms_bool; }
  if (es.has_mw) { if (!ar->mw) return MMST_ERR_ARG; es.mw = ar->mw; }
  if (ar->zero_grads)
    for (int i = 0; i < pl->n_ranges; ++i)
      if (hipMemsetAsync(pl->G + pl->range_lo[i], 0, sizeof(float) * (size_t)(pl->range_hi[i] - pl->range_lo[i]), st) !=
          hipSuccess)
        return MMST_ERR_LAUNCH;
  RUN(forward(pl, ar, eg, &es, st, wst));
  long adam_done_from = pl->n_ranges > 0 ? pl->range_hi[pl->n_ranges - 1] : 0;   // nothing updated yet
  RUN(backward(pl, ar, eg, &es, st, wst, &adam_done_from));
  if (ar->adam_step > 0) RUN(optimizer(pl, ar, st, wst, adam_done_from));
  return MMST_OK;
}

int mmst_sizeof_step_plan(void) { return (int)sizeof(mmst_step_plan); }
int mmst_sizeof_step_args(void) { return (int)sizeof(mmst_step_args); }

}  // extern "C"
