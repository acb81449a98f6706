/* libmmst.so -- C ABI of the MI355X (gfx950) spectral-transformer hot path.
 *
 * Every entry point: plain C, raw DEVICE pointers + sizes + a hipStream_t passed as void*,
 * returns 0 on success or a negative error code (never throws), allocates no device memory and
 * only ENQUEUES kernels on the given stream.  State the library keeps between calls, all of it
 * listed here: (1) a ring of 64 HIP events per device, created on the first two-stream call
 * (*_bwd_streams) and reused for the stream hand-offs; (2) the process-wide kernel-path switches
 * and the diagnostic event tap of mmst_lowlevel.h; (3) the per-device static scratch array of the
 * one *_static entry point; (4) one non-blocking HIP stream per device, created on first use by
 * mmst_decoder_stack_bwd_streams / mmst_train_step: the decoder's parameter gradients and the
 * optimizer updates of finished gradient segments run on it beside the caller's two streams, and it
 * is joined into them before those calls' work counts as complete on the caller's streams.  None of
 * the four is visible in results.
 * The building-block launchers underneath (grouped GEMM, attention, fused feed-forward) are
 * declared in mmst_lowlevel.h.
 *
 * The reference (a PyTorch nn.Module, no FFI of its own) is replaced at the level of the
 * torch.nn calls it makes; each function cites the reference call site it stands in for
 * (paths relative to the reference checkout, utils_MMT/models_MMT_v15_4.py unless noted).
 *
 * Conventions
 *   - activations: bf16, batch-first token rows, row (b, s) at index b*S + s, 128 columns
 *   - weights:     bf16 copies of the fp32 master parameters, torch layout [out][in]
 *   - biases / LayerNorm affine / gradients of parameters: fp32
 *   - parameter gradients are ACCUMULATED (atomically); the caller zeroes them per step
 *   - key-padding masks are additive fp32 biases [B][S] (0/1 reference-style, -inf, or any value)
 *   - dropout: counter-based hash of (seed, element index); a layer uses seeds seed .. seed+7
 *     for its dropout sites; backward must be given the same seed and drop_p as forward
 */
#ifndef MMST_H
#define MMST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMST_D 128        /* hidden_size (hard-wired in the reference's blank-modality branch, :852) */
#define MMST_FF 2048      /* torch default dim_feedforward (config.forward_expansion is never read) */
#define MMST_MAX_GROUPS 6 /* sequences (modalities) processed by one grouped launch */

/* ---- one post-norm encoder layer == nn.TransformerEncoderLayer(d_model=128, nhead) (:510-533) ---- */
typedef struct {
  const void* w_qkv; const float* b_qkv;   /* self_attn.in_proj_weight [384][128], in_proj_bias */
  const void* w_o;   const float* b_o;     /* self_attn.out_proj */
  const void* w_1;   const float* b_1;     /* linear1 [2048][128] */
  const void* w_2;   const float* b_2;     /* linear2 [128][2048] */
  const float* g1; const float* be1;       /* norm1 */
  const float* g2; const float* be2;       /* norm2 */
  const void* w_1t;  const void* w_2t;     /* bf16 transposes linear1^T [128][2048], linear2^T [2048][128]:
                                              read by the backward pass only (mmst_transpose_bf16 makes them) */
} mmst_enc_params;

typedef struct {  /* fp32 gradient buffers, same shapes as the parameters */
  float* w_qkv; float* b_qkv; float* w_o; float* b_o; float* w_1; float* b_1; float* w_2; float* b_2;
  float* g1; float* be1; float* g2; float* be2;
} mmst_enc_grads;

typedef struct {  /* activations of ONE layer for ONE sequence group */
  const void* x;        /* in : [B*S][128] */
  void* y;              /* out: [B*S][128] (or rows of a longer sequence, see y_L) */
  const float* kbias;   /* [B][S] additive key bias, or NULL */
  void* qkv;            /* [B*S][384] */
  void* attn;           /* [B*S][128] attention output before out_proj */
  float* lse;           /* [B][H][S] log2-sum-exp of the attention rows */
  void* z1; float* st1; /* pre-norm1 sum [B*S][128], (mean, rstd) [B*S][2] */
  void* x1;             /* norm1 output */
  void* h;              /* relu(linear1) after dropout, [B*S][2048] */
  void* z2; float* st2; /* pre-norm2 sum, (mean, rstd) */
  int B, S;
  int y_L, y_off;       /* if y_L > 0, output row (b, s) is written at row b*y_L + y_off + s of y */
  void* hsign;          /* optional [B*S][64] uint32: one bit per entry of h (entry > 0, layout in mmst_lowlevel.h).  Used
                           only under setting 3 of mmst_ffn_wgrad_config: _fwd writes it and the feed-forward backward
                           reads these 256 B per row in place of the 4 KiB of h (the weight gradients still read h) */
} mmst_enc_acts;

typedef struct {  /* backward state of ONE layer for ONE sequence group */
  const void* dz2;  /* in : grad at this layer's pre-norm2 sum [B*S][128] */
  const void* dzd2; /* in : the same through the residual-dropout mask (== dz2 when drop_p == 0) */
  void* dh; void* dz1; void* dzd1; void* da; void* dqkv; float* delta; /* scratch, sizes as forward */
  void* dx;         /* out: grad w.r.t. x if prev_z2 == NULL, else grad at the PREVIOUS layer's pre-norm2 sum */
  void* dxd;        /* out: previous layer's dzd2 (only when prev_z2 != NULL and drop_p > 0) */
  const void* prev_z2; const float* prev_st2; const float* prev_g2; /* previous layer's norm2, or NULL */
  float* prev_dg2; float* prev_dbe2;
  uint32_t prev_seed; /* seed the previous layer was run with */
} mmst_enc_bwd;

/* x -> y for n sequence groups with independent parameters (the per-modality encoders run as one
 * grouped launch sequence; encoder_cross is n = 1 with nhead = 4).  Replaces
 * nn.TransformerEncoderLayer.forward as called from :846,:866,:886,:906,:927,:944. */
int mmst_encoder_layer_fwd(int n, const mmst_enc_params* p, const mmst_enc_acts* a, int nhead, float drop_p,
                           uint32_t seed, void* stream);
/* nl layers of ONE encoder stack in sequence (layer l + 1 reads what layer l wrote; p[l] / a[l] as for
 * mmst_encoder_layer_fwd, seeds[l] = that layer's dropout seed): the loop over self.layers of nn.TransformerEncoder
 * (models_MMT_v15_4.py:846,:866,:886,:906,:927,:944).  Same results as nl calls of mmst_encoder_layer_fwd; a layer's QKV
 * projection is chained behind the feed-forward kernel of the layer in front where that kernel can do it. */
int mmst_encoder_stack_fwd(int n, int nl, const mmst_enc_params* const* p, const mmst_enc_acts* const* a, int nhead,
                           float drop_p, const uint32_t* seeds, void* stream);
/* Backward of the above (what autograd does for the reference) = _data followed by _weights. */
int mmst_encoder_layer_bwd(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                           const mmst_enc_bwd* w, int nhead, float drop_p, uint32_t seed, void* stream);
/* The two halves separately, so that a caller can put the parameter gradients on a second stream:
 * _data    = critical path (activation gradients, LayerNorm affine gradients), leaves dh/dzd1/dqkv in w;
 * _weights = every weight/bias gradient of the layer as one grouped split-K launch reading them.
 * w's scratch must stay untouched until _weights has run. */
int mmst_encoder_layer_bwd_data(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                                const mmst_enc_bwd* w, int nhead, float drop_p, uint32_t seed, void* stream);
int mmst_encoder_layer_bwd_weights(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                                   const mmst_enc_bwd* w, float drop_p, void* stream);
/* Both halves on two streams in one call; wstream is made to wait for stream where needed (on the completion event
 * of the kernel it depends on, not on a marker), the caller joins the streams after the last layer.
 * interleave bit 0 clear: all parameter gradients on wstream after the data phase; with recomputed feed-forward
 * gradients (mmst_ffn_wgrad_config) those go first, at the start of the layer, since they read nothing the layer's
 * backward produces.  Bit 0 set: the feed-forward and out_proj weight gradients are enqueued as soon as the
 * feed-forward backward is (they then run beside the attention backward of the same layer), the in_proj gradient after
 * the rest of _data.  Bit 1 set: wstream is already ordered behind the kernel that produced w->dzd2 (true when the
 * previous call on these streams was this function for the layer above), so no marker is recorded on stream for the
 * early start.  Same results as _bwd. */
int mmst_encoder_layer_bwd_streams(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                                   const mmst_enc_bwd* w, int nhead, float drop_p, uint32_t seed, int interleave,
                                   void* stream, void* wstream);

/* ---- one post-norm decoder layer == nn.TransformerDecoderLayer(d_model=128, nhead=16) (:539-541) ---- */
typedef struct {
  const void* w_qkv; const float* b_qkv; const void* w_o; const float* b_o;      /* self_attn */
  const void* w_cqkv; const float* b_cqkv; const void* w_co; const float* b_co;  /* multihead_attn */
  const void* w_1; const float* b_1; const void* w_2; const float* b_2;
  const float* g1; const float* be1; const float* g2; const float* be2; const float* g3; const float* be3;
  const void* w_1t; const void* w_2t;   /* bf16 transposes of linear1 / linear2, backward pass only */
  /* Optional fragment-major images (mmst_fragment_bf16) of the six weight matrices, f_cqkv of all 384 rows, and of their
   * transposes (ft_cqkv: of the first 128 rows, the query projection): with all of f_* present mmst_decoder_stack_fwd may
   * take its two-launches-per-layer form, with all of ft_* the backward entry points theirs; NULL = never. */
  const void* f_qkv; const void* f_o; const void* f_cqkv; const void* f_co; const void* f_1; const void* f_2;
  const void* ft_qkv; const void* ft_o; const void* ft_cqkv; const void* ft_co; const void* ft_1; const void* ft_2;
} mmst_dec_params;

typedef struct {
  float* w_qkv; float* b_qkv; float* w_o; float* b_o; float* w_cqkv; float* b_cqkv; float* w_co; float* b_co;
  float* w_1; float* b_1; float* w_2; float* b_2; float* g1; float* be1; float* g2; float* be2; float* g3; float* be3;
} mmst_dec_grads;

typedef struct {
  const void* x; void* y;             /* [B*T][128] */
  const void* mem;                    /* encoder memory [B*L][128] */
  const float* mem_bias;              /* [B][L] additive, or NULL */
  void* qkv; void* attn1; float* lse1; void* z1; float* st1; void* x1;
  void* qc; void* kvc; void* attn2; float* lse2; void* z2; float* st2; void* x2; /* qc [B*T][128], kvc [B*L][256] */
  void* h; void* z3; float* st3;
  int B, T, L;
  int kv_ready; /* nonzero: kvc already holds this layer's K/V projection of mem (decode loops reuse it) */
  float* ffn_ws; /* optional fp32 workspace [8][B*T][128]: lets the feed-forward block of a SHORT sequence split its
                    hidden dimension over up to 8 workgroups per row tile (NULL: no split) */
  int no_causal; /* nonzero: self-attention without the causal mask (nn.TransformerDecoder called with tgt_mask=None);
                    0, the default, is the reference's generate_square_subsequent_mask (:794-800, :962) */
} mmst_dec_acts;

/* ---- one decoder layer for ONE new target position with cached keys/values (inference) ----
 * The reference's decode loops (validate_generate_MMT_v15_4.py:744-766, models_MMT_v15_4.py:1238-1262) re-run the
 * decoder on the whole prefix for every token; rows of earlier positions do not change under the causal mask, so
 * this entry point computes only the row of position `pos`: its Q|K|V projection is appended to qkv_cache, self-
 * attention reads keys/values 0..pos from the cache, cross-attention reads kvc (projected once per sequence).
 * Same arithmetic per row as mmst_decoder_layer_fwd without dropout. */
typedef struct {
  const void* x; void* y;      /* [B][128] input / output row of this position */
  void* qkv_cache;             /* [B][cap][384] Q|K|V rows of positions 0..pos (row pos is written by the call) */
  const void* kvc;             /* [B*L][256] cross-attention K|V of the encoder memory */
  const float* mem_bias;       /* [B][L] additive key bias, or NULL */
  void* attn; void* x1; void* qc; void* x2;   /* scratch, [B][128] each */
  void* h;                     /* scratch [B][2048] */
  float* lse;                  /* scratch [B][16] */
  float* ffn_ws;               /* optional fp32 [8][B][128], see mmst_dec_acts.ffn_ws */
  int B, pos, cap, L;
} mmst_dec_step;
int mmst_decoder_layer_step(const mmst_dec_params* p, const mmst_dec_step* s, int nhead, void* stream);

/* The same token step for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): mmst_decoder_layer_step with its scratch rows (attn, x1, qc, x2, h, lse,
 * ffn_ws of mmst_dec_step) taken from a static device array inside the library instead of from the caller.  It has NO
 * counterpart in the reference; it exists for those callers only.  The caller still passes the input row, the output
 * row and the caches.  Limits: B <= MMST_STATIC_STEP_MAX_B (larger B returns -1, nothing is launched); the array
 * (about 2.3 MB) exists once per GPU for the life of the process and is shared by every call on that GPU, so two calls
 * that overlap on one GPU (different streams or threads) race -- serialise them.  The call itself allocates nothing
 * and frees nothing; its results equal mmst_decoder_layer_step's bit for bit.  The FIRST call on a device looks the
 * array's address up (hipGetSymbolAddress, which may load the code object): make one eager call per device before
 * capturing the entry point into a HIP graph. */
#define MMST_STATIC_STEP_MAX_B 256
int mmst_decoder_layer_step_static(const mmst_dec_params* p, const void* x, void* y, void* qkv_cache, const void* kvc,
                                   const float* mem_bias, int B, int pos, int cap, int L, int nhead, void* stream);

/* ---- one new target position through ALL decoder layers in one launch (inference) ----
 * Same inputs, caches and arithmetic as n_layers calls of mmst_decoder_layer_step (one workgroup per sequence walks
 * the layers; matrix-vector products instead of GEMM tiles), optionally with the steps around them that the decode
 * loops of the reference run per token (validate_generate_MMT_v15_4.py:747-762, models_MMT_v15_4.py:1240-1257):
 * the embedding of the input token (tok != NULL), the vocabulary projection (logits != NULL) and the greedy choice
 * softmax(logits / temperature) -> argmax, probability (next_tok != NULL).  p = array of n_layers parameter sets. */
#define MMST_MAX_DEC_LAYERS 8
typedef struct {
  const void* x;               /* [B][128] bf16 input rows of this position (ignored when tok != NULL) */
  const int64_t* tok;          /* optional [B] token ids: input row = bf16(embed[tok] + pos_row) */
  const float* embed;          /* fp32 [vocabulary][128] */
  const float* pos_row;        /* fp32 [128]: positional-embedding row of `pos` */
  void* y;                     /* optional [B][128] bf16: output row of the last layer */
  void* qkv_cache[MMST_MAX_DEC_LAYERS];   /* per layer, as mmst_dec_step.qkv_cache */
  const void* kvc[MMST_MAX_DEC_LAYERS];   /* per layer, as mmst_dec_step.kvc ([B / kv_group][L][256]) */
  const float* mem_bias;       /* [B / kv_group][L] additive key bias, or NULL */
  const void* w_out; const float* b_out;  /* optional vocabulary projection, bf16 [n_out][128] / fp32 [n_out] */
  float* logits; int ld_logits; int n_out; /* out: logits[b*ld_logits + 0..n_out) fp32, n_out <= 64 */
  int64_t* next_tok; float* next_prob;    /* optional [B]: choice among logits[0..V) and its probability: greedy, or
                                             (sample != 0) a draw as mmst_sample_select makes it with counter = pos */
  float temperature; int V;
  int B, pos, cap, L;
  int kv_group;                /* >= 1: sequences b*kv_group .. (b+1)*kv_group-1 share row block b of kvc and mem_bias
                                  (the beams of one molecule); 0 is read as 1.  Groups of 2..4 sequences are served
                                  by ONE workgroup: weights and the molecule's memory keys/values are read once for
                                  all of them */
  int rows_per_group;          /* 1: one workgroup per sequence even for kv_group 2..4 (the reference arrangement the
                                  grouped kernel is tested against); 0: as described under kv_group */
  const int32_t* hist;         /* optional [B][cap] (kv_group 1..4): self-attention reads the key/value of position
                                  j < pos of sequence b from the cache of sequence (b / kv_group) * kv_group + hist[b][j],
                                  so a beam that descends from another beam's prefix needs no copy of that prefix's
                                  cache rows; position pos is always the sequence's own new row */
  void* ffn_ws;                /* optional workspace of MMST_DEC_STACK_WS_BYTES(B) bytes, 256-byte aligned: with it the
                                  feed-forward block of every layer runs as a launch of its own
                                  over all sequences, which reads its weights once per call instead of once per
                                  workgroup; same arithmetic, the hidden dimension summed in 16 slices */
  int sample; uint32_t sample_seed;   /* sample != 0: next_tok is drawn from softmax(logits / temperature) instead of
                                         being its argmax (seed, counter = pos, row = sequence) */
} mmst_dec_stack_step;
/* Process-wide: the least number of sequences (B) from which a call with ffn_ws takes the split form; 0 = never,
 * negative = the default (1, or what the environment variable MMST_DECODE_SPLIT_FFN says).  No counterpart in the
 * reference: a tuning switch, used by the tests to run both forms on small batches. */
int mmst_decoder_stack_config(int split_min_sequences);
#define MMST_DEC_STACK_WS_BYTES(B) ((((size_t)(B) * 128 * 2 + 255) & ~(size_t)255) + (size_t)16 * (B) * 128 * 4)
int mmst_decoder_stack_step(int n_layers, const mmst_dec_params* p, const mmst_dec_stack_step* s, int nhead, void* stream);

/* ---- beam search bookkeeping on the device (validate_generate_MMT_v15_4.py:995-1094) ----
 * State of W beams for each of B molecules; two such states alternate between steps.  A step reads the logits the
 * decoder produced for the B*W sequences at position `pos` (mmst_decoder_stack_step with kv_group = W and hist = the
 * CURRENT state's hist), expands every unfinished beam by its W most probable tokens of softmax(logits) (no
 * temperature, :1038), carries finished beams over, multiplies scores in double precision (the reference's Python
 * floats), keeps the first W of a stable descending sort (:1052) and writes the NEXT state plus the tokens the
 * decoder consumes at position pos + 1. */
#define MMST_MAX_BEAM 4
typedef struct {
  int32_t* seqs;     /* [B][W][cap + 1] token ids incl. <SOS> */
  float* probs;      /* [B][W][cap] probability of every chosen token */
  double* scores;    /* [B][W] product of the probabilities */
  int32_t* lens;     /* [B][W] tokens in seqs (incl. <SOS>) */
  int32_t* nbeams;   /* [B] beams that exist (1 before the first step, W afterwards) */
  int32_t* hist;     /* [B*W][cap] see mmst_dec_stack_step.hist */
} mmst_beam_state;
/* one beam (<SOS>) per molecule, tok[b*W + w] = sos for every slot */
int mmst_beam_init(const mmst_beam_state* st, int64_t* tok, int B, int W, int cap, int sos, void* stream);
int mmst_beam_step(const float* logits, int ld_logits, int V, const mmst_beam_state* cur, const mmst_beam_state* next,
                   int64_t* tok, int B, int W, int cap, int pos, int eos, void* stream);

typedef struct {
  const void* dz3; const void* dzd3;
  void* dh; void* dz2; void* dzd2; void* da; void* dqc; void* dkvc; void* dz1; void* dzd1; void* dqkv; float* delta;
  void* da2;        /* scratch [B*T][128]: gradient at the cross-attention output */
  void* dmem;       /* [B*L][128] grad w.r.t. memory */
  int dmem_accumulate; /* 0: overwrite dmem, 1: add to it (layers after the first one processed) */
  void* dx; void* dxd;
  const void* prev_z3; const float* prev_st3; const float* prev_g3; float* prev_dg3; float* prev_dbe3;
  uint32_t prev_seed;
  float* delta2;    /* optional scratch like delta, for the cross-attention: required by _bwd_streams, where the
                       memory's dK/dV pass runs beside the self-attention backward that overwrites delta */
} mmst_dec_bwd;

/* Replaces nn.TransformerDecoderLayer.forward as called from :963 (causal tgt_mask built in-kernel,
 * memory_key_padding_mask = mem_bias) and, with T growing by one per step, from the decode loops
 * (models_MMT_v15_4.py:1244, validate_generate_MMT_v15_4.py:751). */
int mmst_decoder_layer_fwd(const mmst_dec_params* p, const mmst_dec_acts* a, int nhead, float drop_p, uint32_t seed,
                           void* stream);
/* nl decoder layers in sequence (layer l + 1 reads what layer l wrote; p[l] / a[l] / seeds[l] as for
 * mmst_decoder_layer_fwd): the loop over self.layers of nn.TransformerDecoder behind :963.  Same saved activations and
 * dropout masks as nl calls of mmst_decoder_layer_fwd; for target sequences of at most 64 positions under the causal
 * mask a layer takes two launches instead of eight (a workgroup per molecule keeps the molecule's rows in LDS).
 * wstream (optional, may be NULL): a second stream for the K|V projections of the memory, joined into stream. */
int mmst_decoder_stack_fwd(int nl, const mmst_dec_params* const* p, const mmst_dec_acts* const* a, int nhead, float drop_p,
                           const uint32_t* seeds, void* stream, void* wstream);
int mmst_decoder_layer_bwd(const mmst_dec_params* p, const mmst_dec_grads* g, const mmst_dec_acts* a,
                           const mmst_dec_bwd* w, int nhead, float drop_p, uint32_t seed, void* stream);
int mmst_decoder_layer_bwd_data(const mmst_dec_params* p, const mmst_dec_grads* g, const mmst_dec_acts* a,
                                const mmst_dec_bwd* w, int nhead, float drop_p, uint32_t seed, void* stream);
int mmst_decoder_layer_bwd_weights(const mmst_dec_grads* g, const mmst_dec_acts* a, const mmst_dec_bwd* w,
                                   float drop_p, void* stream);
/* _data and _weights on two streams in one call: `stream` carries only what the next layer waits for; the dK/dV of
 * the encoder memory, their projection into w->dmem and all parameter gradients go to `wstream` (made to wait for
 * `stream` where needed).  join_dmem != 0: `stream` finally waits for the dmem update.  Same results as _bwd. */
int mmst_decoder_layer_bwd_streams(const mmst_dec_params* p, const mmst_dec_grads* g, const mmst_dec_acts* a,
                                   const mmst_dec_bwd* w, int nhead, float drop_p, uint32_t seed, int join_dmem,
                                   void* stream, void* wstream);

/* The whole decoder backward on two streams, layers nl - 1 .. 0: what nl calls of mmst_decoder_layer_bwd_streams do
 * from the top layer down (join_dmem set for layer 0), with w[l]->dx = w[l - 1]->dz3 (and dxd = dzd3), w[l]->prev_* =
 * layer l - 1's norm3 and w[l]->prev_seed = seeds[l - 1].  For target sequences of at most 64 positions under the causal
 * mask, with the fragment-major transposed weight images (mmst_dec_params.ft_*) present, the activation gradients of a
 * layer take five launches instead of nine; same results to bf16 rounding (what autograd does for :963). */
int mmst_decoder_stack_bwd_streams(int nl, const mmst_dec_params* const* p, const mmst_dec_grads* const* g,
                                   const mmst_dec_acts* const* a, const mmst_dec_bwd* const* w, int nhead, float drop_p,
                                   const uint32_t* seeds, void* stream, void* wstream);

/* Standalone LayerNorm backward (top of a layer stack): dy -> dz / dzd, accumulates dgamma, dbeta.
 * dy row (b, s) is read at row b*dy_L + dy_off + s when dy_L > 0 (a slice of the concatenated memory). */
int mmst_layernorm_bwd(int rows, const void* dy, int dy_S, int dy_L, int dy_off, const void* z, const float* stats,
                       const float* gamma, void* dz, void* dzd, float* dgamma, float* dbeta, float drop_p,
                       uint32_t seed, void* stream);

/* ---- embeddings: replaces _embed_spectrum_* / _embed_MF / _embed_MS / _embed_MW and
 * _create_embeddings_and_masks_* (:549-792): writes each modality's [B][S][128] input sequence and its
 * additive key bias, plus the modality's slice of the global (cross-encoder) key bias. ---- */
typedef struct {
  void* xseq;            /* out [B][S][128] bf16 */
  float* kbias;          /* out [B][S] */
  float* kbias_glob;     /* out: row b at kbias_glob + b*L_glob + off_glob */
  const float* src;      /* spectrum [B][64][in_dim] fp32 (NULL for IR: its single row comes from a GEMM) */
  const float* mask;     /* [B][64] fp32, nonzero = padded */
  const float* w; const float* b;   /* point embedding Linear(in_dim -> 128), fp32 */
  float* dw; float* db;  /* gradients (backward only) */
  void* ir_dpre;         /* IR only, backward: out [B][128] bf16 = grad before the ReLU of the IR row */
  int in_dim;            /* 2 (1H, HSQC, COSY), 1 (13C), 0 (IR) */
  int n_spec;            /* 64 spectrum rows, or 1 for IR */
  int S, L_glob, off_glob;
  int mod_bool;          /* 1: this modality's concatenated mask is all-bool (-inf semantics) */
} mmst_embed_group;

typedef struct {
  const int64_t* mf_tok; const float* mf_mask; const float* mf_table; float* d_mf_table; int has_mf; int mf_bool;
  const int64_t* ms_tok; const float* ms_mask; const float* ms_table; float* d_ms_table; int has_ms; int ms_bool;
  const float* mw; const float* mw_w; const float* mw_b; float* d_mw_w; float* d_mw_b; int has_mw;
  int B; int glob_bool; /* glob_bool: the concatenated global mask is all-bool */
} mmst_embed_shared;

int mmst_embed_fwd(int n, const mmst_embed_group* g, const mmst_embed_shared* s, void* stream);
/* dxseq[i]: grad w.r.t. group i's input sequence ([B][S][128] bf16) */
int mmst_embed_bwd(int n, const mmst_embed_group* g, const mmst_embed_shared* s, const void* const* dxseq,
                   void* stream);

/* mean over the sequence of memory [B][L][128] -> [B][128] bf16 (torch.mean(memory, dim=0), :946) */
int mmst_seq_mean(const void* mem, void* out, int B, int L, void* stream);

/* target embedding: out[b][t] = dropout(embed_trg[tok[t][b]] + pe_trg[t]) (:955-961); tok is (T,B) int64 with
 * row stride tok_ld */
int mmst_trg_embed_fwd(const int64_t* tok, int tok_ld, const float* embed, const float* pos, void* out, int B, int T,
                       float drop_p, uint32_t seed, void* stream);
/* V = rows of the token table; gradients are ADDED to d_embed / d_pos (one workgroup owns each row). */
int mmst_trg_embed_bwd(const int64_t* tok, int tok_ld, const void* dout, float* d_embed, float* d_pos, int B, int T,
                       int V, float drop_p, uint32_t seed, void* stream);

/* CrossEntropyLoss() over all B*T positions (:70, :268-271): logits fp32 [B*T][ldl] (first V columns valid),
 * target for row (b, t) = trg[b*trg_ld + t + 1].  Adds the mean loss to *loss and writes dlogits (bf16,
 * [B*T][64], zero beyond V), scaled by grad_scale / (B*T). */
int mmst_ce_loss(const float* logits, int ldl, int V, const int64_t* trg, int trg_ld, int B, int T, float* loss,
                 void* dlogits, float grad_scale, void* stream);

/* config.use_real_data branch of MultimodalTransformer.forward (:965-971): logits <- (logits + extra[b]) / 2 with
 * extra = real_data_linear(mean(memory)) [B][64] fp32 (a GEMM on the 64-row padded weight); logits fp32 [B*T][64]. */
int mmst_real_data_mix(float* logits, const float* extra, int B, int T, void* stream);
/* Row part of its backward: d_extra[b][c] = sum_t dlogits[b][t][c] (dlogits bf16 [B*T][64], already scaled by 1/2
 * through mmst_ce_loss's grad_scale), dbias[c] += sum_b d_extra[b][c], mean32 = fp32 copy of mean [B][128].  The
 * weight gradient and d_mean are mmst_sgemm products of d_extra; d_mean reaches the memory through
 * mmst_seq_mean_bwd_add: dmem[b][l][:] += dmean[b][:] / L (bf16 dmem). */
int mmst_real_data_rows_bwd(const void* dlogits, const void* mean, float* d_extra, float* mean32, float* dbias, int B,
                            int T, void* stream);
int mmst_seq_mean_bwd_add(const float* dmean, void* dmem, int B, int L, void* stream);

/* One decode step's token selection (softmax(logits/temperature), argmax, its probability) on the LAST
 * position: validate_generate_MMT_v15_4.py:754-762 / models_MMT_v15_4.py:1246-1256.  logits [B][T][ldl]. */
int mmst_greedy_select(const float* logits, int ldl, int V, int B, int T, float temperature, int64_t* next_tok,
                       float* next_prob, void* stream);
/* The multinomial decode loops' selection, torch.multinomial(softmax(logits / temperature), 1) and the gather of its
 * probability (validate_generate_MMT_v15_4.py:841-880, mmt_result_test_functions_15_4.py:791-829,
 * models_MMT_v15_4.py:1258-1264), without a generator state on the device: row b draws the inverse CDF at a uniform
 * number that is a pure function of (seed, counter, b), so a caller that passes the target position as the counter
 * gets independent draws per position and row and the same tokens again for the same seed.  Same distribution as
 * torch's, not the same stream of numbers.  Never returns an index of probability zero. */
int mmst_sample_select(const float* logits, int ldl, int V, int B, int T, float temperature, uint32_t seed,
                       uint32_t counter, int64_t* next_tok, float* next_prob, void* stream);

/* torch.optim.AdamW step (:1213-1214) on flat fp32 buffers + refresh of the bf16 weight copy. */
int mmst_adamw(float* param, const float* grad, float* m, float* v, void* param_bf16, long n, float lr, float beta1,
               float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* The same update over n_ranges (<= 16) element ranges [lo[i], hi[i]) of the flat buffers in one launch (offsets and
 * lengths multiples of 4); zero_grad != 0 leaves the gradient of those ranges zeroed for the next step. */
int mmst_adamw_ranges(float* param, float* grad, float* m, float* v, void* param_bf16, int n_ranges, const long* lo,
                      const long* hi, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      float grad_scale, int zero_grad, void* stream);
/* fp32 -> bf16 copy of a [rows][cols] matrix into a [rows][ld_out] one (padded weight copies). */
int mmst_cast_bf16(const float* src, void* dst, int rows, int cols, int ld_out, void* stream);

/* bf16 transposes of n equally shaped [rows][cols] matrices: matrix i is read at src + offsets[i] and written as
 * [cols][rows] at dst + offsets[i] (element offsets, host array, n <= 128).  Keeps the transposed feed-forward
 * weights the backward pass reads in step with the bf16 weight copy after an optimizer step. */
int mmst_transpose_bf16(const void* src, void* dst, int n, const uint32_t* offsets, int rows, int cols, void* stream);

/* Fragment-major image of a bf16 matrix A[rows][cols] (rows a multiple of 16, cols of 32) for kernels that take their
 * v_mfma_f32_16x16x32_bf16 weight operand straight from memory: tile (i, j) = rows 16 i .., columns 32 j .. is stored as
 * 64 lanes x 8 elements, lane (r = lane & 15, g = lane >> 4) = A[16 i + r][32 j + 8 g .. + 7], tiles in row-major order,
 * so that one wave instruction reads one contiguous KiB.  transpose == 0: A = src (row stride ld); otherwise A is the
 * transpose of src, A[i][k] = src[k * ld + i].  Keeps the images the decoder kernels read in step with the bf16 weight
 * copy after an optimizer step (no counterpart in the reference: a layout copy like mmst_transpose_bf16). */
typedef struct { const void* src; void* dst; int rows, cols, ld, transpose; } mmst_frag_desc;
#define MMST_FRAG_MAX 32   /* descriptors per launch; longer lists take several launches */
int mmst_fragment_bf16(int n, const mmst_frag_desc* descs, void* stream);

/* ---- CLIP contrastive head on the encoder half (utils_MMT/models_CLIP_v15_4.py), all fp32 ---- */

/* C[M][N] = alpha * sum_k A(i,k) B(k,j) + beta * C + bias[j], with A(i,k) = A[i*sai + k*sak] and
 * B(k,j) = B[k*sbk + j*sbj]: by choice of strides the nn.Linear forward of ProjectionHead (:413-415, :420-422),
 * its dX / dW products, and the B x B similarity products of CLIPModel.forward (:293-295) and their backward. */
int mmst_sgemm(const float* A, long sai, long sak, const float* B, long sbk, long sbj, float* C, int ldc,
               const float* bias, int M, int N, int K, float alpha, float beta, void* stream);
/* nn.GELU() (:414, :421), exact erf form; backward ADDS dy * gelu'(x) to dx. */
int mmst_gelu_fwd(const float* x, float* y, long n, void* stream);
int mmst_gelu_bwd(const float* x, const float* dy, float* dx, long n, void* stream);
/* Tail of ProjectionHead.forward (:423-425): out = LayerNorm(dropout(f) + projected), rows x P (P <= 1024);
 * keeps the normalised rows and 1/std for the backward, which returns df (dropout applied), dproj (the
 * residual branch) and ADDS dgamma / dbeta. */
int mmst_head_out_fwd(const float* f, const float* projected, const float* gamma, const float* beta, float* out,
                      float* xhat, float* rstd, int rows, int P, float drop_p, uint32_t seed, void* stream);
int mmst_head_out_bwd(const float* dout, const float* xhat, const float* rstd, const float* gamma, float* df, float* dproj,
                      float* dgamma, float* dbeta, int rows, int P, float drop_p, uint32_t seed, void* stream);
/* Symmetric soft-target cross entropy (:296-302 with cross_entropy :390-396) and its gradient: logits [B][B],
 * Z = (MT_similarity + CB_similarity) / 2 * temperature [B][B]; targets = softmax(Z) are NOT detached in the
 * reference, so dZ is returned next to dlogits (both for L = mean of the per-sample losses).  ws: scratch of
 * 5*B*B + 3*B floats.  losses [B]; the mean is ADDED to *loss_mean. */
int mmst_clip_loss(const float* logits, const float* Z, int B, float* ws, float* losses, float* loss_mean, float* dlogits,
                   float* dZ, void* stream);
/* out = X + X^T for a [B][B] matrix (gradient of X X^T products). */
int mmst_symmetrize(const float* X, float* out, int B, void* stream);
/* backward of torch.mean(memory, dim=0) (models_MMT_v15_4.py:946): dmem[b][l][:] = dmean[b][:] / L, bf16 out */
int mmst_seq_mean_bwd(const float* dmean, void* dmem, int B, int L, void* stream);

/* ---- one whole training step in ONE call ----
 * What TransformerMultiGPU.training_step + the optimizer do for one batch (models_MMT_v15_4.py:1051-1118, :1213-1214)
 * with gen_SMI_sequence off: embeddings, modality encoders, cross encoder, fingerprint head, target embedding,
 * decoder, vocabulary projection, CrossEntropyLoss, the whole backward pass on two streams, AdamW and the refresh of
 * the bf16 / transposed weight copies.  The launch sequence is exactly the one a caller gets by calling the entry
 * points above layer by layer; this entry point exists because ~330 launches per step issued one by one through a
 * foreign-function interface cost more host time than the GPU needs to run them.
 *
 * mmst_step_plan describes the buffers of one (batch size, target length, training_mode) and is built once;
 * mmst_step_args carries what changes per step (input pointers, seeds, optimizer scalars). */
#define MMST_MAX_LAYERS 8
#define MMST_MAX_RANGES 16
#define MMST_MAX_FFN 128
#define MMST_MAX_FRAG 96
enum { MMST_SEG_DECODER = 0, MMST_SEG_CROSS = 1, MMST_SEG_EMBED = 2, MMST_SEG_MODAL0 = 16 /* + encoder depth */ };

typedef struct {
  int B, T, L;                       /* molecules, target positions (tokens - 1), memory rows per molecule */
  int n_groups, n_enc_layers, n_dec_layers, cross_heads;
  int vocab, in_size, fingerprint_size;
  /* embeddings: groups / shared as for mmst_embed_fwd, input pointers (src, mask, *_tok, *_mask, mw) left NULL */
  mmst_embed_group embed_g[MMST_MAX_GROUPS];
  mmst_embed_shared embed_s;
  int ir_group;                      /* index of the IR group, or -1 */
  void* ir_bf16;                     /* [B][1000] bf16 staging of the IR spectrum */
  const void* ir_w; const float* ir_b; float* ir_dw; float* ir_db;   /* Linear(1000 -> 128) */
  void* dxseq[MMST_MAX_GROUPS];      /* gradient at each group's input sequence */
  /* encoder / decoder layers: arrays as for the layer entry points (modal_*[l] has n_groups elements) */
  const mmst_enc_params* modal_p[MMST_MAX_LAYERS]; const mmst_enc_grads* modal_g[MMST_MAX_LAYERS];
  const mmst_enc_acts* modal_a[MMST_MAX_LAYERS]; const mmst_enc_bwd* modal_w[MMST_MAX_LAYERS];
  const mmst_enc_params* cross_p[MMST_MAX_LAYERS]; const mmst_enc_grads* cross_g[MMST_MAX_LAYERS];
  const mmst_enc_acts* cross_a[MMST_MAX_LAYERS]; const mmst_enc_bwd* cross_w[MMST_MAX_LAYERS];
  const mmst_dec_params* dec_p[MMST_MAX_LAYERS]; const mmst_dec_grads* dec_g[MMST_MAX_LAYERS];
  const mmst_dec_acts* dec_a[MMST_MAX_LAYERS]; const mmst_dec_bwd* dec_w[MMST_MAX_LAYERS];
  /* in the *_w structs dz2 / dz3 is the layer's incoming gradient; *_dzd[l][i] is the buffer that holds the same
   * gradient through the residual-dropout mask (used as dzd2 / dzd3 when dropout is on) */
  void* modal_dzd[MMST_MAX_LAYERS][MMST_MAX_GROUPS]; void* cross_dzd[MMST_MAX_LAYERS]; void* dec_dzd[MMST_MAX_LAYERS];
  /* heads */
  const void* memory; void* mean; const void* fp1_w; const float* fp1_b; float* fingerprint;
  void* trg_emb; const float* embed_trg; const float* pe_trg; float* d_embed_trg; float* d_pe_trg;
  const void* fc_out_w; const float* fc_out_b; float* d_fc_out_w; float* d_fc_out_b;   /* 64-row padded */
  float* logits; void* dlogits; float* loss;     /* [B*T][64] fp32, [B*T][64] bf16, [1] */
  void* dmem; void* dmem_in; void* dtrg;         /* gradients at the memory (decoder side / encoder side), target rows */
  /* flat parameter buffers and the ranges of them that are active in this training_mode */
  float* P; float* G; float* M; float* V; void* Pb; void* PbT;
  int n_ranges; long range_lo[MMST_MAX_RANGES]; long range_hi[MMST_MAX_RANGES];
  int n_ffn; uint32_t ffn1_off[MMST_MAX_FFN]; uint32_t ffn2_off[MMST_MAX_FFN];   /* linear1 / linear2 offsets in Pb */
  /* fragment-major weight images to refresh after the optimizer (the f_* / ft_* of dec_p), or n_frag = 0 */
  int n_frag; mmst_frag_desc frag[MMST_MAX_FRAG];
  /* Element offsets in the flat buffers at which the gradient segments start, in the order backward completes them from
   * the top: decoder + heads [seg_decoder, end), cross encoder [seg_cross, seg_decoder), modality encoders of depth l
   * [seg_modal[l], seg_modal[l + 1]) with seg_modal[n_enc_layers] = seg_cross, embeddings [0, seg_modal[0]).  With
   * them (seg_decoder > 0) the optimizer of a fused step updates a segment as soon as its gradient is final, beside the
   * rest of backward, on a stream of the library's own; all zero: one optimizer launch at the end. */
  long seg_decoder, seg_cross, seg_modal[MMST_MAX_LAYERS + 1];
} mmst_step_plan;

typedef struct {
  const float* src[MMST_MAX_GROUPS]; const float* mask[MMST_MAX_GROUPS];   /* per group, NULL for the IR group */
  const float* src_ir;                /* [B][1000] fp32 */
  const int64_t* mf_tok; const float* mf_mask; const int64_t* ms_tok; const float* ms_mask; const float* mw;
  int mf_bool, ms_bool, mod_bool, glob_bool;   /* mask semantics of this batch, as in mmst_embed_group / _shared */
  const int64_t* trg; int trg_ld;     /* trg_enc_SMI [B][trg_ld] batch-first; input = columns 0..T-1, targets 1..T */
  uint32_t step_seed;                 /* dropout stream of this step */
  float drop_layers, drop_embed;      /* 0.1 / config.drop_out in train mode, 0 in eval mode */
  int zero_grads;                     /* nonzero: clear the active ranges of G first (not needed after a step whose
                                         optimizer ran inside mmst_train_step: it leaves them zeroed) */
  float loss_grad_scale;
  int adam_step;                      /* >= 1: AdamW with this step count after backward; 0: leave the weights alone */
  float lr, beta1, beta2, eps, weight_decay, grad_scale;
  /* called (on the calling thread) when every launch that writes a gradient segment has been enqueued: the caller
   * may start that segment's all-reduce; NULL for a single rank */
  void (*on_segment)(int segment, void* user); void* user;
} mmst_step_args;

/* `stream` carries the forward pass and the critical path of backward, `wstream` (a different stream) the parameter
 * gradients; on return both have been joined into `stream`. */
int mmst_train_step(const mmst_step_plan* plan, const mmst_step_args* args, void* stream, void* wstream);
int mmst_sizeof_step_plan(void);
int mmst_sizeof_step_args(void);

int mmst_version(void);

#ifdef __cplusplus
}
#endif
#endif
