// Layer-level entry points: each transformer layer is a fixed sequence of grouped GEMM and
// attention launches (see gemm.h / attn.h).  Nothing is allocated and no state is kept here.
#include "../../include/mmst.h"
#include "attn.h"
#include "dec_fused.h"
#include "ffn.h"
#include "gemm.h"
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int D = MMST_D, FF = MMST_FF;

inline GemmProb zero_prob() {
  GemmProb p;
  memset(&p, 0, sizeof(p));
  p.ksplit = 1;
  return p;
}

// split-K factor for a weight-gradient GEMM: enough workgroups to fill the chip, few enough that the
// fp32 atomic traffic (ksplit x output bytes) stays small
// (environment MMST_KSPLIT_TARGET).  Default 48 when the feed-forward gradients have their own kernel: a layer's
// projection gradients are then four output tiles over ~37000 rows, and 192 workgroups of ~12 steps each beat 512 of 5
// (7.52 against 7.71 ms per training step; 36..56 measure the same); 128 when the grouped launch also carries the
// feed-forward gradients (stored hidden gradient), the setting it was tuned for.
inline int ksplit_target() {
  const char* e = getenv("MMST_KSPLIT_TARGET");
  const int t = e ? atoi(e) : 0;
  if (t >= 1) return t;
  return ffn_wgrad_recompute() ? 48 : 128;
}
inline int pick_ksplit(int Pn, int Qn, int K) {
  int tiles = ((Pn + 127) / 128) * ((Qn + 127) / 128);
  int nk = (K + 63) / 64;
  int s = (ksplit_target() + tiles - 1) / tiles;
  if (s > nk) s = nk;
  if (s < 1) s = 1;
  return s;
}

// C = P[M,K] * W[N,K]^T + bias (+relu, dropout)
inline GemmProb fwd_linear(const void* x, int M, int K, const void* w, int N, const float* bias, void* out) {
  GemmProb p = zero_prob();
  p.P = (const bf16*)x; p.ldp = K;
  p.Q = (const bf16*)w; p.ldq = K;
  p.C = out; p.ldc = N; p.bias = bias;
  p.Pn = M; p.Qn = N; p.K = K;
  return p;
}
// dW[N,Kw] += dY[M,N]^T X[M,Kw]; db[N] += colsum(dY)
inline GemmProb wgrad(const void* dy, int M, int N, const void* x, int Kw, float* dw, float* db) {
  GemmProb p = zero_prob();
  p.P = (const bf16*)dy; p.ldp = N;
  p.Q = (const bf16*)x; p.ldq = Kw;
  p.C = dw; p.ldc = Kw; p.dbias = db;
  p.Pn = N; p.Qn = Kw; p.K = M;
  p.ksplit = 1;  // set by launch_wgrads
  return p;
}
// dX[M,Kw] = dY[M,N] W[N,Kw]
inline GemmProb dgrad(const void* dy, int M, int N, const void* w, int Kw, void* dx) {
  GemmProb p = zero_prob();
  p.P = (const bf16*)dy; p.ldp = N;
  p.Q = (const bf16*)w; p.ldq = Kw;
  p.C = dx; p.ldc = Kw;
  p.Pn = M; p.Qn = Kw; p.K = N;
  return p;
}

// All weight gradients of a layer in as few launches as the group limit allows.  The K split is chosen
// so that every workgroup runs about the same number of 64-row contraction steps and the whole launch
// has about ksplit_target() workgroups: small outputs then see a 10-20-way split (low atomic contention)
// instead of one sized to fill the chip on their own.
inline int launch_wgrads(GemmProb* probs, int np, hipStream_t st, int lds_pad = 0, int target_blocks = 0) {
  long work = 0;
  for (int i = 0; i < np; ++i) {
    long tiles = ((probs[i].Pn + 127) / 128) * (long)((probs[i].Qn + 127) / 128);
    work += tiles * ((probs[i].K + 63) / 64);
  }
  int target = target_blocks > 0 ? target_blocks : ksplit_target() * 4;
  long iters = (work + target - 1) / target;
  if (iters < 4) iters = 4;
  for (int i = 0; i < np; ++i) {
    int nk = (probs[i].K + 63) / 64;
    int ks = (int)((nk + iters - 1) / iters);
    probs[i].ksplit = ks < 1 ? 1 : ks;
  }
  for (int base = 0; base < np; base += GEMM_MAX_GROUPS) {
    GemmBatch gb;
    gb.n = np - base < GEMM_MAX_GROUPS ? np - base : GEMM_MAX_GROUPS;
    gb.lds_pad = lds_pad;
    for (int i = 0; i < gb.n; ++i) gb.p[i] = probs[base + i];
    int rc = launch_gemm(gb, EPI_ATOMIC, true, true, st);
    if (rc) return rc;
  }
  return MMST_OK;
}

// feed-forward block forward: y = LN(x + drop(linear2(drop(relu(linear1(x))))))
inline FfnProb ffn_fwd_prob(const void* x, int M, const void* w1, const float* b1, const void* w2, const float* b2,
                            const float* gamma, const float* beta, void* h, void* z, float* stats, void* y,
                            float drop_p, uint32_t seed_hidden, uint32_t seed_out) {
  FfnProb f;
  memset(&f, 0, sizeof(f));
  GemmProb g = zero_prob();
  g.Pn = M; g.Qn = D; g.K = FF;
  g.C = y; g.ldc = D; g.bias = b2;
  g.R = (const bf16*)x; g.ldr = D;
  g.gamma = gamma; g.beta = beta; g.Z = (bf16*)z; g.stats = stats;
  g.drop_p = drop_p; g.seed = seed_out;
  f.epi = g;
  f.X = (const bf16*)x; f.W1 = (const bf16*)w1; f.W2 = (const bf16*)w2; f.b1 = b1;
  f.H = (bf16*)h; f.drop_p = drop_p; f.seed = seed_hidden;
  return f;
}
// feed-forward block backward (activations): dx_in = LNbwd_prev(((dzd W2) * mask(h)) W1 + dz)
inline FfnProb ffn_bwd_prob(const void* dzd, const void* dz, int M, const void* w1t, const void* w2t, const void* h,
                            void* dh, float drop_p) {
  FfnProb f;
  memset(&f, 0, sizeof(f));
  GemmProb g = zero_prob();
  g.Pn = M; g.Qn = D; g.K = FF;
  g.R = (const bf16*)dz; g.ldr = D;
  f.epi = g;
  f.X = (const bf16*)dzd; f.W1 = (const bf16*)w2t; f.W2 = (const bf16*)w1t;
  f.H = (bf16*)h; f.dH = (bf16*)dh; f.drop_p = drop_p;
  return f;
}

#define RUN(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__) return rc__;   \
  } while (0)

}  // namespace

namespace {

// One encoder layer.  qkv_ready: the layer's QKV projection exists already (the launch of the layer in front chained
// it).  pn / an (optional): parameters and buffers of the NEXT layer of the same stack, whose input is this layer's
// output: its QKV projection is then chained behind this layer's feed-forward kernel when that kernel can do it
// (*next_qkv_done says so).
int enc_layer_fwd(int n, const mmst_enc_params* p, const mmst_enc_acts* a, int nhead, float drop_p, uint32_t seed,
                  hipStream_t st, bool qkv_ready, const mmst_enc_params* pn, const mmst_enc_acts* an, bool* next_qkv_done) {
  if (next_qkv_done) *next_qkv_done = false;
  if (n < 1 || n > MMST_MAX_GROUPS || !p || !a || (nhead != 16 && nhead != 4)) return MMST_ERR_ARG;
  const int hd = D / nhead;
  GemmBatch gb;
  AttnBatch ab;
  // 1. fused QKV projection
  gb.n = n;
  if (!qkv_ready) {
    for (int i = 0; i < n; ++i) gb.p[i] = fwd_linear(a[i].x, a[i].B * a[i].S, D, p[i].w_qkv, 3 * D, p[i].b_qkv, a[i].qkv);
    RUN(launch_gemm(gb, EPI_BIAS, false, false, st));
  }
  // 2. attention
  memset(&ab, 0, sizeof(ab));
  ab.n = n; ab.hd = hd; ab.causal = 0; ab.drop_p = drop_p;
  for (int i = 0; i < n; ++i) {
    AttnProb& q = ab.p[i];
    const bf16* qkv = (const bf16*)a[i].qkv;
    q.Q = qkv; q.K = qkv + D; q.V = qkv + 2 * D;
    q.ldq = q.ldk = q.ldv = 3 * D;
    q.kbias = a[i].kbias;
    q.O = (bf16*)a[i].attn; q.ldo = D;
    q.lse = a[i].lse;
    q.B = a[i].B; q.H = nhead; q.Tq = a[i].S; q.Sk = a[i].S;
    q.seed = seed + 8 * i + 0;
  }
  RUN(launch_attn_fwd(ab, st));
  // 3. out_proj + dropout + residual + norm1 and 4. feed-forward block + norm2: one launch (the fused feed-forward
  // kernel runs step 3 as a prologue on its row tile; with few rows step 3 is a grouped GEMM launch of its own)
  FfnBatch fb;
  fb.n = n;
  for (int i = 0; i < n; ++i) {
    FfnProb f = ffn_fwd_prob(a[i].x1, a[i].B * a[i].S, p[i].w_1, p[i].b_1, p[i].w_2, p[i].b_2, p[i].g2, p[i].be2, a[i].h,
                             a[i].z2, a[i].st2, a[i].y, drop_p, seed + 8 * i + 2, seed + 8 * i + 3);
    if (a[i].y_L > 0) { f.epi.cS = a[i].S; f.epi.cL = a[i].y_L; f.epi.coff = a[i].y_off; }
    f.Hs = ffn_bwd_signs() ? (uint32_t*)a[i].hsign : nullptr;
    GemmProb g = fwd_linear(a[i].attn, a[i].B * a[i].S, D, p[i].w_o, D, p[i].b_o, a[i].x1);
    g.R = (const bf16*)a[i].x; g.ldr = D;
    g.gamma = p[i].g1; g.beta = p[i].be1; g.Z = (bf16*)a[i].z1; g.stats = a[i].st1;
    g.drop_p = drop_p; g.seed = seed + 8 * i + 1;
    f.pre = g;
    f.A = (const bf16*)a[i].attn; f.Wp = (const bf16*)p[i].w_o;
    fb.p[i] = f;
  }
  bool chain = pn && an && next_qkv_done;
  for (int i = 0; chain && i < n; ++i)
    chain = an[i].x == a[i].y && a[i].y_L <= 0 && an[i].B == a[i].B && an[i].S == a[i].S && an[i].qkv && pn[i].w_qkv;
  if (chain)
    for (int i = 0; i < n; ++i) {
      fb.p[i].Wq = (const bf16*)pn[i].w_qkv; fb.p[i].bq = pn[i].b_qkv; fb.p[i].Cq = (bf16*)an[i].qkv; fb.p[i].nq = 3 * D;
    }
  return launch_ffn_fwd(fb, st, next_qkv_done);
}

}  // namespace

extern "C" {

int mmst_encoder_layer_fwd(int n, const mmst_enc_params* p, const mmst_enc_acts* a, int nhead, float drop_p,
                           uint32_t seed, void* stream_) {
  return enc_layer_fwd(n, p, a, nhead, drop_p, seed, (hipStream_t)stream_, false, nullptr, nullptr, nullptr);
}

// nl layers of one encoder stack in sequence (layer l + 1 reads what layer l wrote): the same launches as nl calls of
// mmst_encoder_layer_fwd, except that a layer's QKV projection is chained behind the feed-forward kernel of the layer in
// front wherever that kernel can do it (large row counts, no row remap of the output).
int mmst_encoder_stack_fwd(int n, int nl, const mmst_enc_params* const* p, const mmst_enc_acts* const* a, int nhead,
                           float drop_p, const uint32_t* seeds, void* stream_) {
  if (nl < 1 || nl > MMST_MAX_LAYERS || !p || !a || !seeds) return MMST_ERR_ARG;
  bool ready = false;
  for (int l = 0; l < nl; ++l) {
    bool next_done = false;
    const bool more = l + 1 < nl;
    RUN(enc_layer_fwd(n, p[l], a[l], nhead, drop_p, seeds[l], (hipStream_t)stream_, ready, more ? p[l + 1] : nullptr,
                      more ? a[l + 1] : nullptr, more ? &next_done : nullptr));
    ready = next_done;
  }
  return MMST_OK;
}

}  // extern "C"

namespace {

hipEvent_t arm_completion_event();   // stream hand-off helpers, defined below

inline bool enc_args_ok(int n, const void* p, const void* g, const void* a, const void* w, int nhead) {
  return n >= 1 && n <= MMST_MAX_GROUPS && p && g && a && w && (nhead == 16 || nhead == 4);
}

// first part of the critical path: feed-forward block backward fused with norm1 backward,
// dz1 = LNbwd(((dzd2 W2) * mask(h)) W1 + dz2).  Leaves dh, dz1 and dzd1 in w.
// *da_done: whether the launch also produced da = dzd1 Wo (the fused kernel chains that GEMM behind its epilogue).
int enc_bwd_ffn(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a, const mmst_enc_bwd* w,
                float drop_p, uint32_t seed, hipStream_t st, bool* da_done) {
  FfnBatch fb;
  fb.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (a[i].B * a[i].S + 127) / 128;
  const bool recompute = ffn_wgrad_recompute() && tiles >= ffn_fuse_min_tiles();   // then dh is not stored
  for (int i = 0; i < n; ++i) {
    FfnProb f = ffn_bwd_prob(w[i].dzd2, w[i].dz2, a[i].B * a[i].S, p[i].w_1t, p[i].w_2t, a[i].h,
                             recompute ? nullptr : w[i].dh, drop_p);
    GemmProb& q = f.epi;
    q.C = w[i].dz1; q.ldc = D;
    q.Z = (bf16*)a[i].z1; q.stats = a[i].st1; q.gamma = p[i].g1;
    q.C2 = drop_p > 0.f ? (bf16*)w[i].dzd1 : nullptr;
    q.dgamma = g[i].g1; q.dbeta = g[i].be1;
    q.drop_p = drop_p; q.seed = seed + 8 * i + 1;
    f.Wn = (const bf16*)p[i].w_o; f.Cn = (bf16*)w[i].da;
    f.Hs = ffn_bwd_signs() ? (uint32_t*)a[i].hsign : nullptr;
    fb.p[i] = f;
  }
  return launch_ffn_bwd(fb, st, da_done);
}

// second part of the critical path: out_proj, attention and in_proj backward.  Leaves dqkv in w.
int enc_bwd_attn(int n, const mmst_enc_params* p, const mmst_enc_acts* a, const mmst_enc_bwd* w, int nhead, float drop_p,
                 uint32_t seed, hipStream_t st, bool da_done, hipEvent_t* done = nullptr, hipEvent_t* attn_done = nullptr) {
  const int hd = D / nhead;
  GemmBatch gb;
  gb.n = n;
  if (!da_done) {   // da = dzd1 Wo
    for (int i = 0; i < n; ++i) {
      const void* dzd1 = drop_p > 0.f ? w[i].dzd1 : w[i].dz1;
      gb.p[i] = dgrad(dzd1, a[i].B * a[i].S, D, p[i].w_o, D, w[i].da);
    }
    RUN(launch_gemm(gb, EPI_BIAS, false, true, st));
  }
  // attention backward -> dqkv
  AttnBatch ab;
  memset(&ab, 0, sizeof(ab));
  ab.n = n; ab.hd = hd; ab.causal = 0; ab.drop_p = drop_p;
  for (int i = 0; i < n; ++i) {
    AttnProb& q = ab.p[i];
    const bf16* qkv = (const bf16*)a[i].qkv;
    bf16* dqkv = (bf16*)w[i].dqkv;
    q.Q = qkv; q.K = qkv + D; q.V = qkv + 2 * D; q.ldq = q.ldk = q.ldv = 3 * D;
    q.kbias = a[i].kbias;
    q.O = (bf16*)a[i].attn; q.ldo = D; q.lse = a[i].lse;
    q.dO = (const bf16*)w[i].da; q.lddo = D; q.delta = w[i].delta;
    q.dQ = dqkv; q.dK = dqkv + D; q.dV = dqkv + 2 * D; q.lddq = q.lddk = q.lddv = 3 * D;
    q.B = a[i].B; q.H = nhead; q.Tq = a[i].S; q.Sk = a[i].S;
    q.seed = seed + 8 * i + 0;
  }
  if (attn_done) *attn_done = arm_completion_event();   // completion event of the last attention kernel: dqkv exists
  RUN(launch_attn_bwd(ab, st));
  // dx = dqkv Wqkv + dz1 ; fused with the previous layer's norm2 backward when there is one
  bool prev = w[0].prev_z2 != nullptr;
  for (int i = 0; i < n; ++i) {
    if ((w[i].prev_z2 != nullptr) != prev) return MMST_ERR_ARG;
    GemmProb q = dgrad(w[i].dqkv, a[i].B * a[i].S, 3 * D, p[i].w_qkv, D, w[i].dx);
    q.R = (const bf16*)w[i].dz1; q.ldr = D;
    if (prev) {
      q.Z = (bf16*)w[i].prev_z2; q.stats = (float*)w[i].prev_st2; q.gamma = w[i].prev_g2;
      q.C2 = drop_p > 0.f ? (bf16*)w[i].dxd : nullptr;
      q.dgamma = w[i].prev_dg2; q.dbeta = w[i].prev_dbe2;
      q.drop_p = drop_p; q.seed = w[i].prev_seed + 8 * i + 3;
    }
    gb.p[i] = q;
  }
  if (done) *done = arm_completion_event();   // completion event of the launch below
  return launch_gemm(gb, prev ? EPI_LNBWD : EPI_BIAS, false, true, st);
}

// Parameter gradients of a layer as grouped split-K launches.  WG_EARLY: linear2, linear1 and out_proj, whose
// inputs exist once enc_bwd_ffn has run; WG_LATE: in_proj, which needs dqkv from enc_bwd_attn.
// WG_FFN_RECOMPUTED: the feed-forward gradients by the recomputing kernel, which reads only what exists BEFORE the
// layer's backward starts (the incoming gradient, the saved activations); WG_SKIP_FFN: they have been launched already.
enum { WG_EARLY = 1, WG_LATE = 2, WG_FFN_RECOMPUTED = 4, WG_SKIP_FFN = 8 };
bool enc_recompute(int n, const mmst_enc_acts* a) {
  int tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (a[i].B * a[i].S + 127) / 128;
  return ffn_wgrad_recompute() && tiles >= ffn_fuse_min_tiles();   // as in enc_bwd_ffn
}
int enc_bwd_weights(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                    const mmst_enc_bwd* w, float drop_p, int which, hipStream_t st) {
  GemmProb probs[4 * MMST_MAX_GROUPS];
  int np = 0;
  const bool recompute = enc_recompute(n, a);
  if (recompute && (which & (WG_EARLY | WG_FFN_RECOMPUTED)) && !(which & WG_SKIP_FFN)) {
    FfnWgradBatch wb;
    wb.n = n;
    for (int i = 0; i < n; ++i) {
      FfnWgradProb& f = wb.p[i];
      f.dZ = (const bf16*)w[i].dzd2; f.H = (const bf16*)a[i].h; f.X = (const bf16*)a[i].x1; f.W2T = (const bf16*)p[i].w_2t;
      f.dW1 = g[i].w_1; f.db1 = g[i].b_1; f.dW2 = g[i].w_2; f.db2 = g[i].b_2;
      f.M = a[i].B * a[i].S; f.drop_p = drop_p;
    }
    RUN(launch_ffn_wgrad(wb, st));
  }
  if (!(which & (WG_EARLY | WG_LATE))) return MMST_OK;
  for (int i = 0; i < n; ++i) {
    const int M = a[i].B * a[i].S;
    const void* dzd1 = drop_p > 0.f ? w[i].dzd1 : w[i].dz1;
    if (which & WG_EARLY) {
      if (!recompute) {
        probs[np++] = wgrad(w[i].dzd2, M, D, a[i].h, FF, g[i].w_2, g[i].b_2);
        probs[np++] = wgrad(w[i].dh, M, FF, a[i].x1, D, g[i].w_1, g[i].b_1);
      }
      probs[np++] = wgrad(dzd1, M, D, a[i].attn, D, g[i].w_o, g[i].b_o);
    }
    if (which & WG_LATE) probs[np++] = wgrad(w[i].dqkv, M, 3 * D, a[i].x, D, g[i].w_qkv, g[i].b_qkv);
  }
  return launch_wgrads(probs, np, st);
}

// Events of a per-device ring created on first use (a wait holds the state its event had when the wait was
// enqueued, so a slot can be used again later).
hipEvent_t ring_event() {
  constexpr int MAX_DEV = 16, RING = 64;
  static hipEvent_t ring[MAX_DEV][RING];
  static bool made[MAX_DEV];
  static int next[MAX_DEV];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!made[dev]) {
    for (int i = 0; i < RING; ++i)
      if (hipEventCreateWithFlags(&ring[dev][i], hipEventDisableTiming) != hipSuccess) return nullptr;
    made[dev] = true;
  }
  hipEvent_t ev = ring[dev][next[dev]];
  next[dev] = (next[dev] + 1) % RING;
  return ev;
}
// One stream of the library's own per device, created on first use: the decoder's backward keeps three chains busy
// (activation gradients, the memory's dK / dV passes, parameter gradients) while most CUs idle.
hipStream_t third_stream() { return mmst_side_stream(); }
}  // namespace
hipStream_t mmst_side_stream() {
  constexpr int MAX_DEV = 16;
  static hipStream_t streams[MAX_DEV];
  static bool made[MAX_DEV];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!made[dev]) {
    if (hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) return nullptr;
    made[dev] = true;
  }
  return streams[dev];
}
namespace {

// `to` (if not null) waits for everything enqueued on `from` so far (a marker on `from`); the event is handed back
// on request.
int stream_fork(hipStream_t from, hipStream_t to, hipEvent_t* recorded = nullptr) {
  hipEvent_t ev = ring_event();
  if (!ev) return MMST_ERR_LAUNCH;
  if (hipEventRecord(ev, from) != hipSuccess) return MMST_ERR_LAUNCH;
  if (to && hipStreamWaitEvent(to, ev, 0) != hipSuccess) return MMST_ERR_LAUNCH;
  if (recorded) *recorded = ev;
  return MMST_OK;
}
// The next kernel launched through MMST_LAUNCH carries the returned event as its completion event (no marker).
hipEvent_t arm_completion_event() {
  hipEvent_t ev = ring_event();
  mmst_armed_stop_event = ev;
  return ev;
}
int wait_for(hipStream_t to, hipEvent_t ev) {
  return (ev && hipStreamWaitEvent(to, ev, 0) == hipSuccess) ? MMST_OK : MMST_ERR_LAUNCH;
}

}  // namespace

// `to` waits for everything enqueued on `from` so far.  Either may be the default stream (a null handle), so unlike
// stream_fork() a null `to` is a stream to wait on, not "nobody waits".
int mmst_stream_fork(hipStream_t from, hipStream_t to) {
  hipEvent_t ev = ring_event();
  if (!ev || from == to) return MMST_ERR_LAUNCH;
  if (hipEventRecord(ev, from) != hipSuccess) return MMST_ERR_LAUNCH;
  return hipStreamWaitEvent(to, ev, 0) == hipSuccess ? MMST_OK : MMST_ERR_LAUNCH;
}

extern "C" {

// ---- encoder layer backward, critical path: gradients of activations only ----
int mmst_encoder_layer_bwd_data(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                                const mmst_enc_bwd* w, int nhead, float drop_p, uint32_t seed, void* stream_) {
  if (!enc_args_ok(n, p, g, a, w, nhead)) return MMST_ERR_ARG;
  bool da_done = false;
  RUN(enc_bwd_ffn(n, p, g, a, w, drop_p, seed, (hipStream_t)stream_, &da_done));
  return enc_bwd_attn(n, p, a, w, nhead, drop_p, seed, (hipStream_t)stream_, da_done);
}

// ---- encoder layer backward, off the critical path: every parameter gradient of the layer as ONE grouped
// split-K launch (all groups x {linear2, linear1, out_proj, in_proj}); reads what the data phase left in w ----
int mmst_encoder_layer_bwd_weights(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                                   const mmst_enc_bwd* w, float drop_p, void* stream_) {
  if (!enc_args_ok(n, p, g, a, w, 16)) return MMST_ERR_ARG;
  return enc_bwd_weights(n, p, g, a, w, drop_p, WG_EARLY | WG_LATE, (hipStream_t)stream_);
}

// ---- both halves on two streams.  interleave == 0: every parameter gradient on wstream once the data phase is
// done (wstream waits for the completion event of its last kernel: no marker on `stream`).  interleave != 0: the
// feed-forward and out_proj weight gradients (nearly all of the layer's weight-gradient work, HBM- and MFMA-heavy)
// start on wstream as soon as the feed-forward backward is done and so run beside the attention backward
// (VALU-bound) of the same layer instead of beside the next layer's feed-forward backward; the in_proj gradient
// follows when dqkv exists.  With recomputed feed-forward gradients (and bit 0 clear) those are launched on wstream
// before anything else of the layer; bit 1 of `interleave` set: the caller knows wstream to be ordered behind the
// producer of the incoming gradient already (the previous call of this function for the layer above), so no marker
// is recorded on `stream` for it ----
int mmst_encoder_layer_bwd_streams(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                                   const mmst_enc_bwd* w, int nhead, float drop_p, uint32_t seed, int interleave,
                                   void* stream_, void* wstream_) {
  if (!enc_args_ok(n, p, g, a, w, nhead) || !wstream_ || stream_ == wstream_) return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_, wst = (hipStream_t)wstream_;
  hipEvent_t done = nullptr;
  // recomputed feed-forward gradients need nothing this layer's backward produces: they start with it, beside the
  // feed-forward backward that streams the same saved activation
  const bool ffn_first = enc_recompute(n, a) && !(interleave & 1);
  if (ffn_first) {
    if (!(interleave & 2)) RUN(stream_fork(st, wst));   // 2: wstream is already ordered behind the incoming gradient
    RUN(enc_bwd_weights(n, p, g, a, w, drop_p, WG_FFN_RECOMPUTED, wst));
  }
  bool da_done = false;
  RUN(enc_bwd_ffn(n, p, g, a, w, drop_p, seed, st, &da_done));
  if (interleave & 1) {
    RUN(stream_fork(st, wst));
    RUN(enc_bwd_weights(n, p, g, a, w, drop_p, WG_EARLY, wst));
  }
  // The projection gradients need dqkv only: they start behind the attention backward, beside the in_proj backward
  // GEMM that ends the layer, instead of behind that GEMM (MMST_WGRAD_AFTER_ATTN=0 restores the later start); wstream
  // is then ordered behind the GEMM as well, whose output the layer below reads first on that stream.
  static const bool after_attn = [] { const char* e = getenv("MMST_WGRAD_AFTER_ATTN"); return e ? atoi(e) != 0 : true; }();
  hipEvent_t attn_done = nullptr;
  RUN(enc_bwd_attn(n, p, a, w, nhead, drop_p, seed, st, da_done, &done, after_attn ? &attn_done : nullptr));
  const int which = ((interleave & 1) ? WG_LATE : (WG_EARLY | WG_LATE)) | (ffn_first ? WG_SKIP_FFN : 0);
  if (after_attn) {
    RUN(wait_for(wst, attn_done));
    RUN(enc_bwd_weights(n, p, g, a, w, drop_p, which, wst));
    return wait_for(wst, done);
  }
  RUN(wait_for(wst, done));
  return enc_bwd_weights(n, p, g, a, w, drop_p, which, wst);
}

int mmst_encoder_layer_bwd(int n, const mmst_enc_params* p, const mmst_enc_grads* g, const mmst_enc_acts* a,
                           const mmst_enc_bwd* w, int nhead, float drop_p, uint32_t seed, void* stream_) {
  RUN(mmst_encoder_layer_bwd_data(n, p, g, a, w, nhead, drop_p, seed, stream_));
  return mmst_encoder_layer_bwd_weights(n, p, g, a, w, drop_p, stream_);
}

int mmst_layernorm_bwd(int rows, const void* dy, int dy_S, int dy_L, int dy_off, const void* z, const float* stats,
                       const float* gamma, void* dz, void* dzd, float* dgamma, float* dbeta, float drop_p,
                       uint32_t seed, void* stream_) {
  if (rows < 1 || !dy || !z || !stats || !gamma || !dz) return MMST_ERR_ARG;
  GemmBatch gb;
  gb.n = 1;
  GemmProb q = zero_prob();
  q.Pn = rows; q.Qn = D; q.K = 0; q.ldp = 8; q.ldq = 8;
  q.C = dz; q.ldc = D;
  q.C2 = (drop_p > 0.f) ? (bf16*)dzd : nullptr;
  q.R = (const bf16*)dy; q.ldr = D;
  if (dy_L > 0) { q.rS = dy_S; q.rL = dy_L; q.roff = dy_off; }
  q.Z = (bf16*)z; q.stats = (float*)stats; q.gamma = gamma;
  q.dgamma = dgamma; q.dbeta = dbeta;
  q.drop_p = drop_p; q.seed = seed;
  gb.p[0] = q;
  return launch_gemm(gb, EPI_LNBWD, false, false, (hipStream_t)stream_);
}

int mmst_decoder_layer_fwd(const mmst_dec_params* p, const mmst_dec_acts* a, int nhead, float drop_p, uint32_t seed,
                           void* stream_) {
  if (!p || !a || nhead != 16) return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  const int hd = D / nhead, M = a->B * a->T, ML = a->B * a->L;
  GemmBatch gb;
  AttnBatch ab;
  gb.n = 1;
  // self-attention block
  gb.p[0] = fwd_linear(a->x, M, D, p->w_qkv, 3 * D, p->b_qkv, a->qkv);
  RUN(launch_gemm(gb, EPI_BIAS, false, false, st));
  memset(&ab, 0, sizeof(ab));
  ab.n = 1; ab.hd = hd; ab.causal = a->no_causal ? 0 : 1; ab.drop_p = drop_p;
  {
    AttnProb& q = ab.p[0];
    const bf16* qkv = (const bf16*)a->qkv;
    q.Q = qkv; q.K = qkv + D; q.V = qkv + 2 * D; q.ldq = q.ldk = q.ldv = 3 * D;
    q.O = (bf16*)a->attn1; q.ldo = D; q.lse = a->lse1;
    q.B = a->B; q.H = nhead; q.Tq = a->T; q.Sk = a->T; q.seed = seed + 0;
  }
  RUN(launch_attn_fwd(ab, st));
  {
    GemmProb g = fwd_linear(a->attn1, M, D, p->w_o, D, p->b_o, a->x1);
    g.R = (const bf16*)a->x; g.ldr = D; g.gamma = p->g1; g.beta = p->be1; g.Z = (bf16*)a->z1; g.stats = a->st1;
    g.drop_p = drop_p; g.seed = seed + 1;
    gb.p[0] = g;
  }
  RUN(launch_gemm(gb, EPI_LN, false, false, st));
  // cross-attention block: q from x1, k/v from memory
  gb.n = a->kv_ready ? 1 : 2;
  gb.p[0] = fwd_linear(a->x1, M, D, p->w_cqkv, D, p->b_cqkv, a->qc);
  gb.p[1] = fwd_linear(a->mem, ML, D, (const bf16*)p->w_cqkv + D * D, 2 * D, p->b_cqkv + D, a->kvc);
  RUN(launch_gemm(gb, EPI_BIAS, false, false, st));
  gb.n = 1;
  ab.causal = 0;
  {
    AttnProb& q = ab.p[0];
    memset(&q, 0, sizeof(q));
    const bf16* kv = (const bf16*)a->kvc;
    q.Q = (const bf16*)a->qc; q.ldq = D;
    q.K = kv; q.V = kv + D; q.ldk = q.ldv = 2 * D;
    q.kbias = a->mem_bias;
    q.O = (bf16*)a->attn2; q.ldo = D; q.lse = a->lse2;
    q.B = a->B; q.H = nhead; q.Tq = a->T; q.Sk = a->L; q.seed = seed + 2;
  }
  RUN(launch_attn_fwd(ab, st));
  {
    GemmProb g = fwd_linear(a->attn2, M, D, p->w_co, D, p->b_co, a->x2);
    g.R = (const bf16*)a->x1; g.ldr = D; g.gamma = p->g2; g.beta = p->be2; g.Z = (bf16*)a->z2; g.stats = a->st2;
    g.drop_p = drop_p; g.seed = seed + 3;
    gb.p[0] = g;
  }
  RUN(launch_gemm(gb, EPI_LN, false, false, st));
  // feed-forward block + norm3, one kernel
  {
    FfnBatch fb;
    fb.n = 1;
    fb.p[0] = ffn_fwd_prob(a->x2, M, p->w_1, p->b_1, p->w_2, p->b_2, p->g3, p->be3, a->h, a->z3, a->st3, a->y, drop_p,
                           seed + 4, seed + 5);
    fb.p[0].ws = a->ffn_ws;
    RUN(launch_ffn_fwd(fb, st));
  }
  return MMST_OK;
}

// ---- nl decoder layers in sequence (layer l + 1 reads what layer l wrote) ----
// Short target sequences (T <= 64, causal): two launches per layer instead of eight.  The K|V projections of the
// memory for all layers are one grouped launch (on wstream when given: it runs beside the first self-attention block);
// per layer the cross-attention launch is followed by ONE launch that finishes the layer (out_proj + norm2, feed-forward
// block + norm3) and starts the next one (QKV, causal self-attention, out_proj + norm1, cross-attention queries) with a
// workgroup per molecule (dec_fused.hip).  Otherwise: nl calls of mmst_decoder_layer_fwd.
int mmst_decoder_stack_fwd(int nl, const mmst_dec_params* const* p, const mmst_dec_acts* const* a, int nhead, float drop_p,
                           const uint32_t* seeds, void* stream_, void* wstream_) {
  if (nl < 1 || nl > MMST_MAX_LAYERS || !p || !a || !seeds || nhead != 16) return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_, wst = (hipStream_t)wstream_;
  if (wst == st) wst = nullptr;
  bool fused = dec_fused_enabled();
  for (int l = 0; l < nl; ++l) {
    if (!p[l] || !a[l]) return MMST_ERR_ARG;
    if (a[l]->T < 1 || a[l]->T > 64 || a[l]->no_causal || a[l]->B != a[0]->B || a[l]->T != a[0]->T || a[l]->L != a[0]->L)
      fused = false;
    if (l > 0 && a[l]->x != a[l - 1]->y) fused = false;
    if (!a[l]->ffn_ws) fused = false;   // [8][B*T][128] fp32: the fused path keeps four partial sums there
    if (!p[l]->f_qkv || !p[l]->f_o || !p[l]->f_cqkv || !p[l]->f_co || !p[l]->f_1 || !p[l]->f_2) fused = false;
  }
  if (!fused) {
    for (int l = 0; l < nl; ++l) RUN(mmst_decoder_layer_fwd(p[l], a[l], nhead, drop_p, seeds[l], stream_));
    return MMST_OK;
  }
  const int B = a[0]->B, T = a[0]->T, L = a[0]->L, ML = B * L;
  GemmBatch gb;
  gb.n = 0;
  for (int l = 0; l < nl; ++l)
    if (!a[l]->kv_ready)
      gb.p[gb.n++] = fwd_linear(a[l]->mem, ML, D, (const bf16*)p[l]->w_cqkv + D * D, 2 * D, p[l]->b_cqkv + D, a[l]->kvc);
  if (gb.n > 0) {
    if (wst) RUN(mmst_stream_fork(st, wst));   // the memory exists
    RUN(launch_gemm(gb, EPI_BIAS, false, false, wst ? wst : st));
  }
  DecFusedFwd f;
  memset(&f, 0, sizeof(f));
  f.B = B; f.T = T; f.drop_p = drop_p;
  f.do_head = 1; f.ph = *p[0]; f.ah = *a[0]; f.seed_h = seeds[0];
  RUN(launch_dec_fused_fwd(f, st));
  if (gb.n > 0 && wst) RUN(mmst_stream_fork(wst, st));
  for (int l = 0; l < nl; ++l) {
    AttnBatch ab;
    memset(&ab, 0, sizeof(ab));
    ab.n = 1; ab.hd = D / nhead; ab.causal = 0; ab.drop_p = drop_p;
    AttnProb& q = ab.p[0];
    const bf16* kv = (const bf16*)a[l]->kvc;
    q.Q = (const bf16*)a[l]->qc; q.ldq = D;
    q.K = kv; q.V = kv + D; q.ldk = q.ldv = 2 * D;
    q.kbias = a[l]->mem_bias;
    q.O = (bf16*)a[l]->attn2; q.ldo = D; q.lse = a[l]->lse2;
    q.B = B; q.H = nhead; q.Tq = T; q.Sk = L; q.seed = seeds[l] + 2;
    RUN(launch_attn_fwd(ab, st));
    memset(&f, 0, sizeof(f));
    f.B = B; f.T = T; f.drop_p = drop_p;
    f.do_tail = 1; f.pt = *p[l]; f.at = *a[l]; f.seed_t = seeds[l]; f.ws = a[l]->ffn_ws;
    if (l + 1 < nl) { f.do_head = 1; f.ph = *p[l + 1]; f.ah = *a[l + 1]; f.seed_h = seeds[l + 1]; }
    RUN(launch_dec_fused_fwd(f, st));
  }
  return MMST_OK;
}

// ---- decoder layer, one new position with cached keys/values ----
int mmst_decoder_layer_step(const mmst_dec_params* p, const mmst_dec_step* s, int nhead, void* stream_) {
  if (!p || !s || nhead != 16 || s->B < 1 || s->pos < 0 || s->pos >= s->cap || s->L < 1) return MMST_ERR_ARG;
  if (!s->x || !s->y || !s->qkv_cache || !s->kvc || !s->attn || !s->x1 || !s->qc || !s->x2 || !s->h || !s->lse)
    return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  const int hd = D / nhead, B = s->B;
  GemmBatch gb;
  AttnBatch ab;
  gb.n = 1;
  // Q|K|V of the new row, straight into its cache slot (row b*cap + pos)
  {
    GemmProb g = fwd_linear(s->x, B, D, p->w_qkv, 3 * D, p->b_qkv, s->qkv_cache);
    g.cS = 1; g.cL = s->cap; g.coff = s->pos;
    gb.p[0] = g;
  }
  RUN(launch_gemm(gb, EPI_BIAS, false, false, st));
  // self-attention of the new row over positions 0..pos (every cached key is visible: no mask needed)
  memset(&ab, 0, sizeof(ab));
  ab.n = 1; ab.hd = hd; ab.causal = 0; ab.drop_p = 0.f;
  {
    AttnProb& q = ab.p[0];
    const bf16* cache = (const bf16*)s->qkv_cache;
    q.Q = cache + (size_t)s->pos * 3 * D; q.ldq = s->cap * 3 * D;   // one query row per batch element
    q.K = cache + D; q.V = cache + 2 * D; q.ldk = q.ldv = 3 * D; q.kv_rows = s->cap;
    q.O = (bf16*)s->attn; q.ldo = D; q.lse = s->lse;
    q.B = B; q.H = nhead; q.Tq = 1; q.Sk = s->pos + 1;
  }
  RUN(launch_attn_fwd(ab, st));
  {
    GemmProb g = fwd_linear(s->attn, B, D, p->w_o, D, p->b_o, s->x1);
    g.R = (const bf16*)s->x; g.ldr = D; g.gamma = p->g1; g.beta = p->be1;
    gb.p[0] = g;
  }
  RUN(launch_gemm(gb, EPI_LN, false, false, st));
  // cross-attention: query from the new row, keys/values of the memory projected once per sequence
  gb.p[0] = fwd_linear(s->x1, B, D, p->w_cqkv, D, p->b_cqkv, s->qc);
  RUN(launch_gemm(gb, EPI_BIAS, false, false, st));
  {
    AttnProb& q = ab.p[0];
    memset(&q, 0, sizeof(q));
    const bf16* kv = (const bf16*)s->kvc;
    q.Q = (const bf16*)s->qc; q.ldq = D;
    q.K = kv; q.V = kv + D; q.ldk = q.ldv = 2 * D;
    q.kbias = s->mem_bias;
    q.O = (bf16*)s->attn; q.ldo = D; q.lse = s->lse;
    q.B = B; q.H = nhead; q.Tq = 1; q.Sk = s->L;
  }
  RUN(launch_attn_fwd(ab, st));
  {
    GemmProb g = fwd_linear(s->attn, B, D, p->w_co, D, p->b_co, s->x2);
    g.R = (const bf16*)s->x1; g.ldr = D; g.gamma = p->g2; g.beta = p->be2;
    gb.p[0] = g;
  }
  RUN(launch_gemm(gb, EPI_LN, false, false, st));
  // feed-forward block + norm3
  {
    FfnBatch fb;
    fb.n = 1;
    fb.p[0] = ffn_fwd_prob(s->x2, B, p->w_1, p->b_1, p->w_2, p->b_2, p->g3, p->be3, s->h, nullptr, nullptr, s->y, 0.f, 0, 0);
    fb.p[0].ws = s->ffn_ws;
    RUN(launch_ffn_fwd(fb, st));
  }
  return MMST_OK;
}

}  // extern "C"

// ---- the token step for callers that cannot allocate: the scratch rows live in a static device array ----
namespace {

constexpr int STATIC_MAX_B = MMST_STATIC_STEP_MAX_B;
// carve-up of the array, every part a multiple of 256 bytes
constexpr size_t SC_ROW = (size_t)STATIC_MAX_B * D * sizeof(bf16);           // attn, x1, qc, x2: [B][128] bf16 each
constexpr size_t SC_H = (size_t)STATIC_MAX_B * FF * sizeof(bf16);            // h: [B][2048] bf16
constexpr size_t SC_LSE = (size_t)STATIC_MAX_B * 16 * sizeof(float);         // lse: [B][16] fp32
constexpr size_t SC_WS = (size_t)FFN_MAX_HSPLIT * STATIC_MAX_B * D * sizeof(float);   // ffn_ws: [8][B][128] fp32
constexpr size_t SC_TOTAL = 4 * SC_ROW + SC_H + SC_LSE + SC_WS;
static_assert(SC_ROW % 256 == 0 && SC_H % 256 == 0 && SC_LSE % 256 == 0 && SC_WS % 256 == 0, "scratch alignment");

}  // namespace

// One copy per GPU, held until the process ends.  The name below is a host-side stand-in: device code and kernel
// arguments use the address hipGetSymbolAddress returns for the current device (scratch_base()).
static __device__ __align__(256) unsigned char scratch[SC_TOTAL];

namespace {

unsigned char* scratch_base() {
  constexpr int MAX_DEV = 16;
  static unsigned char* base[MAX_DEV];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!base[dev]) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return nullptr;
    base[dev] = (unsigned char*)ptr;
  }
  return base[dev];
}

}  // namespace

extern "C" int mmst_decoder_layer_step_static(const mmst_dec_params* p, const void* x, void* y, void* qkv_cache,
                                              const void* kvc, const float* mem_bias, int B, int pos, int cap, int L,
                                              int nhead, void* stream_) {
  if (B < 1 || B > STATIC_MAX_B) return MMST_ERR_ARG;   // the array holds the scratch rows of at most 256 sequences
  unsigned char* base = scratch_base();
  if (!base) return MMST_ERR_LAUNCH;
  mmst_dec_step s;
  memset(&s, 0, sizeof(s));
  s.x = x; s.y = y; s.qkv_cache = qkv_cache; s.kvc = kvc; s.mem_bias = mem_bias;
  s.attn = base;
  s.x1 = base + SC_ROW;
  s.qc = base + 2 * SC_ROW;
  s.x2 = base + 3 * SC_ROW;
  s.h = base + 4 * SC_ROW;
  s.lse = (float*)(base + 4 * SC_ROW + SC_H);
  s.ffn_ws = (float*)(base + 4 * SC_ROW + SC_H + SC_LSE);
  s.B = B; s.pos = pos; s.cap = cap; s.L = L;
  // every scratch row a call reads it has written before (projections, attention rows, hidden tiles and the
  // partial-sum slabs of the split feed-forward block), so nothing needs clearing between calls
  return mmst_decoder_layer_step(p, &s, nhead, stream_);
}

namespace {

// Decoder layer backward, activation gradients.  wst == nullptr: everything on st.  Otherwise the part of the
// cross-attention backward that only feeds the encoder (dK/dV of the memory and their projection into dmem) goes
// to wst: the next decoder layer does not wait for it.  join: st waits for that dmem update at its end (the
// caller is about to read dmem on st).
int dec_bwd_data(const mmst_dec_params* p, const mmst_dec_grads* g, const mmst_dec_acts* a, const mmst_dec_bwd* w,
                 int nhead, float drop_p, uint32_t seed, hipStream_t st, hipStream_t wst, bool join,
                 hipEvent_t* done = nullptr) {
  if (!p || !g || !a || !w || nhead != 16) return MMST_ERR_ARG;
  const int hd = D / nhead, M = a->B * a->T, ML = a->B * a->L;
  const bool dr = drop_p > 0.f;
  GemmBatch gb;
  gb.n = 1;
  // feed-forward block backward fused with norm2 backward
  {
    FfnBatch fb;
    fb.n = 1;
    FfnProb f = ffn_bwd_prob(w->dzd3, w->dz3, M, p->w_1t, p->w_2t, a->h, w->dh, drop_p);
    GemmProb& q = f.epi;
    q.C = w->dz2; q.ldc = D; q.Z = (bf16*)a->z2; q.stats = a->st2; q.gamma = p->g2;
    q.C2 = dr ? (bf16*)w->dzd2 : nullptr; q.dgamma = g->g2; q.dbeta = g->be2;
    q.drop_p = drop_p; q.seed = seed + 3;
    f.ws = a->ffn_ws;
    fb.p[0] = f;
    RUN(launch_ffn_bwd(fb, st));
  }
  const void* dzd2 = dr ? w->dzd2 : w->dz2;
  // cross-attention block
  gb.p[0] = dgrad(dzd2, M, D, p->w_co, D, w->da2);
  RUN(launch_gemm(gb, EPI_BIAS, false, true, st));
  AttnBatch ab;
  memset(&ab, 0, sizeof(ab));
  ab.n = 1; ab.hd = hd; ab.causal = 0; ab.drop_p = drop_p;
  {
    AttnProb& q = ab.p[0];
    const bf16* kv = (const bf16*)a->kvc;
    bf16* dkv = (bf16*)w->dkvc;
    q.Q = (const bf16*)a->qc; q.ldq = D; q.K = kv; q.V = kv + D; q.ldk = q.ldv = 2 * D;
    q.kbias = a->mem_bias; q.O = (bf16*)a->attn2; q.ldo = D; q.lse = a->lse2;
    q.dO = (const bf16*)w->da2; q.lddo = D;
    q.delta = w->delta2 ? w->delta2 : w->delta;   // own row sums when the dK/dV pass may run beside the self-attention
    q.dQ = (bf16*)w->dqc; q.lddq = D; q.dK = dkv; q.dV = dkv + D; q.lddk = q.lddv = 2 * D;
    q.B = a->B; q.H = nhead; q.Tq = a->T; q.Sk = a->L; q.seed = seed + 2;
  }
  if (wst && !w->delta2) return MMST_ERR_ARG;
  hipEvent_t q_done = wst ? arm_completion_event() : nullptr;   // of the dQ kernel: it also leaves delta
  RUN(launch_attn_bwd_q(ab, st));
  hipEvent_t dmem_done = nullptr;
  {
    hipStream_t kst = wst ? wst : st;
    if (wst) RUN(wait_for(wst, q_done));
    RUN(launch_attn_bwd_kv(ab, kst));
    GemmBatch gk;
    gk.n = 1;
    GemmProb q = dgrad(w->dkvc, ML, 2 * D, (const bf16*)p->w_cqkv + D * D, D, w->dmem);
    if (w->dmem_accumulate) { q.R = (const bf16*)w->dmem; q.ldr = D; }
    gk.p[0] = q;
    RUN(launch_gemm(gk, EPI_BIAS, false, true, kst));
    if (wst && join) RUN(stream_fork(wst, nullptr, &dmem_done));
  }
  {
    GemmProb q = dgrad(w->dqc, M, D, p->w_cqkv, D, w->dz1);
    q.R = (const bf16*)w->dz2; q.ldr = D; q.Z = (bf16*)a->z1; q.stats = a->st1; q.gamma = p->g1;
    q.C2 = dr ? (bf16*)w->dzd1 : nullptr; q.dgamma = g->g1; q.dbeta = g->be1;
    q.drop_p = drop_p; q.seed = seed + 1;
    gb.p[0] = q;
  }
  RUN(launch_gemm(gb, EPI_LNBWD, false, true, st));
  const void* dzd1 = dr ? w->dzd1 : w->dz1;
  // self-attention block
  gb.p[0] = dgrad(dzd1, M, D, p->w_o, D, w->da);
  RUN(launch_gemm(gb, EPI_BIAS, false, true, st));
  ab.causal = a->no_causal ? 0 : 1;
  {
    AttnProb& q = ab.p[0];
    memset(&q, 0, sizeof(q));
    const bf16* qkv = (const bf16*)a->qkv;
    bf16* dqkv = (bf16*)w->dqkv;
    q.Q = qkv; q.K = qkv + D; q.V = qkv + 2 * D; q.ldq = q.ldk = q.ldv = 3 * D;
    q.O = (bf16*)a->attn1; q.ldo = D; q.lse = a->lse1;
    q.dO = (const bf16*)w->da; q.lddo = D; q.delta = w->delta;
    q.dQ = dqkv; q.dK = dqkv + D; q.dV = dqkv + 2 * D; q.lddq = q.lddk = q.lddv = 3 * D;
    q.B = a->B; q.H = nhead; q.Tq = a->T; q.Sk = a->T; q.seed = seed + 0;
  }
  RUN(launch_attn_bwd(ab, st));
  {
    GemmProb q = dgrad(w->dqkv, M, 3 * D, p->w_qkv, D, w->dx);
    q.R = (const bf16*)w->dz1; q.ldr = D;
    bool prev = w->prev_z3 != nullptr;
    if (prev) {
      q.Z = (bf16*)w->prev_z3; q.stats = (float*)w->prev_st3; q.gamma = w->prev_g3;
      q.C2 = dr ? (bf16*)w->dxd : nullptr; q.dgamma = w->prev_dg3; q.dbeta = w->prev_dbe3;
      q.drop_p = drop_p; q.seed = w->prev_seed + 5;
    }
    gb.p[0] = q;
    if (done) *done = arm_completion_event();
    RUN(launch_gemm(gb, prev ? EPI_LNBWD : EPI_BIAS, false, true, st));
  }
  if (dmem_done && hipStreamWaitEvent(st, dmem_done, 0) != hipSuccess) return MMST_ERR_LAUNCH;
  return MMST_OK;
}

}  // namespace

extern "C" {

// ---- decoder layer backward, critical path ----
int mmst_decoder_layer_bwd_data(const mmst_dec_params* p, const mmst_dec_grads* g, const mmst_dec_acts* a,
                                const mmst_dec_bwd* w, int nhead, float drop_p, uint32_t seed, void* stream_) {
  return dec_bwd_data(p, g, a, w, nhead, drop_p, seed, (hipStream_t)stream_, nullptr, false);
}

// ---- decoder layer backward, parameter gradients: one grouped split-K launch ----
int mmst_decoder_layer_bwd_weights(const mmst_dec_grads* g, const mmst_dec_acts* a, const mmst_dec_bwd* w,
                                   float drop_p, void* stream_) {
  if (!g || !a || !w) return MMST_ERR_ARG;
  const int M = a->B * a->T, ML = a->B * a->L;
  const bool dr = drop_p > 0.f;
  const void* dzd2 = dr ? w->dzd2 : w->dz2;
  const void* dzd1 = dr ? w->dzd1 : w->dz1;
  GemmProb probs[7];
  probs[0] = wgrad(w->dzd3, M, D, a->h, FF, g->w_2, g->b_2);
  probs[1] = wgrad(w->dh, M, FF, a->x2, D, g->w_1, g->b_1);
  probs[2] = wgrad(dzd2, M, D, a->attn2, D, g->w_co, g->b_co);
  probs[3] = wgrad(w->dqc, M, D, a->x1, D, g->w_cqkv, g->b_cqkv);
  probs[4] = wgrad(w->dkvc, ML, 2 * D, a->mem, D, g->w_cqkv + D * D, g->b_cqkv + D);
  probs[5] = wgrad(dzd1, M, D, a->attn1, D, g->w_o, g->b_o);
  probs[6] = wgrad(w->dqkv, M, 3 * D, a->x, D, g->w_qkv, g->b_qkv);
  // One workgroup per CU: the small, register-heavy kernels of the decoder's data pass (the row epilogue of the
  // split feed-forward block above all) run next to this launch and would otherwise wait for its grid to drain.
  // ... and with few of them (80 instead of the usual ~512, each walking more of the 4032 rows): the launch then takes
  // most of a layer's data pass but leaves two thirds of the CUs to it (8.27 -> 8.18 ms per training step; 32: 8.25,
  // 16: 8.55).  Environment: MMST_DEC_WGRAD_BLOCKS.
  static const int target = [] { const char* e = getenv("MMST_DEC_WGRAD_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 80; }();
  return launch_wgrads(probs, 7, (hipStream_t)stream_, 20000, target);
}

// ---- the whole decoder backward on two streams, layers nl - 1 .. 0 (w[l]->dz3 / dzd3 of the top layer given, w[l]->dx
// of every other layer = w[l - 1]->dz3, w[l]->prev_seed = seeds[l - 1]).  Same results as nl calls of
// mmst_decoder_layer_bwd_streams from the top layer down with join_dmem set for layer 0.  Short target sequences
// (T <= 64, causal, fragment-major transposed weight images present): the activation gradients of a layer take five
// launches instead of nine (dec_fused.hip): feed-forward backward per hidden quarter with the LayerNorm-3 backward of
// the layer's incoming gradient in front, norm2 backward + cross out_proj backward, cross-attention dQ, norm1 backward
// + self out_proj backward, self-attention backward ----
int mmst_decoder_stack_bwd_streams(int nl, const mmst_dec_params* const* p, const mmst_dec_grads* const* g,
                                   const mmst_dec_acts* const* a, const mmst_dec_bwd* const* w, int nhead, float drop_p,
                                   const uint32_t* seeds, void* stream_, void* wstream_) {
  if (nl < 1 || nl > MMST_MAX_LAYERS || !p || !g || !a || !w || !seeds || nhead != 16 || !wstream_ || stream_ == wstream_)
    return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_, wst = (hipStream_t)wstream_;
  const bool dr = drop_p > 0.f;
  bool fused = dec_fused_enabled();
  for (int l = 0; l < nl; ++l) {
    if (!p[l] || !g[l] || !a[l] || !w[l]) return MMST_ERR_ARG;
    const mmst_dec_acts& A = *a[l];
    if (A.T < 1 || A.T > 64 || A.no_causal || A.B != a[0]->B || A.T != a[0]->T || A.L != a[0]->L || !A.ffn_ws) fused = false;
    if (!p[l]->ft_qkv || !p[l]->ft_o || !p[l]->ft_cqkv || !p[l]->ft_co || !p[l]->ft_1 || !p[l]->ft_2 || !w[l]->delta2)
      fused = false;
    if (l > 0 && (w[l]->dx != w[l - 1]->dz3 || w[l]->prev_z3 != a[l - 1]->z3 || (dr && w[l]->dxd != w[l - 1]->dzd3)))
      fused = false;
    if (l > 0 && w[l]->prev_seed != seeds[l - 1]) fused = false;
  }
  if (!fused) {
    for (int l = nl - 1; l >= 0; --l)
      RUN(mmst_decoder_layer_bwd_streams(p[l], g[l], a[l], w[l], nhead, drop_p, seeds[l], l == 0 ? 1 : 0, stream_, wstream_));
    return MMST_OK;
  }
  const int B = a[0]->B, T = a[0]->T, Lm = a[0]->L, M = B * T, ML = B * Lm, hd = D / nhead;
  hipEvent_t dmem_done = nullptr;
  hipEvent_t layer_done[MMST_MAX_LAYERS];
  static const bool use_third = [] { const char* e = getenv("MMST_DEC_THIRD_STREAM"); return e ? atoi(e) != 0 : true; }();
  // norm1 backward, the self-attention backward and the projection down to the layer below in ONE launch per layer
  // (MMST_DEC_SELF_FUSED=0: three launches, the attention among them as a launch of attn.hip)
  static const bool self_fused = [] { const char* e = getenv("MMST_DEC_SELF_FUSED"); return e ? atoi(e) != 0 : true; }();
  hipStream_t xst = use_third ? third_stream() : nullptr;
  for (int l = nl - 1; l >= 0; --l) {
    const mmst_dec_params& P = *p[l];
    const mmst_dec_grads& G = *g[l];
    const mmst_dec_acts& A = *a[l];
    const mmst_dec_bwd& W = *w[l];
    {   // feed-forward backward; below the top layer it first turns the layer above's dqkv into this layer's dz3 / dzd3
      DecFusedBwd m;
      memset(&m, 0, sizeof(m));
      m.p = P; m.g = G; m.a = A; m.w = W; m.seed = seeds[l];
      if (l + 1 < nl && !self_fused) { m.up_dqkv = w[l + 1]->dqkv; m.up_dz1 = w[l + 1]->dz1; m.up_ft_qkv = p[l + 1]->ft_qkv; }
      m.B = B; m.T = T; m.drop_p = drop_p; m.ws = A.ffn_ws;
      RUN(launch_dec_bwd_mid(m, st));
    }
    {   // norm2 backward, cross-attention out_proj backward
      DecFusedLn k;
      memset(&k, 0, sizeof(k));
      k.ws = A.ffn_ws; k.res = W.dz3; k.z = A.z2; k.stats = A.st2; k.gamma = P.g2; k.dgamma = G.g2; k.dbeta = G.be2;
      k.dz = W.dz2; k.dzd = W.dzd2; k.ft_out = P.ft_co; k.out = W.da2; k.seed = seeds[l] + 3;
      k.B = B; k.T = T; k.drop_p = drop_p;
      RUN(launch_dec_bwd_ln(k, st));
    }
    AttnBatch ab;
    memset(&ab, 0, sizeof(ab));
    ab.n = 1; ab.hd = hd; ab.causal = 0; ab.drop_p = drop_p;
    {
      AttnProb& q = ab.p[0];
      const bf16* kv = (const bf16*)A.kvc;
      bf16* dkv = (bf16*)W.dkvc;
      q.Q = (const bf16*)A.qc; q.ldq = D; q.K = kv; q.V = kv + D; q.ldk = q.ldv = 2 * D;
      q.kbias = A.mem_bias; q.O = (bf16*)A.attn2; q.ldo = D; q.lse = A.lse2;
      q.dO = (const bf16*)W.da2; q.lddo = D; q.delta = W.delta2;
      q.dQ = (bf16*)W.dqc; q.lddq = D; q.dK = dkv; q.dV = dkv + D; q.lddk = q.lddv = 2 * D;
      q.B = B; q.H = nhead; q.Tq = T; q.Sk = Lm; q.seed = seeds[l] + 2;
    }
    hipEvent_t q_done = arm_completion_event();   // of the dQ kernel: it also leaves delta2
    hipEvent_t kv_done = nullptr;
    RUN(launch_attn_bwd_q(ab, st));
    {   // the memory's dK / dV and their projection into dmem feed the encoder only: second stream
      RUN(wait_for(wst, q_done));
      if (xst) kv_done = arm_completion_event();   // the parameter gradients read dkvc
      RUN(launch_attn_bwd_kv(ab, wst));
      GemmBatch gk;
      gk.n = 1;
      GemmProb q = dgrad(W.dkvc, ML, 2 * D, (const bf16*)P.w_cqkv + D * D, D, W.dmem);
      if (W.dmem_accumulate) { q.R = (const bf16*)W.dmem; q.ldr = D; }
      gk.p[0] = q;
      RUN(launch_gemm(gk, EPI_BIAS, false, true, wst));
      if (l == 0) RUN(stream_fork(wst, nullptr, &dmem_done));
    }
    hipEvent_t done = nullptr;
    if (self_fused) {
      DecFusedSelf k;
      memset(&k, 0, sizeof(k));
      k.dqc = W.dqc; k.ft_cq = P.ft_cqkv; k.dz2 = W.dz2; k.z1 = A.z1; k.st1 = A.st1; k.g1 = P.g1; k.dg1 = G.g1; k.dbe1 = G.be1;
      k.dz1 = W.dz1; k.dzd1 = W.dzd1; k.ft_o = P.ft_o;
      k.qkv = A.qkv; k.attn1 = A.attn1; k.lse1 = A.lse1; k.dqkv = W.dqkv; k.ft_qkv = P.ft_qkv;
      if (l > 0) {
        k.z3 = a[l - 1]->z3; k.st3 = a[l - 1]->st3; k.g3 = p[l - 1]->g3; k.dg3 = g[l - 1]->g3; k.dbe3 = g[l - 1]->be3;
        k.dz3 = (void*)w[l - 1]->dz3; k.dzd3 = (void*)w[l - 1]->dzd3; k.seed_below = seeds[l - 1];
      } else {
        k.dx = W.dx;
      }
      k.seed = seeds[l]; k.B = B; k.T = T; k.drop_p = drop_p;
      done = arm_completion_event();
      RUN(launch_dec_bwd_self(k, st));
    } else {
    {   // norm1 backward, self-attention out_proj backward
      DecFusedLn k;
      memset(&k, 0, sizeof(k));
      k.in = W.dqc; k.ft_in = P.ft_cqkv; k.res = W.dz2; k.z = A.z1; k.stats = A.st1; k.gamma = P.g1; k.dgamma = G.g1;
      k.dbeta = G.be1; k.dz = W.dz1; k.dzd = W.dzd1; k.ft_out = P.ft_o; k.out = W.da; k.seed = seeds[l] + 1;
      k.B = B; k.T = T; k.drop_p = drop_p;
      RUN(launch_dec_bwd_ln(k, st));
    }
    ab.causal = 1;
    {
      AttnProb& q = ab.p[0];
      memset(&q, 0, sizeof(q));
      const bf16* qkv = (const bf16*)A.qkv;
      bf16* dqkv = (bf16*)W.dqkv;
      q.Q = qkv; q.K = qkv + D; q.V = qkv + 2 * D; q.ldq = q.ldk = q.ldv = 3 * D;
      q.O = (bf16*)A.attn1; q.ldo = D; q.lse = A.lse1;
      q.dO = (const bf16*)W.da; q.lddo = D; q.delta = W.delta;
      q.dQ = dqkv; q.dK = dqkv + D; q.dV = dqkv + 2 * D; q.lddq = q.lddk = q.lddv = 3 * D;
      q.B = B; q.H = nhead; q.Tq = T; q.Sk = T; q.seed = seeds[l] + 0;
    }
    if (l > 0) done = arm_completion_event();
    RUN(launch_attn_bwd(ab, st));
    if (l == 0) {   // the bottom layer: gradient at the target embedding
      GemmBatch gb;
      gb.n = 1;
      GemmProb q = dgrad(W.dqkv, M, 3 * D, P.w_qkv, D, W.dx);
      q.R = (const bf16*)W.dz1; q.ldr = D;
      gb.p[0] = q;
      done = arm_completion_event();
      RUN(launch_gemm(gb, EPI_BIAS, false, true, st));
    }
    }
    // The layer's parameter gradients: not on wstream, where they would sit in front of the next layer's dK / dV pass
    // (the chain of those passes, each waiting for its layer's dQ kernel, is what the encoder's backward waits for),
    // but on a third stream, beside both chains; wstream takes them in at the end.
    if (xst) {
      RUN(wait_for(xst, done));
      RUN(wait_for(xst, kv_done));
      RUN(mmst_decoder_layer_bwd_weights(g[l], a[l], w[l], drop_p, xst));
    } else {
      layer_done[l] = done;
    }
  }
  if (xst) {
    RUN(stream_fork(xst, wst));
  } else {
    for (int l = nl - 1; l >= 0; --l) {
      RUN(wait_for(wst, layer_done[l]));
      RUN(mmst_decoder_layer_bwd_weights(g[l], a[l], w[l], drop_p, wstream_));
    }
  }
  if (dmem_done && hipStreamWaitEvent(st, dmem_done, 0) != hipSuccess) return MMST_ERR_LAUNCH;
  return MMST_OK;
}

int mmst_decoder_layer_bwd(const mmst_dec_params* p, const mmst_dec_grads* g, const mmst_dec_acts* a,
                           const mmst_dec_bwd* w, int nhead, float drop_p, uint32_t seed, void* stream_) {
  RUN(mmst_decoder_layer_bwd_data(p, g, a, w, nhead, drop_p, seed, stream_));
  return mmst_decoder_layer_bwd_weights(g, a, w, drop_p, stream_);
}

// ---- both halves on two streams: the activation gradients the next layer waits for on `stream`; the memory's
// dK/dV, their projection into dmem and every parameter gradient on `wstream`.  join_dmem != 0: `stream` also
// waits for the dmem update (set it for the last layer processed, before dmem is read) ----
int mmst_decoder_layer_bwd_streams(const mmst_dec_params* p, const mmst_dec_grads* g, const mmst_dec_acts* a,
                                   const mmst_dec_bwd* w, int nhead, float drop_p, uint32_t seed, int join_dmem,
                                   void* stream_, void* wstream_) {
  if (!wstream_ || stream_ == wstream_) return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_, wst = (hipStream_t)wstream_;
  hipEvent_t done = nullptr;
  RUN(dec_bwd_data(p, g, a, w, nhead, drop_p, seed, st, wst, join_dmem != 0, &done));
  RUN(wait_for(wst, done));
  return mmst_decoder_layer_bwd_weights(g, a, w, drop_p, wstream_);
}

}  // extern "C"
