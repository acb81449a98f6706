// Beam-search bookkeeping on the device (inference): one step of validate_generate_MMT_v15_4.py:995-1053 for every
// molecule at once, so that the decode loop never waits for the host.  The reference keeps, per molecule, a list of
// (score, token list, probability list); a step expands every unfinished beam by the beam_size most probable next
// tokens of softmax(logits) (no temperature, :1038), carries finished beams (last token <EOS>) over unchanged,
// multiplies scores by probabilities (in Python floats, i.e. double), sorts the candidates by score with a STABLE
// descending sort (:1052) and keeps the first beam_size.  Here a wave owns a molecule and does the same on
// fixed-size arrays; beams also carry a history table that tells the decoder kernel in which sequence's cache the
// key/value rows of their prefix live, so inheriting a prefix copies beam_size*pos integers instead of cache rows.
#include "../../include/mmst.h"
#include "common.h"

namespace {

constexpr int MAXW = MMST_MAX_BEAM, MAXC = MAXW + MAXW * MAXW;

__global__ void beam_init_kernel(mmst_beam_state st, int64_t* tok, int W, int cap, int sos) {
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int w = 0; w < W; ++w) {
    const size_t s = (size_t)b * W + w;
    for (int t = lane; t < cap; t += 64) st.hist[s * cap + t] = w;
    if (lane == 0) {
      st.seqs[s * (cap + 1)] = sos;
      st.lens[s] = 1;
      st.scores[s] = 1.0;
      tok[s] = sos;       // every slot decodes <SOS> at position 0; only beam 0 exists yet
    }
  }
  if (lane == 0) st.nbeams[b] = 1;
}

// one wave per molecule
__global__ __launch_bounds__(64) void beam_step_kernel(const float* logits, int ldl, int V, mmst_beam_state cur,
                                                       mmst_beam_state nxt, int64_t* tok, int W, int cap, int pos, int eos) {
  __shared__ double c_score[MAXC];
  __shared__ int c_parent[MAXC], c_tok[MAXC];
  __shared__ float c_prob[MAXC];
  __shared__ int sel[MAXW];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nb = cur.nbeams[b];
  int nc = 0;   // candidates so far, in the reference's generation order (uniform across the wave)
  for (int j = 0; j < nb; ++j) {
    const size_t s = (size_t)b * W + j;
    const int len = cur.lens[s];
    const double score = cur.scores[s];
    if (cur.seqs[s * (cap + 1) + len - 1] == eos) {   // finished: carried over unchanged
      if (lane == 0) { c_score[nc] = score; c_parent[nc] = j; c_tok[nc] = -1; c_prob[nc] = 0.f; }
      ++nc;
      continue;
    }
    // softmax over the vocabulary in fp32 (lane = token), then the W largest, ties to the lower index
    const float x = lane < V ? logits[s * ldl + lane] : -INFINITY;
    const float mx = wave_max(x);
    const float e = lane < V ? __expf(x - mx) : 0.f;
    const float se = wave_sum(e);
    float p = lane < V ? e / se : -1.f;
    for (int k = 0; k < W; ++k) {
      const float best = wave_max(p);
      const unsigned long long mk = __ballot(p == best);
      const int arg = __ffsll((long long)mk) - 1;
      if (lane == arg) {
        c_score[nc] = score * (double)p;
        c_parent[nc] = j;
        c_tok[nc] = arg;
        c_prob[nc] = p;
        p = -1.f;
      }
      ++nc;
    }
  }
  __syncthreads();
  // stable descending selection of the first W: repeatedly the largest remaining score, earliest candidate on ties
  const int keep = nc < W ? nc : W;
  if (lane == 0) {
    unsigned used = 0;
    for (int r = 0; r < keep; ++r) {
      int best = -1;
      for (int c = 0; c < nc; ++c)
        if (!(used >> c & 1) && (best < 0 || c_score[c] > c_score[best])) best = c;
      used |= 1u << best;
      sel[r] = best;
    }
  }
  __syncthreads();
  for (int r = 0; r < W; ++r) {
    const size_t d = (size_t)b * W + r;
    if (r >= keep) {   // cannot happen after the first step (W * W candidates); keep the slot harmless
      if (lane == 0) { tok[d] = eos; nxt.lens[d] = 1; nxt.scores[d] = 0.0; nxt.seqs[d * (cap + 1)] = eos; }
      for (int t = lane; t < cap; t += 64) nxt.hist[d * cap + t] = r;
      continue;
    }
    const int c = sel[r], j = c_parent[c], tk = c_tok[c];
    const size_t s = (size_t)b * W + j;
    const int len = cur.lens[s];
    for (int t = lane; t < len; t += 64) nxt.seqs[d * (cap + 1) + t] = cur.seqs[s * (cap + 1) + t];
    for (int t = lane; t < len - 1; t += 64) nxt.probs[d * cap + t] = cur.probs[s * cap + t];
    // keys/values of positions < pos live where the parent's prefix lives; position pos was computed in the parent's slot
    for (int t = lane; t < cap; t += 64) nxt.hist[d * cap + t] = t < pos ? cur.hist[s * cap + t] : (t == pos ? j : r);
    if (lane == 0) {
      if (tk >= 0) {
        nxt.seqs[d * (cap + 1) + len] = tk;
        nxt.probs[d * cap + len - 1] = c_prob[c];
        nxt.lens[d] = len + 1;
        tok[d] = tk;
      } else {
        nxt.lens[d] = len;
        tok[d] = eos;   // a finished beam still occupies a slot of the decoder step; its output is never read
      }
      nxt.scores[d] = c_score[c];
    }
  }
  if (lane == 0) nxt.nbeams[b] = keep;
}

bool state_ok(const mmst_beam_state* s) {
  return s && s->seqs && s->probs && s->scores && s->lens && s->nbeams && s->hist;
}

}  // namespace

extern "C" {

int mmst_beam_init(const mmst_beam_state* st, int64_t* tok, int B, int W, int cap, int sos, void* stream) {
  if (!state_ok(st) || !tok || B < 1 || W < 1 || W > MAXW || cap < 1) return MMST_ERR_ARG;
  hipLaunchKernelGGL(beam_init_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *st, tok, W, cap, sos);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_beam_step(const float* logits, int ld_logits, int V, const mmst_beam_state* cur, const mmst_beam_state* next,
                   int64_t* tok, int B, int W, int cap, int pos, int eos, void* stream) {
  if (!logits || !state_ok(cur) || !state_ok(next) || !tok || B < 1 || W < 1 || W > MAXW || V < W || V > 64 ||
      ld_logits < V || pos < 0 || pos >= cap)   // the step writes seqs[.][pos + 1] (of cap + 1), probs[.][pos], hist[.][< cap]
    return MMST_ERR_ARG;
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ld_logits, V, *cur, *next, tok,
                     W, cap, pos, eos);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

}  // extern "C"
