// Bandwidth-bound kernels around the transformer layers: spectrum / formula / weight embeddings and
// sequence assembly, sequence mean, target embedding, cross-entropy, greedy token selection, AdamW.
#include "../../include/mmst.h"
#include "common.h"

namespace {

constexpr int D = MMST_D;

struct EmbedArgs {
  int n;
  int row_begin[MMST_MAX_GROUPS + 1];  // prefix of B*S rows per group
  mmst_embed_group g[MMST_MAX_GROUPS];
  mmst_embed_shared s;
};

// additive bias of one mask component inside a concatenated mask:
// all-bool concatenation -> -inf where set; otherwise bool components promote to 0/1 floats and float
// components are added as they are.
__device__ __forceinline__ float mask_bias(float mv, bool comp_bool, bool target_bool) {
  if (target_bool) return mv != 0.f ? -INFINITY : 0.f;
  return comp_bool ? (mv != 0.f ? 1.f : 0.f) : mv;
}

// one thread = one (row, 8-column chunk); 16 threads per row
__global__ void embed_fwd_kernel(const EmbedArgs A) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = gid >> 4, c0 = (gid & 15) * 8;
  if (row >= A.row_begin[A.n]) return;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < MMST_MAX_GROUPS; ++i)
    if (i < A.n && row >= A.row_begin[i]) gi = i;
  const mmst_embed_group& G = A.g[gi];
  const mmst_embed_shared& S = A.s;
  const int lr = row - A.row_begin[gi];
  const int b = lr / G.S, s = lr % G.S;
  float v[8];
  float mv = 0.f;
  bool comp_bool = true, write_x = true;
  int seg = s;
  if (seg < G.n_spec) {
    if (G.in_dim == 0) {
      write_x = false;  // IR row: produced by the IR GEMM
    } else {
      const float* x = G.src + ((size_t)b * 64 + seg) * G.in_dim;
      const float x0 = x[0], x1 = G.in_dim == 2 ? x[1] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* wr = G.w + (size_t)(c0 + e) * G.in_dim;
        float a = wr[0] * x0 + G.b[c0 + e];
        if (G.in_dim == 2) a += wr[1] * x1;
        v[e] = fmaxf(a, 0.f);
      }
      mv = G.mask[(size_t)b * 64 + seg];
    }
  } else {
    seg -= G.n_spec;
    bool done = false;
    if (S.has_mf) {
      if (seg < 64) {
        const int64_t tok = S.mf_tok[(size_t)b * 64 + seg];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(S.mf_table[(size_t)tok * D + c0 + e], 0.f);
        mv = S.mf_mask[(size_t)b * 64 + seg];
        comp_bool = S.mf_bool != 0;
        done = true;
      } else seg -= 64;
    }
    if (!done && S.has_ms) {
      if (seg < 64) {
        const int64_t tok = S.ms_tok[(size_t)b * 64 + seg];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(S.ms_table[(size_t)tok * D + c0 + e], 0.f);
        mv = S.ms_mask[(size_t)b * 64 + seg];
        comp_bool = S.ms_bool != 0;
        done = true;
      } else seg -= 64;
    }
    if (!done) {  // molecular-weight token
      const float mw = S.mw[b];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(S.mw_w[c0 + e] * mw + S.mw_b[c0 + e], 0.f);
    }
  }
  if (write_x) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
    *reinterpret_cast<bf16x8*>((bf16*)G.xseq + (size_t)lr * D + c0) = o;
  }
  if (c0 == 0) {
    G.kbias[(size_t)b * G.S + s] = mask_bias(mv, comp_bool, G.mod_bool != 0);
    if (G.kbias_glob) G.kbias_glob[(size_t)b * G.L_glob + G.off_glob + s] = mask_bias(mv, comp_bool, S.glob_bool != 0);
  }
}

struct EmbedBwdArgs {
  EmbedArgs e;
  const bf16* dx[MMST_MAX_GROUPS];
};

// one block = one (group, batch element); 8 row groups x 128 columns: the sequence positions are strided over the
// row groups (every load of a thread is independent), point-embedding sums are combined through LDS (one atomic per
// column per block), embedding-table rows are scatter-added.
__global__ __launch_bounds__(1024) void embed_bwd_kernel(const EmbedBwdArgs A) {
  __shared__ float red[3][8][D];
  const int gi = blockIdx.y, b = blockIdx.x, c = threadIdx.x & (D - 1), grp = threadIdx.x >> 7;
  const mmst_embed_group& G = A.e.g[gi];
  const mmst_embed_shared& S = A.e.s;
  const bf16* dx = A.dx[gi] + (size_t)b * G.S * D;
  const bf16* x = (const bf16*)G.xseq + (size_t)b * G.S * D;
  int s = 0;
  if (G.in_dim > 0) {
    float dw0 = 0.f, dw1 = 0.f, db = 0.f;
    for (int r = grp; r < 64; r += 8) {
      float gsc = (float)x[r * D + c] > 0.f ? (float)dx[r * D + c] : 0.f;
      const float* xin = G.src + ((size_t)b * 64 + r) * G.in_dim;
      dw0 += gsc * xin[0];
      if (G.in_dim == 2) dw1 += gsc * xin[1];
      db += gsc;
    }
    red[0][grp][c] = dw0;
    red[1][grp][c] = dw1;
    red[2][grp][c] = db;
    __syncthreads();
    if (grp == 0) {
      dw0 = dw1 = db = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { dw0 += red[0][i][c]; dw1 += red[1][i][c]; db += red[2][i][c]; }
      atomicAdd(G.dw + (size_t)c * G.in_dim, dw0);
      if (G.in_dim == 2) atomicAdd(G.dw + (size_t)c * G.in_dim + 1, dw1);
      atomicAdd(G.db + c, db);
    }
    s = 64;
  } else {
    // IR row: its weight gradient is a GEMM; emit the gradient in front of the ReLU for it
    if (G.ir_dpre && grp == 0) ((bf16*)G.ir_dpre)[(size_t)b * D + c] = (float)x[c] > 0.f ? dx[c] : (bf16)0.f;
    s = G.n_spec;
  }
  if (S.has_mf) {
    for (int t = grp; t < 64; t += 8) {
      const int64_t tok = S.mf_tok[(size_t)b * 64 + t];
      const int r = s + t;
      if (tok != 0 && (float)x[r * D + c] > 0.f) atomicAdd(S.d_mf_table + (size_t)tok * D + c, (float)dx[r * D + c]);
    }
    s += 64;
  }
  if (S.has_ms) {
    for (int t = grp; t < 64; t += 8) {
      const int64_t tok = S.ms_tok[(size_t)b * 64 + t];
      const int r = s + t;
      if (tok != 0 && (float)x[r * D + c] > 0.f) atomicAdd(S.d_ms_table + (size_t)tok * D + c, (float)dx[r * D + c]);
    }
    s += 64;
  }
  if (S.has_mw && grp == 0) {
    float gsc = (float)x[s * D + c] > 0.f ? (float)dx[s * D + c] : 0.f;
    atomicAdd(S.d_mw_w + c, gsc * S.mw[b]);
    atomicAdd(S.d_mw_b + c, gsc);
  }
}

// mean over L rows: block = (batch element), 128 threads = columns, each strides over rows
__global__ void seq_mean_kernel(const bf16* mem, bf16* out, int L) {
  const int b = blockIdx.x, c = threadIdx.x & 127, part = threadIdx.x >> 7, nparts = blockDim.x >> 7;
  __shared__ float red[8][128];
  float s = 0.f;
  const bf16* p = mem + (size_t)b * L * D;
  for (int r = part; r < L; r += nparts) s += (float)p[(size_t)r * D + c];
  red[part][c] = s;
  __syncthreads();
  if (part == 0) {
    for (int i = 1; i < nparts; ++i) s += red[i][c];
    out[(size_t)b * D + c] = (bf16)(s / (float)L);
  }
}

// token of (position t, sequence b) = tok[t * tok_ld + b * tok_bs]: (T, B) row-major is (B, 1), batch-first (B, 64) is (1, 64)
__global__ void trg_embed_fwd_kernel(const int64_t* tok, int tok_ld, int tok_bs, const float* embed, const float* pos,
                                     bf16* out, int B, int T, float drop_p, uint32_t seed, float* clear) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid == 0 && clear) *clear = 0.f;   // the step's loss accumulator, long before the loss kernel adds to it
  const int row = gid >> 4, c0 = (gid & 15) * 8;
  if (row >= B * T) return;
  const int b = row / T, t = row % T;
  const int64_t tk = tok[(size_t)t * tok_ld + (size_t)b * tok_bs];
  const uint32_t thr = drop_threshold(drop_p);
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = embed[(size_t)tk * D + c0 + e] + pos[(size_t)t * D + c0 + e];
    if (drop_p > 0.f) v = drop_keep(seed, (uint32_t)row * D + c0 + e, thr) ? v * sc : 0.f;
    o[e] = (bf16)v;
  }
  *reinterpret_cast<bf16x8*>(out + (size_t)row * D + c0) = o;
}

// Gradient of the two target-embedding tables: one workgroup per (table row, slice of the token rows), 8 row-groups
// x 128 columns per workgroup, token ids staged in LDS; the EMB_SLICES partial sums of a table row meet in fp32 atomics.
// blocks [0, T)                    : d_pos[t]  += sum over the batch of dout[b][t]
// blocks [T, T + V * EMB_SLICES)   : d_embed[v] += sum over the (b, t) of one slice whose token is v
#define EMB_SLICES 8
__global__ void trg_embed_bwd_kernel(const int64_t* tok, int tok_ld, int tok_bs, const bf16* dout, float* d_embed,
                                     float* d_pos, int B, int T, int V, float drop_p, uint32_t seed) {
  __shared__ short toks[1024];
  __shared__ float red[8][D];
  const int c = threadIdx.x & (D - 1), grp = threadIdx.x >> 7;
  const uint32_t thr = drop_threshold(drop_p);
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int rows = B * T;
  float acc = 0.f;
  if ((int)blockIdx.x < T) {
    const int t = blockIdx.x;
    for (int b = grp; b < B; b += 8) {
      const int row = b * T + t;
      float g = (float)dout[(size_t)row * D + c];
      if (drop_p > 0.f) g = drop_keep(seed, (uint32_t)row * D + c, thr) ? g * sc : 0.f;
      acc += g;
    }
  } else {
    const int v = (blockIdx.x - T) / EMB_SLICES, slice = (blockIdx.x - T) % EMB_SLICES;
    const int per = (rows + EMB_SLICES - 1) / EMB_SLICES;
    const int lo = slice * per, hi = min(rows, lo + per);
    for (int base = lo; base < hi; base += 1024) {
      const int n = min(1024, hi - base);
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int row = base + i;
        toks[i] = (short)tok[(size_t)(row % T) * tok_ld + (size_t)(row / T) * tok_bs];
      }
      __syncthreads();
      for (int i = grp; i < n; i += 8) {
        if (toks[i] != v) continue;
        const int row = base + i;
        float g = (float)dout[(size_t)row * D + c];
        if (drop_p > 0.f) g = drop_keep(seed, (uint32_t)row * D + c, thr) ? g * sc : 0.f;
        acc += g;
      }
    }
  }
  red[grp][c] = acc;
  __syncthreads();
  if (grp == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[i][c];
    if ((int)blockIdx.x < T) d_pos[(size_t)blockIdx.x * D + c] += s;
    else if (s != 0.f) atomicAdd(d_embed + (size_t)((blockIdx.x - T) / EMB_SLICES) * D + c, s);
  }
}

// use_real_data branch (models_MMT_v15_4.py:965-971): logits <- (logits + extra[b]) / 2, extra = real_data_linear(mean)
__global__ void real_data_mix_kernel(float* logits, const float* extra, int T, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long b = i / ((long)T * 64);
  logits[i] = 0.5f * (logits[i] + extra[b * 64 + (i & 63)]);
}
// its backward, row part: d_extra[b][c] = sum_t dlogits[b][t][c] (dlogits already carries the factor 1/2),
// db[c] += d_extra[b][c], and an fp32 copy of mean[b] for the fp32 products that follow.  One block of 128 threads per b.
__global__ void real_data_rows_kernel(const bf16* dlogits, const bf16* mean, float* d_extra, float* mean32, float* db,
                                      int T) {
  const int b = blockIdx.x, c = threadIdx.x;
  mean32[(size_t)b * D + c] = (float)mean[(size_t)b * D + c];
  if (c < 64) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += (float)dlogits[((size_t)b * T + t) * 64 + c];
    d_extra[(size_t)b * 64 + c] = s;
    if (s != 0.f) atomicAdd(db + c, s);
  }
}
// dmem[b][l][:] += dmean[b][:] / L  (gradient of torch.mean(memory, dim=0) added to an existing memory gradient)
__global__ void seq_mean_bwd_add_kernel(const float* dmean, bf16* dmem, int L, long n) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  const long b = o / ((long)L * D);
  dmem[o] = (bf16)((float)dmem[o] + dmean[b * D + (o & (D - 1))] / L);
}

// one wave per row of logits (V <= 64), rows strided over the grid
__global__ void ce_loss_kernel(const float* logits, int ldl, int V, const int64_t* trg, int trg_ld, int B, int T,
                               float* loss, bf16* dlogits, float grad_scale) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const float inv_n = 1.0f / (float)(B * T);
  float part = 0.f;
  for (int row = wave; row < B * T; row += nwaves) {
    const int b = row / T, t = row % T;
    const float x = lane < V ? logits[(size_t)row * ldl + lane] : -INFINITY;
    const float mx = wave_max(x);
    const float e = lane < V ? __expf(x - mx) : 0.f;
    const float se = wave_sum(e);
    const int target = (int)trg[(size_t)b * trg_ld + t + 1];
    if (lane == target) part += (mx + __logf(se) - x) * inv_n;
    if (dlogits) {
      float g = lane < V ? (e / se - (lane == target ? 1.f : 0.f)) * inv_n * grad_scale : 0.f;
      dlogits[(size_t)row * 64 + lane] = (bf16)g;
    }
  }
  part = wave_sum(part);
  if (lane == 0) atomicAdd(loss, part);
}

// softmax(logits / temperature) of the last position, first-index argmax, its probability
__global__ void greedy_select_kernel(const float* logits, int ldl, int V, int B, int T, float temperature,
                                     int64_t* next_tok, float* next_prob) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  const float x = lane < V ? logits[((size_t)b * T + (T - 1)) * ldl + lane] / temperature : -INFINITY;
  const float mx = wave_max(x);
  const float e = lane < V ? __expf(x - mx) : 0.f;
  const float se = wave_sum(e);
  // first index attaining the maximum (torch.argmax tie rule)
  unsigned long long m = __ballot(x == mx);
  const int arg = __ffsll((long long)m) - 1;
  if (lane == arg) {
    next_tok[b] = arg;
    next_prob[b] = e / se;
  }
}

// softmax(logits / temperature) of the last position, one draw from it per row (wave_select_token)
__global__ void sample_select_kernel(const float* logits, int ldl, int V, int B, int T, float temperature, uint32_t seed,
                                     uint32_t counter, int64_t* next_tok, float* next_prob) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  const float x = lane < V ? logits[((size_t)b * T + (T - 1)) * ldl + lane] / temperature : -INFINITY;
  float pr;
  const int arg = wave_select_token(x, V, true, seed, counter, (uint32_t)b, &pr);
  if (lane == 0) {
    next_tok[b] = arg;
    next_prob[b] = pr;
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, bf16* __restrict__ pb, long n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    if (i + 4 <= n) {
      f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
      bf16x4 ob;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gr = gg[e] * gscale;
        float pe = pp[e] * (1.f - lr * wd);
        mm[e] = b1 * mm[e] + (1.f - b1) * gr;
        vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
        float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
        pe -= (lr / bc1) * (mm[e] / denom);
        pp[e] = pe;
        ob[e] = (bf16)pe;
      }
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
      if (pb) *reinterpret_cast<bf16x4*>(pb + i) = ob;
    } else {
      for (long j = i; j < n; ++j) {
        float gr = g[j] * gscale;
        float pe = p[j] * (1.f - lr * wd);
        float mj = b1 * m[j] + (1.f - b1) * gr;
        float vj = b2 * v[j] + (1.f - b2) * gr * gr;
        pe -= (lr / bc1) * (mj / (sqrtf(vj) / bc2_sqrt + eps));
        p[j] = pe; m[j] = mj; v[j] = vj;
        if (pb) pb[j] = (bf16)pe;
      }
    }
  }
}

// AdamW over several [lo, hi) ranges of the flat buffers in ONE launch; optionally leaves the gradient zeroed for
// the next step (saves a separate pass over the gradient buffer).  Ranges are multiples of 4 elements at offsets that
// are multiples of 4 (the arena aligns every tensor to 64 elements).
#define ADAMW_MAX_RANGES 16
struct AdamRanges {
  int n;
  long lo[ADAMW_MAX_RANGES];
  long cum[ADAMW_MAX_RANGES + 1];   // cum[i] = number of 4-element groups before range i
};
__global__ void adamw_ranges_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, bf16* __restrict__ pb, const AdamRanges R, float lr, float b1,
                                    float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale, int zero_g) {
  const long total = R.cum[R.n];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
    int r = 0;
    while (r + 1 < R.n && q >= R.cum[r + 1]) ++r;
    const long i = R.lo[r] + (q - R.cum[r]) * 4;
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
    f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
    bf16x4 ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gg[e] * gscale;
      float pe = pp[e] * (1.f - lr * wd);
      mm[e] = b1 * mm[e] + (1.f - b1) * gr;
      vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
      float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pe -= (lr / bc1) * (mm[e] / denom);
      pp[e] = pe;
      ob[e] = (bf16)pe;
    }
    *reinterpret_cast<f32x4*>(p + i) = pp;
    *reinterpret_cast<f32x4*>(m + i) = mm;
    *reinterpret_cast<f32x4*>(v + i) = vv;
    if (pb) *reinterpret_cast<bf16x4*>(pb + i) = ob;
    if (zero_g) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

__global__ void cast_kernel(const float* src, bf16* dst, int rows, int cols, int ld_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  dst[(size_t)r * ld_out + c] = (bf16)src[i];
}

// 64x64 tiles of bf16 matrices transposed through LDS (coalesced on both sides)
#define TRANSPOSE_MAX 128
struct TransposeArgs {
  const bf16* src; bf16* dst;
  int rows, cols;
  uint32_t off[TRANSPOSE_MAX];
};
__global__ __launch_bounds__(256) void transpose_kernel(const TransposeArgs A) {
  __shared__ bf16 tile[64][66];
  const bf16* src = A.src + A.off[blockIdx.z];
  bf16* dst = A.dst + A.off[blockIdx.z];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    if (r0 + r < A.rows && c0 + tx < A.cols) tile[r][tx] = src[(size_t)(r0 + r) * A.cols + c0 + tx];
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < A.cols && r0 + tx < A.rows) dst[(size_t)(c0 + c) * A.rows + r0 + tx] = tile[tx][c];
}

int fill_embed_args(EmbedArgs& A, int n, const mmst_embed_group* g, const mmst_embed_shared* s) {
  if (n < 1 || n > MMST_MAX_GROUPS || !g || !s || s->B < 1) return MMST_ERR_ARG;
  A.n = n;
  A.s = *s;
  int rows = 0;
  for (int i = 0; i < n; ++i) {
    int want = g[i].n_spec + (s->has_mf ? 64 : 0) + (s->has_ms ? 64 : 0) + (s->has_mw ? 1 : 0);
    if (g[i].S != want || (g[i].in_dim != 0 && g[i].n_spec != 64) || g[i].in_dim < 0 || g[i].in_dim > 2)
      return MMST_ERR_ARG;
    A.g[i] = g[i];
    A.row_begin[i] = rows;
    rows += s->B * g[i].S;
  }
  A.row_begin[n] = rows;
  for (int i = n + 1; i <= MMST_MAX_GROUPS; ++i) A.row_begin[i] = rows;
  return MMST_OK;
}

}  // namespace

// target-embedding launchers with a token stride per position and per sequence (see trg_embed_fwd_kernel)
int launch_trg_embed_fwd(const int64_t* tok, int tok_ld, int tok_bs, const float* embed, const float* pos, void* out,
                         int B, int T, float drop_p, uint32_t seed, hipStream_t stream, float* clear) {
  if (!tok || !embed || !pos || !out || B < 1 || T < 1) return MMST_ERR_ARG;
  const int threads = B * T * 16;
  hipLaunchKernelGGL(trg_embed_fwd_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, tok, tok_ld, tok_bs,
                     embed, pos, (bf16*)out, B, T, drop_p, seed, clear);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int launch_trg_embed_bwd(const int64_t* tok, int tok_ld, int tok_bs, const void* dout, float* d_embed, float* d_pos,
                         int B, int T, int V, float drop_p, uint32_t seed, hipStream_t stream) {
  if (!tok || !dout || !d_embed || !d_pos || B < 1 || T < 1 || V < 1) return MMST_ERR_ARG;
  hipLaunchKernelGGL(trg_embed_bwd_kernel, dim3(T + V * EMB_SLICES), dim3(8 * D), 0, stream, tok, tok_ld, tok_bs,
                     (const bf16*)dout, d_embed, d_pos, B, T, V, drop_p, seed);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

extern "C" {

int mmst_embed_fwd(int n, const mmst_embed_group* g, const mmst_embed_shared* s, void* stream) {
  EmbedArgs A;
  int rc = fill_embed_args(A, n, g, s);
  if (rc) return rc;
  const int threads = A.row_begin[n] * 16;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, A);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_embed_bwd(int n, const mmst_embed_group* g, const mmst_embed_shared* s, const void* const* dxseq,
                   void* stream) {
  EmbedBwdArgs A;
  int rc = fill_embed_args(A.e, n, g, s);
  if (rc) return rc;
  if (!dxseq) return MMST_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    if (!dxseq[i]) return MMST_ERR_ARG;
    A.dx[i] = (const bf16*)dxseq[i];
  }
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(s->B, n), dim3(8 * D), 0, (hipStream_t)stream, A);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_seq_mean(const void* mem, void* out, int B, int L, void* stream) {
  if (!mem || !out || B < 1 || L < 1) return MMST_ERR_ARG;
  hipLaunchKernelGGL(seq_mean_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const bf16*)mem, (bf16*)out, L);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_trg_embed_fwd(const int64_t* tok, int tok_ld, const float* embed, const float* pos, void* out, int B, int T,
                       float drop_p, uint32_t seed, void* stream) {
  return launch_trg_embed_fwd(tok, tok_ld, 1, embed, pos, out, B, T, drop_p, seed, (hipStream_t)stream, nullptr);
}

int mmst_trg_embed_bwd(const int64_t* tok, int tok_ld, const void* dout, float* d_embed, float* d_pos, int B, int T,
                       int V, float drop_p, uint32_t seed, void* stream) {
  return launch_trg_embed_bwd(tok, tok_ld, 1, dout, d_embed, d_pos, B, T, V, drop_p, seed, (hipStream_t)stream);
}

int mmst_real_data_mix(float* logits, const float* extra, int B, int T, void* stream) {
  if (!logits || !extra || B < 1 || T < 1) return MMST_ERR_ARG;
  const long n = (long)B * T * 64;
  hipLaunchKernelGGL(real_data_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits,
                     extra, T, n);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_real_data_rows_bwd(const void* dlogits, const void* mean, float* d_extra, float* mean32, float* dbias, int B,
                            int T, void* stream) {
  if (!dlogits || !mean || !d_extra || !mean32 || !dbias || B < 1 || T < 1) return MMST_ERR_ARG;
  hipLaunchKernelGGL(real_data_rows_kernel, dim3(B), dim3(D), 0, (hipStream_t)stream, (const bf16*)dlogits,
                     (const bf16*)mean, d_extra, mean32, dbias, T);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_seq_mean_bwd_add(const float* dmean, void* dmem, int B, int L, void* stream) {
  if (!dmean || !dmem || B < 1 || L < 1) return MMST_ERR_ARG;
  const long n = (long)B * L * D;
  hipLaunchKernelGGL(seq_mean_bwd_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dmean,
                     (bf16*)dmem, L, n);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_ce_loss(const float* logits, int ldl, int V, const int64_t* trg, int trg_ld, int B, int T, float* loss,
                 void* dlogits, float grad_scale, void* stream) {
  if (!logits || !trg || !loss || V < 1 || V > 64 || B < 1 || T < 1 || trg_ld < T + 1) return MMST_ERR_ARG;
  const int rows = B * T;
  int blocks = (rows + 3) / 4;
  if (blocks > 128) blocks = 128;
  hipLaunchKernelGGL(ce_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, trg,
                     trg_ld, B, T, loss, (bf16*)dlogits, grad_scale);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_greedy_select(const float* logits, int ldl, int V, int B, int T, float temperature, int64_t* next_tok,
                       float* next_prob, void* stream) {
  if (!logits || !next_tok || !next_prob || V < 1 || V > 64 || B < 1 || T < 1 || !(temperature > 0.f))
    return MMST_ERR_ARG;
  hipLaunchKernelGGL(greedy_select_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, B, T,
                     temperature, next_tok, next_prob);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_sample_select(const float* logits, int ldl, int V, int B, int T, float temperature, uint32_t seed,
                       uint32_t counter, int64_t* next_tok, float* next_prob, void* stream) {
  if (!logits || !next_tok || !next_prob || V < 1 || V > 64 || B < 1 || T < 1 || !(temperature > 0.f))
    return MMST_ERR_ARG;
  hipLaunchKernelGGL(sample_select_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, B, T,
                     temperature, seed, counter, next_tok, next_prob);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_adamw(float* param, const float* grad, float* m, float* v, void* param_bf16, long n, float lr, float beta1,
               float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  if (!param || !grad || !m || !v || n < 1 || step < 1) return MMST_ERR_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                     (bf16*)param_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_adamw_ranges(float* param, float* grad, float* m, float* v, void* param_bf16, int n_ranges, const long* lo,
                      const long* hi, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      float grad_scale, int zero_grad, void* stream) {
  if (!param || !grad || !m || !v || !lo || !hi || n_ranges < 1 || n_ranges > ADAMW_MAX_RANGES || step < 1)
    return MMST_ERR_ARG;
  AdamRanges R;
  R.n = n_ranges;
  R.cum[0] = 0;
  for (int i = 0; i < n_ranges; ++i) {
    if (lo[i] < 0 || hi[i] <= lo[i] || (lo[i] & 3) || ((hi[i] - lo[i]) & 3)) return MMST_ERR_ARG;
    R.lo[i] = lo[i];
    R.cum[i + 1] = R.cum[i] + (hi[i] - lo[i]) / 4;
  }
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
  long blocks = (R.cum[n_ranges] + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_ranges_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                     (bf16*)param_bf16, R, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, zero_grad);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_cast_bf16(const float* src, void* dst, int rows, int cols, int ld_out, void* stream) {
  if (!src || !dst || rows < 1 || cols < 1 || ld_out < cols) return MMST_ERR_ARG;
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(cast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst,
                     rows, cols, ld_out);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

}  // extern "C"

namespace {
struct FragArgs { int n; mmst_frag_desc d[MMST_FRAG_MAX]; };
// one thread per 16-byte piece of the image: piece p of matrix y = lane p % 64 of tile p / 64 (tiles in row-major order,
// 16 rows x 32 columns each); lane (r = lane & 15, g = lane >> 4) holds A[16 i + r][32 j + 8 g .. + 7]
__global__ __launch_bounds__(256) void fragment_kernel(const FragArgs A) {
  const mmst_frag_desc& d = A.d[blockIdx.y];
  const int pieces = d.rows * d.cols / 8, kt = d.cols / 32;
  const bf16* src = (const bf16*)d.src;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < pieces; p += gridDim.x * 256) {
    const int lane = p & 63, tile = p >> 6, j = tile % kt, i = tile / kt;
    const int row = 16 * i + (lane & 15), col = 32 * j + 8 * (lane >> 4);
    bf16x8 v;
    if (!d.transpose) {
      v = *reinterpret_cast<const bf16x8*>(src + (size_t)row * d.ld + col);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[(size_t)(col + e) * d.ld + row];
    }
    *reinterpret_cast<bf16x8*>((bf16*)d.dst + (size_t)p * 8) = v;
  }
}
}  // namespace

extern "C" {

int mmst_fragment_bf16(int n, const mmst_frag_desc* descs, void* stream) {
  if (n < 0 || (n > 0 && !descs)) return MMST_ERR_ARG;
  for (int base = 0; base < n; base += MMST_FRAG_MAX) {
    FragArgs A;
    A.n = n - base < MMST_FRAG_MAX ? n - base : MMST_FRAG_MAX;
    int most = 0;
    for (int i = 0; i < A.n; ++i) {
      const mmst_frag_desc& d = descs[base + i];
      if (!d.src || !d.dst || d.rows < 16 || (d.rows & 15) || d.cols < 32 || (d.cols & 31) || d.ld < 1) return MMST_ERR_ARG;
      if (!d.transpose && (d.ld & 7)) return MMST_ERR_ARG;
      A.d[i] = d;
      const int pieces = d.rows * d.cols / 8;
      if (pieces > most) most = pieces;
    }
    int gx = (most + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(fragment_kernel, dim3(gx, A.n), dim3(256), 0, (hipStream_t)stream, A);
    MMST_CHECK_LAUNCH();
  }
  return MMST_OK;
}

int mmst_transpose_bf16(const void* src, void* dst, int n, const uint32_t* offsets, int rows, int cols, void* stream) {
  if (!src || !dst || !offsets || n < 1 || n > TRANSPOSE_MAX || rows < 1 || cols < 1) return MMST_ERR_ARG;
  TransposeArgs A;
  A.src = (const bf16*)src; A.dst = (bf16*)dst; A.rows = rows; A.cols = cols;
  for (int i = 0; i < n; ++i) A.off[i] = offsets[i];
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, n), dim3(256), 0, (hipStream_t)stream, A);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

}  // extern "C"
