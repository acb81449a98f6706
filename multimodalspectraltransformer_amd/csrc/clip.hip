// CLIP contrastive head on top of the encoder half (gfx950): the two projection heads and the symmetric
// soft-target cross entropy of models_CLIP_v15_4.py:243-302, 390-426, forward and backward.
//
// The head works on [batch][<=1024] matrices with batch ~ 128: a few hundred MFLOP per step next to the
// encoder's hundreds of GFLOP.  It is therefore kept in fp32 end to end (the loss is a log-softmax over dot
// products of magnitude ~500 and loses digits quickly), as plain FMA kernels: one small strided SGEMM that serves
// every product (forward, dX, dW by choice of strides) and row / column kernels for GELU, dropout + residual +
// LayerNorm and the loss.  Everything is latency-sized; none of it shows up in a profile of the step.
#include "common.h"
#include <math.h>

namespace {

// ---- C[M][N] = alpha * sum_k A(i,k) B(k,j) + beta * C + bias[j];  A(i,k) = A[i*sai + k*sak], B(k,j) = B[k*sbk + j*sbj]
constexpr int TS = 64, TK = 16;
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, long sai, long sak,
                                                    const float* __restrict__ Bm, long sbk, long sbj, float* __restrict__ Cm,
                                                    int ldc, const float* __restrict__ bias, int M, int N, int K, float alpha,
                                                    float beta) {
  __shared__ float As[TK][TS + 1], Bs[TK][TS + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int k0 = 0; k0 < K; k0 += TK) {
    for (int e = tid; e < TS * TK; e += 256) {
      const int kk = e & (TK - 1), r = e >> 4;   // consecutive threads walk k: contiguous for row-major operands
      const int k = k0 + kk;
      As[kk][r] = (i0 + r < M && k < K) ? A[(long)(i0 + r) * sai + (long)k * sak] : 0.f;
      Bs[kk][r] = (j0 + r < N && k < K) ? Bm[(long)k * sbk + (long)(j0 + r) * sbj] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = As[kk][ty * 4 + a];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = Bs[kk][tx * 4 + b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = i0 + ty * 4 + a;
    if (i >= M) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + tx * 4 + b;
      if (j >= N) continue;
      float v = alpha * acc[a][b] + (bias ? bias[j] : 0.f);
      if (beta != 0.f) v += beta * Cm[(long)i * ldc + j];
      Cm[(long)i * ldc + j] = v;
    }
  }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = red[0];
  for (int w = 1; w < nw; ++w) s = fmaxf(s, red[w]);
  return s;
}

// exact (erf) GELU, nn.GELU() default
__global__ void gelu_fwd_kernel(const float* x, float* y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752f));
}
// dx += dy * gelu'(x)
__global__ void gelu_bwd_kernel(const float* x, const float* dy, float* dx, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
  dx[i] += dy[i] * (cdf + v * pdf);
}

// z = dropout(f) + projected; out = LayerNorm(z).  One workgroup per row, P <= 1024.  Saves xhat and rstd.
__global__ __launch_bounds__(256) void head_out_fwd_kernel(const float* f, const float* projected, const float* gamma,
                                                           const float* beta, float* out, float* xhat, float* rstd_out,
                                                           int P, float drop_p, uint32_t seed) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const uint32_t thr = drop_threshold(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  float z[4];
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = threadIdx.x + 256 * e;
    z[e] = 0.f;
    if (j < P) {
      float v = f[(long)row * P + j];
      if (drop_p > 0.f) v = drop_keep(seed, (uint32_t)(row * P + j), thr) ? v * dscale : 0.f;
      z[e] = v + projected[(long)row * P + j];
      s += z[e];
    }
  }
  const float mean = block_sum(s, red) / P;
  float vs = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (threadIdx.x + 256 * e < P) vs += (z[e] - mean) * (z[e] - mean);
  const float rstd = rsqrtf(block_sum(vs, red) / P + 1e-5f);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = threadIdx.x + 256 * e;
    if (j < P) {
      const float xh = (z[e] - mean) * rstd;
      xhat[(long)row * P + j] = xh;
      out[(long)row * P + j] = xh * gamma[j] + beta[j];
    }
  }
  if (threadIdx.x == 0) rstd_out[row] = rstd;
}

// dout -> dz (LayerNorm backward); dproj = dz (residual branch), df = dropout mask applied to dz;
// dgamma/dbeta accumulated with atomics.
__global__ __launch_bounds__(256) void head_out_bwd_kernel(const float* dout, const float* xhat, const float* rstd_in,
                                                           const float* gamma, float* df, float* dproj, float* dgamma,
                                                           float* dbeta, int P, float drop_p, uint32_t seed) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const uint32_t thr = drop_threshold(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  float gy[4], xh[4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = threadIdx.x + 256 * e;
    gy[e] = xh[e] = 0.f;
    if (j < P) {
      const float d = dout[(long)row * P + j];
      xh[e] = xhat[(long)row * P + j];
      gy[e] = d * gamma[j];
      s1 += gy[e];
      s2 += gy[e] * xh[e];
      atomicAdd(dgamma + j, d * xh[e]);
      atomicAdd(dbeta + j, d);
    }
  }
  s1 = block_sum(s1, red) / P;
  s2 = block_sum(s2, red) / P;
  const float rstd = rstd_in[row];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = threadIdx.x + 256 * e;
    if (j < P) {
      const float dz = rstd * (gy[e] - s1 - xh[e] * s2);
      dproj[(long)row * P + j] = dz;
      float v = dz;
      if (drop_p > 0.f) v = drop_keep(seed, (uint32_t)(row * P + j), thr) ? dz * dscale : 0.f;
      df[(long)row * P + j] = v;
    }
  }
}

// ---- loss, pass 1 (one workgroup per row i): T = softmax(Z_i), LR = log_softmax(logits_i), SR = softmax(logits_i),
// texts_i = -sum_j T_ij LR_ij
__global__ __launch_bounds__(256) void clip_rows_kernel(const float* logits, const float* Z, float* T, float* LR, float* SR,
                                                        float* texts, int B) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  float mz = -INFINITY, ml = -INFINITY;
  for (int j = threadIdx.x; j < B; j += 256) {
    mz = fmaxf(mz, Z[(long)i * B + j]);
    ml = fmaxf(ml, logits[(long)i * B + j]);
  }
  mz = block_max(mz, red);
  ml = block_max(ml, red);
  float sz = 0.f, sl = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    sz += expf(Z[(long)i * B + j] - mz);
    sl += expf(logits[(long)i * B + j] - ml);
  }
  sz = block_sum(sz, red);
  sl = block_sum(sl, red);
  const float lsl = logf(sl);
  float t = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float tv = expf(Z[(long)i * B + j] - mz) / sz;
    const float lr = logits[(long)i * B + j] - ml - lsl;
    T[(long)i * B + j] = tv;
    LR[(long)i * B + j] = lr;
    SR[(long)i * B + j] = expf(lr);
    t -= tv * lr;
  }
  t = block_sum(t, red);
  if (threadIdx.x == 0) texts[i] = t;
}
// ---- pass 2 (one workgroup per column j): LC = log_softmax over rows of column j, SC, csum_j = sum_i T_ij,
// images_j = -sum_i T_ij LC_ij
__global__ __launch_bounds__(256) void clip_cols_kernel(const float* logits, const float* T, float* LC, float* SC,
                                                        float* csum, float* images, int B) {
  __shared__ float red[4];
  const int j = blockIdx.x;
  float ml = -INFINITY;
  for (int i = threadIdx.x; i < B; i += 256) ml = fmaxf(ml, logits[(long)i * B + j]);
  ml = block_max(ml, red);
  float sl = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) sl += expf(logits[(long)i * B + j] - ml);
  sl = block_sum(sl, red);
  const float lsl = logf(sl);
  float cs = 0.f, im = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float lc = logits[(long)i * B + j] - ml - lsl;
    const float tv = T[(long)i * B + j];
    LC[(long)i * B + j] = lc;
    SC[(long)i * B + j] = expf(lc);
    cs += tv;
    im -= tv * lc;
  }
  cs = block_sum(cs, red);
  im = block_sum(im, red);
  if (threadIdx.x == 0) { csum[j] = cs; images[j] = im; }
}
// ---- pass 3 (per row i): loss_i, mean loss, dlogits and dZ of L = mean_i loss_i
__global__ __launch_bounds__(256) void clip_grad_kernel(const float* T, const float* LR, const float* SR, const float* LC,
                                                        const float* SC, const float* csum, const float* texts,
                                                        const float* images, float* losses, float* loss_mean,
                                                        float* dlogits, float* dZ, int B) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  const float inv2B = 0.5f / B;
  float dot = 0.f;   // sum_k T_ik G_ik with G = -(LR + LC) / 2B
  for (int j = threadIdx.x; j < B; j += 256)
    dot += T[(long)i * B + j] * (-(LR[(long)i * B + j] + LC[(long)i * B + j]) * inv2B);
  dot = block_sum(dot, red);
  for (int j = threadIdx.x; j < B; j += 256) {
    const long o = (long)i * B + j;
    const float t = T[o];
    dlogits[o] = -inv2B * ((t - SR[o]) + (t - SC[o] * csum[j]));
    dZ[o] = t * (-(LR[o] + LC[o]) * inv2B - dot);
  }
  if (threadIdx.x == 0) {
    const float l = 0.5f * (texts[i] + images[i]);
    losses[i] = l;
    atomicAdd(loss_mean, l / B);
  }
}

// out = X + X^T (square)
__global__ void symmetrize_kernel(const float* X, float* out, int B) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (long)B * B) return;
  const int i = o / B, j = o % B;
  out[o] = X[o] + X[(long)j * B + i];
}

// d(memory)[b][l][:] = dmean[b][:] / L  (backward of torch.mean(memory, dim=0), bf16 out)
__global__ void seq_mean_bwd_kernel(const float* dmean, bf16* dmem, int L, long n) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  const long b = o / ((long)L * 128);
  dmem[o] = (bf16)(dmean[b * 128 + (o & 127)] / L);
}

}  // namespace

extern "C" {

int mmst_sgemm(const float* A, long sai, long sak, const float* B, long sbk, long sbj, float* C, int ldc,
               const float* bias, int M, int N, int K, float alpha, float beta, void* stream) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || ldc < N) return MMST_ERR_ARG;
  dim3 grid((N + TS - 1) / TS, (M + TS - 1) / TS);
  hipLaunchKernelGGL(sgemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, sai, sak, B, sbk, sbj, C, ldc, bias, M, N, K,
                     alpha, beta);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_gelu_fwd(const float* x, float* y, long n, void* stream) {
  if (!x || !y || n < 1) return MMST_ERR_ARG;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
int mmst_gelu_bwd(const float* x, const float* dy, float* dx, long n, void* stream) {
  if (!x || !dy || !dx || n < 1) return MMST_ERR_ARG;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_head_out_fwd(const float* f, const float* projected, const float* gamma, const float* beta, float* out,
                      float* xhat, float* rstd, int rows, int P, float drop_p, uint32_t seed, void* stream) {
  if (!f || !projected || !gamma || !beta || !out || !xhat || !rstd || rows < 1 || P < 1 || P > 1024) return MMST_ERR_ARG;
  hipLaunchKernelGGL(head_out_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, f, projected, gamma, beta, out, xhat,
                     rstd, P, drop_p, seed);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
int mmst_head_out_bwd(const float* dout, const float* xhat, const float* rstd, const float* gamma, float* df, float* dproj,
                      float* dgamma, float* dbeta, int rows, int P, float drop_p, uint32_t seed, void* stream) {
  if (!dout || !xhat || !rstd || !gamma || !df || !dproj || !dgamma || !dbeta || rows < 1 || P < 1 || P > 1024)
    return MMST_ERR_ARG;
  hipLaunchKernelGGL(head_out_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dout, xhat, rstd, gamma, df, dproj,
                     dgamma, dbeta, P, drop_p, seed);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

// logits, Z: [B][B] fp32.  ws: 5*B*B + 3*B floats of scratch.  losses [B], *loss_mean is ADDED to.
int mmst_clip_loss(const float* logits, const float* Z, int B, float* ws, float* losses, float* loss_mean, float* dlogits,
                   float* dZ, void* stream) {
  if (!logits || !Z || !ws || !losses || !loss_mean || !dlogits || !dZ || B < 1) return MMST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * B;
  float *T = ws, *LR = ws + n, *SR = ws + 2 * n, *LC = ws + 3 * n, *SC = ws + 4 * n;
  float *csum = ws + 5 * n, *texts = csum + B, *images = texts + B;
  hipLaunchKernelGGL(clip_rows_kernel, dim3(B), dim3(256), 0, st, logits, Z, T, LR, SR, texts, B);
  hipLaunchKernelGGL(clip_cols_kernel, dim3(B), dim3(256), 0, st, logits, T, LC, SC, csum, images, B);
  hipLaunchKernelGGL(clip_grad_kernel, dim3(B), dim3(256), 0, st, T, LR, SR, LC, SC, csum, texts, images, losses, loss_mean,
                     dlogits, dZ, B);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_symmetrize(const float* X, float* out, int B, void* stream) {
  if (!X || !out || B < 1) return MMST_ERR_ARG;
  const long n = (long)B * B;
  hipLaunchKernelGGL(symmetrize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, out, B);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int mmst_seq_mean_bwd(const float* dmean, void* dmem, int B, int L, void* stream) {
  if (!dmean || !dmem || B < 1 || L < 1) return MMST_ERR_ARG;
  const long n = (long)B * L * 128;
  hipLaunchKernelGGL(seq_mean_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dmean,
                     (bf16*)dmem, L, n);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

}  // extern "C"
