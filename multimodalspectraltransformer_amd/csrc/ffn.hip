// Fused feed-forward kernels, see ffn.h.
//
// Workgroup = 128 token rows, 4 waves, each wave owns 32 whole rows (the layout of the LayerNorm epilogues).
// The hidden dimension is walked in 32 chunks of 64 columns; per chunk two 16 KiB weight images stream
// through a double-buffered LDS slot (one barrier per image, next image prefetched into registers):
//   image 0  first GEMM: 64 hidden columns x all 128 model columns (the two contraction halves packed
//            side by side into one image)
//   image 1  second GEMM: all 128 model columns x the chunk's 64 hidden columns (a standard image)
// The token-side operand of both GEMMs lives in registers: the input rows are loaded once as MFMA fragments,
// and the first GEMM's accumulator layout (token row on the lane, 8 consecutive hidden columns per pair of
// interleaved tiles) IS the fragment layout of the second GEMM's contraction, so the hidden chunk is
// converted to bf16 in place and never touches LDS.
#include "ffn.h"
#include "gemm_tile.h"

namespace {

constexpr int DM = 128, FF = 2048, HC = 64, NCHUNK = FF / HC;

// 64 index rows x 128 contraction columns of a row-major matrix as ONE row-major image:
// image row r = matrix row i0 + (r & 63), contraction columns (r >> 6) * 64 ... + 63
__device__ __forceinline__ void g2r_n64(uint4 (&r)[4], const bf16* __restrict__ base, int ld, int i0, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + 256 * i, row = c >> 3;
    r[i] = *reinterpret_cast<const uint4*>(base + (size_t)(i0 + (row & 63)) * ld + (row >> 6) * 64 + (c & 7) * 8);
  }
}
// 128 contraction rows x 64 index columns of a contraction-major matrix as ONE contraction-major image:
// image row c, image index j = matrix row (j >> 6) * 64 + c, matrix column i0 + (j & 63)
__device__ __forceinline__ void g2r_t64(uint4 (&r)[4], const bf16* __restrict__ base, int ld, int i0, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + 256 * i, row = c >> 4, j = (c & 15) * 8;
    r[i] = *reinterpret_cast<const uint4*>(base + (size_t)((j >> 6) * 64 + row) * ld + i0 + (j & 63));
  }
}

template <bool BWD>
__global__ __launch_bounds__(256, 2) void ffn_kernel(const FfnBatch batch) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BQ * BK * 2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g = 0;
#pragma unroll
  for (int i = 1; i < FFN_MAX_GROUPS; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.p[i].tile_begin) g = i;
  const FfnProb& pr = batch.p[g];
  const int M = pr.epi.Pn;
  const int p0 = ((int)blockIdx.x - pr.tile_begin) * BP;
  const int pw0 = wave * 32;
  const int q = lane >> 4;

  // token-side fragments of the first GEMM, resident for the whole kernel
  bf16x8 xf[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < M) v = *reinterpret_cast<const uint4*>(pr.X + (size_t)m * DM + ks * 32 + 8 * q);
      xf[mt][ks] = __builtin_bit_cast(bf16x8, v);
    }
  }

  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight image `ph` of chunk `c` -> registers
  auto fetch = [&](uint4 (&r)[4], int c, int ph) {
    if (!BWD) {
      if (ph == 0) g2r_n64(r, pr.W1, DM, c * HC, tid);                       // rows = hidden chunk, k = model
      else g2r<false>(r, pr.W2, FF, 0, DM, c * HC, FF, tid);                 // rows = model, k = hidden chunk
    } else {
      if (ph == 0) g2r_t64(r, pr.W2, FF, c * HC, tid);                       // k = model (rows of W2), index = hidden chunk
      else g2r<true>(r, pr.W1, DM, 0, DM, c * HC, FF, tid);                  // k = hidden chunk (rows of W1), index = model
    }
  };
  // weight-side fragment of output-column tile nt; `half` selects the packed half of a first-GEMM image
  auto qfrag = [&](const unsigned char* img, int nt, int half, int ks) -> bf16x8 {
    if (BWD) return frag_t(img, half * 64 + (nt >> 1) * 32 + 8 * (lane & 3) + 4 * (nt & 1), ks, lane);
    return frag_n<true>(img, half * 64 + (nt >> 1) * 32 + 8 * ((lane & 15) >> 2) + 4 * (nt & 1) + (lane & 3), ks, lane);
  };

  const float dscale = pr.drop_p > 0.f ? 1.0f / (1.0f - pr.drop_p) : 1.0f;
  const uint32_t thr = drop_threshold(pr.drop_p);

  uint4 rq[4];
  fetch(rq, 0, 0);
  for (int c = 0; c < NCHUNK; ++c) {
    // backward: the saved hidden activation of this chunk (ReLU/dropout mask), issued ahead of the first GEMM
    uint4 hsave[2][2];
    if (BWD) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = p0 + pw0 + mt * 16 + (lane & 15);
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (m < M) v = *reinterpret_cast<const uint4*>(pr.H + (size_t)m * FF + c * HC + tp * 32 + 8 * q);
          hsave[mt][tp] = v;
        }
      }
    }
    // ---- first GEMM: a1[32 rows][64 hidden] over the 128 model columns
    f32x4 a1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) a1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      unsigned char* img = smem;
      if (BWD) r2s<true, true>(img, rq, tid);
      else r2s<false, true>(img, rq, tid);
      __syncthreads();
      fetch(rq, c, 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 qf[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) qf[nt] = qfrag(img, nt, ks >> 1, ks & 1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            a1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], xf[mt][ks], a1[mt][nt], 0, 0, 0);
      }
    }
    // ---- hidden chunk: epilogue of the first GEMM, re-packed as the token-side operand of the second
    bf16x8 hf[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = p0 + pw0 + mt * 16 + (lane & 15);
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const int n = c * HC + tp * 32 + 8 * q;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a1[mt][2 * tp][e]; v[4 + e] = a1[mt][2 * tp + 1][e]; }
        if (!BWD) {
          float b[8];
          load8f(b, pr.b1 + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + b[e], 0.f);
#ifndef FFN_NODROP
          if (pr.drop_p > 0.f) drop8(v, pr.seed, (uint32_t)m * (uint32_t)FF + n, thr, dscale);
#endif
        } else {
          bf16x8 h = __builtin_bit_cast(bf16x8, hsave[mt][tp]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)h[e] > 0.f ? v[e] * dscale : 0.f;
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        hf[mt][tp] = o;
#ifndef FFN_NOSTORE
        if (m < M) {
          if (!BWD) *reinterpret_cast<bf16x8*>(pr.H + (size_t)m * FF + n) = o;
          else if (pr.dH) *reinterpret_cast<bf16x8*>(pr.dH + (size_t)m * FF + n) = o;
        }
#endif
      }
    }
    // ---- second GEMM: acc[32 rows][128 model] += hidden chunk x weights
    {
      unsigned char* img = smem + BQ * BK * 2;
      if (BWD) r2s<true, true>(img, rq, tid);
      else r2s<false, true>(img, rq, tid);
      __syncthreads();
      if (c + 1 < NCHUNK) fetch(rq, c + 1, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 qf[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) qf[nt] = qfrag(img, nt, 0, ks);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 8; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], hf[mt][ks], acc[mt][nt], 0, 0, 0);
      }
    }
  }

  if (!BWD) {
    epi_ln(acc, pr.epi, p0, pw0, lane);
  } else {
    __syncthreads();  // the LDS slot doubles as the column-sum scratch of the LayerNorm backward
    epi_lnbwd(acc, pr.epi, p0, pw0, lane, wave, tid, smem);
  }
}

int prepare(FfnBatch& b, bool bwd) {
  if (b.n < 1 || b.n > FFN_MAX_GROUPS) return -1;
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    FfnProb& p = b.p[i];
    if (p.epi.Pn <= 0 || !p.X || !p.W1 || !p.W2 || !p.H || !p.epi.C) return -1;
    if (!bwd && (!p.b1 || !p.epi.gamma || !p.epi.beta)) return -1;
    if (bwd && (!p.epi.gamma || !p.epi.Z || !p.epi.stats)) return -1;
    p.epi.Qn = DM;
    p.tile_begin = total;
    total += (p.epi.Pn + BP - 1) / BP;
  }
  return total;
}

}  // namespace

int launch_ffn_fwd(FfnBatch& b, hipStream_t stream) {
  int total = prepare(b, false);
  if (total < 0) return MMST_ERR_ARG;
  hipLaunchKernelGGL((ffn_kernel<false>), dim3(total), dim3(256), 0, stream, b);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int launch_ffn_bwd(FfnBatch& b, hipStream_t stream) {
  int total = prepare(b, true);
  if (total < 0) return MMST_ERR_ARG;
  hipLaunchKernelGGL((ffn_kernel<true>), dim3(total), dim3(256), 0, stream, b);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
