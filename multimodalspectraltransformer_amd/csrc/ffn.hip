// Fused feed-forward kernels, see ffn.h.
//
// Workgroup = 128 token rows, 4 waves, each wave owns 32 whole rows (the layout of the LayerNorm epilogues).
// The hidden dimension is walked in 32 chunks of 64 columns.  Per chunk a fixed sequence of 16 KiB LDS images
// is consumed, one barrier each:
//   forward   [W1 chunk: 64 hidden x 128 model, both contraction halves side by side] [W2 chunk: 128 model x 64 hidden]
//   backward  [W2 chunk, contraction-major] [saved hidden tile 128 rows x 64 (ReLU/dropout mask)] [W1 chunk, contraction-major]
// Images are written by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a ring of four slots, three
// images ahead of the consumer; the XOR swizzles that make the fragment reads conflict-free are applied to the
// per-lane SOURCE address, since the DMA destination is lane-linear.  Synchronisation per image: a counted
// s_waitcnt vmcnt (own DMA portions landed), one raw s_barrier (everyone's landed, and the slot that is re-filled
// next has been read by all waves), then the next DMA is issued.  There is no ordinary global load inside the loop.
// The token-side operand of both GEMMs lives in registers: the input rows are loaded once as MFMA fragments,
// and the first GEMM's accumulator layout (token row on the lane, 8 consecutive hidden columns per pair of
// interleaved tiles) IS the fragment layout of the second GEMM's contraction, so the hidden chunk is
// converted to bf16 in place and never touches LDS.
#include "ffn.h"
#include "gemm_tile.h"

namespace {

constexpr int DM = 128, FF = 2048, HC = 64, NCHUNK = FF / HC;
constexpr int SLOT = 16384, NSLOT = 4, AHEAD = 3;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 64 lanes x 16 B, global (per-lane address) -> LDS (wave-uniform base + 16 * lane)
__device__ __forceinline__ void glds16(const bf16* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

enum ImgKind { IMG_W1_FWD, IMG_W2_FWD, IMG_W2_BWD, IMG_H_BWD, IMG_W1_BWD };

// Per-lane element offset (relative to chunk 0) of the 16 bytes lane `lane` of wave `wave` contributes to DMA
// instruction j (0..3) of an image.  Row-major images: 8 rows of 128 B per instruction; contraction-major
// images: 4 rows of 256 B per instruction.
template <int KIND>
__device__ __forceinline__ uint32_t src_offset(int wave, int lane, int j, int p0, int M) {
  if (KIND == IMG_W1_FWD || KIND == IMG_W2_FWD || KIND == IMG_H_BWD) {
    const int r = wave * 32 + j * 8 + (lane >> 3), s = lane & 7;
    if (KIND == IMG_W1_FWD) return (uint32_t)((r & 63) * DM + (r >> 6) * 64 + ((s ^ fsw(r)) << 3));
    if (KIND == IMG_W2_FWD) return (uint32_t)(r * FF + ((s ^ fsw(r)) << 3));
    const int m = (p0 + r < M) ? p0 + r : M - 1;
    return (uint32_t)m * FF + (uint32_t)((s ^ (r & 7)) << 3);   // M * FF < 2^32 is checked by the launcher
  } else {
    const int r = wave * 16 + j * 4 + (lane >> 4), s = lane & 15;
    const int ch = (((s >> 1) ^ fsw(r)) << 1) | (s & 1);
    if (KIND == IMG_W2_BWD) return (uint32_t)((((ch * 8) >> 6) * 64 + r) * FF + ((ch * 8) & 63));
    return (uint32_t)(r * DM + ch * 8);
  }
}
// element stride of one hidden chunk for each image kind
template <int KIND> __device__ __forceinline__ constexpr uint32_t chunk_stride() {
  return (KIND == IMG_W1_FWD || KIND == IMG_W1_BWD) ? (uint32_t)HC * DM : (uint32_t)HC;
}

template <bool BWD>
__global__ __launch_bounds__(256, 2) void ffn_kernel(const FfnBatch batch) {
  // one array: DMA ring (4 x 16 KiB), then the first-layer bias (forward) / column-sum scratch (backward)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSLOT * SLOT + FF * 4];
  float* b1s = reinterpret_cast<float*>(smem + NSLOT * SLOT);

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
  if (!BWD) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<f32x4*>(b1s + (tid + 256 * i) * 4) = *reinterpret_cast<const f32x4*>(pr.b1 + (tid + 256 * i) * 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NIMG = BWD ? 3 : 2, TOTAL = NCHUNK * NIMG;
  // per-lane source offsets of the 4 DMA instructions of each image kind (loop invariant)
  uint32_t offA[4], offB[4], offH[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    offA[j] = BWD ? src_offset<IMG_W2_BWD>(wave, lane, j, p0, M) : src_offset<IMG_W1_FWD>(wave, lane, j, p0, M);
    offB[j] = BWD ? src_offset<IMG_W1_BWD>(wave, lane, j, p0, M) : src_offset<IMG_W2_FWD>(wave, lane, j, p0, M);
    offH[j] = BWD ? src_offset<IMG_H_BWD>(wave, lane, j, p0, M) : 0u;
  }
  // image i of the sequence = phase (i % NIMG) of chunk (i / NIMG), into ring slot (i % 4)
  auto issue = [&](int c, int ph, int i) {
    unsigned char* dst = smem + (i & (NSLOT - 1)) * SLOT + wave * (SLOT / 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16* src;
      if (!BWD) {
        src = ph == 0 ? pr.W1 + offA[j] + (uint32_t)c * chunk_stride<IMG_W1_FWD>()
                      : pr.W2 + offB[j] + (uint32_t)c * chunk_stride<IMG_W2_FWD>();
      } else {
        src = ph == 0 ? pr.W2 + offA[j] + (uint32_t)c * chunk_stride<IMG_W2_BWD>()
            : ph == 1 ? pr.H + offH[j] + (uint32_t)c * chunk_stride<IMG_H_BWD>()
                      : pr.W1 + offB[j] + (uint32_t)c * chunk_stride<IMG_W1_BWD>();
      }
      glds16(src, dst + j * 1024);
    }
  };
  // own DMA portions of image i have landed once at most the (up to two) younger images are outstanding;
  // the hidden-activation stores issued in between only make the count conservative
  auto wait_image = [&](int i) {
    if (i + 2 < TOTAL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (i + 1 < TOTAL) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // weight-side fragment of output-column tile nt; `half` selects the packed half of a first-GEMM image
  // (backward: contraction-major images, transposing reads issued by hand, see frag_t_issue)
  auto qfrag = [&](const unsigned char* img, int nt, int half, int ks) -> bf16x8 {
    return frag_n<true>(img, half * 64 + (nt >> 1) * 32 + 8 * ((lane & 15) >> 2) + 4 * (nt & 1) + (lane & 3), ks, lane);
  };

  const float dscale = pr.drop_p > 0.f ? 1.0f / (1.0f - pr.drop_p) : 1.0f;
  const uint32_t thr = drop_threshold(pr.drop_p);

  issue(0, 0, 0);
  issue(0, 1, 1);
  if (BWD) issue(0, 2, 2);
  else issue(1, 0, 2);

  for (int c = 0; c < NCHUNK; ++c) {
    int i = c * NIMG;
    // ---- first GEMM: a1[32 rows][64 hidden] over the 128 model columns
    f32x4 a1[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) a1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      wait_image(i);
      __builtin_amdgcn_s_barrier();
      if (i + AHEAD < TOTAL) { if (BWD) issue(c + 1, 0, i + AHEAD); else issue(c + 1, 1, i + AHEAD); }
      const unsigned char* img = smem + (i & (NSLOT - 1)) * SLOT;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 qf[4];
        if (BWD) {
          s16x4 lo[4], hi[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            frag_t_issue(lo[nt], hi[nt], img, (ks >> 1) * 64 + (nt >> 1) * 32 + 8 * (lane & 3) + 4 * (nt & 1), ks & 1, lane);
          lds_wait4(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) qf[nt] = frag_join(lo[nt], hi[nt]);
        } else {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) qf[nt] = qfrag(img, nt, ks >> 1, ks & 1);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            a1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], xf[mt][ks], a1[mt][nt], 0, 0, 0);
      }
      ++i;
    }
    // ---- hidden chunk: epilogue of the first GEMM, re-packed as the token-side operand of the second
    const unsigned char* himg = nullptr;
    if (BWD) {
      wait_image(i);
      __builtin_amdgcn_s_barrier();
      if (i + AHEAD < TOTAL) issue(c + 1, 1, i + AHEAD);
      himg = smem + (i & (NSLOT - 1)) * SLOT;
      ++i;
    }
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
          load8f(b, b1s + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + b[e], 0.f);
          if (pr.drop_p > 0.f) drop8(v, pr.seed, (uint32_t)m * (uint32_t)FF + n, thr, dscale);
        } else {
          bf16x8 h = frag_n<false>(himg, pw0 + mt * 16 + (lane & 15), tp, lane);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)h[e] > 0.f ? v[e] * dscale : 0.f;
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        hf[mt][tp] = o;
        if (m < M) {
          if (!BWD) *reinterpret_cast<bf16x8*>(pr.H + (size_t)m * FF + n) = o;
          else if (pr.dH) *reinterpret_cast<bf16x8*>(pr.dH + (size_t)m * FF + n) = o;
        }
      }
    }
    // ---- second GEMM: acc[32 rows][128 model] += hidden chunk x weights
    {
      wait_image(i);
      __builtin_amdgcn_s_barrier();
      if (i + AHEAD < TOTAL) { if (BWD) issue(c + 1, 2, i + AHEAD); else issue(c + 2, 0, i + AHEAD); }
      const unsigned char* img = smem + (i & (NSLOT - 1)) * SLOT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 qf[8];
        if (BWD) {
          s16x4 lo[8], hi[8];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt)
            frag_t_issue(lo[nt], hi[nt], img, (nt >> 1) * 32 + 8 * (lane & 3) + 4 * (nt & 1), ks, lane);
          lds_wait4(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]);
          lds_wait4(lo[4], hi[4], lo[5], hi[5], lo[6], hi[6], lo[7], hi[7]);
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) qf[nt] = frag_join(lo[nt], hi[nt]);
        } else {
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) qf[nt] = qfrag(img, nt, 0, ks);
        }
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
    __syncthreads();  // the ring doubles as the column-sum scratch of the LayerNorm backward
    epi_lnbwd(acc, pr.epi, p0, pw0, lane, wave, tid, smem);
  }
}

int prepare(FfnBatch& b, bool bwd) {
  if (b.n < 1 || b.n > FFN_MAX_GROUPS) return -1;
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    FfnProb& p = b.p[i];
    if ((long)p.epi.Pn * FF >= (1L << 32)) return -1;
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
