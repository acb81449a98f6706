// Fused feed-forward kernels, see ffn.h.
//
// Workgroup = 128 token rows as 8 waves of 16 whole rows (or 160 rows / 10 waves, or 4 waves of 32 rows: see
// Shape<>); a wave owns whole rows, the layout of the LayerNorm epilogues.
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
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#ifdef MMST_WG_LOG
// timing build: one record per workgroup (start, end in 100 MHz ticks; kind << 32 | block; hardware ids)
__device__ unsigned int g_wglog_n;
__device__ unsigned long long g_wglog[4 * 32768];
extern "C" int mmst_debug_wglog(unsigned long long* out, unsigned int* n) {
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_wglog_n), 4) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wglog), sizeof(unsigned long long) * 4 * 32768) != hipSuccess) return -1;
  return 0;
}
#define WGLOG_BEGIN() const unsigned long long wg_t0__ = wall_clock64()
#define WGLOG_END(kind)                                                                                      \
  if (threadIdx.x == 0) {                                                                                    \
    const unsigned int ix__ = atomicAdd(&g_wglog_n, 1u) & 32767u;                                            \
    g_wglog[4 * ix__] = wg_t0__;                                                                             \
    g_wglog[4 * ix__ + 1] = wall_clock64();                                                                  \
    g_wglog[4 * ix__ + 2] = ((unsigned long long)(kind) << 32) | blockIdx.x;                                 \
    g_wglog[4 * ix__ + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
  }
#else
#define WGLOG_BEGIN()
#define WGLOG_END(kind)
#endif

namespace {

constexpr int DM = 128, FF = 2048, HC = 64, NCHUNK = FF / HC;
constexpr int SLOT = 16384, NSLOT = 4, AHEAD = 3;

// Workgroup shapes (NW = waves): 4 waves x 32 rows, 8 waves x 16 rows (128-row tiles) or 10 waves x 16 rows
// (160-row tiles: a launch of 257..320 128-row tiles would need a second, mostly empty round of workgroups).
// The 16 DMA instructions of a weight image are issued by the first 16/WD waves, WD each; every wave brings in
// its own rows of a hidden-activation image.
template <int NW> struct Shape {
  static constexpr int MT = NW == 4 ? 2 : 1;        // 16-row MFMA tiles per wave
  static constexpr int BPX = NW * 16 * MT;          // rows per workgroup
  static constexpr int WD = NW == 4 ? 4 : 2;        // weight-image DMA instructions per issuing wave
  static constexpr int NWD = 16 / WD;               // issuing waves
  static constexpr int HD = 2 * MT;                 // hidden-image DMA instructions (and hidden stores) per wave
  static constexpr int RSLOT = BPX * 128 > SLOT ? BPX * 128 : SLOT;   // ring slot of the backward kernel
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 64 lanes x 16 B, global (per-lane address) -> LDS (wave-uniform base + 16 * lane)
__device__ __forceinline__ void glds16(const bf16* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

enum ImgKind { IMG_W1_FWD, IMG_W2_FWD, IMG_H_BWD };

// Per-lane element offset (relative to chunk 0) of the 16 bytes lane `lane` contributes to the DMA instruction
// that fills image rows r0 .. r0+7 (8 rows of 128 B per instruction).
template <int KIND>
__device__ __forceinline__ uint32_t src_offset(int r0, int lane, int p0, int M) {
  {
    const int r = r0 + (lane >> 3), s = lane & 7;
    if (KIND == IMG_W1_FWD) return (uint32_t)((r & 63) * DM + (r >> 6) * 64 + ((s ^ fsw(r)) << 3));
    if (KIND == IMG_W2_FWD) return (uint32_t)(r * FF + ((s ^ fsw(r)) << 3));
    const int m = (p0 + r < M) ? p0 + r : M - 1;
    return (uint32_t)m * FF + (uint32_t)((s ^ (r & 7)) << 3);   // M * FF < 2^32 is checked by the launcher
  }
}
// element stride of one hidden chunk for each image kind
template <int KIND> __device__ __forceinline__ constexpr uint32_t chunk_stride() {
  return KIND == IMG_W1_FWD ? (uint32_t)HC * DM : (uint32_t)HC;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Sign words (FfnProb::Hs): word 16 q + (c >> 1) of a row covers the hidden columns 64 c + 32 t + 8 q + e of the chunks
// c = 2 j and 2 j + 1 (t = 0, 1; e = 0..7); the flag of (c, t, e) is bit sign_bit(c & 1, t, e).
__host__ __device__ constexpr int tp_e_index(int t, int e) { return t * 4 + (e >> 1); }
__host__ __device__ constexpr int sign_bit(int odd, int t, int e) { return ((e & 1) ? 31 : 15) - tp_e_index(t, e) - 8 * odd; }

// Row epilogue of the fused backward kernels: LayerNorm backward of the block in front (epi_lnbwd, scratch = the first
// NW KiB of smem), then optionally the chained GEMM Cn = E Wn with E = the rows the epilogue has just produced (the
// gradient through the linear layer in front of the block), so those rows are not re-read by a separate launch.  Wn
// [k][j] arrives by LDS-DMA as two contraction-major 16 KiB images (k 0..63, 64..127) at wimg while the epilogue runs.
template <int NW, int MT>
__device__ __forceinline__ void bwd_rows_epilogue(f32x4 (&acc)[MT][8], const FfnProb& pr, int p0, int pw0, int lane,
                                                  int wave, int tid, unsigned char* smem, unsigned char* wimg) {
  const int M = pr.epi.Pn, q = lane >> 4;
  __syncthreads();  // the weight ring doubles as the column-sum scratch of the LayerNorm backward
  if (pr.Wn == nullptr) {
    epi_lnbwd<NW>(acc, pr.epi, p0, pw0, lane, wave, tid, smem);
    return;
  }
  {
    constexpr int PER = 32 / (NW < 8 ? NW : 8);   // 32 instructions of 4 rows x 256 B over the first 8 waves
    if (wave < 8) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int piece = wave * PER + j, row = piece * 4 + (lane >> 4), s = lane & 15;
        const int ch = (((s >> 1) ^ fsw(row & 63)) << 1) | (s & 1);   // logical 16-byte chunk that lands at position s
        glds16(pr.Wn + row * DM + ch * 8, wimg + piece * 1024);
      }
    }
  }
  epi_lnbwd<NW>(acc, pr.epi, p0, pw0, lane, wave, tid, smem);
  // E as MFMA fragments: every lane reads back the 16-byte pieces it has just stored (its own writes, L2 hits);
  // holding them in registers across the epilogue instead costs 32 registers there and 64 spilled ones
  const bf16* E = pr.epi.C2 ? pr.epi.C2 : (const bf16*)pr.epi.C;
  const int lde = pr.epi.C2 ? DM : pr.epi.ldc;
  bf16x8 keep[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < M) v = *reinterpret_cast<const uint4*>(E + (size_t)m * lde + ks * 32 + 8 * q);
      keep[mt][ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned char* img = wimg + (ks >> 1) * SLOT;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const bf16x8 wf = frag_t(img, (nt >> 1) * 32 + 8 * (lane & 3) + 4 * (nt & 1), ks & 1, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, keep[mt][ks], acc[mt][nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[mt][2 * tp][e]; v[4 + e] = acc[mt][2 * tp + 1][e]; }
      store8(pr.Cn + (size_t)m * DM + tp * 32 + 8 * q, v);
    }
  }
}

// ---- backward (activation gradients) ----------------------------------------------------------------------
// Same chained-GEMM structure with the transposed weight copies, so both GEMMs read row-major images:
//   g  = dz W2        first GEMM, weight image from W2^T [2048][128]
//   dh = g * mask(h)  mask = saved hidden activation > 0 (ReLU and dropout at once), h tile by LDS-DMA
//   dx += dh W1       second GEMM, weight image from W1^T [128][2048]
// Three images per chunk through a ring of four slots, three images ahead, one barrier per image.
template <bool SPLIT, int NW, bool SIGNS = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void ffn_bwd_kernel(const FfnBatch batch) {
  typedef Shape<NW> S;
  constexpr int MT = S::MT, WD = S::WD, HD = S::HD, RSLOT = S::RSLOT;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSLOT * RSLOT];
  WGLOG_BEGIN();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g = 0;
#pragma unroll
  for (int i = 1; i < FFN_MAX_GROUPS; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.p[i].tile_begin) g = i;
  const FfnProb& pr = batch.p[g];
  const int M = pr.epi.Pn;
  const int local = (int)blockIdx.x - pr.tile_begin;
  const int hsplit = SPLIT ? pr.hsplit : 1;
  const int p0 = (local / hsplit) * S::BPX;
  const int hs = local % hsplit;                      // this workgroup walks hidden chunks [c0, c1)
  const int c0 = SPLIT ? hs * NCHUNK / hsplit : 0, c1 = SPLIT ? (hs + 1) * NCHUNK / hsplit : NCHUNK;
  const int pw0 = wave * 16 * MT;
  const int q = lane >> 4;

  bf16x8 xf[MT][4];
  uint32_t hoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < M) v = *reinterpret_cast<const uint4*>(pr.X + (size_t)m * DM + ks * 32 + 8 * q);
      xf[mt][ks] = __builtin_bit_cast(bf16x8, v);
    }
    hoff[mt] = ((uint32_t)m * FF + 8 * q) * 2;
  }
  // SIGNS: the mask comes from the sign words of the rows (FfnProb::Hs), held in registers for the whole hidden
  // dimension and shifted down 16 bits per chunk; there is no hidden-activation image and nothing is stored in the loop
  uint32_t mk[SIGNS ? MT : 1][16];
  if constexpr (SIGNS) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = p0 + pw0 + mt * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 v = *reinterpret_cast<const uint4*>(pr.Hs + (size_t)(m < M ? m : M - 1) * 64 + q * 16 + j * 4);
        mk[mt][4 * j] = v.x; mk[mt][4 * j + 1] = v.y; mk[mt][4 * j + 2] = v.z; mk[mt][4 * j + 3] = v.w;
      }
    }
  }
  // optional copy of dh for a separate weight-gradient pass; zero records = every store dropped
  const auto drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(pr.dH ? pr.dH : pr.H), 0,
                                                       pr.dH ? (int)((uint32_t)M * FF * 2) : 0, 0x00020000);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[MT][8];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NIMG = SIGNS ? 2 : 3;
  const int TOTAL = (c1 - c0) * NIMG;
  const bool wdma = NW == S::NWD || wave < S::NWD;   // this wave issues weight-image DMA
  uint32_t offA[WD], offB[WD], offH[HD];
#pragma unroll
  for (int j = 0; j < WD; ++j) {
    offA[j] = src_offset<IMG_W1_FWD>(wave * 8 * WD + j * 8, lane, p0, M);
    offB[j] = src_offset<IMG_W2_FWD>(wave * 8 * WD + j * 8, lane, p0, M);
  }
#pragma unroll
  for (int j = 0; j < HD; ++j) offH[j] = src_offset<IMG_H_BWD>(pw0 + j * 8, lane, p0, M);
  // image i of the sequence = phase (i % 3) of chunk (i / 3), into ring slot (i % 4).  (The base pointers as values:
  // read through `pr` they would be fetched from the kernel arguments again at every issue, the loop's waits being
  // memory barriers to the compiler.)
  const bf16* const gH = pr.H;
  const bf16* const gW1 = pr.W1;
  const bf16* const gW2 = pr.W2;
  auto issue = [&](int c, int ph, int i) {
    unsigned char* slot = smem + (i & (NSLOT - 1)) * RSLOT;
    if (ph == 1) {
#pragma unroll
      for (int j = 0; j < HD; ++j)
        glds16(gH + offH[j] + (uint32_t)c * chunk_stride<IMG_H_BWD>(), slot + (pw0 + j * 8) * 128);
    } else if (wdma) {
#pragma unroll
      for (int j = 0; j < WD; ++j) {
        const bf16* src = ph == 0 ? gW1 + offA[j] + (uint32_t)c * chunk_stride<IMG_W1_FWD>()
                                  : gW2 + offB[j] + (uint32_t)c * chunk_stride<IMG_W2_FWD>();
        glds16(src, slot + (wave * WD + j) * 1024);
      }
    }
  };
  // Own DMA portions of image i have landed once only what was issued after them is outstanding: the two younger
  // images and the hidden-gradient stores of the one mask step among the last three steps (all three kinds count
  // WD = HD instructions on a wave that issues weight images).  A wave that only brings in hidden-activation rows
  // has, behind such an image, just the stores of the previous mask step.
  auto wait_image = [&](int i, int ph) {
    if constexpr (SIGNS) {   // weight images only: two younger ones in flight
      if (!wdma) return;
      if (i + 2 < TOTAL) {
        if (WD == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else if (i + 1 < TOTAL) {
        if (WD == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      return;
    }
    if (!wdma) {
      if (ph == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      return;
    }
    if (i + 2 < TOTAL) {
      if (WD == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (i + 1 < TOTAL) {
      if (WD == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  auto qrow = [&](int nt) { return (nt >> 1) * 32 + 8 * ((lane & 15) >> 2) + 4 * (nt & 1) + (lane & 3); };
  const float dscale = pr.drop_p > 0.f ? 1.0f / (1.0f - pr.drop_p) : 1.0f;

  issue(c0, 0, 0);
  if constexpr (SIGNS) {
    issue(c0, 2, 1);
    issue(c0 + 1, 0, 2);
  } else {
    issue(c0, 1, 1);
    issue(c0, 2, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the counted waits below assume a full pipeline behind them

  for (int c = c0; c < c1; ++c) {
    int i = (c - c0) * NIMG;
    f32x4 a1[MT][4];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) a1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      wait_image(i, 0);
      __builtin_amdgcn_s_barrier();
      if (i + AHEAD < TOTAL) issue(c + 1, SIGNS ? 2 : 0, i + AHEAD);
      const unsigned char* img = smem + (i & (NSLOT - 1)) * RSLOT;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 qf[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) qf[nt] = frag_n<true>(img, (ks >> 1) * 64 + qrow(nt), ks & 1, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            a1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], xf[mt][ks], a1[mt][nt], 0, 0, 0);
      }
      ++i;
    }
    bf16x8 hf[MT][2];
    if constexpr (SIGNS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const uint32_t mw = mk[mt][0] << (8 * (c & 1));   // the flags of an odd chunk sit 8 bits below those of an even one
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          // on pairs: the two flags of a pair sit 16 bits apart (sign_bit()), so shifted to bits 15 / 31 an arithmetic
          // shift of each 16-bit half by 15 spreads them into the mask word of the pair's two bf16 values
          u32x4 ow;
#pragma unroll
          for (int pr2 = 0; pr2 < 4; ++pr2) {
            const int e = 2 * pr2;
            const float g0 = e < 4 ? a1[mt][2 * tp][e] : a1[mt][2 * tp + 1][e - 4];
            const float g1 = e < 4 ? a1[mt][2 * tp][e + 1] : a1[mt][2 * tp + 1][e - 3];
            const f32x2 gs = {g0 * dscale, g1 * dscale};
            const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(gs, bf16x2));
            static_assert(sign_bit(0, 0, 1) - sign_bit(0, 0, 0) == 16, "flags of a pair 16 bits apart");
            uint32_t on = mw << (15 - sign_bit(0, tp, e));
            asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(on) : "v"(on));   // (the constant's low half for both shifts)
            ow[pr2] = pk & on;
          }
          hf[mt][tp] = __builtin_bit_cast(bf16x8, ow);
        }
        if (c & 1) {
#pragma unroll
          for (int j = 0; j < 15; ++j) mk[mt][j] = mk[mt][j + 1];
        }
      }
    } else {
      wait_image(i, 1);
      __builtin_amdgcn_s_barrier();
      if (i + AHEAD < TOTAL) issue(c + 1, 1, i + AHEAD);
      const unsigned char* himg = smem + (i & (NSLOT - 1)) * RSLOT;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          // dh = (h > 0 ? g / (1 - p) : 0) on PAIRS of elements: the scaled pair is converted to two bf16 in one
          // instruction and ANDed with a mask built from the saved activation's bit patterns (h >= 0, so a half is
          // "on" when its magnitude bits are not zero: min(bits, 1) * 0xFFFF per half, packed 16-bit instructions):
          // 3 instructions per element instead of 4.7 with a compare and a select per element.
          const u32x4 hw = __builtin_bit_cast(u32x4, frag_n<false>(himg, pw0 + mt * 16 + (lane & 15), tp, lane));
          u32x4 ow;
#pragma unroll
          for (int pr2 = 0; pr2 < 4; ++pr2) {
            const int e = 2 * pr2;
            const float g0 = e < 4 ? a1[mt][2 * tp][e] : a1[mt][2 * tp + 1][e - 4];
            const float g1 = e < 4 ? a1[mt][2 * tp][e + 1] : a1[mt][2 * tp + 1][e - 3];
            const f32x2 gs = {g0 * dscale, g1 * dscale};
            const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(gs, bf16x2));
            // (as instructions: written with vector types the two packed operations come back as a compare and a
            // select per half)
            uint32_t on = hw[pr2] & 0x7FFF7FFFu;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(on) : "v"(on), "s"(0x00010001u));
            asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(on) : "v"(on), "s"(0xFFFFFFFFu));
            ow[pr2] = pk & on;
          }
          const bf16x8 o = __builtin_bit_cast(bf16x8, ow);
          hf[mt][tp] = o;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), drsrc,
                                                 (int)(hoff[mt] + (uint32_t)(c * HC + tp * 32) * 2), 0, 0);
        }
      ++i;
    }
    {
      wait_image(i, 2);
      __builtin_amdgcn_s_barrier();
      if (i + AHEAD < TOTAL) issue(SIGNS ? c + 2 : c + 1, SIGNS ? 0 : 2, i + AHEAD);
      const unsigned char* img = smem + (i & (NSLOT - 1)) * RSLOT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 qf[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) qf[nt] = frag_n<true>(img, qrow(nt), ks, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 8; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], hf[mt][ks], acc[mt][nt], 0, 0, 0);
      }
    }
  }

  if constexpr (SPLIT) {   // partial sums; the row epilogue runs as a second launch over all slabs
    store_slab(acc, pr.ws + (size_t)hs * M * DM, M, p0, pw0, lane);
  } else {
    bwd_rows_epilogue<NW>(acc, pr, p0, pw0, lane, wave, tid, smem, smem + 2 * RSLOT);
  }
  WGLOG_END(1);
}

// Sign words from a stored hidden activation, for the forward paths that do not produce them in the kernel (hidden
// split, two-GEMM form): one thread per word.
__global__ __launch_bounds__(256) void ffn_signs_kernel(const FfnBatch batch) {
  for (int gi = 0; gi < batch.n; ++gi) {
    const FfnProb& pr = batch.p[gi];
    if (!pr.Hs) continue;
    const long words = (long)pr.epi.Pn * 64;
    for (long w = (long)blockIdx.x * 256 + threadIdx.x; w < words; w += (long)gridDim.x * 256) {
      const long m = w >> 6;
      const int qq = (int)(w >> 4) & 3, j = (int)w & 15;
      uint32_t bits = 0;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          const uint4 v = *reinterpret_cast<const uint4*>(pr.H + m * FF + (2 * j + cc) * HC + tp * 32 + 8 * qq);
          const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t hv = (d[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
            if (hv != 0 && !(hv & 0x8000u)) bits |= 1u << sign_bit(cc, tp, e);
          }
        }
      pr.Hs[w] = bits;
    }
  }
}

// ---- forward, software-pipelined across chunks ------------------------------------------------------------
// One barrier per chunk.  Step c runs, in one straight-line region,
//   first GEMM of chunk c            (MFMA, image W1(c))
//   second GEMM of chunk c-1         (MFMA, image W2(c-1), operand = bf16 hidden fragments of the previous step)
//   bias/ReLU/dropout of chunk c     (VALU, fills the MFMA shadow) -> store + fragments for the next step
// LDS: two slots per weight (the image being read, the image in flight), filled right after the barrier.
template <bool DROP, int NW, bool SIGNS>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void ffn_fwd_kernel(const FfnBatch batch) {
  typedef Shape<NW> S;
  constexpr int MT = S::MT, WD = S::WD;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * SLOT + FF * 4 + 9 * DM * 4];
  unsigned char* w1s = smem;             // 2 slots
  unsigned char* w2s = smem + 2 * SLOT;  // 2 slots
  WGLOG_BEGIN();
  float* b1s = reinterpret_cast<float*>(smem + 4 * SLOT);
  float* vecs = b1s + FF;                // bias | gamma | beta of the final row epilogue, then of the prologue's,
                                         // then the bias of the chained linear layer (up to 384 columns)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g = 0;
#pragma unroll
  for (int i = 1; i < FFN_MAX_GROUPS; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.p[i].tile_begin) g = i;
  const FfnProb& pr = batch.p[g];
  const int M = pr.epi.Pn;
  const int local = (int)blockIdx.x - pr.tile_begin;
  const int p0 = (local / pr.hsplit) * S::BPX;
  const int hs = local % pr.hsplit;                   // this workgroup walks hidden chunks [c0, c1)
  const int c0 = hs * NCHUNK / pr.hsplit, c1 = (hs + 1) * NCHUNK / pr.hsplit;
  const int pw0 = wave * 16 * MT;
  const int q = lane >> 4;

  const bool has_pre = pr.Wp != nullptr && pr.hsplit == 1;
  const bool wdma = NW == S::NWD || wave < S::NWD;   // this wave issues weight-image DMA
  uint32_t offA[WD], offB[WD];
#pragma unroll
  for (int j = 0; j < WD; ++j) {
    offA[j] = src_offset<IMG_W1_FWD>(wave * 8 * WD + j * 8, lane, p0, M);
    offB[j] = src_offset<IMG_W2_FWD>(wave * 8 * WD + j * 8, lane, p0, M);
  }
  // (base pointers as values: read through `pr` they would be fetched from the kernel arguments again at every issue)
  const bf16* const gW1 = pr.W1;
  const bf16* const gW2 = pr.W2;
  auto issue_w1 = [&](int c) {
    unsigned char* dst = w1s + (c & 1) * SLOT + wave * WD * 1024;
    if (wdma)
#pragma unroll
      for (int j = 0; j < WD; ++j) glds16(gW1 + offA[j] + (uint32_t)c * chunk_stride<IMG_W1_FWD>(), dst + j * 1024);
  };
  auto issue_w2 = [&](int c) {
    unsigned char* dst = w2s + (c & 1) * SLOT + wave * WD * 1024;
    if (wdma)
#pragma unroll
      for (int j = 0; j < WD; ++j) glds16(gW2 + offB[j] + (uint32_t)c * chunk_stride<IMG_W2_FWD>(), dst + j * 1024);
  };
  // Everything the first step needs is requested before anything is waited for: the first linear1 image and, with a
  // prologue, the two 64-column halves of Wp as first-GEMM images in the (still unused) slots of the second weight.
  if (wdma && has_pre) {
#pragma unroll
    for (int j = 0; j < WD; ++j) {
      glds16(pr.Wp + offA[j], w2s + wave * WD * 1024 + j * 1024);
      glds16(pr.Wp + offA[j] + chunk_stride<IMG_W1_FWD>(), w2s + SLOT + wave * WD * 1024 + j * 1024);
    }
  }
  issue_w1(c0);
  // token-side fragments: the block input, or (prologue) the input of the linear layer in front of it
  bf16x8 xf[MT][4];
  uint32_t hoff[MT], rseed[MT];
  auto load_rows = [&](const bf16* src) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = p0 + pw0 + mt * 16 + (lane & 15);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m < M) v = *reinterpret_cast<const uint4*>(src + (size_t)m * DM + ks * 32 + 8 * q);
        xf[mt][ks] = __builtin_bit_cast(bf16x8, v);
      }
    }
  };
  load_rows(has_pre ? pr.A : pr.X);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
    hoff[mt] = ((uint32_t)m * FF + 8 * q) * 2;   // byte offset; rows >= M fall outside the descriptor and are dropped
    rseed[mt] = attn_row_seed(pr.seed, (uint32_t)m);
  }
  // relu(a + b) * s == relu(a * s + b * s) for s > 0: the bias is staged pre-multiplied by the dropout scale
  const float dscale = (DROP && pr.drop_p > 0.f) ? 1.0f / (1.0f - pr.drop_p) : 1.0f;
  for (int i = tid; i < FF / 4; i += NW * 64)
    *reinterpret_cast<f32x4*>(b1s + i * 4) = *reinterpret_cast<const f32x4*>(pr.b1 + i * 4) * dscale;
  // the row epilogues read their vectors from LDS, not from memory at the end of a dependent chain
  for (int i = tid; i < (has_pre ? 6 : 3) * DM; i += NW * 64) {
    const GemmProb& e = i < 3 * DM ? pr.epi : pr.pre;
    const int w = (i / DM) % 3, col = i % DM;
    const float* src = w == 0 ? e.bias : (w == 1 ? e.gamma : e.beta);
    vecs[i] = src ? src[col] : 0.f;
  }
  const bool has_post = pr.Wq != nullptr && pr.hsplit == 1;
  if (has_post)
    for (int i = tid; i < pr.nq; i += NW * 64) vecs[6 * DM + i] = pr.bq ? pr.bq[i] : 0.f;
  // hidden activation goes out through a buffer descriptor: the range check replaces a per-row branch
  const auto hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pr.H, 0, (int)((uint32_t)M * FF * 2), 0x00020000);
  // sign words of the hidden activation (see FfnProb::Hs): the 16 flags of a chunk are gathered with three operations
  // per pair of entries, two chunks make a word, and four words go out as one 16-byte store every 8 chunks; zero
  // records = every store dropped
  const auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(pr.Hs ? (void*)pr.Hs : (void*)pr.H), 0,
                                                       pr.Hs ? (int)((uint32_t)M * 256) : 0, 0x00020000);
  uint32_t fm[MT][4], fcur[MT], soff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    soff[mt] = (uint32_t)(p0 + pw0 + mt * 16 + (lane & 15)) * 256 + q * 64;
    fcur[mt] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) fm[mt][j] = 0;
  }

  f32x4 acc[MT][8];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto qrow = [&](int nt) { return (nt >> 1) * 32 + 8 * ((lane & 15) >> 2) + 4 * (nt & 1) + (lane & 3); };

  const uint32_t thr = drop_threshold(pr.drop_p);
  const uint32_t lane_pt = (uint32_t)(4 * q) * ATTN_PAIR_MUL;

  f32x4 a1[MT][4];
  bf16x8 hfp[MT][2];
  auto gemm1_img = [&](const unsigned char* img) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) a1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 qf[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) qf[nt] = frag_n<true>(img, (ks >> 1) * 64 + qrow(nt), ks & 1, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          a1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], xf[mt][ks], a1[mt][nt], 0, 0, 0);
    }
  };
  auto gemm1 = [&](int c) __attribute__((always_inline)) { gemm1_img(w1s + (c & 1) * SLOT); };
  // Second GEMM of chunk c2 and hidden-chunk arithmetic of chunk c in ONE hand-ordered region: the 16 MT MFMAs go
  // out MT at a time, each group followed by one 2-column piece of the bias/ReLU/dropout/convert work (8 MT pieces
  // over 16 slots), so the vector ALU runs in the shadow of the matrix pipe.  has2 / hasc are compile-time (the region
  // stays free of branches): whether there is a second GEMM (chunk c2) / hidden-chunk arithmetic (chunk c).
  auto gemm2_hidden = [&](auto has2_, auto hasc_, int c2, int c) __attribute__((always_inline)) {
    constexpr bool has2 = decltype(has2_)::value, hasc = decltype(hasc_)::value;
    const unsigned char* img = w2s + (c2 & 1) * SLOT;
    const uint32_t chunk_pt = (uint32_t)(c * (HC / 2)) * ATTN_PAIR_MUL;
    bf16x8 qf[2][4];
    float bv[2][8];
    uint32_t hn[MT][2][4], nb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) nb[mt] = 0;
    if constexpr (has2) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) qf[0][nt] = frag_n<true>(img, qrow(nt), 0, lane);
    }
    if constexpr (hasc) {
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) load8f(bv[tp], b1s + c * HC + tp * 32 + 8 * q);
    }
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      const int ks = grp >> 1, nt0 = (grp & 1) * 4;
      if constexpr (has2) {
        if (grp + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            qf[(grp + 1) & 1][nt] = frag_n<true>(img, qrow(((grp + 1) & 1) * 4 + nt), (grp + 1) >> 1, lane);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (has2) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][nt0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[grp & 1][j], hfp[mt][ks], acc[mt][nt0 + j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int slot = grp * 4 + j;
        if (hasc && (MT == 2 || (slot & 1) == 0)) {
          const int piece = MT == 2 ? slot : slot >> 1;
          const int tp = MT == 2 ? piece >> 3 : piece >> 2, mt = MT == 2 ? (piece >> 2) & 1 : 0, e = (piece & 3) * 2;
          float v0 = a1[mt][2 * tp + (e >> 2)][e & 3], v1 = a1[mt][2 * tp + (e >> 2)][(e & 3) + 1];
          v0 = fmaxf(fmaf(v0, dscale, bv[tp][e]), 0.f);
          v1 = fmaxf(fmaf(v1, dscale, bv[tp][e + 1]), 0.f);
          if (DROP) {
            const uint32_t hb = attn_pair_bits(rseed[mt] + lane_pt + chunk_pt, (uint32_t)(tp * 16 + (e >> 1)) * ATTN_PAIR_MUL);
            v0 = ((hb & 0xFFFFu) >= thr) ? v0 : 0.f;
            v1 = ((hb >> 16) >= thr) ? v1 : 0.f;
          }
          const f32x2 v2 = {v0, v1};
          const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, bf16x2));   // one packed convert
          hn[mt][tp][e >> 1] = pk;
          if constexpr (SIGNS) {
            // both halves are bf16 values >= 0: adding 0x7FFF carries into bit 15 / 31 exactly when a half is not zero
            const uint32_t t = pk + 0x7FFF7FFFu;
            const int i = tp_e_index(tp, e);   // a constant once the region is unrolled
            nb[mt] = (__builtin_amdgcn_alignbit(t, t, i) & (0x80008000u >> i)) | nb[mt];
          }
          if ((piece & 3) == 3) {
            u32x4 o = {hn[mt][tp][0], hn[mt][tp][1], hn[mt][tp][2], hn[mt][tp][3]};
            __builtin_amdgcn_raw_buffer_store_b128(o, hrsrc, (int)(hoff[mt] + (uint32_t)(c * HC + tp * 32) * 2), 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (hasc) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          u32x4 o = {hn[mt][tp][0], hn[mt][tp][1], hn[mt][tp][2], hn[mt][tp][3]};
          hfp[mt][tp] = __builtin_bit_cast(bf16x8, o);
        }
      if constexpr (SIGNS) {
        if ((c & 1) == 0) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) fcur[mt] = nb[mt];
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            fm[mt][0] = fm[mt][1]; fm[mt][1] = fm[mt][2]; fm[mt][2] = fm[mt][3];
            fm[mt][3] = fcur[mt] | (nb[mt] >> 8);
          }
          if ((c & 7) == 7) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              u32x4 o = {fm[mt][0], fm[mt][1], fm[mt][2], fm[mt][3]};
              __builtin_amdgcn_raw_buffer_store_b128(o, srsrc, (int)(soff[mt] + (uint32_t)(c >> 3) * 16), 0, 0);
            }
          }
        }
      }
    }
  };
  constexpr std::true_type yes{};
  constexpr std::false_type no{};

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (has_pre) {
    // X = LN(R + drop(A Wp^T + bias)) on this tile's rows: the 128 outputs of a row land in the accumulators in the row
    // epilogue's layout, and the epilogue hands the rows back as the block's input fragments
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      gemm1_img(w2s + c * SLOT);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][4 * c + nt] = a1[mt][nt];
    }
    epi_ln(acc, pr.pre, p0, pw0, lane, xf, vecs + 3 * DM);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();   // every wave has read the Wp images: the second weight's slots may be filled
  }
  issue_w1(c0 + 1);   // a workgroup walks at least four chunks
  issue_w2(c0);
  gemm1(c0);
  gemm2_hidden(no, yes, -1, c0);

  // everything older than the 2 MT hidden-activation stores of the previous step (the MT sign-word stores that follow
  // them every eighth step make this wait stricter than needed there, never weaker)
  auto wait_images = [&]() {
    if (MT == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  };
  // last_ (compile-time): the final chunk, no further first-GEMM image to request
  auto step = [&](auto last_, int c) __attribute__((always_inline)) {
    wait_images();   // both images of this step have landed
    __builtin_amdgcn_s_barrier();
    if constexpr (!decltype(last_)::value) issue_w1(c + 1);
    issue_w2(c);
    gemm1(c);
    __builtin_amdgcn_sched_barrier(0);
    gemm2_hidden(yes, yes, c - 1, c);
  };
  for (int c = c0 + 1; c + 1 < c1; ++c) step(no, c);
  step(yes, c1 - 1);
  wait_images();
  __builtin_amdgcn_s_barrier();
  // Chained linear layer: its weight arrives as first-GEMM images, 64 output columns each; the first two are requested
  // here (every wave is past its last first GEMM) and land during the last second GEMM and the row epilogue.
  const int nqc = has_post ? pr.nq / HC : 0;
  auto issue_wq = [&](int c) {
    unsigned char* dst = w1s + (c & 1) * SLOT + wave * WD * 1024;
    if (wdma)
#pragma unroll
      for (int j = 0; j < WD; ++j) glds16(pr.Wq + offA[j] + (uint32_t)c * chunk_stride<IMG_W1_FWD>(), dst + j * 1024);
  };
  if (nqc > 0) issue_wq(0);
  if (nqc > 1) issue_wq(1);
  gemm2_hidden(yes, no, c1 - 1, -1);

  if (pr.hsplit > 1) {
    store_slab(acc, pr.ws + (size_t)hs * M * DM, M, p0, pw0, lane);
    return;
  }
  epi_ln(acc, pr.epi, p0, pw0, lane, has_post ? xf : nullptr, vecs);
  for (int c = 0; c < nqc; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own portions of image c (and the stores behind it)
    __builtin_amdgcn_s_barrier();                        // everyone's; and image c - 1 has been read by every wave
    if (c >= 1 && c + 1 < nqc) issue_wq(c + 1);
    gemm1_img(w1s + (c & 1) * SLOT);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = p0 + pw0 + mt * 16 + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        float v[8], bb[8];
        load8f(bb, vecs + 6 * DM + c * HC + tp * 32 + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a1[mt][2 * tp][e] + bb[e]; v[4 + e] = a1[mt][2 * tp + 1][e] + bb[4 + e]; }
        store8(pr.Cq + (size_t)m * pr.nq + c * HC + tp * 32 + 8 * q, v);
      }
    }
  }
  WGLOG_END(2);
}

int prepare(FfnBatch& b, bool bwd, int bpx) {
  if (b.n < 1 || b.n > FFN_MAX_GROUPS) return -1;
  int tiles = 0;
  bool have_ws = true;
  for (int i = 0; i < b.n; ++i) {
    FfnProb& p = b.p[i];
    if (((long)p.epi.Pn + bpx) * FF * 2 >= (1L << 31)) return -1;   // 32-bit byte offsets into the hidden activation
    if (p.epi.Pn <= 0 || !p.X || !p.W1 || !p.W2 || !p.H || !p.epi.C) return -1;
    if (!bwd && (!p.b1 || !p.epi.gamma || !p.epi.beta)) return -1;
    if (bwd && (!p.epi.gamma || !p.epi.Z || !p.epi.stats)) return -1;
    p.epi.Qn = DM;
    tiles += (p.epi.Pn + bpx - 1) / bpx;
    have_ws = have_ws && p.ws != nullptr;
  }
  // few row tiles: split the hidden dimension over up to FFN_MAX_HSPLIT workgroups per tile (needs the workspace)
  int hsplit = 1;
  if (have_ws)
    while (hsplit < FFN_MAX_HSPLIT && tiles * hsplit * 2 <= 256) hsplit *= 2;
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    FfnProb& p = b.p[i];
    p.hsplit = hsplit;
    p.tile_begin = total;
    total += ((p.epi.Pn + bpx - 1) / bpx) * hsplit;
  }
  return total;
}

// Second stage of a hidden-split launch, 32 rows per workgroup.  Few rows by construction, so the kernel is
// built around memory latency, not throughput: its 4 waves each fetch a quarter of the fp32 slabs in ONE round of
// independent loads, three of them hand their partial sums over through LDS and end, and the first wave (the
// layout in which a wave owns whole rows) adds them and runs the row epilogue.
template <bool BWD>
__global__ __launch_bounds__(256) void ffn_rows_kernel(const FfnBatch batch) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 16 * 64 * 16];   // 3 partial accumulators, 48 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g = 0, first = 0, acc_blocks = 0;
#pragma unroll
  for (int i = 0; i < FFN_MAX_GROUPS; ++i) {
    if (i < batch.n) {
      if (i > 0 && (int)blockIdx.x >= acc_blocks) { g = i; first = acc_blocks; }
      acc_blocks += (batch.p[i].epi.Pn + 31) / 32;
    }
  }
  const FfnProb& pr = batch.p[g];
  const int M = pr.epi.Pn, p0 = ((int)blockIdx.x - first) * 32;
  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  add_slabs<FFN_MAX_HSPLIT / 4>(acc, pr.ws, wave, 4, pr.hsplit, M, p0, 0, lane);
  f32x4* part = reinterpret_cast<f32x4*>(smem);
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) part[((wave - 1) * 16 + i * 8 + j) * 64 + lane] = acc[i][j];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] += part[(w * 16 + i * 8 + j) * 64 + lane];
  if (BWD) epi_lnbwd<1>(acc, pr.epi, p0, 0, lane, 0, lane, smem);
  else epi_ln(acc, pr.epi, p0, 0, lane);
}

int row_epilogue(const FfnBatch& b, bool bwd, hipStream_t stream) {
  int blocks = 0;
  for (int i = 0; i < b.n; ++i) blocks += (b.p[i].epi.Pn + 31) / 32;
  if (bwd) hipLaunchKernelGGL((ffn_rows_kernel<true>), dim3(blocks), dim3(256), 0, stream, b);
  else hipLaunchKernelGGL((ffn_rows_kernel<false>), dim3(blocks), dim3(256), 0, stream, b);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

// Below this many 128-row tiles the fused kernels cannot fill the chip (one workgroup walks the whole hidden
// dimension serially), so short sequences (the decoder) run the block as two grouped GEMM launches whose
// first GEMM also spreads over the hidden dimension.
int g_fuse_min_tiles = -1;
int fuse_min_tiles() {
  if (g_fuse_min_tiles < 0) {
    const char* e = getenv("MMST_FFN_FUSE_MIN_TILES");
    g_fuse_min_tiles = e ? atoi(e) : 96;
    if (g_fuse_min_tiles < 0) g_fuse_min_tiles = 96;
  }
  return g_fuse_min_tiles;
}

// Rows per wave of the fused kernels: 16 (8 or 10 waves per tile: two to three waves per SIMD to hide the
// dependent LDS/MFMA latencies behind, at twice the LDS fragment traffic) or 32 (4 waves per 128-row tile).
// The forward pass runs alone on the GPU and takes the 16-row shape; the backward kernels share the CUs with the
// weight-gradient stream, which already fills their idle issue slots, and measured slightly better with the
// lighter 4-wave shape.  Environment: MMST_FFN_WAVE_ROWS_FWD / MMST_FFN_WAVE_ROWS_BWD (16 or 32).
int wave_rows(bool bwd) {
  static const int f = [] { const char* e = getenv("MMST_FFN_WAVE_ROWS_FWD"); return (e && atoi(e) == 32) ? 32 : 16; }();
  static const int b = [] { const char* e = getenv("MMST_FFN_WAVE_ROWS_BWD"); return (e && atoi(e) == 16) ? 16 : 32; }();
  return bwd ? b : f;
}

int ffn_fwd_unfused(FfnBatch& b, hipStream_t stream) {
  GemmBatch g1, g2;
  g1.n = g2.n = b.n;
  for (int i = 0; i < b.n; ++i) {
    const FfnProb& f = b.p[i];
    GemmProb a;
    memset(&a, 0, sizeof(a));
    a.ksplit = 1;
    a.P = f.X; a.ldp = DM; a.Q = f.W1; a.ldq = DM; a.C = f.H; a.ldc = FF; a.bias = f.b1;
    a.Pn = f.epi.Pn; a.Qn = FF; a.K = DM;
    a.flags = GF_RELU; a.drop_p = f.drop_p; a.seed = f.seed;
    g1.p[i] = a;
    GemmProb c = f.epi;
    c.P = f.H; c.ldp = FF; c.Q = f.W2; c.ldq = FF; c.Qn = DM; c.K = FF;
    g2.p[i] = c;
  }
  int rc = launch_gemm(g1, EPI_BIAS, false, false, stream);
  if (rc) return rc;
  return launch_gemm(g2, EPI_LN, false, false, stream);
}

int ffn_bwd_unfused(FfnBatch& b, hipStream_t stream) {
  GemmBatch g1, g2;
  g1.n = g2.n = b.n;
  for (int i = 0; i < b.n; ++i) {
    const FfnProb& f = b.p[i];
    if (!f.dH) return MMST_ERR_ARG;
    GemmProb a;
    memset(&a, 0, sizeof(a));
    a.ksplit = 1;
    a.P = f.X; a.ldp = DM; a.Q = f.W1; a.ldq = DM; a.C = f.dH; a.ldc = FF;   // W1 here = linear2^T [2048][128]
    a.R = f.H; a.ldr = FF; a.drop_p = f.drop_p;
    a.Pn = f.epi.Pn; a.Qn = FF; a.K = DM;
    g1.p[i] = a;
    GemmProb c = f.epi;
    c.P = f.dH; c.ldp = FF; c.Q = f.W2; c.ldq = FF; c.Qn = DM; c.K = FF;      // W2 here = linear1^T [128][2048]
    g2.p[i] = c;
  }
  int rc = launch_gemm(g1, EPI_DRELU, false, false, stream);
  if (rc) return rc;
  return launch_gemm(g2, EPI_LNBWD, false, false, stream);
}

// Workgroup shape of a fused launch: 8 waves of 16 rows; 10 waves when that brings a launch of more than 256
// 128-row tiles down to one workgroup per CU; or the 4-wave shape, see wave_rows().
int pick_waves(const FfnBatch& b, bool bwd) {
  if (wave_rows(bwd) == 32) return 4;
  int t128 = 0, t160 = 0;
  for (int i = 0; i < b.n; ++i) {
    t128 += (b.p[i].epi.Pn + 127) / 128;
    t160 += (b.p[i].epi.Pn + 159) / 160;
  }
  return (t128 > 256 && t160 <= 256) ? 10 : 8;
}

template <int NW>
void launch_fwd_shape(const FfnBatch& b, int total, bool drop, bool signs, hipStream_t stream) {
  if (signs) {
    if (drop) hipLaunchKernelGGL((ffn_fwd_kernel<true, NW, true>), dim3(total), dim3(NW * 64), 0, stream, b);
    else hipLaunchKernelGGL((ffn_fwd_kernel<false, NW, true>), dim3(total), dim3(NW * 64), 0, stream, b);
  } else {
    if (drop) hipLaunchKernelGGL((ffn_fwd_kernel<true, NW, false>), dim3(total), dim3(NW * 64), 0, stream, b);
    else hipLaunchKernelGGL((ffn_fwd_kernel<false, NW, false>), dim3(total), dim3(NW * 64), 0, stream, b);
  }
}
template <int NW>
void launch_bwd_shape(const FfnBatch& b, int total, bool split, bool signs, hipStream_t stream) {
  if (split) hipLaunchKernelGGL((ffn_bwd_kernel<true, NW>), dim3(total), dim3(NW * 64), 0, stream, b);
  else if (signs) hipLaunchKernelGGL((ffn_bwd_kernel<false, NW, true>), dim3(total), dim3(NW * 64), 0, stream, b);
  else hipLaunchKernelGGL((ffn_bwd_kernel<false, NW>), dim3(total), dim3(NW * 64), 0, stream, b);
}
// sign words for the forward paths whose kernels do not write them
int signs_from_hidden(const FfnBatch& b, hipStream_t stream) {
  long words = 0;
  for (int i = 0; i < b.n; ++i)
    if (b.p[i].Hs) words += (long)b.p[i].epi.Pn * 64;
  if (words == 0) return MMST_OK;
  const long blocks = (words + 255) / 256;
  hipLaunchKernelGGL(ffn_signs_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, b);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

}  // namespace

// The linear layer + LayerNorm in front of the block (FfnProb::Wp) as a grouped GEMM launch of its own, for the
// paths whose kernels have no prologue; clears Wp.
int pre_as_gemm(FfnBatch& b, hipStream_t stream) {
  GemmBatch g;
  g.n = b.n;
  for (int i = 0; i < b.n; ++i) {
    FfnProb& f = b.p[i];
    GemmProb q = f.pre;
    q.P = f.A; q.ldp = DM; q.Q = f.Wp; q.ldq = DM;
    q.Pn = f.epi.Pn; q.Qn = DM; q.K = DM; q.ksplit = 1;
    g.p[i] = q;
    f.Wp = nullptr;
  }
  return launch_gemm(g, EPI_LN, false, false, stream);
}

int launch_ffn_fwd(FfnBatch& b, hipStream_t stream, bool* post_done) {
  if (post_done) *post_done = false;
  if (b.n < 1 || b.n > FFN_MAX_GROUPS) return MMST_ERR_ARG;
  bool pre = b.p[0].Wp != nullptr;
  // the chained linear layer is all problems or none, and needs the rows where the next launch reads them
  static const bool post_fused = [] { const char* e = getenv("MMST_FFN_POST_FUSED"); return e ? atoi(e) != 0 : true; }();
  bool post = post_fused;
  for (int i = 0; i < b.n; ++i) {
    const FfnProb& f = b.p[i];
    post = post && f.Wq != nullptr && f.Cq != nullptr && f.nq >= HC && f.nq <= 3 * DM && f.nq % HC == 0 && f.epi.cS == 0;
  }
  for (int i = 0; i < b.n; ++i) {
    FfnProb& f = b.p[i];
    if ((f.Wp != nullptr) != pre) return MMST_ERR_ARG;   // all problems or none
    if (pre && (!f.A || f.pre.C != (void*)f.X || f.pre.ldc != DM || f.pre.cS > 0 || !f.pre.gamma || !f.pre.beta))
      return MMST_ERR_ARG;
  }
  static const bool pre_fused = [] { const char* e = getenv("MMST_FFN_PRE_FUSED"); return e ? atoi(e) != 0 : true; }();
  const int nw = pick_waves(b, false);
  int total = prepare(b, false, nw == 10 ? 160 : 128);
  if (total < 0) return MMST_ERR_ARG;
  const bool split = b.p[0].hsplit > 1;
  if (pre && (split || total < fuse_min_tiles() || !pre_fused)) {
    const int rc = pre_as_gemm(b, stream);
    if (rc) return rc;
  }
  post = post && !split && total >= fuse_min_tiles();
  if (!post)
    for (int i = 0; i < b.n; ++i) b.p[i].Wq = nullptr;
  else if (post_done) *post_done = true;
  if (!split && total < fuse_min_tiles()) {
    const int rc = ffn_fwd_unfused(b, stream);
    return rc ? rc : signs_from_hidden(b, stream);
  }
  bool drop = false;
  for (int i = 0; i < b.n; ++i) drop = drop || b.p[i].drop_p > 0.f;
  bool signs = false;   // written in the kernel only by workgroups that walk the whole hidden dimension
  for (int i = 0; i < b.n; ++i) signs = signs || (b.p[i].Hs != nullptr && !split);
  if (nw == 4) launch_fwd_shape<4>(b, total, drop, signs, stream);
  else if (nw == 8) launch_fwd_shape<8>(b, total, drop, signs, stream);
  else launch_fwd_shape<10>(b, total, drop, signs, stream);
  MMST_CHECK_LAUNCH();
  if (split) {   // the kernel's sign-word stores assume a workgroup that walks the whole hidden dimension
    const int rc = signs_from_hidden(b, stream);
    return rc ? rc : row_epilogue(b, false, stream);
  }
  return MMST_OK;
}

int launch_ffn_bwd(FfnBatch& b, hipStream_t stream, bool* next_done) {
  if (next_done) *next_done = false;
  if (b.n < 1 || b.n > FFN_MAX_GROUPS) return MMST_ERR_ARG;
  bool chained = true;
  for (int i = 0; i < b.n; ++i) {
    if ((b.p[i].Wn == nullptr) != (b.p[i].Cn == nullptr)) return MMST_ERR_ARG;
    chained = chained && b.p[i].Wn != nullptr;
  }
  const int nw = pick_waves(b, true);
  int total = prepare(b, true, nw == 10 ? 160 : 128);
  if (total < 0) return MMST_ERR_ARG;
  const bool split = b.p[0].hsplit > 1;
  if (!split && total < fuse_min_tiles()) return ffn_bwd_unfused(b, stream);
  if (split || !chained)   // the chained GEMM is all problems or none, and lives in the kernel without hidden split
    for (int i = 0; i < b.n; ++i) b.p[i].Wn = nullptr;
  else if (next_done) *next_done = true;
  // sign words for every problem and no stored hidden gradient: the kernel that never reads the hidden activation
  bool signs = true;
  for (int i = 0; i < b.n; ++i) signs = signs && b.p[i].Hs != nullptr && b.p[i].dH == nullptr;
  if (nw == 4) launch_bwd_shape<4>(b, total, split, signs, stream);
  else if (nw == 8) launch_bwd_shape<8>(b, total, split, signs, stream);
  else launch_bwd_shape<10>(b, total, split, signs, stream);
  MMST_CHECK_LAUNCH();
  if (split) return row_epilogue(b, true, stream);
  return MMST_OK;
}

// Weight gradients of fused launches: 1 (default), 2 or 3 = ffn_wgrad.hip recomputes the hidden gradient, which is then
// never written or re-read as a [rows][2048] matrix (2.3 GB less HBM traffic per training step at 64 molecules, the
// backward kernel 45 us shorter per layer), with its 4-wave (1, 3) or 8-wave (2) shape; 0 = the backward kernel stores
// it and the grouped split-K GEMM consumes it.  3 = as 1, and the layer entry points also pass the sign words of the
// hidden activation (FfnProb::Hs) from the forward to the backward launch, which then does not read the hidden
// activation at all: another 2.4 GB less traffic per step, but 0.03 ms more time (the forward kernel is bound by its
// vector instructions and gathering the flags costs 5 us per launch, the backward kernel gains 4 us).
// Environment: MMST_FFN_WGRAD_RECOMPUTE.
int g_wgrad_recompute = -1;
int ffn_wgrad_recompute() {
  if (g_wgrad_recompute < 0) {
    const char* e = getenv("MMST_FFN_WGRAD_RECOMPUTE");
    const int v = e ? atoi(e) : 1;
    g_wgrad_recompute = (v >= 0 && v <= 3) ? v : 1;
  }
  return g_wgrad_recompute;
}
void ffn_set_wgrad_recompute(int on) { g_wgrad_recompute = on < 0 ? -1 : (on > 3 ? 1 : on); }   // negative: back to the default
bool ffn_bwd_signs() { return ffn_wgrad_recompute() == 3; }
void ffn_set_fuse_min_tiles(int tiles) { g_fuse_min_tiles = tiles; }
int ffn_fuse_min_tiles() { return fuse_min_tiles(); }
