// Grouped bf16 MFMA GEMM for gfx950 with fused bias / ReLU / dropout / residual /
// LayerNorm (forward and backward) / split-K atomic epilogues.  See gemm.h.
//
// Tile: 128 (P) x 128 (Q) x 64 (contraction); NW = 8 waves (or 4: MMST_GEMM_WAVES=4).
//   wave layout 4x2 (32x64 per wave; 2x2 of 64x64 with 4 waves) for the plain epilogues,
//   wave layout 8x1 (16 P-rows x all 128 Q-columns per wave; 4x1 of 32 rows with 4 waves) for the LayerNorm
//   epilogues, so a wave owns whole rows and row statistics need no LDS.
// Eight waves at <= 128 registers give four waves per SIMD (two workgroups per CU): the dependent
// barrier -> LDS read -> MFMA steps of one wave run in the shadow of the others.
// MFMA: v_mfma_f32_16x16x32_bf16 with the WEIGHT side as the A operand and the token
// side as the B operand: the accumulator then has the token row on the lane
// (lane & 15) and 4 consecutive output columns in its 4 registers.  Two adjacent
// MFMA column tiles are interleaved (row i of tile t maps to column
// 8*(i>>2) + 4*(t&1) + (i&3)) so that every lane owns 8 CONSECUTIVE output columns:
// all epilogue traffic is 16-byte bf16 vectors.
// LDS images (16 KiB each, XOR-swizzled, conflict-free for the reads below):
//   row-major operand      [128 index][64 c]  read with ds_read_b128
//   contraction-major      [64 c][128 index]  read with ds_read_b64_tr_b16
#include "gemm_tile.h"
#include <cstdio>
#include <cstdlib>

namespace {

#ifdef MMST_WG_LOG
}  // namespace
// timing build: one record per workgroup of the LayerNorm-backward GEMM (start, end in 100 MHz ticks; kind << 32 | block; hardware ids)
__device__ unsigned int g_wglog4_n;
__device__ unsigned long long g_wglog4[4 * 32768];
extern "C" int mmst_debug_wglog4(unsigned long long* out, unsigned int* n) {
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_wglog4_n), 4) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wglog4), sizeof(unsigned long long) * 4 * 32768) != hipSuccess) return -1;
  return 0;
}
namespace {
#endif
template <int EPI, bool PT, bool QT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void gemm_kernel(const GemmBatch batch) {
#ifdef MMST_WG_LOG
  const unsigned long long wg_t0 = wall_clock64();
#endif
  constexpr bool ROWEPI = (EPI == EPI_LN || EPI == EPI_LNBWD);
  constexpr int MT = (ROWEPI ? 2 : 4) * 4 / NW;
  constexpr int NT = ROWEPI ? 8 : 4;
  constexpr int NL = 1024 / (NW * 64);   // 16-byte staging slots per thread and operand tile
  constexpr int SMEM = (EPI == EPI_ATOMIC) ? NW * 32 * 65 * 4 : 2 * BP * BK * 2;  // images, or the atomic staging
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned char* Pimg = smem;
  unsigned char* Qimg = smem + BP * BK * 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: consecutive workgroup ids rotate over the 8 XCDs (each with its own L2), so id % 8
  // names the XCD.  Renumber so that each XCD works on a CONTIGUOUS run of logical tiles: tiles that share an
  // operand (the Q-tiles of one row tile; the P-tiles of one split-K slice) are neighbours in logical order and
  // now fetch that operand into one L2 instead of eight.  Bijective for any grid size.
  int bid;
  {
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
  }
  int g = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_GROUPS; ++i)
    if (i < batch.n && bid >= batch.p[i].tile_begin) g = i;
  const GemmProb& pr = batch.p[g];
  int local = bid - pr.tile_begin;
  const int qt = local % pr.qtiles;
  local /= pr.qtiles;
  const int ptile = local % pr.ptiles;
  const int ks_id = local / pr.ptiles;
  const int p0 = ptile * BP, q0 = qt * BQ;
  const int pw0 = ROWEPI ? wave * 16 * MT : (wave >> 1) * 16 * MT;
  const int qw0 = ROWEPI ? 0 : (wave & 1) * 64;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (pr.K + BK - 1) / BK;
  int kbeg = 0, kend = nk;
  if (EPI == EPI_ATOMIC) {
    kbeg = (int)(((long)nk * ks_id) / pr.ksplit);
    kend = (int)(((long)nk * (ks_id + 1)) / pr.ksplit);
  }
  // bias gradient (column sums of the P-side operand over the contraction): one extra MFMA per
  // P tile against an all-ones Q fragment, on the waves of wave-column 0 of Q-tile 0 only
  const bool do_colsum = (EPI == EPI_ATOMIC) && PT && pr.dbias != nullptr && qt == 0 && (wave & 1) == 0;
  f32x4 bacc[MT];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < MT; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;

  // Main loop: register-staged tiles, TWO tiles in flight (named register sets A/B, statically indexed):
  // the loads of tile k+2 are issued before tile k is computed, so each tile's global latency is covered
  // by two compute phases even when the workgroup is alone on its CU (small grids, short K).
  auto compute_tile = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pf[MT], qf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (PT) pf[mt] = frag_t(Pimg, pw0 + mt * 16 + 4 * (lane & 3), ks, lane);
        else pf[mt] = frag_n<false>(Pimg, pw0 + mt * 16 + (lane & 15), ks, lane);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (QT) qf[nt] = frag_t(Qimg, qw0 + (nt >> 1) * 32 + 8 * (lane & 3) + 4 * (nt & 1), ks, lane);
        else qf[nt] = frag_n<true>(Qimg, qw0 + (nt >> 1) * 32 + 8 * ((lane & 15) >> 2) + 4 * (nt & 1) + (lane & 3), ks, lane);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], pf[mt], acc[mt][nt], 0, 0, 0);
      if (EPI == EPI_ATOMIC && do_colsum) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          bacc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[mt], bacc[mt], 0, 0, 0);
      }
    }
  };

  if (kbeg < kend) {
    // (the problem's fields as values: read through `pr` inside the loop they are fetched from the kernel arguments
    // again after every barrier, seven scalar loads and a wait for them per k step)
    const bf16* const gP = pr.P;
    const bf16* const gQ = pr.Q;
    const int ldp = pr.ldp, ldq = pr.ldq, Pn = pr.Pn, Qn = pr.Qn, Kd = pr.K;
    uint4 rpA[NL], rqA[NL], rpB[NL], rqB[NL];
    g2r<PT, NL>(rpA, gP, ldp, p0, Pn, kbeg * BK, Kd, tid);
    g2r<QT, NL>(rqA, gQ, ldq, q0, Qn, kbeg * BK, Kd, tid);
    if (kbeg + 1 < kend) {
      g2r<PT, NL>(rpB, gP, ldp, p0, Pn, (kbeg + 1) * BK, Kd, tid);
      g2r<QT, NL>(rqB, gQ, ldq, q0, Qn, (kbeg + 1) * BK, Kd, tid);
    }
    for (int kt = kbeg; kt < kend; kt += 2) {
      __syncthreads();
      r2s<PT, false, NL>(Pimg, rpA, tid);
      r2s<QT, true, NL>(Qimg, rqA, tid);
      __syncthreads();
      if (kt + 2 < kend) {
        g2r<PT, NL>(rpA, gP, ldp, p0, Pn, (kt + 2) * BK, Kd, tid);
        g2r<QT, NL>(rqA, gQ, ldq, q0, Qn, (kt + 2) * BK, Kd, tid);
      }
      compute_tile();
      if (kt + 1 < kend) {
        __syncthreads();
        r2s<PT, false, NL>(Pimg, rpB, tid);
        r2s<QT, true, NL>(Qimg, rqB, tid);
        __syncthreads();
        if (kt + 3 < kend) {
          g2r<PT, NL>(rpB, gP, ldp, p0, Pn, (kt + 3) * BK, Kd, tid);
          g2r<QT, NL>(rqB, gQ, ldq, q0, Qn, (kt + 3) * BK, Kd, tid);
        }
        compute_tile();
      }
    }
  }

  const int q = lane >> 4;
  auto crow = [&](int m) -> size_t { return pr.cS > 0 ? (size_t)(m / pr.cS) * pr.cL + pr.coff + (m % pr.cS) : (size_t)m; };
  auto rrow = [&](int m) -> size_t { return pr.rS > 0 ? (size_t)(m / pr.rS) * pr.rL + pr.roff + (m % pr.rS) : (size_t)m; };
  const uint32_t thr = drop_threshold(pr.drop_p);
  const float dscale = pr.drop_p > 0.f ? 1.0f / (1.0f - pr.drop_p) : 1.0f;

  if constexpr (EPI == EPI_BIAS || EPI == EPI_DRELU) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = p0 + pw0 + mt * 16 + (lane & 15);
#pragma unroll
      for (int tp = 0; tp < NT / 2; ++tp) {
        const int n = q0 + qw0 + tp * 32 + 8 * q;
        if (m >= pr.Pn || n >= pr.Qn) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[mt][2 * tp][e]; v[4 + e] = acc[mt][2 * tp + 1][e]; }
        if constexpr (EPI == EPI_BIAS) {
          if (pr.bias) {
            float b[8];
            load8f(b, pr.bias + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e];
          }
          if (pr.flags & GF_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (pr.drop_p > 0.f) drop8(v, pr.seed, (uint32_t)m * (uint32_t)pr.Qn + n, thr, dscale);
          if (pr.R) {
            float r[8];
            load8(r, pr.R + rrow(m) * pr.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
          }
          if (pr.flags & GF_OUT_F32) store8f((float*)pr.C + crow(m) * pr.ldc + n, v);
          else store8((bf16*)pr.C + crow(m) * pr.ldc + n, v);
        } else if constexpr (EPI == EPI_DRELU) {
          float r[8];
          load8(r, pr.R + rrow(m) * pr.ldr + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = r[e] > 0.f ? v[e] * dscale : 0.f;
          store8((bf16*)pr.C + (size_t)m * pr.ldc + n, v);
        }
      }
    }
  }

  if constexpr (EPI == EPI_ATOMIC) {
    // Split-K accumulation into the fp32 gradient.  Float atomics only run at full rate when one
    // wave-instruction covers 256 contiguous bytes, so each wave transposes its 64x64 sub-tile through
    // LDS (passes of 32 rows) and adds whole 64-float row segments.
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 65);
#pragma unroll
    for (int half = 0; half < MT / 2; ++half) {
      __syncthreads();
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const int mt = half * 2 + mh;
        const int rl = mh * 16 + (lane & 15);
#pragma unroll
        for (int tp = 0; tp < NT / 2; ++tp)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            stage[rl * 65 + tp * 32 + 8 * q + e] = acc[mt][2 * tp][e];
            stage[rl * 65 + tp * 32 + 8 * q + 4 + e] = acc[mt][2 * tp + 1][e];
          }
      }
      __syncthreads();
      const int n = q0 + qw0 + lane;
      if (n < pr.Qn) {
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
          const int m = p0 + pw0 + half * 32 + r;
          if (m < pr.Pn) atomicAdd((float*)pr.C + (size_t)m * pr.ldc + n, stage[r * 65 + lane]);
        }
      }
    }
    if (do_colsum && q == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = p0 + pw0 + mt * 16 + (lane & 15);
        if (m < pr.Pn) atomicAdd(pr.dbias + m, bacc[mt][0]);
      }
    }
  }

  if constexpr (EPI == EPI_LN) {
    if constexpr (ROWEPI) epi_ln(acc, pr, p0, pw0, lane);
  }

  if constexpr (EPI == EPI_LNBWD) {
    if constexpr (ROWEPI) epi_lnbwd<NW>(acc, pr, p0, pw0, lane, wave, tid, smem);
  }
#ifdef MMST_WG_LOG
  if (EPI == EPI_LNBWD && threadIdx.x == 0) {
    const unsigned int ix = atomicAdd(&g_wglog4_n, 1u) & 32767u;
    g_wglog4[4 * ix] = wg_t0;
    g_wglog4[4 * ix + 1] = wall_clock64();
    g_wglog4[4 * ix + 2] = (9ull << 32) | blockIdx.x;
    g_wglog4[4 * ix + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
  }
#endif
}


// ------------------------------------------------------------------------------------------------------------
// Split-K weight gradients with an LDS-DMA pipeline.  dW[Pn][Qn] += dY[K][Pn]^T X[K][Qn] streams two long
// [rows][128] operands once and is bound by how many bytes it keeps in flight, so the operand tiles go
// global -> LDS directly (global_load_lds_dwordx4: no staging registers, no LDS-write instructions), NS stages
// deep, one raw barrier per 64-row step and counted vmcnt waits.  The XOR swizzle that makes the transposing
// fragment reads conflict-free is applied to the per-lane SOURCE address (the DMA destination is lane-linear).
// Fragment reads are issued as inline assembly: the compiler would drain every pending DMA (vmcnt(0)) in front of
// a transposing LDS read it can see.  Full 128 x 128 output tiles only (the launcher checks); a ragged last
// 64-row step reads clamped rows and zeroes the surplus rows of both images before the barrier.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NS, int KB>   // NS pipeline stages of KB (64 or 32) contraction rows
__global__ __launch_bounds__(512, NS * KB <= 128 ? 2 : 1) void wgrad_dma_kernel(const GemmBatch batch) {
  constexpr int NW = 8, MT = 2, NT = 4, IMG = BP * KB * 2, STAGE = 2 * IMG, AHEAD = NS - 1, NJ = KB / 32;
  constexpr int SMEM = NS * STAGE > NW * 32 * 65 * 4 ? NS * STAGE : NW * 32 * 65 * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid;
  {
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
  }
  int g = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_GROUPS; ++i)
    if (i < batch.n && bid >= batch.p[i].tile_begin) g = i;
  const GemmProb& pr = batch.p[g];
  int local = bid - pr.tile_begin;
  const int qt = local % pr.qtiles;
  local /= pr.qtiles;
  const int ptile = local % pr.ptiles;
  const int ks_id = local / pr.ptiles;
  const int p0 = ptile * BP, q0 = qt * BQ;
  const int pw0 = (wave >> 1) * 32, qw0 = (wave & 1) * 64;
  const int K = pr.K, nk64 = (K + BK - 1) / BK;   // the K split is planned in 64-row units
  const int kbeg = (int)(((long)nk64 * ks_id) / pr.ksplit) * (BK / KB);
  int kend = (int)(((long)nk64 * (ks_id + 1)) / pr.ksplit) * (BK / KB);
  if (kend * KB >= K + KB) kend = (K + KB - 1) / KB;   // no empty trailing step
  const int nsteps = kend - kbeg;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = pr.dbias != nullptr && qt == 0 && (wave & 1) == 0;
  f32x4 bacc[MT];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < MT; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;

  // this lane's share of a step: DMA instruction j of the wave fills image rows (2 wave + j) * 4 .. + 3
  int rrow[NJ], rcol[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    rrow[j] = (wave * NJ + j) * 4 + (lane >> 4);
    const int c16 = lane & 15;
    rcol[j] = (((c16 >> 1) ^ fsw(rrow[j])) << 4) + ((c16 & 1) << 3);   // logical column (elements) of this LDS slot
  }
  // (base pointers and strides as values: read through `pr` they are fetched from the kernel arguments at every issue)
  const bf16* const gP = pr.P;
  const bf16* const gQ = pr.Q;
  const int ldp = pr.ldp, ldq = pr.ldq;
  auto issue = [&](int step) {
    const int k0 = (kbeg + step) * KB;
    unsigned char* st = smem + (step % NS) * STAGE;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int row = k0 + rrow[j];
      row = row < K ? row : K - 1;
      __builtin_amdgcn_global_load_lds((gptr_t)(gP + (size_t)row * ldp + p0 + rcol[j]),
                                       (lptr_t)(st + (wave * NJ + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gQ + (size_t)row * ldq + q0 + rcol[j]),
                                       (lptr_t)(st + IMG + (wave * NJ + j) * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int s0 = 0; s0 < AHEAD; ++s0)
    if (s0 < nsteps) issue(s0);

  for (int step = 0; step < nsteps; ++step) {
    // own DMA of this step has landed once only the younger steps' (2 NJ instructions each) are outstanding
    if (step + AHEAD - 1 < nsteps) {
      constexpr int YOUNGER = (AHEAD - 1) * 2 * NJ;
      if (YOUNGER == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (YOUNGER == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (YOUNGER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (YOUNGER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (YOUNGER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      static_assert(YOUNGER == 0 || YOUNGER == 2 || YOUNGER == 4 || YOUNGER == 6 || YOUNGER == 8 || YOUNGER == 12, "wait count");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned char* st = smem + (step % NS) * STAGE;
    if ((kbeg + step + 1) * KB > K) {   // ragged last step: rows >= K must not contribute (products, column sums)
      const int k0 = (kbeg + step) * KB;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (k0 + rrow[j] >= K) {
          *reinterpret_cast<uint4*>(st + (wave * NJ + j) * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(st + IMG + (wave * NJ + j) * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_barrier();
    if (step + AHEAD < nsteps) issue(step + AHEAD);
    const unsigned char* Pimg = st;
    const unsigned char* Qimg = st + IMG;
#pragma unroll
    for (int ks = 0; ks < KB / 32; ++ks) {
      s16x4 plo[MT], phi[MT], qlo[NT], qhi[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) frag_t_issue(plo[mt], phi[mt], Pimg, pw0 + mt * 16 + 4 * (lane & 3), ks, lane);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        frag_t_issue(qlo[nt], qhi[nt], Qimg, qw0 + (nt >> 1) * 32 + 8 * (lane & 3) + 4 * (nt & 1), ks, lane);
      lds_wait4(plo[0], phi[0], plo[1], phi[1], qlo[0], qhi[0], qlo[1], qhi[1]);
      lds_wait4(qlo[2], qhi[2], qlo[3], qhi[3], plo[0], phi[0], plo[1], phi[1]);
      bf16x8 pf[MT], qf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) pf[mt] = frag_join(plo[mt], phi[mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) qf[nt] = frag_join(qlo[nt], qhi[nt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], pf[mt], acc[mt][nt], 0, 0, 0);
      if (do_colsum) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          bacc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[mt], bacc[mt], 0, 0, 0);
      }
    }
  }

  // accumulation into the fp32 gradient: as in gemm_kernel (256 contiguous bytes per atomic wave-instruction)
  const int q = lane >> 4;
  float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 65);
  __syncthreads();
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
    const int rl = mh * 16 + (lane & 15);
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        stage[rl * 65 + tp * 32 + 8 * q + e] = acc[mh][2 * tp][e];
        stage[rl * 65 + tp * 32 + 8 * q + 4 + e] = acc[mh][2 * tp + 1][e];
      }
  }
  __syncthreads();
  {
    const int n = q0 + qw0 + lane;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) atomicAdd((float*)pr.C + (size_t)(p0 + pw0 + r) * pr.ldc + n, stage[r * 65 + lane]);
  }
  if (do_colsum && q == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) atomicAdd(pr.dbias + p0 + pw0 + mt * 16 + (lane & 15), bacc[mt][0]);
  }
}

// 0 = off (register-staged gemm_kernel), 2 / 3 / 4 = pipeline stages (environment MMST_WGRAD_DMA, default 2).
// Measured at 37248 rows (feed-forward weight gradients of one layer): alone on the GPU 90 us register-staged,
// 96 / 74 / 72 us with 2 / 3 / 4 stages; inside a training step 8.54 ms register-staged, 8.51 / 8.74 / 8.67 ms:
// the deeper pipelines hold 96-128 KiB of LDS per CU and keep the critical-path kernels of the other stream
// (64 KiB feed-forward backward, 50-70 KiB attention) off the CUs they occupy.
int wgrad_dma_stages() {
  static const int v = [] { const char* e = getenv("MMST_WGRAD_DMA"); return e ? atoi(e) : 2; }();
  return (v >= 2 && v <= 4) || v == 34 || v == 36 ? v : 0;   // 34 / 36: 4 / 6 stages of 32 rows
}
bool wgrad_dma_ok(const GemmBatch& b) {
  for (int i = 0; i < b.n; ++i) {
    const GemmProb& p = b.p[i];
    if ((p.Pn & 127) || (p.Qn & 127) || p.K < 1 || (p.ldp & 7) || (p.ldq & 7)) return false;
    if (((uintptr_t)p.P & 15) || ((uintptr_t)p.Q & 15)) return false;
  }
  return true;
}

// Waves per workgroup by epilogue.  Eight everywhere except the LayerNorm-backward epilogue: its 160 registers
// allow one 8-wave workgroup per CU only, and a 291-tile launch then needs two rounds (45 us against 30 us with
// two resident 4-wave workgroups).  Environment: MMST_GEMM_WAVES, MMST_GEMM_WAVES_WGRAD, MMST_GEMM_WAVES_LNBWD.
int gemm_waves(int epi) {
  static const int v = [] { const char* e = getenv("MMST_GEMM_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
  static const int w = [] { const char* e = getenv("MMST_GEMM_WAVES_WGRAD"); return (e && atoi(e) == 4) ? 4 : 8; }();
  static const int l = [] { const char* e = getenv("MMST_GEMM_WAVES_LNBWD"); return (e && atoi(e) == 8) ? 8 : 4; }();
  return epi == EPI_ATOMIC ? w : epi == EPI_LNBWD ? l : v;
}

// Experiment switch: MMST_WGRAD_LDS_PAD=<bytes> pads EVERY split-K launch (see GemmBatch::lds_pad).
int wgrad_lds_pad() {
  static const int v = [] { const char* e = getenv("MMST_WGRAD_LDS_PAD"); return e ? atoi(e) : 0; }();
  return v;
}

template <int EPI, int NW>
int dispatch_nw(const GemmBatch& b, bool pt, bool qt, int grid, hipStream_t s) {
  const int dyn = EPI == EPI_ATOMIC ? (b.lds_pad > wgrad_lds_pad() ? b.lds_pad : wgrad_lds_pad()) : 0;
  if (!pt && !qt) MMST_LAUNCH((gemm_kernel<EPI, false, false, NW>), dim3(grid), dim3(NW * 64), dyn, s, b);
  else if (!pt && qt) MMST_LAUNCH((gemm_kernel<EPI, false, true, NW>), dim3(grid), dim3(NW * 64), dyn, s, b);
  else if (pt && qt) MMST_LAUNCH((gemm_kernel<EPI, true, true, NW>), dim3(grid), dim3(NW * 64), dyn, s, b);
  else return MMST_ERR_ARG;
  return MMST_OK;
}
template <int EPI>
int dispatch(const GemmBatch& b, bool pt, bool qt, int grid, hipStream_t s) {
  if (EPI == EPI_ATOMIC && pt && qt && wgrad_dma_stages() && wgrad_dma_ok(b)) {
    const int dyn = b.lds_pad > wgrad_lds_pad() ? b.lds_pad : wgrad_lds_pad();
    switch (wgrad_dma_stages()) {
      case 2: hipLaunchKernelGGL((wgrad_dma_kernel<2, 64>), dim3(grid), dim3(512), dyn, s, b); break;
      case 3: hipLaunchKernelGGL((wgrad_dma_kernel<3, 64>), dim3(grid), dim3(512), dyn, s, b); break;
      case 4: hipLaunchKernelGGL((wgrad_dma_kernel<4, 64>), dim3(grid), dim3(512), dyn, s, b); break;
      case 34: hipLaunchKernelGGL((wgrad_dma_kernel<4, 32>), dim3(grid), dim3(512), dyn, s, b); break;
      default: hipLaunchKernelGGL((wgrad_dma_kernel<6, 32>), dim3(grid), dim3(512), dyn, s, b); break;
    }
    return MMST_OK;
  }
  return gemm_waves(EPI) == 4 ? dispatch_nw<EPI, 4>(b, pt, qt, grid, s) : dispatch_nw<EPI, 8>(b, pt, qt, grid, s);
}

}  // namespace

int gemm_waves_for(int epi) { return gemm_waves(epi); }

// name of the kernel that serves (epi, pt, qt) launches with full tiles, as a profiler prints it
void gemm_kernel_name(int epi, bool pt, bool qt, char* out, int cap) {
  if (epi == TAP_FFN_WGRAD)
    snprintf(out, cap, "ffn_wgrad_kernel<%d>", g_gemm_tap.pt == 8 ? 1 : 2);
  else if (epi == EPI_ATOMIC && pt && qt && wgrad_dma_stages())
    snprintf(out, cap, "wgrad_dma_kernel<%d, %d>", wgrad_dma_stages() > 30 ? wgrad_dma_stages() - 30 : wgrad_dma_stages(),
             wgrad_dma_stages() > 30 ? 32 : 64);
  else
    snprintf(out, cap, "gemm_kernel<%d, %s, %s, %d>", epi, pt ? "true" : "false", qt ? "true" : "false", gemm_waves(epi));
}

thread_local hipEvent_t mmst_armed_stop_event = nullptr;

GemmTap g_gemm_tap = {-1, 0, 0, 0, 0, nullptr, nullptr, 0.0, 0.0};

int launch_gemm(GemmBatch& b, int epi, bool pt, bool qt, hipStream_t stream) {
  if (b.n < 1 || b.n > GEMM_MAX_GROUPS) return MMST_ERR_ARG;
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    GemmProb& p = b.p[i];
    if (p.Pn <= 0 || p.Qn <= 0 || p.K < 0) return MMST_ERR_ARG;
    // 16-byte vector loads: the contiguous extent of each operand must be a multiple of 8 elements
    if ((p.Qn & 7) || (p.ldc & 3)) return MMST_ERR_ARG;
    if (p.K > 0) {
      if ((p.ldp & 7) || (p.ldq & 7)) return MMST_ERR_ARG;
      if ((!pt || !qt) && (p.K & 7)) return MMST_ERR_ARG;
      if (pt && (p.Pn & 7)) return MMST_ERR_ARG;
    }
    if (epi == EPI_LN || epi == EPI_LNBWD) {
      if (p.Qn != 128) return MMST_ERR_ARG;
    }
    if (epi != EPI_ATOMIC || p.ksplit < 1) p.ksplit = 1;
    int nk = (p.K + BK - 1) / BK;
    if (p.ksplit > nk && nk > 0) p.ksplit = nk;
    p.ptiles = (p.Pn + BP - 1) / BP;
    p.qtiles = (p.Qn + BQ - 1) / BQ;
    p.tile_begin = total;
    total += p.ptiles * p.qtiles * p.ksplit;
  }
  GemmTap& tap = g_gemm_tap;
  // the tap times ONE kernel: with the LDS-DMA weight-gradient kernel on, only the launches that take it
  const bool dma = epi == EPI_ATOMIC && pt && qt && wgrad_dma_stages() != 0;
  const bool timed = tap.epi == epi && tap.pt == (int)pt && tap.qt == (int)qt && tap.count < tap.capacity &&
                     (!dma || wgrad_dma_ok(b));
  if (timed) {
    for (int i = 0; i < b.n; ++i) {
      const GemmProb& p = b.p[i];
      const double out_bytes = (epi == EPI_ATOMIC) ? 8.0 : ((p.flags & GF_OUT_F32) ? 4.0 : 2.0);
      tap.flops += 2.0 * p.Pn * (double)p.Qn * p.K;
      tap.bytes += 2.0 * p.K * ((double)p.Pn + p.Qn) + out_bytes * p.Pn * (double)p.Qn +
                   ((epi == EPI_DRELU || epi == EPI_LN) ? 2.0 * p.Pn * (double)p.Qn : 0.0);
    }
    (void)hipEventRecord(tap.start[tap.count], stream);
  }
  int rc;
  switch (epi) {
    case EPI_BIAS: rc = dispatch<EPI_BIAS>(b, pt, qt, total, stream); break;
    case EPI_DRELU: rc = dispatch<EPI_DRELU>(b, pt, qt, total, stream); break;
    case EPI_LN: rc = dispatch<EPI_LN>(b, pt, qt, total, stream); break;
    case EPI_LNBWD: rc = dispatch<EPI_LNBWD>(b, pt, qt, total, stream); break;
    case EPI_ATOMIC: rc = dispatch<EPI_ATOMIC>(b, pt, qt, total, stream); break;
    default: return MMST_ERR_ARG;
  }
  if (timed) (void)hipEventRecord(tap.stop[tap.count++], stream);
  if (rc != MMST_OK) return rc;
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
