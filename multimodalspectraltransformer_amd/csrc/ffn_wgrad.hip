// Weight gradients of the feed-forward block without a materialised hidden gradient (gfx950).
//
//   dW2[n][j] += sum_m dz[m][n] * h[m][j]            db2[n] += sum_m dz[m][n]
//   dW1[j][k] += sum_m dh[m][j] * x[m][k]            db1[j] += sum_m dh[m][j]
//   dh[m][j]   = (sum_n dz[m][n] * W2[n][j]) * (h[m][j] > 0 ? 1/(1-p) : 0)
//
// dh is recomputed here from dz and the saved hidden activation, so the backward pass never writes or re-reads a
// [rows][2048] gradient: the only large HBM stream of this kernel is ONE read of h.
//
// Decomposition: a workgroup owns 128 hidden columns j and one row split (2000 to 3100 rows; 12 splits at 37248 rows); each of its waves
// owns 16 of those columns for everything: g and dh for them, the [128 n][16 j] slice of dW2 and the [16 j][128 k]
// slice of dW1, both in accumulators for the workgroup's whole life; they leave through LDS-staged fp32 atomics of
// 256 contiguous bytes at the end (one add per gradient element and row split).  Rows are walked 32 at a time: the dz, h
// and x tiles arrive by LDS-DMA into a ring of three (two steps in flight while one is consumed), one barrier per
// step.  Nothing is written back to LDS inside a step: g is computed as D[m][j] (rows on the accumulator registers),
// and the contraction index of the two gradient GEMMs is the step's 32 rows in exactly that register order
// (k = 0..3 -> rows 4q..4q+3, k = 4..7 -> rows 16+4q..16+4q+3 for lane quarter q), so the masked g IS the
// operand fragment of dW1, and the transposing LDS reads of h, dz and x use the same row order.
// Workgroups of one row split sit on one XCD (two with 12 splits; consecutive ids rotate over the 8 XCDs), so the
// dz / x tiles they share are served by that L2.  The splits' sizes differ on purpose (FfnWgradProb::skew).
#include "ffn.h"
#include "gemm_tile.h"
#include <stdlib.h>

#ifdef MMST_WG_LOG
// timing build: one record per workgroup (start, end in 100 MHz ticks; block; hardware ids)
__device__ unsigned int g_wglog2_n;
__device__ unsigned long long g_wglog2[4 * 32768];
extern "C" int mmst_debug_wglog2(unsigned long long* out, unsigned int* n) {
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_wglog2_n), 4) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wglog2), sizeof(unsigned long long) * 4 * 32768) != hipSuccess) return -1;
  return 0;
}
#endif

namespace {

constexpr int DM = 128, FF = 2048, JC = 128, NJ = FF / JC, RS = 16, STEP = 32;
// Tile ring: a step's tiles are requested two steps ahead.  Measured alternatives: four buffers (no faster alone; 96 KiB
// of LDS keep the other stream's 64 KiB launches off the CU: slower in a step); three buffers for h and two for the L2-
// resident dz / x tiles (56 KiB, room for two attention workgroups beside: 145 us alone against 135, 7.56 against 7.50 ms
// per step: what the kernels beside this one lack is registers, not LDS).
// (Four operand tiles in flight instead of two inside a step, the extra two requested once g is computed: 9 spilled
// registers in the loop, 190 us instead of 133.)
constexpr int NBUF = 3;
constexpr int IMG = STEP * 256;   // one 32-row x 128-column bf16 tile, rows of 256 B

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Swizzle key of image row r (XORed into the index of a row's 32-byte granules): the key of the GEMM images.  With it
// the transposing reads below are 2-way bank-conflicted (lane quarters 0/1 share an LDS cycle and read rows r, r + 4,
// which have equal keys; 40 % of the kernel's LDS cycles by SQ_LDS_BANK_CONFLICT).  The conflict-free key
// (r & 3) | ((r >> 2) & 1) << 2 was measured: no conflicts left, LDS cycles down by 38 %, yet the kernel alone takes
// 171 us instead of 134 and a training step the same time, so the LDS array is not what bounds it; kept simple.
__device__ __forceinline__ int rsw(int r) { return fsw(r); }
// byte offset of 16-byte chunk `ch` (8 columns) of row `r` inside a tile image
__device__ __forceinline__ int img_off(int r, int ch) { return r * 256 + (((((ch >> 1) ^ rsw(r)) << 1) | (ch & 1)) << 4); }

// Transposing fragment read over a 32-row tile in the row order above: lane i of each 16 receives column col0 + i,
// lo = rows 4q..4q+3, hi = rows 16+4q..16+4q+3.  Inline assembly for the reason given at frag_t_issue().
__device__ __forceinline__ void frag_rows32_issue(s16x4& lo, s16x4& hi, const unsigned char* img, int col0, int lane) {
  const int krow = 4 * (lane >> 4) + ((lane & 15) >> 2), col = col0 + 4 * (lane & 3);
  const int off = krow * 256 + (((col >> 4) ^ rsw(krow)) << 5) + (col & 15) * 2;   // rsw(krow + 16) == rsw(krow)
  uint32_t a = (uint32_t)(uintptr_t)(LDS_AS const unsigned char*)(img + off);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(hi) : "v"(a));
}
__device__ __forceinline__ void lds_wait2(s16x4& a, s16x4& b, s16x4& c, s16x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

template <int JT>   // 16-column tiles per wave: 1 (8 waves) or 2 (4 waves)
__global__ __launch_bounds__(512 / JT, JT) void ffn_wgrad_kernel(const FfnWgradBatch batch) {
  constexpr int NWAVE = 8 / JT;
  // [NBUF buffers][dz | h | x images] = 72 KiB, reused as the fp32 staging tile of the final atomics (128 x 129 floats)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NBUF * 3 * IMG];
  static_assert(NBUF * 3 * IMG >= 128 * 129 * 4, "staging tile");

#ifdef MMST_WG_LOG
  const unsigned long long wg_t0 = wall_clock64();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < FFN_MAX_GROUPS; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.p[i].tile_begin) gi = i;
  const FfnWgradProb& pr = batch.p[gi];
  const int local = (int)blockIdx.x - pr.tile_begin;
  const int nrs = pr.rs;                                    // row splits of this problem, <= RS
  // consecutive workgroup ids rotate over the 8 XCDs: ids equal mod 8 (mod nrs when there are fewer splits) share a
  // row split, so its dz / x tiles are fetched into one L2 (8 / nrs of them)
  // (12 splits: an XCD holds 3 splits x 8 hidden blocks; any other count: no XCD affinity)
  int rs, jb;
  if (nrs == RS) { rs = (local & 7) | (((local >> 3) & 1) << 3); jb = local >> 4; }
  else if (nrs == 12) { const int x = local & 7, idx = local >> 3; rs = 3 * (x & 3) + idx % 3; jb = 8 * (x >> 2) + idx / 3; }
  else if ((nrs & (nrs - 1)) == 0) { rs = local & (nrs - 1); jb = local / nrs; }
  else { rs = local % nrs; jb = local / nrs; }
  const int j0 = jb * JC;
  const int M = pr.M;
  const int nsteps = (M + STEP - 1) / STEP;
  // The row splits are not equal: every workgroup ends with 128 KB of fp32 atomics, and the rate of those is what
  // the launch's last 26 us were spent on when all workgroups reached them together; with sizes that grow linearly over
  // the splits the early finishers' atomics drain while the others still compute (pr.skew, see the launcher).
  const float skew = pr.skew;
  auto cut = [&](int r) {   // first step of split r: nsteps * F(r / nrs), F(x) = x + skew x (x - 1)
    const float x = (float)r / (float)nrs;
    return r >= nrs ? nsteps : (int)((float)nsteps * (x + skew * x * (x - 1.f)));
  };
  const int s0 = cut(rs), s1 = cut(rs + 1);
  const int q = lane >> 4, li = lane & 15;
  const int jw = wave * 16 * JT;                             // this wave's hidden columns inside the block
  const float dscale = pr.drop_p > 0.f ? 1.0f / (1.0f - pr.drop_p) : 1.0f;

  // W2^T rows of this wave's 16 hidden columns as the B operand of g = dz W2 (lane <-> column j, k = n)
  bf16x8 wt[JT][4];
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      wt[jt][ks] = *reinterpret_cast<const bf16x8*>(pr.W2T + (size_t)(j0 + jw + jt * 16 + li) * DM + ks * 32 + 8 * q);

  f32x4 aw2[JT][8], aw1[JT][8];   // dW2[n = 16 t + 4q + reg][j = li], dW1[j = 4q + reg][k = 16 t + li] per column tile
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int t = 0; t < 8; ++t) { aw2[jt][t] = f32x4{0.f, 0.f, 0.f, 0.f}; aw1[jt][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // bias gradients: db1[j] from the dh values of this lane (column li, 8 rows per step; the 4 lane quarters are
  // added at the end); db2[n] = column sums of dz: the first wave of hidden block t < 8 of each row split multiplies the
  // dz^T fragment of n-tile t, which it loads anyway, with an all-ones operand (one more MFMA per step; every column of
  // the result holds the sums).  (Summed by the first 128 threads of hidden block 0 with one LDS read per row and
  // thread, that block's workgroups were the last to finish: 132 instead of 105 us for the launch.)
  float b1p[JT];
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) b1p[jt] = 0.f;
  // (wave-uniform by construction; read through a scalar so that the MFMA sits behind a branch, not an EXEC mask, which it would ignore)
  const int b2_tile = __builtin_amdgcn_readfirstlane((wave == 0 && j0 < 8 * JC) ? j0 / JC : -1);
  f32x4 b2acc = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  // this lane's share of a step's DMA: JT instructions per image, 4 image rows each, logical chunk sch[]
  int srow[JT], sch[JT];
#pragma unroll
  for (int jj = 0; jj < JT; ++jj) {
    srow[jj] = (wave * JT + jj) * 4 + q;
    sch[jj] = (((li >> 1) ^ rsw(srow[jj])) << 1) | (li & 1);
  }
  // The three base pointers as values: read through `pr` inside the loop they would be fetched from the kernel
  // arguments again in every step (the waits of the loop are memory barriers to the compiler), a scalar load and a wait
  // for it in front of each step's DMA.
  const bf16* const gZ = pr.dZ;
  const bf16* const gH = pr.H + j0;
  const bf16* const gX = pr.X;
  auto stage = [&](int s, int b) {
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
      unsigned char* dst = smem + b * 3 * IMG + (wave * JT + jj) * 4 * 256;
      const size_t m = (size_t)(s * STEP + srow[jj]);
      __builtin_amdgcn_global_load_lds((gptr_t)(gZ + m * DM + sch[jj] * 8), (lptr_t)(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gH + m * FF + sch[jj] * 8), (lptr_t)(dst + IMG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gX + m * DM + sch[jj] * 8), (lptr_t)(dst + 2 * IMG), 16, 0, 0);
    }
  };
  // the ragged last step of the matrix goes through registers: rows past the end read as zero
  auto stage_ragged = [&](int s, int b) {
    unsigned char* base = smem + b * 3 * IMG;
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
      const int m = s * STEP + srow[jj];
      uint4 a = make_uint4(0, 0, 0, 0), h = a, x = a;
      if (m < M) {
        a = *reinterpret_cast<const uint4*>(pr.dZ + (size_t)m * DM + sch[jj] * 8);
        h = *reinterpret_cast<const uint4*>(pr.H + (size_t)m * FF + j0 + sch[jj] * 8);
        x = *reinterpret_cast<const uint4*>(pr.X + (size_t)m * DM + sch[jj] * 8);
      }
      unsigned char* dst = base + srow[jj] * 256 + li * 16;
      *reinterpret_cast<uint4*>(dst) = a;
      *reinterpret_cast<uint4*>(dst + IMG) = h;
      *reinterpret_cast<uint4*>(dst + 2 * IMG) = x;
    }
  };

  // one 32-row step on buffer b (tiles landed and visible to every wave)
  auto step = [&](int b) {
    const unsigned char* zimg = smem + b * 3 * IMG;
    const unsigned char* himg = zimg + IMG;
    const unsigned char* ximg = zimg + 2 * IMG;
    // row-major dz fragments of both 16-row halves (A operand of g), then the transposed reads, four tiles ahead
    bf16x8 zf[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        zf[u][ks] = *reinterpret_cast<const bf16x8*>(zimg + img_off(u * 16 + li, ks * 4 + q));
    // transposed fragments: h for this wave's columns, then 16 operand tiles (dz^T 0..7 for dW2, x^T 0..7 for dW1)
    // through a ring of PF register pairs, PF tiles in flight
    constexpr int PF = JT == 2 ? 2 : 4;
    s16x4 hlo[JT], hhi[JT], tlo[PF], thi[PF];
    auto issue_tile = [&](int i, s16x4& lo, s16x4& hi) { frag_rows32_issue(lo, hi, i < 8 ? zimg : ximg, (i & 7) * 16, lane); };
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) frag_rows32_issue(hlo[jt], hhi[jt], himg, jw + jt * 16, lane);
#pragma unroll
    for (int i = 0; i < PF; ++i) issue_tile(i, tlo[i], thi[i]);
    // ---- g[m][j] = sum_n dz[m][n] W2^T[j][n]: rows m = 16 u + 4q + reg on the registers, column j = li
    f32x4 g[JT][2];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int u = 0; u < 2; ++u) g[jt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          g[jt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[u][ks], wt[jt][ks], g[jt][u], 0, 0, 0);
    bf16x8 hf[JT], dh[JT];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s16x4& lo = tlo[i % PF];
      s16x4& hi = thi[i % PF];
      // tile i has landed once only the younger tiles' reads (two each) are outstanding; LDS reads return in order
      const int younger = 2 * ((PF - 1) < (15 - i) ? (PF - 1) : (15 - i));
      if (i == 0) {   // the h fragments are older than tile 0
        if (younger == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo), "+v"(hi), "+v"(hlo[0]), "+v"(hhi[0]));
        else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(lo), "+v"(hi), "+v"(hlo[0]), "+v"(hhi[0]));
        if constexpr (JT == 2) asm volatile("" : "+v"(hlo[1]), "+v"(hhi[1]));   // readable from here on, like hlo[0]
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) hf[jt] = frag_join(hlo[jt], hhi[jt]);
      } else if (younger == 6) {
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(lo), "+v"(hi));
      } else if (younger == 4) {
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo), "+v"(hi));
      } else if (younger == 2) {
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo), "+v"(hi));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
      }
      const bf16x8 af = frag_join(lo, hi);
      if (i + PF < 16) issue_tile(i + PF, lo, hi);
      if (i == 8) {   // dh = g * mask(h): the A fragment of dW1 (lane <-> column j, k = the step's rows)
        // The mask is read from the bf16 bit patterns two at a time (h >= 0: set where the magnitude bits are not
        // zero), the pairs are converted together, and 1/(1-p) is applied once to the finished sums instead of to
        // every element: 4.5 instead of 8 instructions per element of this block, a fifth of the loop.
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
          const u32x4 hw = __builtin_bit_cast(u32x4, hf[jt]);
          float sel[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float gv = e < 4 ? g[jt][0][e] : g[jt][1][e - 4];
            const bool on = (hw[e >> 1] & ((e & 1) ? 0x7FFF0000u : 0x7FFFu)) != 0u;
            sel[e] = on ? gv : 0.f;
            b1p[jt] += sel[e];
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)sel[e];
          dh[jt] = o;
        }
      }
      if (i < 8 && i == b2_tile) b2acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones, b2acc, 0, 0, 0);       // dz^T 1
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) {
        if (i < 8) aw2[jt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, hf[jt], aw2[jt][i], 0, 0, 0);          // dz^T h
        else aw1[jt][i - 8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dh[jt], af, aw1[jt][i - 8], 0, 0, 0);        // dh^T x
      }
    }
  };

  const int s1f = (s1 * STEP <= M) ? s1 : s1 - 1;   // steps [s0, s1f) are full; at most one ragged step follows
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i)
    if (s0 + i < s1f) stage(s0 + i, i);
  int b = 0;
  for (int s = s0; s < s1f; ++s) {
    // Tiles of this step have landed: own portions once only the 3 JT DMA instructions of each younger step in flight
    // are still outstanding, everyone's after the barrier, which also frees the buffer of step s-1.
    const int younger = s1f - 1 - s;   // steps after this one
    if (younger >= NBUF - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * JT * (NBUF - 2)) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * JT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    static_assert(NBUF == 3 || NBUF == 4, "tail waits");
    __builtin_amdgcn_s_barrier();
    const int bn = b == 0 ? NBUF - 1 : b - 1;         // (b + NBUF - 1) % NBUF: the buffer of step s - 1
    if (s + NBUF - 1 < s1f) stage(s + NBUF - 1, bn);
    step(b);
    b = b == NBUF - 1 ? 0 : b + 1;
  }
  if (s1f < s1) {
    __syncthreads();
    stage_ragged(s1f, 0);
    __syncthreads();
    step(0);
  }

  // ---- both 128 x 128 tiles leave as contiguous fp32 atomics through one staging tile in LDS: every wave writes its
  // slice, then adds 16 whole rows (two 256-byte wave instructions per row)
  float* stg = reinterpret_cast<float*>(smem);
  constexpr int RW = 128 / NWAVE;   // staging rows each wave adds
  if (s0 < s1) {
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) stg[(16 * t + 4 * q + e) * 129 + jw + jt * 16 + li] = aw2[jt][t][e];   // [n][j]
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < RW; ++r) {
      const int n = wave * RW + r;
      atomicAdd(pr.dW2 + (size_t)n * FF + j0 + lane, stg[n * 129 + lane]);
      atomicAdd(pr.dW2 + (size_t)n * FF + j0 + 64 + lane, stg[n * 129 + 64 + lane]);
    }
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) stg[(jw + jt * 16 + 4 * q + e) * 129 + 16 * t + li] = aw1[jt][t][e] * dscale;   // [j][k]
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < RW; ++r) {
      const int j = wave * RW + r;
      atomicAdd(pr.dW1 + (size_t)(j0 + j) * DM + lane, stg[j * 129 + lane]);
      atomicAdd(pr.dW1 + (size_t)(j0 + j) * DM + 64 + lane, stg[j * 129 + 64 + lane]);
    }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      float v = b1p[jt] * dscale;
      v = xor32_sum(xor16_sum(v));
      if (q == 0) atomicAdd(pr.db1 + j0 + jw + jt * 16 + li, v);
    }
    if (b2_tile >= 0 && li == 0) {   // column 0 of the tile: n = 16 tile + 4 q + reg
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(pr.db2 + 16 * b2_tile + 4 * q + e, b2acc[e]);
    }
  }
#ifdef MMST_WG_LOG
  if (tid == 0) {
    const unsigned int ix = atomicAdd(&g_wglog2_n, 1u) & 32767u;
    g_wglog2[4 * ix] = wg_t0;
    g_wglog2[4 * ix + 1] = wall_clock64();
    g_wglog2[4 * ix + 2] = (3ull << 32) | blockIdx.x;
    g_wglog2[4 * ix + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
  }
#endif
}

}  // namespace

int launch_ffn_wgrad(FfnWgradBatch& b, hipStream_t stream) {
  if (b.n < 1 || b.n > FFN_MAX_GROUPS) return MMST_ERR_ARG;
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    FfnWgradProb& p = b.p[i];
    if (p.M < 1 || !p.dZ || !p.H || !p.X || !p.W2T || !p.dW1 || !p.db1 || !p.dW2 || !p.db2) return MMST_ERR_ARG;
    // about 2300 rows (72 steps) per row split: a workgroup's fixed cost (its two 128 x 128 tiles leave as fp32
    // atomics) stays a small part of its life, whatever the problem's size.  (16 splits for every problem made the
    // grouped launch of the five modality encoders 1280 workgroups for the rows of one cross-encoder layer: 175 us
    // against 131, and 7.93 against 7.69 ms per training step; 1200 or 4700 rows per split measure 7.73 / 7.86 ms.)
    int rs = 1;
    while (rs < RS && p.M > rs * 2328) rs *= 2;
    // 16 x 16 workgroups want every CU, but the feed-forward backward kernel that runs beside this one has 257..320
    // row tiles for such a problem and puts two of its workgroups on the CUs beyond 256, which then cannot take a
    // workgroup of this kernel as well: the surplus ones started when that kernel ended (35 of 256, 85 us late, at 37248
    // rows).  Twelve splits = 192 workgroups all start at once, and a quarter of the fp32 atomics is gone
    // (6.30 -> 6.26 ms per training step; 10, 13 and 14 splits measure like 16).  MMST_WGRAD_SPLITS overrides.
    const int tiles = (p.M + 127) / 128;
    if (rs == RS && tiles > 256 && tiles <= 320) rs = 12;
    static const int forced = [] { const char* e = getenv("MMST_WGRAD_SPLITS"); return e ? atoi(e) : 0; }();
    if (forced > 0 && forced <= RS && p.M > 30000) rs = forced;
    p.rs = rs;
    // sizes from 0.86 to 1.14 of the average: 6.255 -> 6.215 ms per training step (8 %: 6.23, 20 %: 6.23, 30 %: 6.28)
    static const float skew = [] { const char* e = getenv("MMST_WGRAD_SKEW"); return e ? 0.01f * (float)atoi(e) : 0.14f; }();
    p.skew = (skew >= 0.f && skew <= 0.5f) ? skew : 0.14f;
    p.tile_begin = total;
    total += NJ * rs;
  }
  // 4 waves of 32 hidden columns (default) or 8 waves of 16 (setting 2 of mmst_ffn_wgrad_config).  Alone on the GPU the
  // 8-wave shape was faster when both were measured (120 against 137 us at 37248 rows; the 4-wave shape has since gone to 127 us); inside a training step its two 168-register waves per
  // SIMD leave no room for the 240-register feed-forward backward of the other stream (8.30 against 7.95 ms per step).
  const int waves = ffn_wgrad_recompute() == 2 ? 8 : 4;
  GemmTap& tap = g_gemm_tap;
  const bool timed = tap.epi == TAP_FFN_WGRAD && tap.count < tap.capacity;
  if (timed) {   // algorithmic work: three GEMMs over the rows; h, dz and x read once, both gradients read and written
    for (int i = 0; i < b.n; ++i) {
      tap.flops += 3.0 * 2.0 * b.p[i].M * (double)DM * FF;
      tap.bytes += 2.0 * b.p[i].M * (double)(FF + 2 * DM) + 2.0 * DM * FF + 2.0 * 8.0 * DM * FF;
    }
    tap.pt = waves;
    (void)hipEventRecord(tap.start[tap.count], stream);
  }
  // the 4-wave shape would fit a CU twice by registers and LDS; 10 KiB of unused dynamic LDS keep it to one workgroup
  // per CU, so that a launch of the activation-gradient stream (64 KiB, one 240-register wave per SIMD) always finds
  // room beside it
  if (waves == 4) hipLaunchKernelGGL(ffn_wgrad_kernel<2>, dim3(total), dim3(256), 10240, stream, b);
  else hipLaunchKernelGGL(ffn_wgrad_kernel<1>, dim3(total), dim3(512), 0, stream, b);
  if (timed) (void)hipEventRecord(tap.stop[tap.count++], stream);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
