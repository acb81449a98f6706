// Weight gradients of the feed-forward block without a materialised hidden gradient (gfx950).
//
//   dW2[n][j] += sum_m dz[m][n] * h[m][j]            db2[n] += sum_m dz[m][n]
//   dW1[j][k] += sum_m dh[m][j] * x[m][k]            db1[j] += sum_m dh[m][j]
//   dh[m][j]   = (sum_n dz[m][n] * W2[n][j]) * (h[m][j] > 0 ? 1/(1-p) : 0)
//
// dh is recomputed here from dz and the saved hidden activation, so the backward pass never writes or re-reads a
// [rows][2048] gradient: the only large HBM stream of this kernel is ONE read of h.
//
// Decomposition: a workgroup owns 128 hidden columns j (16 chunks) and 1/16 of the rows (16 row splits), and
// keeps both 128x128 fp32 gradient tiles in accumulators for its whole life; they leave through LDS-staged,
// 256-byte-contiguous fp32 atomics at the end (16 adds per gradient element per launch).  Rows are walked 32 at a
// time: the dz, h and x tiles arrive by LDS-DMA into a double buffer (the tile of the next step is in flight
// while this one is consumed); dh goes registers -> LDS -> transposing reads.  Workgroups of one row split sit on
// one XCD (consecutive ids rotate over the 8 XCDs), so the dz / x tiles they share are served by that L2.
#include "ffn.h"
#include "gemm_tile.h"

namespace {

constexpr int DM = 128, FF = 2048, JC = 128, NJ = FF / JC, RS = 16, STEP = 32;
constexpr int NBUF = 3;
constexpr int IMG = STEP * 256;   // one 32-row x 128-column bf16 tile, rows of 256 B

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// byte offset of 16-byte chunk `ch` (8 columns) of row `r` inside a tile image: the swizzle of frag_t()
__device__ __forceinline__ int img_off(int r, int ch) { return r * 256 + (((((ch >> 1) ^ fsw(r)) << 1) | (ch & 1)) << 4); }

__global__ __launch_bounds__(256, 2) void ffn_wgrad_kernel(const FfnWgradBatch batch) {
  // [3 buffers][dz | h | x images] then the dh image (reused as the atomic staging area at the end): 80 KiB.
  // The launch adds 2 KiB of unused dynamic LDS so that at most ONE workgroup of this kernel fits a CU, while the
  // register budget stays that of two waves per SIMD: the launches of the activation-gradient stream this kernel
  // runs beside always find LDS and registers on every CU.
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NBUF * 3 * IMG + IMG];
  unsigned char* dhimg = smem + NBUF * 3 * IMG;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = NJ * RS;
  const FfnWgradProb& pr = batch.p[blockIdx.x / per];
  const int local = blockIdx.x % per;
  const int rs = (local & 7) | (((local >> 3) & 1) << 3);   // ids equal mod 8 share a row split (one XCD)
  const int j0 = (local >> 4) * JC;
  const int M = pr.M;
  const int nsteps = (M + STEP - 1) / STEP;
  const int s0 = (int)((long)nsteps * rs / RS), s1 = (int)((long)nsteps * (rs + 1) / RS);
  const int q = lane >> 4;
  const int pw0 = (wave >> 1) * 64, qw0 = (wave & 1) * 64;
  const float dscale = pr.drop_p > 0.f ? 1.0f / (1.0f - pr.drop_p) : 1.0f;

  // this wave's 32 rows of W2^T (hidden j0 + wave*32 ...), as MFMA A fragments with interleaved rows so that a
  // lane ends up with 8 CONSECUTIVE hidden columns of dh
  bf16x8 wt[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = lane & 15;
    const int j = j0 + wave * 32 + 8 * (i >> 2) + 4 * t + (i & 3);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      wt[t][ks] = *reinterpret_cast<const bf16x8*>(pr.W2T + (size_t)j * DM + ks * 32 + 8 * q);
  }

  f32x4 aw2[4][4], aw1[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { aw2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; aw1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // bias gradients: db1 from the dh values this lane produces (8 hidden columns, summed over rows across the
  // 16 lanes of a quarter-wave at the end); db2 = column sums of dz, by the first hidden chunk of each row split
  float b1p[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) b1p[e] = 0.f;
  float b2p = 0.f;
  const bool do_b2 = j0 == 0 && tid < DM;

  // per-lane source coordinates of this wave's 2 DMA instructions per image: image row r, logical chunk ch
  int srow[2], sch[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int r = wave * 8 + jj * 4 + (lane >> 4), s = lane & 15;
    srow[jj] = r;
    sch[jj] = (((s >> 1) ^ fsw(r)) << 1) | (s & 1);
  }
  // the three 32-row tiles of (full) step s -> buffer b, by LDS-DMA
  auto stage = [&](int s, int b) {
    unsigned char* base = smem + b * 3 * IMG;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const size_t m = (size_t)(s * STEP + srow[jj]);
      unsigned char* dst = base + (wave * 8 + jj * 4) * 256;
      __builtin_amdgcn_global_load_lds((gptr_t)(pr.dZ + m * DM + sch[jj] * 8), (lptr_t)(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(pr.H + m * FF + j0 + sch[jj] * 8), (lptr_t)(dst + IMG), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(pr.X + m * DM + sch[jj] * 8), (lptr_t)(dst + 2 * IMG), 16, 0, 0);
    }
  };
  // the ragged last step of the matrix goes through registers: rows past the end read as zero
  auto stage_ragged = [&](int s, int b) {
    unsigned char* base = smem + b * 3 * IMG;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int m = s * STEP + srow[jj];
      uint4 a = make_uint4(0, 0, 0, 0), h = a, x = a;
      if (m < M) {
        a = *reinterpret_cast<const uint4*>(pr.dZ + (size_t)m * DM + sch[jj] * 8);
        h = *reinterpret_cast<const uint4*>(pr.H + (size_t)m * FF + j0 + sch[jj] * 8);
        x = *reinterpret_cast<const uint4*>(pr.X + (size_t)m * DM + sch[jj] * 8);
      }
      unsigned char* dst = base + srow[jj] * 256 + (lane & 15) * 16;
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
    // ---- g^T[j][m] = sum_n W2^T[j][n] dz[m][n] for this wave's 32 hidden columns and the step's 32 rows
    f32x4 gacc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) gacc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 zf[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
        zf[u] = *reinterpret_cast<const bf16x8*>(zimg + img_off(u * 16 + (lane & 15), ks * 4 + q));
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          gacc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[t][ks], zf[u], gacc[t][u], 0, 0, 0);
    }
    // ---- dh = g * mask(h) -> bf16 -> dh image (row m, hidden columns wave*32 + 8q .. +7)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int off = img_off(u * 16 + (lane & 15), wave * 4 + q);
      const bf16x8 h = *reinterpret_cast<const bf16x8*>(himg + off);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gv = e < 4 ? gacc[0][u][e] : gacc[1][u][e - 4];
        o[e] = (bf16)((float)h[e] > 0.f ? gv * dscale : 0.f);
        b1p[e] += (float)o[e];
      }
      *reinterpret_cast<bf16x8*>(dhimg + off) = o;
    }
    if (do_b2) {
#pragma unroll 8
      for (int r = 0; r < STEP; ++r)
        b2p += (float)*reinterpret_cast<const bf16*>(zimg + img_off(r, tid >> 3) + (tid & 7) * 2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- dW2 += dz^T h ;  dW1 += dh^T x   (contraction over the step's 32 rows: one k-step each)
    {
      s16x4 plo[4], phi[4], qlo[4], qhi[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) frag_t_issue(plo[mt], phi[mt], zimg, pw0 + mt * 16 + 4 * (lane & 3), 0, lane);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        frag_t_issue(qlo[nt], qhi[nt], himg, qw0 + (nt >> 1) * 32 + 8 * (lane & 3) + 4 * (nt & 1), 0, lane);
      lds_wait4(plo[0], phi[0], plo[1], phi[1], plo[2], phi[2], plo[3], phi[3]);
      lds_wait4(qlo[0], qhi[0], qlo[1], qhi[1], qlo[2], qhi[2], qlo[3], qhi[3]);
      bf16x8 pf[4], qf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { pf[i] = frag_join(plo[i], phi[i]); qf[i] = frag_join(qlo[i], qhi[i]); }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          aw2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], pf[mt], aw2[mt][nt], 0, 0, 0);
    }
    {
      s16x4 plo[4], phi[4], qlo[4], qhi[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) frag_t_issue(plo[mt], phi[mt], dhimg, pw0 + mt * 16 + 4 * (lane & 3), 0, lane);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        frag_t_issue(qlo[nt], qhi[nt], ximg, qw0 + (nt >> 1) * 32 + 8 * (lane & 3) + 4 * (nt & 1), 0, lane);
      lds_wait4(plo[0], phi[0], plo[1], phi[1], plo[2], phi[2], plo[3], phi[3]);
      lds_wait4(qlo[0], qhi[0], qlo[1], qhi[1], qlo[2], qhi[2], qlo[3], qhi[3]);
      bf16x8 pf[4], qf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { pf[i] = frag_join(plo[i], phi[i]); qf[i] = frag_join(qlo[i], qhi[i]); }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          aw1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[nt], pf[mt], aw1[mt][nt], 0, 0, 0);
    }
  };

  const int s1f = (s1 * STEP <= M) ? s1 : s1 - 1;   // steps [s0, s1f) are full; at most one ragged step follows
  if (s0 < s1f) stage(s0, 0);
  if (s0 + 1 < s1f) stage(s0 + 1, 1);
  int b = 0;
  for (int s = s0; s < s1f; ++s) {
    // Tiles of this step have landed: own portions once only the 6 DMA instructions of the next step are still
    // outstanding, everyone's after the barrier, which also frees the buffer of step s-1 and the dh image.
    if (s + 1 < s1f) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int b2 = b == 0 ? 2 : b - 1;                // (b + 2) % 3
    if (s + 2 < s1f) stage(s + 2, b2);
    step(b);
    b = b == NBUF - 1 ? 0 : b + 1;
  }
  if (s1f < s1) {
    __syncthreads();
    stage_ragged(s1f, 0);
    __syncthreads();
    step(0);
  }

  // ---- both tiles leave as contiguous fp32 atomics: each wave transposes its 64x64 sub-tile through LDS
  // (two passes of 32 rows) and adds whole 64-float row segments (256 bytes per wave instruction)
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 65);
  auto flush = [&](f32x4 (&acc)[4][4], float* C, int ldc, int row_base, int col_base) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const int mt = half * 2 + mh;
        const int rl = mh * 16 + (lane & 15);
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            stg[rl * 65 + tp * 32 + 8 * q + e] = acc[mt][2 * tp][e];
            stg[rl * 65 + tp * 32 + 8 * q + 4 + e] = acc[mt][2 * tp + 1][e];
          }
      }
      __syncthreads();
#pragma unroll 8
      for (int r = 0; r < 32; ++r)
        atomicAdd(C + (size_t)(row_base + pw0 + half * 32 + r) * ldc + col_base + qw0 + lane, stg[r * 65 + lane]);
    }
  };
  if (s0 < s1) {
    flush(aw2, pr.dW2, FF, 0, j0);      // dW2[n][j0 + j]
    flush(aw1, pr.dW1, DM, j0, 0);      // dW1[j0 + j][k]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = b1p[e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
      if ((lane & 15) == 0) atomicAdd(pr.db1 + j0 + wave * 32 + 8 * q + e, v);
    }
    if (do_b2) atomicAdd(pr.db2 + tid, b2p);
  }
}

}  // namespace

int launch_ffn_wgrad(FfnWgradBatch& b, hipStream_t stream) {
  if (b.n < 1 || b.n > FFN_MAX_GROUPS) return MMST_ERR_ARG;
  for (int i = 0; i < b.n; ++i) {
    const FfnWgradProb& p = b.p[i];
    if (p.M < 1 || !p.dZ || !p.H || !p.X || !p.W2T || !p.dW1 || !p.db1 || !p.dW2 || !p.db2) return MMST_ERR_ARG;
  }
  hipLaunchKernelGGL(ffn_wgrad_kernel, dim3(b.n * NJ * RS), dim3(256), 2048, stream, b);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
