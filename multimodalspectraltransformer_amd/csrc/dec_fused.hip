// Fused decoder-layer kernels for short target sequences: one workgroup per molecule.  See dec_fused.h.
//
// Workgroup = 8 waves, 64 token rows (T <= 64, rows beyond T are zero on input and never stored).  Every GEMM is
// OUT[64 tokens][N] = X[64][K] W[N][K]^T with v_mfma_f32_16x16x32_bf16, the weight side as the A operand: a wave owns
// N/8 output columns for all 64 tokens, so every weight element is needed by exactly one wave and goes from memory
// (L2) straight into that wave's registers in fragment layout (lane (r, g) reads 16 bytes of weight row r at
// k = 32 ks + 8 g) without an LDS image; the token side comes from a row-major LDS image whose rows are 4 banks apart
// (16-byte fragment reads without conflicts).  The accumulator has the token on the lane (lane & 15) and 4 consecutive
// output columns in its registers.  Row-wise work (residual, LayerNorm) runs from an fp32 LDS copy of the GEMM output
// with 8 lanes per row.  The feed-forward block walks the hidden dimension in chunks of 256 columns: the chunk's
// activation goes through an LDS image (double buffered, one barrier per chunk) to become the token side of the second
// GEMM and leaves for memory from that image with full-line stores; the weights of chunk c + 1 are requested before
// chunk c is computed.
#include "dec_fused.h"
#include <cstdlib>

namespace {

constexpr int D = MMST_D, FF = MMST_FF;
constexpr int TM = 64;                          // token rows of a workgroup
constexpr int NTHR = 512;
constexpr int XS = 272;                         // [64][128] bf16 image: bytes per row
constexpr int HS = 528;                         // [64][256] bf16 image
constexpr int QS = 784;                         // [64][384] bf16 image
constexpr int VS = 144;                         // transposed V image [128][64] bf16
constexpr int ZS = 132;                         // fp32 scratch [64][128]: floats per row
constexpr int HC = 256;                         // hidden columns per feed-forward chunk
constexpr int X_BYTES = TM * XS;
constexpr int U_BYTES = TM * QS + D * VS;       // QKV image + V^T image; also two hidden-chunk images / the fp32 scratch
static_assert(U_BYTES >= 2 * TM * HS && U_BYTES >= TM * ZS * 4, "union region");
constexpr int SMEM = 2 * X_BYTES + U_BYTES;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int MID_VEC = 3 * D + 2 * HC;
static_assert(MID_VEC <= 2 * 512, "two staged values per thread");    // floats staged by the middle kernel
constexpr int HEAD_VEC = 10 * D;           // ... by the head kernel

__device__ __forceinline__ bf16x8 lds8(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
  return z;
}
__device__ __forceinline__ int acc_row(int r16, int h) { return (r16 & 3) + 8 * (r16 >> 2) + 4 * h; }

// rows m0 .. m0 + T - 1 of a [.][128] bf16 matrix -> image (rows >= T zero)
__device__ __forceinline__ void load_rows(unsigned char* img, const bf16* src, int T, int tid) {
  for (int i = tid; i < TM * 16; i += NTHR) {
    const int t = i >> 4, c = i & 15;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t < T) v = *reinterpret_cast<const uint4*>(src + (size_t)t * D + c * 8);
    *reinterpret_cast<uint4*>(img + t * XS + c * 16) = v;
  }
}
// the same in two steps, so that other loads can be requested between the request and the LDS stores that wait for it
__device__ __forceinline__ void rows_issue(uint4 (&rv)[2], const bf16* src, int T, int tid) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {   // unconditional loads (a load under a branch is waited for where the branch ends)
    const int i = tid + k * NTHR, t = i >> 4, c = i & 15;
    rv[k] = *reinterpret_cast<const uint4*>(src + (size_t)(t < T ? t : T - 1) * D + c * 8);
  }
}
__device__ __forceinline__ void rows_commit(unsigned char* img, const uint4 (&rv)[2], int T, int tid) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + k * NTHR, t = i >> 4, c = i & 15;
    *reinterpret_cast<uint4*>(img + t * XS + c * 16) = t < T ? rv[k] : make_uint4(0, 0, 0, 0);
  }
}
// image rows -> memory, 16 bytes per thread and piece, whole rows contiguous
template <int COLS, int STRIDE>
__device__ __forceinline__ void store_rows(bf16* dst, int ld, const unsigned char* img, int T, int tid) {
  constexpr int PPR = COLS / 8;
  for (int i = tid; i < TM * PPR; i += NTHR) {
    const int t = i / PPR, c = i % PPR;
    if (t < T) *reinterpret_cast<uint4*>(dst + (size_t)t * ld + c * 8) = *reinterpret_cast<const uint4*>(img + t * STRIDE + c * 16);
  }
}

// Weight fragments of NT row tiles x KS contraction steps from a fragment-major image (mmst_fragment_bf16): tile
// (tile0 + nt, ks0 + ks) of a matrix with kt contraction tiles per row tile; one wave instruction reads one contiguous KiB
template <int NT, int KS>
__device__ __forceinline__ void load_w(bf16x8 (&w)[NT][KS], const bf16* F, int tile0, int kt, int ks0, int lane) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      w[nt][ks] = *reinterpret_cast<const bf16x8*>(F + ((size_t)((tile0 + nt) * kt + ks0 + ks) * 64 + lane) * 8);
}
// acc[nt][mt] += W-tile nt x token tile mt, tokens from an image with row stride xs
template <int NT, int KS>
__device__ __forceinline__ void mma_tok(f32x4 (&acc)[NT][4], const bf16x8 (&w)[NT][KS], const unsigned char* X, int xs, int lane) {
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 xf[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xf[mt] = lds8(X + (mt * 16 + r) * xs + (ks * 32 + 8 * g) * 2);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt][ks], xf[mt], acc[nt][mt], 0, 0, 0);
  }
}
template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NT][4]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ uint2 pack4(const float (&v)[4]) {
  const f32x2 a = {v[0], v[1]}, c = {v[2], v[3]};
  uint2 o;
  o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2));
  o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(c, bf16x2));
  return o;
}

// Output of a 128-column projection (the wave's 16 columns from n0) with bias and the element dropout of the GEMM row
// epilogues (pair hash of element index m * 128 + n) -> fp32 scratch
__device__ __forceinline__ void proj_to_scratch(const f32x4 (&acc)[1][4], float* Zf, const float* bias, int n0, int m0,
                                                float dp, uint32_t seed, uint32_t thr, float dscale, int lane) {
  const int r = lane & 15, g = lane >> 4, n = n0 + 4 * g;
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int t = mt * 16 + r;
    f32x4 v = (acc[0][mt] + bv) * dscale;
    {   // no branch on the dropout probability: without dropout the threshold is 0 and every comparison keeps
      const uint32_t idx = (uint32_t)(m0 + t) * 128u + (uint32_t)n;
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const uint32_t h = drop_bits(seed, (idx + e) >> 1);
        v[e] = ((h & 0xFFFFu) >= thr) ? v[e] : 0.f;
        v[e + 1] = ((h >> 16) >= thr) ? v[e + 1] : 0.f;
      }
    }
    *reinterpret_cast<f32x4*>(Zf + t * ZS + n) = v;
  }
}

// Workgroup barrier that waits for this wave's LDS traffic only: weight fragments requested for a later phase stay in
// flight across it (__syncthreads() would wait for every outstanding load as well)
// keeps the compiler from moving memory operations across this point (no instruction)
__device__ __forceinline__ void order_point() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// y = LayerNorm(z) for the 16 columns c0.. of row t held by this lane (8 consecutive lanes share a row).  Writes z
// (bf16) and (mean, rstd) to memory for the backward pass if `store`, y to memory if `store` and to an LDS image.
__device__ __forceinline__ void ln_core(const float (&z)[16], const float* gamma, const float* beta, bf16* Zg, float* stats,
                                        unsigned char* out_img, bf16* out_g, int t, int c0, bool store, bool first) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += z[e];
  const float mean = row8_sum(s) * (1.0f / 128.0f);
  float vs = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) { const float d = z[e] - mean; vs += d * d; }
  const float rstd = rsqrtf(row8_sum(vs) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float gam[8], bet[8];
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0 + 8 * j), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 8 * j + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c0 + 8 * j), b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 8 * j + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gam[e] = g0[e]; gam[4 + e] = g1[e]; bet[e] = b0[e]; bet[4 + e] = b1[e]; }
    bf16x8 zo, yo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      zo[e] = (bf16)z[8 * j + e];
      yo[e] = (bf16)((z[8 * j + e] - mean) * rstd * gam[e] + bet[e]);
    }
    *reinterpret_cast<bf16x8*>(out_img + t * XS + (c0 + 8 * j) * 2) = yo;
    if (store) {
      *reinterpret_cast<bf16x8*>(Zg + (size_t)t * D + c0 + 8 * j) = zo;
      *reinterpret_cast<bf16x8*>(out_g + (size_t)t * D + c0 + 8 * j) = yo;
    }
  }
  if (store && first) { stats[2 * t] = mean; stats[2 * t + 1] = rstd; }
}

// z = scratch + residual; y = LayerNorm(z).  8 lanes per row, 16 columns each.
template <bool RES_IMG>   // residual rows from an LDS image / in registers (res: the lane's 16 columns)
__device__ __forceinline__ void ln_rows(const float* Zf, const unsigned char* res_img, const bf16x8 (&res)[2], const float* gamma,
                                        const float* beta, bf16* Zg, float* stats, unsigned char* out_img, bf16* out_g, int T,
                                        int tid, bool store = true) {
  const int t = tid >> 3, c0 = (tid & 7) * 16;
  const bool ok = t < T;
  float z[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(Zf + t * ZS + c0 + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) z[4 * j + e] = v[e];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bf16x8 rv = res[j];
    if constexpr (RES_IMG) rv = lds8(res_img + t * XS + (c0 + 8 * j) * 2);
#pragma unroll
    for (int e = 0; e < 8; ++e) z[8 * j + e] += (float)rv[e];
  }
  ln_core(z, gamma, beta, Zg, stats, out_img, out_g, t, c0, ok && store, (tid & 7) == 0);
}
// the lane's 16 columns of row t of a [.][128] bf16 matrix (rows past the end: the last row, never stored)
__device__ __forceinline__ void load_res(bf16x8 (&res)[2], const bf16* rows, int T, int tid) {
  const int t = tid >> 3, c0 = (tid & 7) * 16;
#pragma unroll
  for (int j = 0; j < 2; ++j) res[j] = *reinterpret_cast<const bf16x8*>(rows + (size_t)(t < T ? t : T - 1) * D + c0 + 8 * j);
}

// weights of feed-forward chunk c for this wave: 32 rows of linear1, the chunk's 256 columns of 16 rows of linear2
__device__ __forceinline__ void ffn_load_w(bf16x8 (&w1)[2][4], bf16x8 (&w2)[1][8], const bf16* F1, const bf16* F2, int c,
                                           int wave, int lane) {
  load_w<2, 4>(w1, F1, c * (HC / 16) + 2 * wave, D / 32, 0, lane);
  load_w<1, 8>(w2, F2, wave, FF / 32, c * (HC / 32), lane);
}
// One chunk of the feed-forward block: hidden columns c * 256 .. + 255 of all 64 rows (first GEMM, bias, ReLU, dropout)
// into the chunk image Hc and out to memory, then their contribution to the block's output (second GEMM).
__device__ __forceinline__ void ffn_chunk(f32x4 (&acc2)[1][4], const bf16x8 (&w1)[2][4], const bf16x8 (&w2)[1][8],
                                          const unsigned char* X2, unsigned char* Hc, bf16* Hg, const float* b1c, int c,
                                          const uint32_t (&rseed)[4], float dp, uint32_t thr, float dscale, int T, int tid) {
  const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  f32x4 acc1[2][4];
  zero_acc<2>(acc1);
  mma_tok<2, 4>(acc1, w1, X2, XS, lane);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int lc = 32 * wave + nt * 16 + 4 * g, col = c * HC + lc;
    // relu(a + b) * s == relu(a * s + b * s): what is stored carries the dropout scale
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b1c + lc) * dscale;   // b1c: the chunk's 256 bias values
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(acc1[nt][mt][e], dscale, bv[e]), 0.f);
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const uint32_t hb = attn_pair_bits(rseed[mt], (uint32_t)((col + e) >> 1) * ATTN_PAIR_MUL);
        v[e] = ((hb & 0xFFFFu) >= thr) ? v[e] : 0.f;
        v[e + 1] = ((hb >> 16) >= thr) ? v[e + 1] : 0.f;
      }
      *reinterpret_cast<uint2*>(Hc + (mt * 16 + r) * HS + lc * 2) = pack4(v);
    }
  }
  lds_barrier();
  for (int i = tid; i < TM * 32; i += NTHR) {
    const int t = i >> 5, cc = i & 31;
    if (t < T) *reinterpret_cast<uint4*>(Hg + (size_t)t * FF + c * HC + cc * 8) = *reinterpret_cast<const uint4*>(Hc + t * HS + cc * 16);
  }
  mma_tok<1, 8>(acc2, w2, Hc, HS, lane);
}

// Timing builds (-DDEC_X_STAMP): the first lane of workgroup 0 leaves shader-clock stamps of its phases behind the used
// part of the hidden-activation buffer (the plan's buffers have room for 128 positions per molecule)
#ifdef DEC_X_STAMP
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) stamp_base[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// ---- middle of a layer, row-local: x2 = LN2(x1 + drop(attn2 Wco^T + b)) and this workgroup's quarter of the hidden
// dimension of the feed-forward block.  Grid = molecules x DEC_FFN_PARTS; the quarters' fp32 partial sums of the block
// output go to a workspace and are added by the kernel that follows (a launch boundary is the cheapest hand-off between
// workgroups); every workgroup of a molecule computes x2 for itself, the first one stores it ----
__global__ __launch_bounds__(NTHR) void dec_mid_kernel(const DecFusedFwd a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * X_BYTES + 2 * TM * HS + MID_VEC * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / DEC_FFN_PARTS, part = blockIdx.x % DEC_FFN_PARTS, T = a.T, m0 = b * T;
  const float dp = a.drop_p;
  const uint32_t thr = drop_threshold(dp);
  const float dscale = dp > 0.f ? 1.0f / (1.0f - dp) : 1.0f;
  const mmst_dec_params& p = a.pt;
  const mmst_dec_acts& s = a.at;
  unsigned char* const X2 = smem;
  unsigned char* const A2 = smem + X_BYTES;
  unsigned char* const U = smem + 2 * X_BYTES;
  float* const Zf = reinterpret_cast<float*>(U);
  float* const vec = reinterpret_cast<float*>(smem + 2 * X_BYTES + 2 * TM * HS);   // b_co | g2 | be2 | b1 of this quarter
#ifdef DEC_X_STAMP
  unsigned long long* const stamp_base = reinterpret_cast<unsigned long long*>((bf16*)s.h + (size_t)a.B * T * FF);
#endif
  STAMP(0);
  constexpr int CPP = FF / HC / DEC_FFN_PARTS;   // chunks per workgroup
  static_assert(CPP == 2, "two chunks per workgroup: both weight sets are requested at the start");
  const int c0 = part * CPP;
  // everything this workgroup will read from the weights is requested before anything is waited for
  bf16x8 wo[1][4], w1A[2][4], w2A[1][8], w1B[2][4], w2B[1][8], res[2];
  uint4 rv[2];
  float vv[2];
  rows_issue(rv, (const bf16*)s.attn2 + (size_t)m0 * D, T, tid);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + k * NTHR < MID_VEC ? tid + k * NTHR : MID_VEC - 1;
    const float* src = i < D ? p.b_co + i : i < 2 * D ? p.g2 + (i - D) : i < 3 * D ? p.be2 + (i - 2 * D) : p.b_1 + c0 * HC + (i - 3 * D);
    vv[k] = *src;
  }
  load_w<1, 4>(wo, (const bf16*)p.f_co, wave, D / 32, 0, lane);
  load_res(res, (const bf16*)s.x1 + (size_t)m0 * D, T, tid);
  // The feed-forward weights (256 KiB per workgroup, every workgroup of the chip at once) are requested only when the
  // few loads the first phases wait for have come back: behind that burst they would take as long as the burst.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const bf16* const W1 = (const bf16*)p.f_1;
  const bf16* const W2 = (const bf16*)p.f_2;
  ffn_load_w(w1A, w2A, W1, W2, c0, wave, lane);
  ffn_load_w(w1B, w2B, W1, W2, c0 + 1, wave, lane);
  order_point();
  rows_commit(A2, rv, T, tid);
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (tid + k * NTHR < MID_VEC) vec[tid + k * NTHR] = vv[k];
  lds_barrier();
  STAMP(1);
  {
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_tok<1, 4>(acc, wo, A2, XS, lane);
    proj_to_scratch(acc, Zf, vec, 16 * wave, m0, dp, a.seed_t + 3, thr, dscale, lane);
  }
  lds_barrier();
  STAMP(2);
  ln_rows<false>(Zf, nullptr, res, vec + D, vec + 2 * D, (bf16*)s.z2 + (size_t)m0 * D, s.st2 + 2 * (size_t)m0, X2,
                 (bf16*)s.x2 + (size_t)m0 * D, T, tid, part == 0);
  uint32_t rseed[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) rseed[mt] = attn_row_seed(a.seed_t + 4, (uint32_t)(m0 + mt * 16 + r));
  f32x4 acc2[1][4];
  zero_acc<1>(acc2);
  lds_barrier();   // x2 image complete (and the scratch, which the first chunk image overlays, has been read)
  STAMP(3);
  bf16* const Hg = (bf16*)s.h + (size_t)m0 * FF;
  ffn_chunk(acc2, w1A, w2A, X2, U, Hg, vec + 3 * D, c0, rseed, dp, thr, dscale, T, tid);
  STAMP(4);
  ffn_chunk(acc2, w1B, w2B, X2, U + TM * HS, Hg, vec + 3 * D + HC, c0 + 1, rseed, dp, thr, dscale, T, tid);
  STAMP(5);
  float* const ws = a.ws + ((size_t)part * a.B * T + m0) * D;
  const int n = 16 * wave + 4 * g;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int t = mt * 16 + r;
    if (t < T) *reinterpret_cast<f32x4*>(ws + (size_t)t * D + n) = acc2[0][mt];
  }
  STAMP(6);
}

// ---- end of a layer and head of the next one, one workgroup per molecule:
//   tail: y = LN3(x2 + drop(sum of the feed-forward partial sums + b2))
//   head: qkv = x Wqkv^T + b, causal self-attention, x1 = LN1(x + drop(attn1 Wo^T + b)), qc = x1 Wcq^T + b ----
// (TAIL / HEAD as template parameters: loads requested under a run-time branch are waited for where the branch ends)
template <bool TAIL, bool HEAD>
__global__ __launch_bounds__(NTHR) void dec_head_kernel(const DecFusedFwd a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM + HEAD_VEC * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, T = a.T, m0 = b * T;
  const float dp = a.drop_p;
  const uint32_t thr = drop_threshold(dp);
  const float dscale = dp > 0.f ? 1.0f / (1.0f - dp) : 1.0f;
  unsigned char* const Xin = smem;             // the head's input rows
  unsigned char* const Oth = smem + X_BYTES;   // attention output, then x1
  unsigned char* const U = smem + 2 * X_BYTES;
  float* const Zf = reinterpret_cast<float*>(U);
  unsigned char* const Vt = U + TM * QS;
  float* const vec = reinterpret_cast<float*>(smem + SMEM);   // tail: b_2 | g3 | be3; head: b_qkv | b_o | g1 | be1 | b_cq
  const mmst_dec_params& p = a.ph;
  const mmst_dec_acts& s = a.ah;
#ifdef DEC_X_STAMP
  unsigned long long* const stamp_base = reinterpret_cast<unsigned long long*>((bf16*)(HEAD ? s.qkv : a.at.h) + (size_t)a.B * T * (HEAD ? 3 * D : FF)) + 16;
#endif
  STAMP(0);
  // what the tail adds up and the small vectors go first, then the head's weights: all requested before anything is
  // waited for, and returned in that order
  const int t_ = tid >> 3, c0_ = (tid & 7) * 16;
  const bool ok_ = t_ < T;
  f32x4 part[DEC_FFN_PARTS][4];
  bf16x8 res[2];
  constexpr bool do_tail = TAIL;
  uint4 rv[2];
  float vt = 0.f, vh[2] = {0.f, 0.f};
  if constexpr (do_tail) {
#pragma unroll
    for (int q = 0; q < DEC_FFN_PARTS; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        part[q][j] = *reinterpret_cast<const f32x4*>(a.ws + ((size_t)q * a.B * T + m0 + (ok_ ? t_ : T - 1)) * D + c0_ + 4 * j);
      }
    load_res(res, (const bf16*)a.at.x2 + (size_t)m0 * D, T, tid);
    {
      const int i = tid < 3 * D ? tid : 3 * D - 1;
      const float* src = i < D ? a.pt.b_2 + i : i < 2 * D ? a.pt.g3 + (i - D) : a.pt.be3 + (i - 2 * D);
      vt = *src;
    }
  } else {
    rows_issue(rv, (const bf16*)s.x + (size_t)m0 * D, T, tid);
  }
  if constexpr (do_tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // as in the middle kernel
  else order_point();
  bf16x8 wq[3][4], wo[1][4], wc[1][4];
  if constexpr (HEAD) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * NTHR < 7 * D ? tid + k * NTHR : 7 * D - 1;
      const float* src = i < 3 * D ? p.b_qkv + i : i < 4 * D ? p.b_o + (i - 3 * D) : i < 5 * D ? p.g1 + (i - 4 * D)
                       : i < 6 * D ? p.be1 + (i - 5 * D) : p.b_cqkv + (i - 6 * D);
      vh[k] = *src;
    }
    order_point();
    load_w<3, 4>(wq, (const bf16*)p.f_qkv, 3 * wave, D / 32, 0, lane);
    load_w<1, 4>(wo, (const bf16*)p.f_o, wave, D / 32, 0, lane);
    load_w<1, 4>(wc, (const bf16*)p.f_cqkv, wave, D / 32, 0, lane);
    order_point();
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + k * NTHR < 7 * D) vec[3 * D + tid + k * NTHR] = vh[k];
  }
  if constexpr (do_tail) {
    if (tid < 3 * D) vec[tid] = vt;
  } else {
    rows_commit(Xin, rv, T, tid);
  }
  if constexpr (do_tail) {
    lds_barrier();   // the staged vectors
    float z[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(vec + c0_ + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) z[4 * j + e] = bv[e] + ((part[0][j][e] + part[1][j][e]) + (part[2][j][e] + part[3][j][e]));
    }
    static_assert(DEC_FFN_PARTS == 4, "sum of the partial sums");
    {
      const uint32_t idx = (uint32_t)(m0 + t_) * 128u + (uint32_t)c0_;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const uint32_t h = drop_bits(a.seed_t + 5, (idx + e) >> 1);
        z[e] = ((h & 0xFFFFu) >= thr) ? z[e] * dscale : 0.f;
        z[e + 1] = ((h >> 16) >= thr) ? z[e + 1] * dscale : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) z[8 * j + e] += (float)res[j][e];
    ln_core(z, vec + D, vec + 2 * D, (bf16*)a.at.z3 + (size_t)m0 * D, a.at.st3 + 2 * (size_t)m0, Xin, (bf16*)a.at.y + (size_t)m0 * D,
            t_, c0_, ok_, (tid & 7) == 0);
  }
  if constexpr (!HEAD) return;
  lds_barrier();
  STAMP(1);
  // ---- qkv = x Wqkv^T + b: rows into the QKV image (and memory), V also transposed ----
  {
    f32x4 acc[3][4];
    zero_acc<3>(acc);
    mma_tok<3, 4>(acc, wq, Xin, XS, lane);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int n = 48 * wave + nt * 16 + 4 * g;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(vec + 3 * D + n);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int t = mt * 16 + r;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][e] + bv[e];
        const uint2 pk = pack4(v);
        *reinterpret_cast<uint2*>(U + t * QS + n * 2) = pk;
        if (n >= 2 * D) {
          unsigned short* vt = reinterpret_cast<unsigned short*>(Vt + (n - 2 * D) * VS + t * 2);
          vt[0] = (unsigned short)(pk.x & 0xFFFFu);
          vt[VS / 2] = (unsigned short)(pk.x >> 16);
          vt[2 * (VS / 2)] = (unsigned short)(pk.y & 0xFFFFu);
          vt[3 * (VS / 2)] = (unsigned short)(pk.y >> 16);
        }
      }
    }
  }
  lds_barrier();
  STAMP(2);
  store_rows<3 * D, QS>((bf16*)s.qkv + (size_t)m0 * 3 * D, 3 * D, U, T, tid);
  STAMP(3);
  // ---- causal self-attention, head_dim 8: wave w serves heads 2 w, 2 w + 1, two 32-query tiles each.  S^T = K Q^T
  // (v_mfma_f32_32x32x16_bf16, contraction zero-padded to 16) puts a query's 32 scores of a key tile into the 16
  // registers of its two lanes; the whole row (<= 64 keys) is in registers, so the softmax is one pass ----
  {
    const int r32 = lane & 31, hf = lane >> 5;
    const float scale2 = rsqrtf(8.0f) * LOG2E;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      const int hh = 2 * wave + hi, hoff = hh * 8;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int q = 32 * qt + r32;
        const bf16x8 qf = hf == 0 ? lds8(U + q * QS + hoff * 2) : zero8();
        f32x16 sa[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) sa[kt][i] = 0.f;
          if (kt <= qt) {
            const bf16x8 kf = hf == 0 ? lds8(U + (32 * kt + r32) * QS + (D + hoff) * 2) : zero8();
            sa[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, sa[kt], 0, 0, 0);
          }
        }
        float pv[2][16];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = 32 * kt + acc_row(i, hf);
            const float v = key <= q ? sa[kt][i] * scale2 : -INFINITY;
            pv[kt][i] = v;
            mx = fmaxf(mx, v);
          }
        mx = xor32_max(mx);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) { pv[kt][i] = __builtin_amdgcn_exp2f(pv[kt][i] - mx); l += pv[kt][i]; }
        l = xor32_sum(l);
        {
          const uint32_t rs = attn_row_seed(a.seed_h + 0, (uint32_t)((b * 16 + hh) * T + q));
#pragma unroll
          for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              const int key = 32 * kt + acc_row(i, hf);
              const uint32_t hb = attn_pair_bits(rs, (uint32_t)(key >> 1) * ATTN_PAIR_MUL);
              pv[kt][i] = ((hb & 0xFFFFu) >= thr) ? pv[kt][i] : 0.f;
              pv[kt][i + 1] = ((hb >> 16) >= thr) ? pv[kt][i + 1] : 0.f;
            }
        }
        f32x16 oacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
#pragma unroll
        for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
          for (int st2 = 0; st2 < 2; ++st2) {
            bf16x8 vf = zero8();
            if (r32 < 8) {
              const unsigned char* vp = Vt + (hoff + r32) * VS + (32 * kt + 16 * st2 + 4 * hf) * 2;
              const uint2 lo = *reinterpret_cast<const uint2*>(vp), hi2 = *reinterpret_cast<const uint2*>(vp + 16);
              const uint4 both = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
              vf = __builtin_bit_cast(bf16x8, both);
            }
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16)pv[kt][8 * st2 + j];
            oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc, 0, 0, 0);
          }
        const float inv = l > 0.f ? dscale / l : 0.f;
        float o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = oacc[j] * inv;
        *reinterpret_cast<uint2*>(Oth + q * XS + (hoff + 4 * hf) * 2) = pack4(o4);
        if (hf == 0 && q < T) s.lse1[(size_t)(b * 16 + hh) * T + q] = mx + __builtin_amdgcn_logf(l);
      }
    }
  }
  lds_barrier();
  STAMP(4);
  store_rows<D, XS>((bf16*)s.attn1 + (size_t)m0 * D, D, Oth, T, tid);
  // ---- x1 = LN1(x + drop(attn1 Wo^T + b)) ----
  {
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_tok<1, 4>(acc, wo, Oth, XS, lane);
    proj_to_scratch(acc, Zf, vec + 6 * D, 16 * wave, m0, dp, a.seed_h + 1, thr, dscale, lane);
  }
  lds_barrier();
  STAMP(5);
  ln_rows<true>(Zf, Xin, res, vec + 7 * D, vec + 8 * D, (bf16*)s.z1 + (size_t)m0 * D, s.st1 + 2 * (size_t)m0, Oth, (bf16*)s.x1 + (size_t)m0 * D, T,
                tid);
  lds_barrier();
  STAMP(6);
  // ---- qc = x1 Wcq^T + b (the query rows of the cross-attention) ----
  {
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_tok<1, 4>(acc, wc, Oth, XS, lane);
    const int n = 16 * wave + 4 * g;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(vec + 9 * D + n);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int t = mt * 16 + r;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[0][mt][e] + bv[e];
      if (t < T) *reinterpret_cast<uint2*>((bf16*)s.qc + (size_t)(m0 + t) * D + n) = pack4(v);
    }
  }
  STAMP(7);
}


// =====================================================================================================================
// Backward (activation gradients).  Same arrangement as the forward kernels; the weight side of every GEMM is the
// fragment-major image of the TRANSPOSED weight (dX = dY W has W^T as its [out][contraction] operand).
// =====================================================================================================================

// acc of a 128-column product (the wave's 16 columns from n0) as it stands -> fp32 scratch
__device__ __forceinline__ void acc_to_scratch(const f32x4 (&acc)[1][4], float* Zf, int n0, int lane) {
  const int r = lane & 15, g = lane >> 4, n = n0 + 4 * g;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(Zf + (mt * 16 + r) * ZS + n) = acc[0][mt];
}

// LayerNorm backward for the lane's 16 columns of row t (8 consecutive lanes share a row): dy -> dz; the lane's
// contributions to dgamma / dbeta are ADDED to pg / pb.  zr: the saved pre-norm row (bf16), gamma: 128 floats in LDS.
__device__ __forceinline__ void lnbwd_core(const float (&dy)[16], const bf16x8 (&zr)[2], float mean, float rstd, const float* gamma,
                                           int c0, bool ok, float (&dz)[16], float (&pg)[16], float (&pb)[16]) {
  float xh[16], gy[16];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c0 + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * j + e;
      xh[i] = ok ? ((float)zr[i >> 3][i & 7] - mean) * rstd : 0.f;
      gy[i] = dy[i] * gv[e];
      s1 += gy[i];
      s2 += gy[i] * xh[i];
    }
  }
  s1 = row8_sum(s1) * (1.0f / 128.0f);
  s2 = row8_sum(s2) * (1.0f / 128.0f);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    dz[i] = rstd * (gy[i] - s1 - xh[i] * s2);
    pg[i] += dy[i] * xh[i];
    pb[i] += dy[i];
  }
}
// column sums of pg / pb over the workgroup's 64 rows -> atomically into dgamma / dbeta.  red: 8 KiB of LDS free of live data.
__device__ __forceinline__ void lnbwd_colsums(float (&pg)[16], float (&pb)[16], float* dgamma, float* dbeta, float* red, int tid,
                                              bool store) {
  const int lane = tid & 63, wave = tid >> 6;
  // the 8 rows of a wave are the lanes with equal lane & 7: rotate by 8 inside the row of 16 lanes, then across the rows
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    pg[i] += dpp_f32<0x128>(pg[i]);   // row_ror:8
    pb[i] += dpp_f32<0x128>(pb[i]);
    pg[i] = xor32_sum(xor16_sum(pg[i]));
    pb[i] = xor32_sum(xor16_sum(pb[i]));
  }
  if (lane < 8) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      red[(wave * 2 + 0) * 128 + lane * 16 + i] = pg[i];
      red[(wave * 2 + 1) * 128 + lane * 16 + i] = pb[i];
    }
  }
  lds_barrier();
  if (store && tid < 256) {
    const int which = tid >> 7, col = tid & 127;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[(w * 2 + which) * 128 + col];
    atomicAdd((which ? dbeta : dgamma) + col, s);
  }
}
// dz (16 columns of row t) -> memory as bf16, and through the element dropout of the forward's row epilogue (pair hash of
// element index m * 128 + n) into an LDS image and memory
__device__ __forceinline__ void store_dz(const float (&dz)[16], bf16* dz_g, bf16* dzd_g, unsigned char* dzd_img, int m, int t, int c0,
                                         bool store, uint32_t seed, uint32_t thr, float dscale) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bf16x8 a, d;
    const uint32_t idx = (uint32_t)m * 128u + (uint32_t)(c0 + 8 * j);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const uint32_t h = drop_bits(seed, (idx + e) >> 1);
      const float v0 = dz[8 * j + e], v1 = dz[8 * j + e + 1];
      a[e] = (bf16)v0;
      a[e + 1] = (bf16)v1;
      d[e] = (bf16)(((h & 0xFFFFu) >= thr) ? v0 * dscale : 0.f);
      d[e + 1] = (bf16)(((h >> 16) >= thr) ? v1 * dscale : 0.f);
    }
    *reinterpret_cast<bf16x8*>(dzd_img + t * XS + (c0 + 8 * j) * 2) = d;
    if (store) {
      *reinterpret_cast<bf16x8*>(dz_g + (size_t)t * D + c0 + 8 * j) = a;
      if (dzd_g) *reinterpret_cast<bf16x8*>(dzd_g + (size_t)t * D + c0 + 8 * j) = d;
    }
  }
}

// ---- feed-forward block backward, row-local, this workgroup's quarter of the hidden dimension:
//   [PRO: dy3 = dqkv' Wqkv' + dz1' (the layer ABOVE, primed), (dz3, dzd3) = LayerNorm-3 backward of this layer]
//   dh = ((dzd3 W2) * mask(h)) / (1 - p)  (stored for the weight gradients),  partial sums of dh W1 -> workspace ----
constexpr int BU_BYTES = TM * QS + TM * ZS * 4;   // dqkv image + fp32 scratch; later the two hidden-chunk images
static_assert(BU_BYTES >= 2 * TM * HS, "union region of the backward middle kernel");
template <bool PRO>
__global__ __launch_bounds__(NTHR) void dec_bwd_mid_kernel(const DecFusedBwd a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[X_BYTES + BU_BYTES + (D + 2048) * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / DEC_FFN_PARTS, part = blockIdx.x % DEC_FFN_PARTS, T = a.T, m0 = b * T;
  const float dp = a.drop_p;
  const uint32_t thr = drop_threshold(dp);
  const float dscale = dp > 0.f ? 1.0f / (1.0f - dp) : 1.0f;
  const mmst_dec_params& p = a.p;
  const mmst_dec_acts& s = a.a;
  const mmst_dec_bwd& w = a.w;
  unsigned char* const D3 = smem;                 // dzd3 rows
  unsigned char* const U = smem + X_BYTES;
  float* const Zf = reinterpret_cast<float*>(U + TM * QS);
  float* const vec = reinterpret_cast<float*>(smem + X_BYTES + BU_BYTES);   // gamma3
  float* const red = vec + D;
  constexpr int CPP = FF / HC / DEC_FFN_PARTS;
  const int c0 = part * CPP;
  const int t_ = tid >> 3, c0_ = (tid & 7) * 16;
  const bool ok_ = t_ < T;
  const bf16* const dzd3_g = (const bf16*)(dp > 0.f ? w.dzd3 : w.dz3);
  bf16x8 w1A[2][4], w2A[1][8], w1B[2][4], w2B[1][8];
  if constexpr (PRO) {
    // small loads first: the rows of dqkv', the residual dz1', this layer's saved z3 and statistics, gamma3, Wqkv'^T
    uint4 qv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = tid + k * NTHR, t = i / 48, c = i % 48;
      qv[k] = *reinterpret_cast<const uint4*>((const bf16*)a.up_dqkv + (size_t)(m0 + (t < T ? t : T - 1)) * 3 * D + c * 8);
    }
    bf16x8 res[2], zr[2];
    load_res(res, (const bf16*)a.up_dz1 + (size_t)m0 * D, T, tid);
    load_res(zr, (const bf16*)s.z3 + (size_t)m0 * D, T, tid);
    const float mean = s.st3[2 * (size_t)(m0 + (ok_ ? t_ : T - 1))], rstd = s.st3[2 * (size_t)(m0 + (ok_ ? t_ : T - 1)) + 1];
    const float gv = p.g3[tid & 127];
    bf16x8 wq[1][12];
    load_w<1, 12>(wq, (const bf16*)a.up_ft_qkv, wave, 3 * D / 32, 0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ffn_load_w(w1A, w2A, (const bf16*)p.ft_2, (const bf16*)p.ft_1, c0, wave, lane);
    ffn_load_w(w1B, w2B, (const bf16*)p.ft_2, (const bf16*)p.ft_1, c0 + 1, wave, lane);
    order_point();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = tid + k * NTHR, t = i / 48, c = i % 48;
      *reinterpret_cast<uint4*>(U + t * QS + c * 16) = t < T ? qv[k] : make_uint4(0, 0, 0, 0);
    }
    if (tid < D) vec[tid] = gv;
    lds_barrier();
    {
      f32x4 acc[1][4];
      zero_acc<1>(acc);
      mma_tok<1, 12>(acc, wq, U, QS, lane);
      acc_to_scratch(acc, Zf, 16 * wave, lane);
    }
    lds_barrier();
    float dy[16], dz[16], pg[16], pb[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Zf + t_ * ZS + c0_ + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * j + e;
        dy[i] = ok_ ? v[e] + (float)res[i >> 3][i & 7] : 0.f;
        pg[i] = 0.f;
        pb[i] = 0.f;
      }
    }
    lnbwd_core(dy, zr, mean, rstd, vec, c0_, ok_, dz, pg, pb);
    store_dz(dz, (bf16*)w.dz3 + (size_t)m0 * D, dp > 0.f ? (bf16*)w.dzd3 + (size_t)m0 * D : nullptr, D3, m0 + t_, t_, c0_,
             ok_ && part == 0, a.seed + 5, thr, dscale);
    lnbwd_colsums(pg, pb, a.g.g3, a.g.be3, red, tid, part == 0);   // (its barrier also completes the dzd3 image)
  } else {
    uint4 rv[2];
    rows_issue(rv, dzd3_g + (size_t)m0 * D, T, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ffn_load_w(w1A, w2A, (const bf16*)p.ft_2, (const bf16*)p.ft_1, c0, wave, lane);
    ffn_load_w(w1B, w2B, (const bf16*)p.ft_2, (const bf16*)p.ft_1, c0 + 1, wave, lane);
    order_point();
    rows_commit(D3, rv, T, tid);
    lds_barrier();
  }
  // ---- two chunks of 256 hidden columns: the saved activation's chunk is staged into the chunk image, the epilogue
  // replaces each entry by its gradient in place ----
  const bf16* const Hg = (const bf16*)s.h + (size_t)m0 * FF;
  bf16* const dHg = (bf16*)w.dh + (size_t)m0 * FF;
  f32x4 acc2[1][4];
  zero_acc<1>(acc2);
#pragma unroll
  for (int cc = 0; cc < CPP; ++cc) {
    const int c = c0 + cc;
    unsigned char* const Hc = U + cc * (TM * HS);
    uint4 hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * NTHR, t = i >> 5, q = i & 31;
      hv[k] = *reinterpret_cast<const uint4*>(Hg + (size_t)(t < T ? t : T - 1) * FF + c * HC + q * 8);
    }
    f32x4 acc1[2][4];
    zero_acc<2>(acc1);
    if (cc == 0) mma_tok<2, 4>(acc1, w1A, D3, XS, lane);
    else mma_tok<2, 4>(acc1, w1B, D3, XS, lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * NTHR, t = i >> 5, q = i & 31;
      *reinterpret_cast<uint4*>(Hc + t * HS + q * 16) = hv[k];
    }
    lds_barrier();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int lc = 32 * wave + nt * 16 + 4 * g;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        unsigned char* const hp = Hc + (mt * 16 + r) * HS + lc * 2;
        const uint2 hb = *reinterpret_cast<const uint2*>(hp);
        float v[4];
        v[0] = (hb.x & 0xFFFFu) ? acc1[nt][mt][0] * dscale : 0.f;   // saved entries are >= 0: nonzero bits = positive
        v[1] = (hb.x >> 16) ? acc1[nt][mt][1] * dscale : 0.f;
        v[2] = (hb.y & 0xFFFFu) ? acc1[nt][mt][2] * dscale : 0.f;
        v[3] = (hb.y >> 16) ? acc1[nt][mt][3] * dscale : 0.f;
        *reinterpret_cast<uint2*>(hp) = pack4(v);
      }
    }
    lds_barrier();
    for (int i = tid; i < TM * 32; i += NTHR) {
      const int t = i >> 5, q = i & 31;
      if (t < T) *reinterpret_cast<uint4*>(dHg + (size_t)t * FF + c * HC + q * 8) = *reinterpret_cast<const uint4*>(Hc + t * HS + q * 16);
    }
    if (cc == 0) mma_tok<1, 8>(acc2, w2A, Hc, HS, lane);
    else mma_tok<1, 8>(acc2, w2B, Hc, HS, lane);
  }
  float* const ws = a.ws + ((size_t)part * a.B * T + m0) * D;
  const int n = 16 * wave + 4 * g;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int t = mt * 16 + r;
    if (t < T) *reinterpret_cast<f32x4*>(ws + (size_t)t * D + n) = acc2[0][mt];
  }
}

// ---- LayerNorm backward between two projections, one workgroup per molecule:
//   dy = (PARTS ? sum of the feed-forward partial sums : in Win) + residual;  (dz, dzd) = LayerNorm backward;
//   out = dzd Wout.   PARTS: norm2 (in front of the cross-attention's out_proj);  otherwise norm1 with in = dqc ----
template <bool PARTS>
__global__ __launch_bounds__(NTHR) void dec_bwd_ln_kernel(const DecFusedLn a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * X_BYTES + TM * ZS * 4 + (D + 2048) * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, T = a.T, m0 = b * T;
  const float dp = a.drop_p;
  const uint32_t thr = drop_threshold(dp);
  const float dscale = dp > 0.f ? 1.0f / (1.0f - dp) : 1.0f;
  unsigned char* const IN = smem;                 // input rows of the first projection
  unsigned char* const DZ = smem + X_BYTES;       // dzd rows
  float* const Zf = reinterpret_cast<float*>(smem + 2 * X_BYTES);
  float* const vec = reinterpret_cast<float*>(smem + 2 * X_BYTES + TM * ZS * 4);
  float* const red = vec + D;
  const int t_ = tid >> 3, c0_ = (tid & 7) * 16;
  const bool ok_ = t_ < T;
  const size_t row = (size_t)(m0 + (ok_ ? t_ : T - 1));
  f32x4 part[DEC_FFN_PARTS][4];
  uint4 rv[2];
  if constexpr (PARTS) {
#pragma unroll
    for (int q = 0; q < DEC_FFN_PARTS; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        part[q][j] = *reinterpret_cast<const f32x4*>(a.ws + ((size_t)q * a.B * T + row) * D + c0_ + 4 * j);
  } else {
    rows_issue(rv, (const bf16*)a.in + (size_t)m0 * D, T, tid);
  }
  bf16x8 res[2], zr[2];
  load_res(res, (const bf16*)a.res + (size_t)m0 * D, T, tid);
  load_res(zr, (const bf16*)a.z + (size_t)m0 * D, T, tid);
  const float mean = a.stats[2 * row], rstd = a.stats[2 * row + 1];
  const float gv = a.gamma[tid & 127];
  bf16x8 wi[1][4], wo[1][4];
  if constexpr (!PARTS) load_w<1, 4>(wi, (const bf16*)a.ft_in, wave, D / 32, 0, lane);
  load_w<1, 4>(wo, (const bf16*)a.ft_out, wave, D / 32, 0, lane);
  order_point();
  if (tid < D) vec[tid] = gv;
  float dy[16], dz[16], pg[16], pb[16];
  if constexpr (!PARTS) {
    rows_commit(IN, rv, T, tid);
    lds_barrier();
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_tok<1, 4>(acc, wi, IN, XS, lane);
    acc_to_scratch(acc, Zf, 16 * wave, lane);
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Zf + t_ * ZS + c0_ + 4 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) dy[4 * j + e] = v[e];
    }
  } else {
    lds_barrier();   // gamma staged
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) dy[4 * j + e] = (part[0][j][e] + part[1][j][e]) + (part[2][j][e] + part[3][j][e]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    dy[i] = ok_ ? dy[i] + (float)res[i >> 3][i & 7] : 0.f;
    pg[i] = 0.f;
    pb[i] = 0.f;
  }
  lnbwd_core(dy, zr, mean, rstd, vec, c0_, ok_, dz, pg, pb);
  store_dz(dz, (bf16*)a.dz + (size_t)m0 * D, dp > 0.f ? (bf16*)a.dzd + (size_t)m0 * D : nullptr, DZ, m0 + t_, t_, c0_, ok_, a.seed, thr,
           dscale);
  lnbwd_colsums(pg, pb, a.dgamma, a.dbeta, red, tid, true);   // (its barrier also completes the dzd image)
  {
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_tok<1, 4>(acc, wo, DZ, XS, lane);
    const int n = 16 * wave + 4 * g;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int t = mt * 16 + r;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[0][mt][e];
      if (t < T) *reinterpret_cast<uint2*>((bf16*)a.out + (size_t)(m0 + t) * D + n) = pack4(v);
    }
  }
}


// ---- start of a layer's backward below the cross-attention, one workgroup per molecule:
//   (dz1, dzd1) = LayerNorm-1 backward of (dqc Wcq + dz2);  da = dzd1 Wo
//   causal self-attention backward from da, the saved Q|K|V rows, attention output and row statistics -> dqkv
//   dx = dqkv Wqkv + dz1;  DOWN: (dz3, dzd3) of the layer BELOW = its LayerNorm-3 backward of dx;  else dx is the output
// Self-attention, head_dim 8, on v_mfma_f32_32x32x16_bf16 with the contraction zero-padded to 16; wave w serves heads
// 2 w and 2 w + 1.  The probabilities are recomputed in BOTH orientations, so that no score matrix is ever transposed:
//   query on the lane, 32 keys of a tile on the registers:  S^T = K Q^T, dP^T = V dO^T  ->  dQ^T += K^T dS
//   key on the lane, 32 queries of a tile on the registers: S = Q K^T,   dP = dO V^T    ->  dK^T += Q^T dS, dV^T += dO^T P
// (the second operand of each product is the score registers themselves; the first comes from transposed images
// K^T / Q^T / dO^T [feature][token], read in the register order of the scores).  dQ / dK / dV replace the head's
// Q / K / V columns of the row image in place (no other wave reads them), which then is the token side of the
// projection down.
constexpr int SELF_TAB = 16 * TM * 4;   // one [head][token] table
constexpr int SELF_SMEM = TM * QS + X_BYTES + D * VS + 2 * D * VS + 3 * SELF_TAB + (2 * D + 2048) * 4;
static_assert(2 * D * VS >= TM * ZS * 4, "fp32 scratch inside the transposed images' region");
static_assert(SELF_SMEM <= 160 * 1024, "LDS of one compute unit");

// 8 values of row `row` of a transposed image for the contraction step (tile kt, half st2) in the register order of a
// score tile: tokens 32 kt + 16 st2 + 4 hf + {0..3} and the same + 8
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* img, int row, int kt, int st2, int hf) {
  const unsigned char* vp = img + row * VS + (32 * kt + 16 * st2 + 4 * hf) * 2;
  const uint2 lo = *reinterpret_cast<const uint2*>(vp), hi = *reinterpret_cast<const uint2*>(vp + 16);
  return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

template <bool DOWN>
__global__ __launch_bounds__(NTHR) void dec_bwd_self_kernel(const DecFusedSelf a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SELF_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, T = a.T, m0 = b * T;
  const float dp = a.drop_p;
  const uint32_t thr = drop_threshold(dp);
  const float dscale = dp > 0.f ? 1.0f / (1.0f - dp) : 1.0f;
  unsigned char* const QKV = smem;                       // Q|K|V rows, later dQ|dK|dV
  unsigned char* const R1 = QKV + TM * QS;               // dqc rows, then dO rows
  unsigned char* const R2 = R1 + X_BYTES;                // dzd1 rows, then dO^T
  unsigned char* const R3 = R2 + D * VS;                 // fp32 scratch; K^T | Q^T while the attention runs
  float* const Zf = reinterpret_cast<float*>(R3);
  unsigned char* const Kt = R3;
  unsigned char* const Qt = R3 + D * VS;
  float* const lse_s = reinterpret_cast<float*>(R3 + 2 * D * VS);   // [16][64]
  float* const delta_s = lse_s + 16 * TM;
  uint32_t* const rs_s = reinterpret_cast<uint32_t*>(delta_s + 16 * TM);
  float* const vec = reinterpret_cast<float*>(rs_s + 16 * TM);     // gamma1 | gamma3
  float* const red = vec + 2 * D;
  const int t_ = tid >> 3, c0_ = (tid & 7) * 16;
  const bool ok_ = t_ < T;
  const size_t row = (size_t)(m0 + (ok_ ? t_ : T - 1));
#ifdef DEC_X_STAMP
  unsigned long long* const stamp_base = reinterpret_cast<unsigned long long*>((bf16*)a.dqkv + (size_t)a.B * T * 3 * D);
#endif
  STAMP(0);

  // ---- everything the first part reads is requested before anything is waited for ----
  uint4 rv[2], qv[6];
  rows_issue(rv, (const bf16*)a.dqc + (size_t)m0 * D, T, tid);
  bf16x8 res[2], zr[2], orow[2];
  load_res(res, (const bf16*)a.dz2 + (size_t)m0 * D, T, tid);
  load_res(zr, (const bf16*)a.z1 + (size_t)m0 * D, T, tid);
  const float mean = a.st1[2 * row], rstd = a.st1[2 * row + 1];
  const float gv = tid < D ? a.g1[tid] : (DOWN ? a.g3[tid & 127] : 0.f);
  bf16x8 wi[1][4], wo[1][4];
  load_w<1, 4>(wi, (const bf16*)a.ft_cq, wave, D / 32, 0, lane);
  load_w<1, 4>(wo, (const bf16*)a.ft_o, wave, D / 32, 0, lane);
  order_point();
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int i = tid + k * NTHR, t = i / 48, c = i % 48;
    qv[k] = *reinterpret_cast<const uint4*>((const bf16*)a.qkv + (size_t)(m0 + (t < T ? t : T - 1)) * 3 * D + c * 8);
  }
  load_res(orow, (const bf16*)a.attn1 + (size_t)m0 * D, T, tid);
  float lsev[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + k * NTHR, hh = i >> 6, q = i & 63;
    lsev[k] = a.lse1[(size_t)(b * 16 + hh) * T + (q < T ? q : T - 1)];
  }
  order_point();
  if (tid < 2 * D) vec[tid] = gv;
  rows_commit(R1, rv, T, tid);
  lds_barrier();   // (the Q|K|V rows and the tables are stored further down: their loads land beside the first product)
  STAMP(1);
  // ---- norm1 backward ----
  float dy[16], dz[16], pg[16], pb[16];
  {
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_tok<1, 4>(acc, wi, R1, XS, lane);
    acc_to_scratch(acc, Zf, 16 * wave, lane);
  }
  lds_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(Zf + t_ * ZS + c0_ + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * j + e;
      dy[i] = ok_ ? v[e] + (float)res[i >> 3][i & 7] : 0.f;
      pg[i] = 0.f;
      pb[i] = 0.f;
    }
  }
  lnbwd_core(dy, zr, mean, rstd, vec, c0_, ok_, dz, pg, pb);
  store_dz(dz, (bf16*)a.dz1 + (size_t)m0 * D, dp > 0.f ? (bf16*)a.dzd1 + (size_t)m0 * D : nullptr, R2, m0 + t_, t_, c0_, ok_,
           a.seed + 1, thr, dscale);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int i = tid + k * NTHR, t = i / 48, c = i % 48;
    *reinterpret_cast<uint4*>(QKV + t * QS + c * 16) = t < T ? qv[k] : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + k * NTHR, hh = i >> 6, q = i & 63;
    lse_s[i] = q < T ? lsev[k] : 1.0e30f;    // rows past the end: every probability 0
    rs_s[i] = attn_row_seed(a.seed + 0, (uint32_t)((b * 16 + hh) * T + q));
  }
  lnbwd_colsums(pg, pb, a.dg1, a.dbe1, red, tid, true);   // (its barrier also completes the dzd1 and the Q|K|V image)
  STAMP(2);
  // ---- da = dzd1 Wo: rows into R1, transposed into R2 (which the product reads: a barrier in between) ----
  {
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_tok<1, 4>(acc, wo, R2, XS, lane);
    lds_barrier();
    const int n = 16 * wave + 4 * g;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int t = mt * 16 + r;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[0][mt][e];
      const uint2 pk = pack4(v);
      *reinterpret_cast<uint2*>(R1 + t * XS + n * 2) = pk;
      unsigned short* tp = reinterpret_cast<unsigned short*>(R2 + n * VS + t * 2);
      tp[0] = (unsigned short)(pk.x & 0xFFFFu);
      tp[VS / 2] = (unsigned short)(pk.x >> 16);
      tp[2 * (VS / 2)] = (unsigned short)(pk.y & 0xFFFFu);
      tp[3 * (VS / 2)] = (unsigned short)(pk.y >> 16);
    }
  }
  STAMP(3);
  // K^T and Q^T from the row image (the fp32 scratch is free since the rows were read)
  // (a thread takes two neighbouring tokens of one 8-column chunk, so that it writes whole 32-bit words, and the lanes of
  // a wave take neighbouring token pairs of the same chunk, so that a write instruction covers contiguous words)
  for (int i = tid; i < (TM / 2) * 32; i += NTHR) {
    const int tp2 = i & 31, c = i >> 5;                // chunk c: columns 8 c .. of Q (c < 16) or K
    const uint4 u0 = __builtin_bit_cast(uint4, lds8(QKV + (2 * tp2) * QS + c * 16));
    const uint4 u1 = __builtin_bit_cast(uint4, lds8(QKV + (2 * tp2 + 1) * QS + c * 16));
    uint32_t* wp = reinterpret_cast<uint32_t*>((c < 16 ? Qt : Kt) + ((c & 15) * 8) * VS + tp2 * 4);
    const uint32_t a0[4] = {u0.x, u0.y, u0.z, u0.w}, a1[4] = {u1.x, u1.y, u1.z, u1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wp[(2 * e) * (VS / 4)] = (a0[e] & 0xFFFFu) | (a1[e] << 16);
      wp[(2 * e + 1) * (VS / 4)] = (a0[e] >> 16) | (a1[e] & 0xFFFF0000u);
    }
  }
  lds_barrier();
  // delta[head][q] = sum over the head's 8 columns of dO * O (O: the saved attention output)
  {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16x8 dov = lds8(R1 + t_ * XS + (c0_ + 8 * j) * 2);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)dov[e] * (float)orow[j][e];
      delta_s[((c0_ >> 3) + j) * TM + t_] = ok_ ? s : 0.f;
    }
  }
  lds_barrier();
  STAMP(4);
  // ---- self-attention backward ----
  {
    const int r32 = lane & 31, hf = lane >> 5;
    const float scale = rsqrtf(8.0f), scale2 = scale * LOG2E;
#pragma unroll 1
    for (int hi = 0; hi < 2; ++hi) {
      const int hh = 2 * wave + hi, hoff = hh * 8;
      float dq4[2][4], dk4[2][4], dv4[2][4];
      // query on the lane -> dQ
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int q = 32 * qt + r32;
        const bf16x8 qf = hf == 0 ? lds8(QKV + q * QS + hoff * 2) : zero8();
        const bf16x8 dof = hf == 0 ? lds8(R1 + q * XS + hoff * 2) : zero8();
        const float lq = lse_s[hh * TM + q], dl = delta_s[hh * TM + q];
        const uint32_t rs = rs_s[hh * TM + q];
        f32x16 oacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
#pragma unroll
        for (int kt = 0; kt <= qt; ++kt) {
          const bf16x8 kf = hf == 0 ? lds8(QKV + (32 * kt + r32) * QS + (D + hoff) * 2) : zero8();
          const bf16x8 vf = hf == 0 ? lds8(QKV + (32 * kt + r32) * QS + (2 * D + hoff) * 2) : zero8();
          f32x16 sa, da_;
#pragma unroll
          for (int i = 0; i < 16; ++i) { sa[i] = 0.f; da_[i] = 0.f; }
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, sa, 0, 0, 0);
          da_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof, da_, 0, 0, 0);
          float ds[16];
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const int key = 32 * kt + acc_row(i, hf);
            const uint32_t hb = attn_pair_bits(rs, (uint32_t)(key >> 1) * ATTN_PAIR_MUL);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float pr = (key + e) <= q ? __builtin_amdgcn_exp2f(sa[i + e] * scale2 - lq) : 0.f;
              const bool keep = (e ? (hb >> 16) : (hb & 0xFFFFu)) >= thr;
              ds[i + e] = pr * ((keep ? da_[i + e] * dscale : 0.f) - dl) * scale;
            }
          }
#pragma unroll
          for (int st2 = 0; st2 < 2; ++st2) {
            const bf16x8 ktf = r32 < 8 ? tr_frag(Kt, hoff + r32, kt, st2, hf) : zero8();
            bf16x8 dsf;
#pragma unroll
            for (int j = 0; j < 8; ++j) dsf[j] = (bf16)ds[8 * st2 + j];
            oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, oacc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) dq4[qt][j] = oacc[j];
      }
      // key on the lane -> dK, dV
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int key = 32 * kt + r32;
        const bf16x8 kf = hf == 0 ? lds8(QKV + key * QS + (D + hoff) * 2) : zero8();
        const bf16x8 vf = hf == 0 ? lds8(QKV + key * QS + (2 * D + hoff) * 2) : zero8();
        f32x16 kacc, vacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) { kacc[i] = 0.f; vacc[i] = 0.f; }
#pragma unroll
        for (int qt = kt; qt < 2; ++qt) {
          const bf16x8 qf = hf == 0 ? lds8(QKV + (32 * qt + r32) * QS + hoff * 2) : zero8();
          const bf16x8 dof = hf == 0 ? lds8(R1 + (32 * qt + r32) * XS + hoff * 2) : zero8();
          f32x16 sb, db;
#pragma unroll
          for (int i = 0; i < 16; ++i) { sb[i] = 0.f; db[i] = 0.f; }
          sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf, sb, 0, 0, 0);
          db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, vf, db, 0, 0, 0);
          float pd[16], ds[16];
          // the registers' queries come in runs of four (acc_row): their table entries are read as 16-byte pieces
          typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
          f32x4 lq4[4], dl4[4];
          u32x4_t rs4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q0 = hh * TM + 32 * qt + 8 * j + 4 * hf;
            lq4[j] = *reinterpret_cast<const f32x4*>(lse_s + q0);
            dl4[j] = *reinterpret_cast<const f32x4*>(delta_s + q0);
            rs4[j] = *reinterpret_cast<const u32x4_t*>(rs_s + q0);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int q = 32 * qt + acc_row(i, hf);
            const float lq = lq4[i >> 2][i & 3], dl = dl4[i >> 2][i & 3];
            const uint32_t hb = attn_pair_bits(rs4[i >> 2][i & 3], (uint32_t)(key >> 1) * ATTN_PAIR_MUL);
            const bool keep = ((key & 1) ? (hb >> 16) : (hb & 0xFFFFu)) >= thr;
            const float pr = key <= q ? __builtin_amdgcn_exp2f(sb[i] * scale2 - lq) : 0.f;
            pd[i] = keep ? pr * dscale : 0.f;
            ds[i] = pr * ((keep ? db[i] * dscale : 0.f) - dl) * scale;
          }
#pragma unroll
          for (int st2 = 0; st2 < 2; ++st2) {
            const bf16x8 qtf = r32 < 8 ? tr_frag(Qt, hoff + r32, qt, st2, hf) : zero8();
            const bf16x8 dotf = r32 < 8 ? tr_frag(R2, hoff + r32, qt, st2, hf) : zero8();
            bf16x8 dsf, pdf;
#pragma unroll
            for (int j = 0; j < 8; ++j) { dsf[j] = (bf16)ds[8 * st2 + j]; pdf[j] = (bf16)pd[8 * st2 + j]; }
            kacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, kacc, 0, 0, 0);
            vacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pdf, vacc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { dk4[kt][j] = kacc[j]; dv4[kt][j] = vacc[j]; }
      }
      // the head's gradients replace its Q / K / V columns (this wave is their only reader)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 32 * tt + r32;
        *reinterpret_cast<uint2*>(QKV + t * QS + (hoff + 4 * hf) * 2) = pack4(dq4[tt]);
        *reinterpret_cast<uint2*>(QKV + t * QS + (D + hoff + 4 * hf) * 2) = pack4(dk4[tt]);
        *reinterpret_cast<uint2*>(QKV + t * QS + (2 * D + hoff + 4 * hf) * 2) = pack4(dv4[tt]);
      }
    }
  }
  STAMP(5);
  // ---- dx = dqkv Wqkv + dz1 (this lane's dz1 columns: its own stores of the first part) ----
  // (requested only now: held across the attention they cost 48 registers, which then spills)
  bf16x8 wq[1][12];
  load_w<1, 12>(wq, (const bf16*)a.ft_qkv, wave, 3 * D / 32, 0, lane);
  bf16x8 res1[2], z3r[2];
  load_res(res1, (const bf16*)a.dz1 + (size_t)m0 * D, T, tid);
  float mean3 = 0.f, rstd3 = 0.f;
  if constexpr (DOWN) {
    load_res(z3r, (const bf16*)a.z3 + (size_t)m0 * D, T, tid);
    mean3 = a.st3[2 * row];
    rstd3 = a.st3[2 * row + 1];
  }
  lds_barrier();
  store_rows<3 * D, QS>((bf16*)a.dqkv + (size_t)m0 * 3 * D, 3 * D, QKV, T, tid);
  {
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_tok<1, 12>(acc, wq, QKV, QS, lane);
    acc_to_scratch(acc, Zf, 16 * wave, lane);
  }
  lds_barrier();
  STAMP(6);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(Zf + t_ * ZS + c0_ + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * j + e;
      dy[i] = ok_ ? v[e] + (float)res1[i >> 3][i & 7] : 0.f;
      pg[i] = 0.f;
      pb[i] = 0.f;
    }
  }
  if constexpr (DOWN) {
    lnbwd_core(dy, z3r, mean3, rstd3, vec + D, c0_, ok_, dz, pg, pb);
    store_dz(dz, (bf16*)a.dz3 + (size_t)m0 * D, dp > 0.f ? (bf16*)a.dzd3 + (size_t)m0 * D : nullptr, R1, m0 + t_, t_, c0_, ok_,
             a.seed_below + 5, thr, dscale);
    lnbwd_colsums(pg, pb, a.dg3, a.dbe3, red, tid, true);
  } else if (ok_) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)dy[8 * j + e];
      *reinterpret_cast<bf16x8*>((bf16*)a.dx + (size_t)(m0 + t_) * D + c0_ + 8 * j) = o;
    }
  }
  STAMP(7);
}

}  // namespace

namespace {
int g_dec_fused = -1;   // -1: the default (on, or what the environment variable MMST_DEC_FUSED says)
}
bool dec_fused_enabled() {
  static const bool def = [] { const char* e = getenv("MMST_DEC_FUSED"); return e ? atoi(e) != 0 : true; }();
  return g_dec_fused < 0 ? def : g_dec_fused != 0;
}
void dec_fused_set(int on) { g_dec_fused = on < 0 ? -1 : (on != 0); }

int launch_dec_fused_fwd(const DecFusedFwd& a, hipStream_t stream) {
  if (a.B < 1 || a.T < 1 || a.T > TM || (!a.do_tail && !a.do_head) || (a.do_tail && !a.ws)) return MMST_ERR_ARG;
  if (a.do_tail) {
    hipLaunchKernelGGL(dec_mid_kernel, dim3(a.B * DEC_FFN_PARTS), dim3(NTHR), 0, stream, a);
    MMST_CHECK_LAUNCH();
  }
  if (a.do_tail && a.do_head) MMST_LAUNCH((dec_head_kernel<true, true>), dim3(a.B), dim3(NTHR), 0, stream, a);
  else if (a.do_tail) MMST_LAUNCH((dec_head_kernel<true, false>), dim3(a.B), dim3(NTHR), 0, stream, a);
  else MMST_LAUNCH((dec_head_kernel<false, true>), dim3(a.B), dim3(NTHR), 0, stream, a);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

int launch_dec_bwd_mid(const DecFusedBwd& a, hipStream_t stream) {
  if (a.B < 1 || a.T < 1 || a.T > TM || !a.ws) return MMST_ERR_ARG;
  if (a.up_dqkv) MMST_LAUNCH((dec_bwd_mid_kernel<true>), dim3(a.B * DEC_FFN_PARTS), dim3(NTHR), 0, stream, a);
  else MMST_LAUNCH((dec_bwd_mid_kernel<false>), dim3(a.B * DEC_FFN_PARTS), dim3(NTHR), 0, stream, a);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
int launch_dec_bwd_self(const DecFusedSelf& a, hipStream_t stream) {
  if (a.B < 1 || a.T < 1 || a.T > TM || !a.dqc || !a.qkv || !a.dqkv || !a.ft_qkv) return MMST_ERR_ARG;
  if (!a.ft_cq || !a.dz2 || !a.z1 || !a.st1 || !a.g1 || !a.dg1 || !a.dbe1 || !a.dz1 || !a.ft_o || !a.attn1 || !a.lse1)
    return MMST_ERR_ARG;
  if (a.drop_p > 0.f && !a.dzd1) return MMST_ERR_ARG;
  if (a.z3 && (!a.st3 || !a.g3 || !a.dg3 || !a.dbe3 || !a.dz3 || (a.drop_p > 0.f && !a.dzd3))) return MMST_ERR_ARG;
  if (a.z3) MMST_LAUNCH((dec_bwd_self_kernel<true>), dim3(a.B), dim3(NTHR), 0, stream, a);
  else if (a.dx) MMST_LAUNCH((dec_bwd_self_kernel<false>), dim3(a.B), dim3(NTHR), 0, stream, a);
  else return MMST_ERR_ARG;
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
int launch_dec_bwd_ln(const DecFusedLn& a, hipStream_t stream) {
  if (a.B < 1 || a.T < 1 || a.T > TM) return MMST_ERR_ARG;
  if (a.ws) MMST_LAUNCH((dec_bwd_ln_kernel<true>), dim3(a.B), dim3(NTHR), 0, stream, a);
  else MMST_LAUNCH((dec_bwd_ln_kernel<false>), dim3(a.B), dim3(NTHR), 0, stream, a);
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
