// Multi-head attention forward / backward, head_dim 8 or 32, for gfx950.
//
// One 64-lane wave owns a 32-row tile of one side ("owner" rows live on the MFMA
// lane); the other side is streamed through LDS in 128-row chunks shared by the
// workgroup's waves.  All products are v_mfma_f32_32x32x16_bf16:
//   forward / dQ pass : owner = queries.  S^T = K Q^T puts a query's scores for 32
//     keys into the 16 accumulator registers of the two lanes (l, l+32) of that query,
//     so softmax statistics are lane-local plus one cross-half shuffle.  The
//     probability registers are converted to bf16 in place and fed back as the B
//     operand of O^T += V^T P^T (dQ^T += K^T dS^T); the A operand comes from a
//     transposed LDS image whose k order matches the accumulator's row order.
//   dK/dV pass        : owner = keys; S = Q K^T and dP = dO V^T are produced with the
//     key on the lane, and P / dS feed dV^T += dO^T P and dK^T += Q^T dS.
// head_dim 8 uses the same tiles with the contraction zero-padded to 16 (half of the
// lanes supply zeros): the blocks are exp/VALU bound, the matrix pipe has slack.
// A workgroup is 8 waves and covers a GROUP of heads whose columns are 128 contiguous bytes of every
// token row (8 heads of 8, 2 heads of 32): the streamed chunk is staged with full-line coalesced loads
// and every byte fetched from HBM is used (one (batch, head) per workgroup would fetch 16 B of each
// 128 B line and pay 4-8x the traffic).  Wave w works on head (w % HG) and owner tile (w / HG).
// Dropout on the attention probabilities is a counter hash of (b, h, q, key), so the
// backward passes regenerate the forward's mask.
#include "attn.h"
#include <cstdlib>

namespace {

constexpr int KC = 128;             // streamed rows per LDS chunk
constexpr int TSTR = KC * 2 + 8;    // byte stride of a transposed-image row (== 8 mod 256)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

constexpr int GC = 64;              // columns of a head group (= 128 bytes of a token row)
constexpr int NWAVES = 8;

template <int HD> struct Cfg {
  static constexpr int KS = HD == 32 ? 2 : 1;       // contraction steps over head_dim
  static constexpr int HG = GC / HD;                // heads per workgroup
  static constexpr int TPB = NWAVES / HG;           // owner tiles per workgroup
  static constexpr int RM_BYTES = KC * GC * 2;      // row-major image [KC][64]
  static constexpr int TR_BYTES = GC * TSTR;        // transposed image [64][KC]
  static constexpr int OREGS = HD == 32 ? 16 : 4;   // valid accumulator registers of a d-major output
};

__device__ __forceinline__ int acc_row(int r16, int h) { return (r16 & 3) + 8 * (r16 >> 2) + 4 * h; }

// 128-byte rows, 16-byte chunk ch XOR-swizzled so that the 16 rows of a ds_read_b128 lane group
// (same chunk, rows r .. r+3 / r+12 .. / r+20 ..) fall on 16 different bank slots
__device__ __forceinline__ int rm_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// Stage `nvalid` rows (zero-filled up to KC) of a [rows][64] head-group slice into a row-major
// image and/or a transposed image.  8 consecutive lanes read one full 128-byte line.
template <bool RM, bool TR>
__device__ __forceinline__ void stage(unsigned char* rm, unsigned char* tr, const bf16* src, int ld, int nvalid,
                                      int tid, int nthr) {
  constexpr int CPR = GC / 8;
  for (int c = tid; c < KC * CPR; c += nthr) {
    int row = c / CPR, ch = c % CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nvalid) v = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + ch * 8);
    if (RM) *reinterpret_cast<uint4*>(rm + rm_off(row, ch)) = v;
    if (TR) {
      const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<unsigned short*>(tr + (ch * 8 + j) * TSTR + row * 2) = e[j];
    }
  }
}

constexpr int NTHR = 64 * 8;

// Row-major image of a streamed chunk by LDS-DMA (no registers, no LDS store instructions): 128 rows x 128 B, a wave
// instruction fills 8 rows (1 KiB, contiguous in LDS); the chunk swizzle of rm_off() is applied on the global side (the
// lane that fills 16-byte slot `slot` of a row fetches logical chunk slot ^ key(row)).  Rows past `nvalid` repeat the
// last valid row: finite data, whose scores the callers mask through their bias / log-sum-exp vectors.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void stage_dma(unsigned char* img, const bf16* src, int ld, int nvalid, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < KC / 8 / NWAVES; ++i) {
    const int grp = i * NWAVES + wave;
    const int row = grp * 8 + (lane >> 3), slot = lane & 7;
    const int ch = slot ^ ((row >> 1) & 7);
    const int srow = row < nvalid ? row : nvalid - 1;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)srow * ld + ch * 8), (lptr_t)(img + grp * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
  return z;
}

// A operand (32 rows x 16 k) from a row-major image: lane (r, h) takes row `row`, k = 16 s + 8 h ..
template <int HD>
__device__ __forceinline__ bf16x8 afrag_rm(const unsigned char* img, int row, int s, int h, int hoff) {
  if (HD == 32) return *reinterpret_cast<const bf16x8*>(img + rm_off(row, (hoff >> 3) + 2 * s + h));
  bf16x8 v = *reinterpret_cast<const bf16x8*>(img + rm_off(row, hoff >> 3));
  return h == 0 ? v : zero8();
}

// A operand (rows = d, k = 16 streamed rows from c0 + 16 s) from a transposed image, with k in the
// order of an accumulator's rows: element j of lane half h is streamed row 16 s + 8 (j>>2) + 4 h + (j&3).
template <int HD>
__device__ __forceinline__ bf16x8 afrag_tr(const unsigned char* img, int c0, int s, int lane, int hoff) {
  const int r = lane & 31, h = lane >> 5;
  const int d = HD == 32 ? r : (r & 7);
  const unsigned char* p = img + (hoff + d) * TSTR + (c0 + 16 * s + 4 * h) * 2;
  s16x4 lo = *reinterpret_cast<const s16x4*>(p);
  s16x4 hi = *reinterpret_cast<const s16x4*>(p + 16);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  bf16x8 o = __builtin_bit_cast(bf16x8, v);
  if (HD == 8 && r >= 8) o = zero8();
  return o;
}

// The same operand from a ROW-MAJOR image through transposing LDS reads (ds_read_b64_tr_b16: within 16 lanes, lane j
// addresses 4 columns 4 (j & 3).. of row j >> 2, lane i receives column i of those 4 rows), so no transposed image has
// to be written.  head_dim 8: only lanes 0..7 of a half keep their result, the other lanes' addresses stay inside the
// head's own 16 bytes.  The lane's two byte offsets inside a 16-row group are fixed for a kernel's life (the swizzle
// key of a row repeats every 16 rows): tr_offsets() once, then one add per read.
struct TrOff { int lo, hi; };
template <int HD>
__device__ __forceinline__ TrOff tr_offsets(int lane, int hoff) {
  const int i = lane & 15, h = lane >> 5;
  const int rl = 4 * h + (i >> 2);
  const int col = HD == 32 ? hoff + 16 * ((lane >> 4) & 1) + 4 * (i & 3) : hoff + 4 * (i & 1);
  const int key = (rl >> 1) & 7, ch = col >> 3, within = (col & 7) * 2;
  TrOff o;
  o.lo = rl * 128 + ((ch ^ key) << 4) + within;
  o.hi = (rl + 8) * 128 + ((ch ^ key ^ 4) << 4) + within;   // rows 8 further: key ^ 4
  return o;
}
template <int HD>
__device__ __forceinline__ bf16x8 afrag_trd(const unsigned char* img, int row0, const TrOff& off, int lane) {   // row0: multiple of 16
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + row0 * 128 + off.lo));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + row0 * 128 + off.hi));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  bf16x8 o = __builtin_bit_cast(bf16x8, v);
  if (HD == 8 && (lane & 31) >= 8) o = zero8();
  return o;
}
// Both operands of a 32-row tile (rows row0.. and row0 + 16..) issued as inline assembly, for loops that keep an LDS-DMA
// in flight: the compiler puts s_waitcnt vmcnt(0) in front of every transposing read it can see while a DMA is pending
// (which would make each wave wait for the NEXT chunk before it consumes this one); here the reads are waited for by
// count.  No LDS store may sit between a barrier and a call of this.
template <int HD>
__device__ __forceinline__ void afrag_trd_pair(bf16x8& f0, bf16x8& f1, const unsigned char* img, int row0, const TrOff& off, int lane) {
  const uint32_t base = (uint32_t)(uintptr_t)(LDS_AS const unsigned char*)(img + row0 * 128);
  const uint32_t a0 = base + off.lo, a1 = base + off.hi;
  s16x4 l0, h0, l1, h1;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(l0) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h0) : "v"(a1));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(l1) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(h1) : "v"(a1));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  f0 = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7));
  f1 = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7));
  if (HD == 8 && (lane & 31) >= 8) { f0 = zero8(); f1 = zero8(); }
}
// B operand of owner row `row` taken from a row-major image (what load_owner() reads from memory)
template <int HD>
__device__ __forceinline__ void owner_from_img(bf16x8 (&f)[Cfg<HD>::KS], const unsigned char* img, int row, int h, int hoff, bool valid) {
  if (HD == 32) {
#pragma unroll
    for (int s = 0; s < Cfg<HD>::KS; ++s)
      f[s] = valid ? *reinterpret_cast<const bf16x8*>(img + rm_off(row, (hoff >> 3) + 2 * s + h)) : zero8();
  } else {
    f[0] = (valid && h == 0) ? *reinterpret_cast<const bf16x8*>(img + rm_off(row, hoff >> 3)) : zero8();
  }
}

// B operand of an owner row kept in registers: lane (r, h) holds row r, k = 16 s + 8 h ..
template <int HD>
__device__ __forceinline__ void load_owner(bf16x8 (&f)[Cfg<HD>::KS], const bf16* rowptr, int h, bool valid) {
  if (HD == 32) {
#pragma unroll
    for (int s = 0; s < Cfg<HD>::KS; ++s)
      f[s] = valid ? *reinterpret_cast<const bf16x8*>(rowptr + 16 * s + 8 * h) : zero8();
  } else {
    f[0] = (valid && h == 0) ? *reinterpret_cast<const bf16x8*>(rowptr) : zero8();
  }
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
  return o;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// store a d-major accumulator (lane = owner row, registers = d) as a bf16 row slice
template <int HD>
__device__ __forceinline__ void store_rows(bf16* rowptr, const f32x16& acc, float mul, int h) {
  constexpr int G = HD == 32 ? 4 : 1;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[4 * g + j] * mul);
    *reinterpret_cast<bf16x4*>(rowptr + 8 * g + 4 * h) = o;
  }
}

__device__ __forceinline__ bf16x8 ldrow8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ float dot8(const bf16x8& a, const bf16x8& b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)b[j];
  return s;
}
__device__ __forceinline__ void store_row8(bf16* p, const float* v, float mul) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)(v[j] * mul);
  *reinterpret_cast<bf16x8*>(p) = o;
}
// delta = dO . O of one query row and head
template <int HD> __device__ __forceinline__ float row_delta(const AttnProb& pr, size_t row, int hh) {
  const bf16* dor = pr.dO + row * pr.lddo + hh * HD;
  const bf16* orow = pr.O + row * pr.ldo + hh * HD;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HD / 8; ++c) s += dot8(ldrow8(dor + 8 * c), ldrow8(orow + 8 * c));
  return s;
}
// dropout keep flag of (query-row seed, key): the same hash the tile code evaluates pairwise
// attn_pair_bits() with the 24-bit multiply written as an instruction: where a lane hashes once per score (the key-owner
// tiles) the compiler otherwise spends an AND per hash on clearing the upper bits the instruction ignores anyway
__device__ __forceinline__ uint32_t attn_pair_bits_i(uint32_t row_seed, uint32_t pair_term) {
  uint32_t t = row_seed + pair_term;
  t ^= t >> 16;
  asm("v_mul_u32_u24 %0, 0x9e3779, %1" : "=v"(t) : "v"(t));
  static_assert(ATTN_PAIR_K24 == 0x9E3779U, "constant of the instruction above");
  t ^= t >> 16;
  return t;
}
__device__ __forceinline__ bool keep_of(uint32_t row_seed, int key, uint32_t thr) {
  const uint32_t hb = attn_pair_bits(row_seed, (uint32_t)(key >> 1) * ATTN_PAIR_MUL);
  return ((hb >> ((key & 1) * 16)) & 0xFFFFu) >= thr;
}

// ------------------------------------------------------------------------------------------
// Tail rows (head_dim 8, not causal).  The modality sequences are 129 (or 66) tokens long: 1 or 2 rows
// beyond a multiple of 32 would cost a whole 32-row tile on either side of the score matrix (25 tiles instead
// of 16 for 129 x 129).  Those rows never enter the MFMA tiling:
//   * a tail row of the STREAMED side is folded in by every owner lane with plain FMAs (one score per lane);
//   * a tail row of the OWNER side is computed by one wave with a lane per streamed row and wave reductions.
// ------------------------------------------------------------------------------------------
struct TailCtx {
  float scale, scale2, dscale;
  uint32_t thr;
  bool dropping;
};

// forward (BWD = false: O row and lse) or dQ pass (BWD = true: dQ row and delta) of query row q of head hh
template <bool BWD>
__device__ __forceinline__ void q_owner_tail(const AttnProb& pr, const TailCtx& tc, int b, int hh, int q, int lane) {
  constexpr int HD = 8;
  const int H = pr.H, Tq = pr.Tq, Sk = pr.Sk;
  const int kvr = (!BWD && pr.kv_rows > 0) ? pr.kv_rows : Sk;
  const size_t stat = (size_t)(b * H + hh) * Tq + q;
  const bf16x8 qrow = ldrow8(pr.Q + (size_t)(b * Tq + q) * pr.ldq + hh * HD);
  const uint32_t rseed = attn_row_seed(pr.seed, (uint32_t)stat);
  const bf16* kbase = pr.K + (size_t)b * kvr * pr.ldk + hh * HD;
  const bf16* vbase = pr.V + (size_t)b * kvr * pr.ldv + hh * HD;
  float acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) acc[d] = 0.f;
  if (!BWD) {
    float m = NEG_BIG, l = 0.f;
    for (int key = lane; key < Sk; key += 64) {
      const bf16x8 kt = ldrow8(kbase + (size_t)key * pr.ldk), vt = ldrow8(vbase + (size_t)key * pr.ldv);
      const float bt = pr.kbias ? pr.kbias[(size_t)b * Sk + key] * LOG2E : 0.f;
      const float sc = dot8(qrow, kt) * tc.scale2 + bt;
      const float mn = fmaxf(m, sc);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      float p = __builtin_amdgcn_exp2f(sc - mn);
      l = l * alpha + p;
      m = mn;
      if (tc.dropping) p = keep_of(rseed, key, tc.thr) ? p * tc.dscale : 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[d] = acc[d] * alpha + p * (float)vt[d];
    }
    const float M = wave_max(m);
    const float f = __builtin_amdgcn_exp2f(m - M);
    float red = 0.f;   // lane 8 of a row keeps l, lanes 0..7 the output elements: one trip through the crossbar
    {
      const float a = row16_sum(l * f);
      red = (lane & 15) == 8 ? a : red;
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float a = row16_sum(acc[d] * f);
      red = (lane & 15) == d ? a : red;
    }
    red = xor32_sum(xor16_sum(red));
    l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, red), 8));
    if (lane < 8) pr.O[(size_t)(b * Tq + q) * pr.ldo + hh * HD + lane] = (bf16)(red * (l > 0.f ? 1.0f / l : 0.f));
    if (lane == 0) pr.lse[stat] = M + __builtin_amdgcn_logf(l);
  } else {
    const bf16x8 dorow = ldrow8(pr.dO + (size_t)(b * Tq + q) * pr.lddo + hh * HD);
    const float delta = dot8(dorow, ldrow8(pr.O + (size_t)(b * Tq + q) * pr.ldo + hh * HD));
    const float lse2 = pr.lse[stat];
    for (int key = lane; key < Sk; key += 64) {
      const bf16x8 kt = ldrow8(kbase + (size_t)key * pr.ldk), vt = ldrow8(vbase + (size_t)key * pr.ldv);
      const float bt = pr.kbias ? pr.kbias[(size_t)b * Sk + key] * LOG2E : 0.f;
      const float p = __builtin_amdgcn_exp2f(dot8(qrow, kt) * tc.scale2 + bt - lse2);
      float dp = dot8(dorow, vt);
      if (tc.dropping) dp = keep_of(rseed, key, tc.thr) ? dp * tc.dscale : 0.f;
      const float ds = p * (dp - delta) * tc.scale;
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[d] += ds * (float)kt[d];
    }
    float red = 0.f;   // as in kv_owner_tail(): one trip through the crossbar for the eight sums
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float a = row16_sum(acc[d]);
      red = (lane & 15) == d ? a : red;
    }
    red = xor32_sum(xor16_sum(red));
    if (lane < 8) pr.dQ[(size_t)(b * Tq + q) * pr.lddq + hh * HD + lane] = (bf16)red;
    if (lane == 0) pr.delta[stat] = delta;
  }
}

// dK and dV rows of key `key` of head hh: a lane per query (INLINE_DELTA: delta = dO . O computed here instead of read)
template <bool INLINE_DELTA = false>
__device__ __forceinline__ void kv_owner_tail(const AttnProb& pr, const TailCtx& tc, int b, int hh, int key, int lane) {
  constexpr int HD = 8;
  const int H = pr.H, Tq = pr.Tq, Sk = pr.Sk;
  const bf16x8 kt = ldrow8(pr.K + (size_t)(b * Sk + key) * pr.ldk + hh * HD);
  const bf16x8 vt = ldrow8(pr.V + (size_t)(b * Sk + key) * pr.ldv + hh * HD);
  const float bt = pr.kbias ? pr.kbias[(size_t)b * Sk + key] * LOG2E : 0.f;
  float dk[8], dv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) dk[d] = dv[d] = 0.f;
  for (int q = lane; q < Tq; q += 64) {
    const size_t stat = (size_t)(b * H + hh) * Tq + q;
    const bf16x8 qrow = ldrow8(pr.Q + (size_t)(b * Tq + q) * pr.ldq + hh * HD);
    const bf16x8 dorow = ldrow8(pr.dO + (size_t)(b * Tq + q) * pr.lddo + hh * HD);
    const float p = __builtin_amdgcn_exp2f(dot8(qrow, kt) * tc.scale2 + bt - pr.lse[stat]);
    float dp = dot8(dorow, vt), pk = p;
    if (tc.dropping) {
      const bool keep = keep_of(attn_row_seed(pr.seed, (uint32_t)stat), key, tc.thr);
      dp = keep ? dp * tc.dscale : 0.f;
      pk = keep ? p * tc.dscale : 0.f;
    }
    const float dlt = INLINE_DELTA ? dot8(dorow, ldrow8(pr.O + (size_t)(b * Tq + q) * pr.ldo + hh * HD)) : pr.delta[stat];
    const float ds = p * (dp - dlt) * tc.scale;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      dv[d] += pk * (float)dorow[d];
      dk[d] += ds * (float)qrow[d];
    }
  }
  // sixteen wave-wide sums: inside the rows of 16 lanes each by itself, then lane i of every row keeps sum i and the
  // four rows are added through the LDS crossbar ONCE (2 exchanges instead of 32); lanes 0..15 store one element each
  float red = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const float a = row16_sum(dk[d]), c = row16_sum(dv[d]);
    red = (lane & 15) == d ? a : red;
    red = (lane & 15) == 8 + d ? c : red;
  }
  red = xor32_sum(xor16_sum(red));
  if (lane < 8) pr.dK[(size_t)(b * Sk + key) * pr.lddk + hh * HD + lane] = (bf16)red;
  else if (lane < 16) pr.dV[(size_t)(b * Sk + key) * pr.lddv + hh * HD + lane - 8] = (bf16)red;
}

struct BlockCoord { int g, b, hg, blk; };   // hg = head-group index
template <int HD>
__device__ __forceinline__ BlockCoord locate(const AttnBatch& batch, int block) {
  int g = 0;
#pragma unroll
  for (int i = 1; i < ATTN_MAX_GROUPS; ++i)
    if (i < batch.n && block >= batch.p[i].block_begin) g = i;
  const AttnProb& pr = batch.p[g];
  // Block index major, (batch, head group) minor: the partly filled last block of every sequence is dispatched
  // last (a short final round instead of a full one), and the blocks that share one (batch, head group)'s K/V are
  // B * ngroups ids apart, i.e. on one XCD's L2 whenever that is a multiple of 8.
  int local = block - pr.block_begin;
  const int ngroups = pr.H / Cfg<HD>::HG;
  const int nbg = pr.B * ngroups;
  const int bg = local % nbg;
  BlockCoord c;
  c.g = g;
  c.blk = local / nbg;
  c.b = bg / ngroups;
  c.hg = bg % ngroups;
  return c;
}

// ------------------------------------------------------------------------------------------
// forward (BWD = false) and dQ pass (BWD = true): owner = queries
// ------------------------------------------------------------------------------------------
template <int HD, bool BWD> constexpr int q_smem_bytes() {
  return 4 * Cfg<HD>::RM_BYTES + KC * 4;   // two stages of (K | V) row-major images, key bias
}
template <int HD, bool CAUSAL, bool BWD>
__device__ __forceinline__ void attn_q_body(const AttnBatch& batch, int block, unsigned char* smem) {
  typedef Cfg<HD> C;
  float* bias = reinterpret_cast<float*>(smem + 4 * C::RM_BYTES);

  const BlockCoord bc = locate<HD>(batch, block);
  const AttnProb& pr = batch.p[bc.g];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int hoff = (wave % C::HG) * HD;          // this wave's head inside the staged group
  const int gcol = bc.hg * GC;                   // first column of the head group
  const int b = bc.b, hh = bc.hg * C::HG + wave % C::HG, H = pr.H, Tq = pr.Tq, Sk = pr.Sk;
  constexpr bool TAILS = HD == 8 && !CAUSAL;
  const int tail_o = TAILS ? pr.tail_o : 0, tail_s = TAILS ? pr.tail_s : 0;
  const int Tq_main = Tq - tail_o, Sk_main = Sk - tail_s;   // rows covered by MFMA tiles
  const int q0 = (bc.blk * C::TPB + wave / C::HG) * 32;
  const bool active = q0 < Tq_main;
  const int q = q0 + r;
  const bool qvalid = active && q < Tq_main;
  const float scale = rsqrtf((float)HD);
  const float scale2 = scale * LOG2E;
  const uint32_t thr = drop_threshold(batch.drop_p);
  const float dscale = batch.drop_p > 0.f ? 1.0f / (1.0f - batch.drop_p) : 1.0f;
  const float inv_dscale = batch.drop_p > 0.f ? 1.0f - batch.drop_p : 1.0f, inv_scale2 = 1.0f / scale2;
  const bool dropping = batch.drop_p > 0.f;
  if (TAILS && tail_o > 0 && bc.blk == 0) {
    // the first workgroup of every (batch, head group) also serves the tail query rows, one head per wave
    const TailCtx tc = {scale, scale2, dscale, thr, dropping};
    for (int t = 0; t < tail_o; ++t) q_owner_tail<BWD>(pr, tc, b, hh, Tq_main + t, lane);
  }
  if (TAILS && Tq_main == 0) return;   // uniform over the workgroup: nothing left for the tiled path
  const uint32_t rseed = attn_row_seed(pr.seed, (uint32_t)((b * H + hh) * Tq + (qvalid ? q : 0)));
  const TrOff troff = tr_offsets<HD>(lane, hoff);

  bf16x8 qf[C::KS], dof[C::KS];
  load_owner<HD>(qf, pr.Q + (size_t)(b * Tq + (qvalid ? q : 0)) * pr.ldq + hh * HD, h, qvalid);
  float lse2 = 0.f, delta = 0.f;
  if (BWD) {
    load_owner<HD>(dof, pr.dO + (size_t)(b * Tq + (qvalid ? q : 0)) * pr.lddo + hh * HD, h, qvalid);
    bf16x8 of[C::KS];
    load_owner<HD>(of, pr.O + (size_t)(b * Tq + (qvalid ? q : 0)) * pr.ldo + hh * HD, h, qvalid);
#pragma unroll
    for (int s = 0; s < C::KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)dof[s][j] * (float)of[s][j];
    delta = xor32_sum(delta);
    if (qvalid) {
      lse2 = pr.lse[(size_t)(b * H + hh) * Tq + q];
      if (h == 0) pr.delta[(size_t)(b * H + hh) * Tq + q] = delta;
    }
    // Constant factors leave the per-score arithmetic: with keep in {0, 1} and c = 1/(1-p),
    // dS = P (keep c dP - delta) scale = [P (keep dP - delta / c)] (c scale); the bracket is what the tiles accumulate,
    // the factor is applied once to the output row.
    delta *= inv_dscale;
    lse2 = -lse2;
  }

  float m = NEG_BIG, l = 0.f;
  f32x16 oacc = zero16();

  const int kvr = (!BWD && pr.kv_rows > 0) ? pr.kv_rows : Sk;
  const bf16* kbatch = pr.K + (size_t)b * kvr * pr.ldk + gcol;
  const bf16* vbatch = pr.V + (size_t)b * kvr * pr.ldv + gcol;
  // The K and V chunks arrive by LDS-DMA (row-major images only; the transposed operand comes from transposing LDS
  // reads) into one of two stages: the next chunk is in flight while this one is consumed.
  // (`t` is the thread index: the call inside the chunk loop passes a copy the compiler cannot see through, so that
  // its address arithmetic is redone there instead of being hoisted out of the loop into long-lived registers)
  float bst = -INFINITY;
  auto fetch = [&](int c0, int stg, int t) {
    const int nvalid = min(KC, Sk_main - c0);
    // the bias load goes first: it is waited for by count, and a wait for a load younger than the DMA would wait for
    // the DMA as well
    bst = -INFINITY;
    if (t < nvalid) bst = pr.kbias ? pr.kbias[(size_t)b * Sk + c0 + t] : 0.f;
    stage_dma(smem + stg * 2 * C::RM_BYTES, kbatch + (size_t)c0 * pr.ldk, pr.ldk, nvalid, t);
    stage_dma(smem + stg * 2 * C::RM_BYTES + C::RM_BYTES, vbatch + (size_t)c0 * pr.ldv, pr.ldv, nvalid, t);
  };
  if (Sk_main > 0) fetch(0, 0, tid);
  int stg = 0;
  for (int c0 = 0; c0 < Sk_main; c0 += KC, stg ^= 1) {
    const int nvalid = min(KC, Sk_main - c0);
    if (c0 > 0) __syncthreads();                  // every wave is done with the previous chunk's bias (and the other stage)
    if (tid < KC) bias[tid] = bst * (LOG2E * inv_scale2);   // in score units: the accumulator of S starts from it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c0 + KC < Sk_main) {
      int t2 = tid;
      asm volatile("" : "+v"(t2));
      fetch(c0 + KC, stg ^ 1, t2);
    }
    if (!active) continue;
    const unsigned char* Kimg = smem + stg * 2 * C::RM_BYTES;
    const unsigned char* Vimg = Kimg + C::RM_BYTES;
    const unsigned char* Timg = BWD ? Kimg : Vimg;   // transposed operand: K^T for dQ, V^T for the forward
    const int ntile = (nvalid + 31) >> 5;
    for (int t = 0; t < ntile; ++t) {
      const int lk0 = t * 32;
      if (CAUSAL && c0 + lk0 > q0 + 31) break;
      // S accumulates on top of the key bias (read from LDS straight into the accumulator registers): the score in
      // log2 units is sv * scale2, one FMA together with the subtraction of the row's maximum / log-sum-exp
      f32x16 sacc;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 bb = *reinterpret_cast<const f32x4*>(bias + lk0 + 8 * g4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc[4 * g4 + j] = bb[j];
      }
#pragma unroll
      for (int s = 0; s < C::KS; ++s)
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_rm<HD>(Kimg, lk0 + r, s, h, hoff), qf[s], sacc, 0, 0, 0);
      float sv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = sacc[i];
        if (CAUSAL && (c0 + lk0 + acc_row(i, h)) > q) v = -INFINITY;
        sv[i] = v;
      }
      float pv[16];
      if (!BWD) {
        float tmax = sv[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, sv[i]);
        tmax = xor32_max(tmax) * scale2;
        const float mn = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { pv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], scale2, -mn)); ps += pv[i]; }
        l = l * alpha + ps;
#pragma unroll
        for (int i = 0; i < C::OREGS; ++i) oacc[i] *= alpha;
        if (dropping) {
          const uint32_t tbase = rseed + (uint32_t)((c0 + lk0) >> 1) * ATTN_PAIR_MUL + (uint32_t)(2 * h) * ATTN_PAIR_MUL;
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            uint32_t hb = attn_pair_bits(tbase, (uint32_t)(acc_row(i, 0) >> 1) * ATTN_PAIR_MUL);
            pv[i] = ((hb & 0xFFFFu) >= thr) ? pv[i] : 0.f;       // 1/(1-p) is applied once, to the output row
            pv[i + 1] = ((hb >> 16) >= thr) ? pv[i + 1] : 0.f;
          }
        }
      } else {
        f32x16 dpacc = zero16();
#pragma unroll
        for (int s = 0; s < C::KS; ++s)
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_rm<HD>(Vimg, lk0 + r, s, h, hoff), dof[s], dpacc, 0, 0, 0);
        const uint32_t tbase = rseed + (uint32_t)((c0 + lk0) >> 1) * ATTN_PAIR_MUL + (uint32_t)(2 * h) * ATTN_PAIR_MUL;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          float p0 = __builtin_amdgcn_exp2f(fmaf(sv[i], scale2, lse2)), p1 = __builtin_amdgcn_exp2f(fmaf(sv[i + 1], scale2, lse2));
          // no branch on `dropping` inside the tile (it would fence the instruction schedule at every pair): without
          // dropout the threshold is 0 and every comparison keeps
          const uint32_t hb = attn_pair_bits(tbase, (uint32_t)(acc_row(i, 0) >> 1) * ATTN_PAIR_MUL);
          const float d0 = ((hb & 0xFFFFu) >= thr) ? dpacc[i] : 0.f;
          const float d1 = ((hb >> 16) >= thr) ? dpacc[i + 1] : 0.f;
          pv[i] = p0 * (d0 - delta);
          pv[i + 1] = p1 * (d1 - delta);
        }
      }
      const bf16x8 pf0 = pack8(pv), pf1 = pack8(pv + 8);
      bf16x8 tf0, tf1;
      afrag_trd_pair<HD>(tf0, tf1, Timg, lk0, troff, lane);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf0, pf0, oacc, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf1, pf1, oacc, 0, 0, 0);
    }
  }
  if (!qvalid) return;
  if (TAILS && tail_s > 0) {
    // tail keys: one score per query, both lane halves of a query compute it, each updates its own output half
    const bf16x8 qrow = ldrow8(pr.Q + (size_t)(b * Tq + q) * pr.ldq + hh * HD);
    bf16x8 dorow = qrow;
    if (BWD) dorow = ldrow8(pr.dO + (size_t)(b * Tq + q) * pr.lddo + hh * HD);
    for (int t = 0; t < tail_s; ++t) {
      const int key = Sk_main + t;
      const bf16x8 kt = ldrow8(pr.K + ((size_t)b * kvr + key) * pr.ldk + hh * HD);
      const bf16x8 vt = ldrow8(pr.V + ((size_t)b * kvr + key) * pr.ldv + hh * HD);
      const float bt = pr.kbias ? pr.kbias[(size_t)b * Sk + key] * LOG2E : 0.f;
      const float sc = dot8(qrow, kt) * scale2 + bt;
      const bool keep = !dropping || keep_of(rseed, key, thr);
      if (!BWD) {
        const float mn = fmaxf(m, sc);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        float p = __builtin_amdgcn_exp2f(sc - mn);
        l = l * alpha + (h == 0 ? p : 0.f);
        m = mn;
        p = keep ? p : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[j] = oacc[j] * alpha + p * (float)(h ? vt[4 + j] : vt[j]);
      } else {
        const float p = __builtin_amdgcn_exp2f(sc + lse2);
        float dp = dot8(dorow, vt);
        dp = keep ? dp : 0.f;
        const float ds = p * (dp - delta);
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[j] += ds * (float)(h ? kt[4 + j] : kt[j]);
      }
    }
  }
  if (!BWD) {
    l = xor32_sum(l);
    const float inv = l > 0.f ? dscale / l : 0.f;
    store_rows<HD>(pr.O + (size_t)(b * Tq + q) * pr.ldo + hh * HD, oacc, inv, h);
    if (h == 0) pr.lse[(size_t)(b * H + hh) * Tq + q] = m + __builtin_amdgcn_logf(l);
  } else {
    store_rows<HD>(pr.dQ + (size_t)(b * Tq + q) * pr.lddq + hh * HD, oacc, scale * dscale, h);
  }
}

#ifdef MMST_WG_LOG
}  // namespace
// timing build: one record per workgroup (start, end in 100 MHz ticks; kind << 32 | block; hardware ids)
__device__ unsigned int g_wglog3_n;
__device__ unsigned long long g_wglog3[4 * 32768];
extern "C" int mmst_debug_wglog3(unsigned long long* out, unsigned int* n) {
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_wglog3_n), 4) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wglog3), sizeof(unsigned long long) * 4 * 32768) != hipSuccess) return -1;
  return 0;
}
namespace {
#endif
#ifdef MMST_WG_LOG
#define ATTN_WGLOG_END(kind)                                                                                  \
  if (threadIdx.x == 0) {                                                                                    \
    const unsigned int ix = atomicAdd(&g_wglog3_n, 1u) & 32767u;                                             \
    g_wglog3[4 * ix] = wg_t0;                                                                                \
    g_wglog3[4 * ix + 1] = wall_clock64();                                                                   \
    g_wglog3[4 * ix + 2] = ((unsigned long long)(kind) << 32) | blockIdx.x;                                  \
    g_wglog3[4 * ix + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
  }
#endif
template <int HD, bool CAUSAL, bool BWD>
__global__ void attn_q_kernel(const AttnBatch batch) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[q_smem_bytes<HD, BWD>()];
#ifdef MMST_WG_LOG
  const unsigned long long wg_t0 = wall_clock64();
#endif
  attn_q_body<HD, CAUSAL, BWD>(batch, (int)blockIdx.x, smem);
#ifdef MMST_WG_LOG
  ATTN_WGLOG_END(BWD ? 8 : (HD == 32 ? 6 : 7));
#endif
}

// ------------------------------------------------------------------------------------------
// dK / dV pass: owner = keys, queries streamed.  OWN_DELTA: delta = dO . O is computed while a chunk is staged instead
// of read from what the dQ pass wrote (the two passes are then independent of each other).
// ------------------------------------------------------------------------------------------
template <int HD> constexpr int kv_smem_bytes() {
  return 2 * Cfg<HD>::RM_BYTES + 2 * Cfg<HD>::TR_BYTES + 3 * Cfg<HD>::HG * KC * 4;
}
template <int HD, bool CAUSAL, bool OWN_DELTA>
__device__ __forceinline__ void attn_kv_body(const AttnBatch& batch, int block, unsigned char* smem) {
  typedef Cfg<HD> C;
  unsigned char* Qimg = smem;
  unsigned char* Dimg = smem + C::RM_BYTES;
  unsigned char* Qt = smem + 2 * C::RM_BYTES;
  unsigned char* Dt = Qt + C::TR_BYTES;
  float* lse_all = reinterpret_cast<float*>(Dt + C::TR_BYTES);   // [HG][KC]
  float* del_all = lse_all + C::HG * KC;
  uint32_t* rs_all = reinterpret_cast<uint32_t*>(del_all + C::HG * KC);   // per-row dropout seeds

  const BlockCoord bc = locate<HD>(batch, block);
  const AttnProb& pr = batch.p[bc.g];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int hl = wave % C::HG, hoff = hl * HD, gcol = bc.hg * GC;
  const int b = bc.b, hh = bc.hg * C::HG + hl, H = pr.H, Tq = pr.Tq, Sk = pr.Sk;
  const float* lse_s = lse_all + hl * KC;
  const float* del_s = del_all + hl * KC;
  const uint32_t* rs_s = rs_all + hl * KC;
  constexpr bool TAILS = HD == 8 && !CAUSAL;
  const int tail_o = TAILS ? pr.tail_o : 0, tail_s = TAILS ? pr.tail_s : 0;
  const int Sk_main = Sk - tail_o, Tq_main = Tq - tail_s;   // rows covered by MFMA tiles
  const int k0 = (bc.blk * C::TPB + wave / C::HG) * 32;
  const bool active = k0 < Sk_main;
  const int key = k0 + r;
  const bool kvalid = active && key < Sk_main;
  const float scale = rsqrtf((float)HD);
  const float scale2 = scale * LOG2E;
  const uint32_t thr = drop_threshold(batch.drop_p);
  const float dscale = batch.drop_p > 0.f ? 1.0f / (1.0f - batch.drop_p) : 1.0f;
  const float inv_dscale = batch.drop_p > 0.f ? 1.0f - batch.drop_p : 1.0f, inv_scale2 = 1.0f / scale2;
  const bool dropping = batch.drop_p > 0.f;
  if (TAILS && tail_o > 0 && bc.blk == 0) {
    const TailCtx tc = {scale, scale2, dscale, thr, dropping};
    for (int t = 0; t < tail_o; ++t) kv_owner_tail<OWN_DELTA>(pr, tc, b, hh, Sk_main + t, lane);
  }
  if (TAILS && Sk_main == 0) return;
  const uint32_t pair_term = (uint32_t)(key >> 1) * ATTN_PAIR_MUL;
  const uint32_t pair_shift = (key & 1) ? 16u : 0u;

  bf16x8 kf[C::KS], vf[C::KS];
  load_owner<HD>(kf, pr.K + (size_t)(b * Sk + (kvalid ? key : 0)) * pr.ldk + hh * HD, h, kvalid);
  load_owner<HD>(vf, pr.V + (size_t)(b * Sk + (kvalid ? key : 0)) * pr.ldv + hh * HD, h, kvalid);
  float bias2 = -INFINITY;
  if (kvalid) bias2 = pr.kbias ? pr.kbias[(size_t)b * Sk + key] * LOG2E : 0.f;

  f32x16 dkacc = zero16(), dvacc = zero16();

  for (int c0 = 0; c0 < Tq_main; c0 += KC) {
    const int nvalid = min(KC, Tq_main - c0);
    __syncthreads();
    stage<true, true>(Qimg, Qt, pr.Q + (size_t)(b * Tq + c0) * pr.ldq + gcol, pr.ldq, nvalid, tid, nthr);
    stage<true, true>(Dimg, Dt, pr.dO + (size_t)(b * Tq + c0) * pr.lddo + gcol, pr.lddo, nvalid, tid, nthr);
    for (int j = tid; j < C::HG * KC; j += nthr) {
      const int hj = j / KC, i = j % KC;
      float lv = 1.0e30f, dv = 0.f;
      if (i < nvalid) {
        lv = pr.lse[(size_t)(b * H + bc.hg * C::HG + hj) * Tq + c0 + i];
        dv = OWN_DELTA ? row_delta<HD>(pr, (size_t)b * Tq + c0 + i, bc.hg * C::HG + hj)
                       : pr.delta[(size_t)(b * H + bc.hg * C::HG + hj) * Tq + c0 + i];
      }
      // Staged in the form the tiles use (constant factors applied once to the output rows, see the dQ pass):
      // -lse in score units is the start value of the S accumulator, delta comes divided by 1/(1-p)
      lse_all[j] = -lv * inv_scale2;
      del_all[j] = dv * inv_dscale;
      if (dropping) rs_all[j] = attn_row_seed(pr.seed, (uint32_t)((b * H + bc.hg * C::HG + hj) * Tq + c0 + i));
    }
    __syncthreads();
    if (!active) continue;
    const int ntile = (nvalid + 31) >> 5;
    for (int t = 0; t < ntile; ++t) {
      const int lq0 = t * 32;
      if (CAUSAL && c0 + lq0 + 31 < k0) continue;  // every query of the tile precedes every key
      f32x16 sacc, dpacc = zero16();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 ll = *reinterpret_cast<const f32x4*>(lse_s + lq0 + 8 * g4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc[4 * g4 + j] = ll[j];
      }
#pragma unroll
      for (int s = 0; s < C::KS; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_rm<HD>(Qimg, lq0 + r, s, h, hoff), kf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_rm<HD>(Dimg, lq0 + r, s, h, hoff), vf[s], dpacc, 0, 0, 0);
      }
      float pd[16], ds[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 dd = *reinterpret_cast<const f32x4*>(del_s + lq0 + 8 * g4 + 4 * h);
        const uint4 rr = *reinterpret_cast<const uint4*>(rs_s + lq0 + 8 * g4 + 4 * h);
        const uint32_t rsv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * g4 + j;
          const int qg = c0 + lq0 + 8 * g4 + 4 * h + j;
          float v = fmaf(sacc[i], scale2, bias2);
          if (CAUSAL && key > qg) v = -INFINITY;
          const float p = __builtin_amdgcn_exp2f(v);
          // (no branch on `dropping` here: threshold 0 keeps everything, whatever the seed words hold)
          // (head_dim 32: the instruction form costs that kernel registers it does not have)
          const uint32_t hb = HD == 8 ? attn_pair_bits_i(rsv[j], pair_term) : attn_pair_bits(rsv[j], pair_term);
          const float pk = ((hb >> pair_shift) & 0xFFFFu) >= thr ? p : 0.f;
          pd[i] = pk;
          ds[i] = pk * dpacc[i] - p * dd[j];
        }
        if (HD == 8) __builtin_amdgcn_sched_barrier(0);   // four scores at a time: hoisting every hash of the tile spills registers
      }
      const bf16x8 pf0 = pack8(pd), pf1 = pack8(pd + 8), sf0 = pack8(ds), sf1 = pack8(ds + 8);
      dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_tr<HD>(Dt, lq0, 0, lane, hoff), pf0, dvacc, 0, 0, 0);
      dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_tr<HD>(Dt, lq0, 1, lane, hoff), pf1, dvacc, 0, 0, 0);
      dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_tr<HD>(Qt, lq0, 0, lane, hoff), sf0, dkacc, 0, 0, 0);
      dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_tr<HD>(Qt, lq0, 1, lane, hoff), sf1, dkacc, 0, 0, 0);
    }
  }
  if (!kvalid) return;
  if (TAILS && tail_s > 0) {
    // tail queries: one score per key lane pair
    const bf16x8 krow = ldrow8(pr.K + (size_t)(b * Sk + key) * pr.ldk + hh * HD);
    const bf16x8 vrow = ldrow8(pr.V + (size_t)(b * Sk + key) * pr.ldv + hh * HD);
    for (int t = 0; t < tail_s; ++t) {
      const int qt = Tq_main + t;
      const size_t stat = (size_t)(b * H + hh) * Tq + qt;
      const bf16x8 qrow = ldrow8(pr.Q + (size_t)(b * Tq + qt) * pr.ldq + hh * HD);
      const bf16x8 dorow = ldrow8(pr.dO + (size_t)(b * Tq + qt) * pr.lddo + hh * HD);
      const float p = __builtin_amdgcn_exp2f(dot8(krow, qrow) * scale2 + bias2 - pr.lse[stat]);
      const float dp = dot8(vrow, dorow);
      float pk = p;
      if (dropping) pk = keep_of(attn_row_seed(pr.seed, (uint32_t)stat), key, thr) ? p : 0.f;
      const float dlt = OWN_DELTA ? dot8(dorow, ldrow8(pr.O + (size_t)(b * Tq + qt) * pr.ldo + hh * HD)) : pr.delta[stat];
      const float ds = pk * dp - p * dlt * inv_dscale;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dvacc[j] += pk * (float)(h ? dorow[4 + j] : dorow[j]);
        dkacc[j] += ds * (float)(h ? qrow[4 + j] : qrow[j]);
      }
    }
  }
  store_rows<HD>(pr.dK + (size_t)(b * Sk + key) * pr.lddk + hh * HD, dkacc, scale * dscale, h);
  store_rows<HD>(pr.dV + (size_t)(b * Sk + key) * pr.lddv + hh * HD, dvacc, dscale, h);
}

template <int HD, bool CAUSAL>
__global__ void attn_kv_kernel(const AttnBatch batch) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kv_smem_bytes<HD>()];
  attn_kv_body<HD, CAUSAL, false>(batch, (int)blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// Self-attention backward of SHORT sequences: everything but the tail rows fits one 128-row chunk (the modality
// encoders: 129 and 66 tokens).  One workgroup per (batch, head group) stages Q, K, V and dO ONCE (LDS-DMA, row-major
// images only; transposed operands come from transposing LDS reads) and runs the dK/dV pass and the dQ pass of all its
// owner tiles from them: the per-workgroup fixed cost of the two-pass kernels (staging, per-row vectors, tail rows; 70 %
// of their time at these lengths) is paid once per (batch, head group) instead of once per owner tile and pass, and
// the tail rows are computed while the DMA is in flight.  The arithmetic is that of attn_kv_body / attn_q_body.
// ------------------------------------------------------------------------------------------
template <int HD> constexpr int self_smem_bytes() { return 4 * Cfg<HD>::RM_BYTES + 3 * Cfg<HD>::HG * KC * 4 + KC * 4; }

template <int HD>
__global__ void attn_bwd_self_kernel(const AttnBatch batch) {
  typedef Cfg<HD> C;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[self_smem_bytes<HD>()];
  unsigned char* Qimg = smem;
  unsigned char* Kimg = smem + C::RM_BYTES;
  unsigned char* Vimg = smem + 2 * C::RM_BYTES;
  unsigned char* Dimg = smem + 3 * C::RM_BYTES;
  float* lse_all = reinterpret_cast<float*>(smem + 4 * C::RM_BYTES);   // [HG][KC]: -lse in score units
  float* del_all = lse_all + C::HG * KC;                               // delta (1 - p)
  uint32_t* rs_all = reinterpret_cast<uint32_t*>(del_all + C::HG * KC);   // per-row dropout seeds
  float* bias = reinterpret_cast<float*>(rs_all + C::HG * KC);         // [KC]: key bias in score units

  const BlockCoord bc = locate<HD>(batch, (int)blockIdx.x);
  const AttnProb& pr = batch.p[bc.g];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int hl = wave % C::HG, hoff = hl * HD, gcol = bc.hg * GC;
  const int b = bc.b, hh = bc.hg * C::HG + hl, H = pr.H, S = pr.Tq;
  constexpr bool TAILS = HD == 8;
  const int tail = TAILS ? pr.tail_o : 0;
  const int S_main = S - tail;   // rows covered by MFMA tiles, <= KC
  const float scale = rsqrtf((float)HD);
  const float scale2 = scale * LOG2E;
  const uint32_t thr = drop_threshold(batch.drop_p);   // 0 without dropout: every comparison keeps
  const float dscale = batch.drop_p > 0.f ? 1.0f / (1.0f - batch.drop_p) : 1.0f;
  const float inv_dscale = batch.drop_p > 0.f ? 1.0f - batch.drop_p : 1.0f, inv_scale2 = 1.0f / scale2;
  const bool dropping = batch.drop_p > 0.f;

  // ---- staging: four images by DMA, the per-row vectors by ordinary loads, the tail rows meanwhile
  if (S_main > 0) {
    stage_dma(Qimg, pr.Q + (size_t)b * S * pr.ldq + gcol, pr.ldq, S_main, tid);
    stage_dma(Kimg, pr.K + (size_t)b * S * pr.ldk + gcol, pr.ldk, S_main, tid);
    stage_dma(Vimg, pr.V + (size_t)b * S * pr.ldv + gcol, pr.ldv, S_main, tid);
    stage_dma(Dimg, pr.dO + (size_t)b * S * pr.lddo + gcol, pr.lddo, S_main, tid);
  }
  for (int j = tid; j < C::HG * KC; j += NTHR) {
    const int hj = j / KC, i = j % KC;
    float lv = 1.0e30f, dv = 0.f;
    if (i < S_main) {
      lv = pr.lse[(size_t)(b * H + bc.hg * C::HG + hj) * S + i];
      dv = row_delta<HD>(pr, (size_t)b * S + i, bc.hg * C::HG + hj);
    }
    lse_all[j] = -lv * inv_scale2;
    del_all[j] = dv * inv_dscale;
    rs_all[j] = attn_row_seed(pr.seed, (uint32_t)((b * H + bc.hg * C::HG + hj) * S + i));
  }
  if (tid < KC) {
    float bv = -INFINITY;
    if (tid < S_main) bv = pr.kbias ? pr.kbias[(size_t)b * S + tid] * LOG2E * inv_scale2 : 0.f;
    bias[tid] = bv;
  }
  if (TAILS && tail > 0) {
    const TailCtx tc = {scale, scale2, dscale, thr, dropping};
    for (int t = 0; t < tail; ++t) kv_owner_tail<true>(pr, tc, b, hh, S_main + t, lane);
    for (int t = 0; t < tail; ++t) q_owner_tail<true>(pr, tc, b, hh, S_main + t, lane);
  }
  if (S_main == 0) return;   // uniform over the workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float* lse_s = lse_all + hl * KC;
  const float* del_s = del_all + hl * KC;
  const uint32_t* rs_s = rs_all + hl * KC;
  const TrOff troff = tr_offsets<HD>(lane, hoff);
  const int ntile = (S_main + 31) >> 5;

  // ---- dK / dV: owner = keys, queries streamed
  for (int ot = wave / C::HG; ot < ntile; ot += C::TPB) {
    const int key = ot * 32 + r;
    const bool kvalid = key < S_main;
    bf16x8 kf[C::KS], vf[C::KS];
    owner_from_img<HD>(kf, Kimg, key, h, hoff, kvalid);
    owner_from_img<HD>(vf, Vimg, key, h, hoff, kvalid);
    const float bias2 = kvalid ? bias[key] * scale2 : -INFINITY;
    const uint32_t pair_term = (uint32_t)(key >> 1) * ATTN_PAIR_MUL;
    const uint32_t pair_shift = (key & 1) ? 16u : 0u;
    f32x16 dkacc = zero16(), dvacc = zero16();
    for (int t = 0; t < ntile; ++t) {
      const int lq0 = t * 32;
      f32x16 sacc, dpacc = zero16();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 ll = *reinterpret_cast<const f32x4*>(lse_s + lq0 + 8 * g4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc[4 * g4 + j] = ll[j];
      }
#pragma unroll
      for (int s = 0; s < C::KS; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_rm<HD>(Qimg, lq0 + r, s, h, hoff), kf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_rm<HD>(Dimg, lq0 + r, s, h, hoff), vf[s], dpacc, 0, 0, 0);
      }
      float pd[16], ds[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 dd = *reinterpret_cast<const f32x4*>(del_s + lq0 + 8 * g4 + 4 * h);
        const uint4 rr = *reinterpret_cast<const uint4*>(rs_s + lq0 + 8 * g4 + 4 * h);
        const uint32_t rsv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * g4 + j;
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[i], scale2, bias2));
          // (head_dim 32: the instruction form costs that kernel registers it does not have)
          const uint32_t hb = HD == 8 ? attn_pair_bits_i(rsv[j], pair_term) : attn_pair_bits(rsv[j], pair_term);
          const float pk = ((hb >> pair_shift) & 0xFFFFu) >= thr ? p : 0.f;
          pd[i] = pk;
          ds[i] = pk * dpacc[i] - p * dd[j];
        }
        if (HD == 8) __builtin_amdgcn_sched_barrier(0);   // four scores at a time: hoisting every hash of the tile spills registers
      }
      const bf16x8 pf0 = pack8(pd), pf1 = pack8(pd + 8), sf0 = pack8(ds), sf1 = pack8(ds + 8);
      dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_trd<HD>(Dimg, lq0, troff, lane), pf0, dvacc, 0, 0, 0);
      dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_trd<HD>(Dimg, lq0 + 16, troff, lane), pf1, dvacc, 0, 0, 0);
      dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_trd<HD>(Qimg, lq0, troff, lane), sf0, dkacc, 0, 0, 0);
      dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_trd<HD>(Qimg, lq0 + 16, troff, lane), sf1, dkacc, 0, 0, 0);
    }
    if (kvalid) {
      if (TAILS && tail > 0) {
        // tail queries: one score per key lane pair
        const bf16x8 krow = ldrow8(pr.K + (size_t)(b * S + key) * pr.ldk + hh * HD);
        const bf16x8 vrow = ldrow8(pr.V + (size_t)(b * S + key) * pr.ldv + hh * HD);
        for (int t = 0; t < tail; ++t) {
          const int qt = S_main + t;
          const size_t stat = (size_t)(b * H + hh) * S + qt;
          const bf16x8 qrow = ldrow8(pr.Q + (size_t)(b * S + qt) * pr.ldq + hh * HD);
          const bf16x8 dorow = ldrow8(pr.dO + (size_t)(b * S + qt) * pr.lddo + hh * HD);
          const float p = __builtin_amdgcn_exp2f(dot8(krow, qrow) * scale2 + bias2 - pr.lse[stat]);
          const float dp = dot8(vrow, dorow);
          float pk = p;
          if (dropping) pk = keep_of(attn_row_seed(pr.seed, (uint32_t)stat), key, thr) ? p : 0.f;
          const float dlt = dot8(dorow, ldrow8(pr.O + (size_t)(b * S + qt) * pr.ldo + hh * HD));
          const float dsv = pk * dp - p * dlt * inv_dscale;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dvacc[j] += pk * (float)(h ? dorow[4 + j] : dorow[j]);
            dkacc[j] += dsv * (float)(h ? qrow[4 + j] : qrow[j]);
          }
        }
      }
      store_rows<HD>(pr.dK + (size_t)(b * S + key) * pr.lddk + hh * HD, dkacc, scale * dscale, h);
      store_rows<HD>(pr.dV + (size_t)(b * S + key) * pr.lddv + hh * HD, dvacc, dscale, h);
    }
  }

  // ---- dQ: owner = queries, keys streamed
  for (int ot = wave / C::HG; ot < ntile; ot += C::TPB) {
    const int q = ot * 32 + r;
    const bool qvalid = q < S_main;
    bf16x8 qf[C::KS], dof[C::KS];
    owner_from_img<HD>(qf, Qimg, q, h, hoff, qvalid);
    owner_from_img<HD>(dof, Dimg, q, h, hoff, qvalid);
    const float lse2 = qvalid ? lse_s[q] * scale2 : 0.f;   // -lse in log2 units
    const float delta = del_s[q];
    const uint32_t rseed = rs_s[q];
    f32x16 oacc = zero16();
    for (int t = 0; t < ntile; ++t) {
      const int lk0 = t * 32;
      f32x16 sacc, dpacc = zero16();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 bb = *reinterpret_cast<const f32x4*>(bias + lk0 + 8 * g4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc[4 * g4 + j] = bb[j];
      }
#pragma unroll
      for (int s = 0; s < C::KS; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_rm<HD>(Kimg, lk0 + r, s, h, hoff), qf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_rm<HD>(Vimg, lk0 + r, s, h, hoff), dof[s], dpacc, 0, 0, 0);
      }
      float pv[16];
      const uint32_t tbase = rseed + (uint32_t)(lk0 >> 1) * ATTN_PAIR_MUL + (uint32_t)(2 * h) * ATTN_PAIR_MUL;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[i], scale2, lse2)), p1 = __builtin_amdgcn_exp2f(fmaf(sacc[i + 1], scale2, lse2));
        const uint32_t hb = attn_pair_bits(tbase, (uint32_t)(acc_row(i, 0) >> 1) * ATTN_PAIR_MUL);
        const float d0 = ((hb & 0xFFFFu) >= thr) ? dpacc[i] : 0.f;
        const float d1 = ((hb >> 16) >= thr) ? dpacc[i + 1] : 0.f;
        pv[i] = p0 * (d0 - delta);
        pv[i + 1] = p1 * (d1 - delta);
      }
      const bf16x8 pf0 = pack8(pv), pf1 = pack8(pv + 8);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_trd<HD>(Kimg, lk0, troff, lane), pf0, oacc, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_trd<HD>(Kimg, lk0 + 16, troff, lane), pf1, oacc, 0, 0, 0);
    }
    if (qvalid) {
      if (TAILS && tail > 0) {
        // tail keys: one score per query, both lane halves of a query compute it, each updates its own output half
        const bf16x8 qrow = ldrow8(pr.Q + (size_t)(b * S + q) * pr.ldq + hh * HD);
        const bf16x8 dorow = ldrow8(pr.dO + (size_t)(b * S + q) * pr.lddo + hh * HD);
        for (int t = 0; t < tail; ++t) {
          const int key = S_main + t;
          const bf16x8 kt = ldrow8(pr.K + (size_t)(b * S + key) * pr.ldk + hh * HD);
          const bf16x8 vt = ldrow8(pr.V + (size_t)(b * S + key) * pr.ldv + hh * HD);
          const float bt = pr.kbias ? pr.kbias[(size_t)b * S + key] * LOG2E : 0.f;
          const float p = __builtin_amdgcn_exp2f(dot8(qrow, kt) * scale2 + bt + lse2);
          float dp = dot8(dorow, vt);
          if (dropping) dp = keep_of(rseed, key, thr) ? dp : 0.f;
          const float dsv = p * (dp - delta);
#pragma unroll
          for (int j = 0; j < 4; ++j) oacc[j] += dsv * (float)(h ? kt[4 + j] : kt[j]);
        }
      }
      store_rows<HD>(pr.dQ + (size_t)(b * S + q) * pr.lddq + hh * HD, oacc, scale * dscale, h);
    }
  }
}

// Both passes of the backward of SELF-attention (Tq == Sk: the two passes tile the same rows, so one block plan serves
// both) in ONE launch: blocks [0, nblocks) run the dK/dV pass (with its own delta), blocks [nblocks, 2 nblocks) the dQ
// pass.  No launch boundary between the passes, and the tail of one overlaps the head of the other: -0.03 ms per training
// step at 64 molecules (dQ blocks first: -0.02 less; the two kinds interleaved block by block: +0.36 ms).
template <int HD, bool CAUSAL>
__global__ void attn_bwd_both_kernel(const AttnBatch batch, int nblocks) {
  constexpr int SM = kv_smem_bytes<HD>() > q_smem_bytes<HD, true>() ? kv_smem_bytes<HD>() : q_smem_bytes<HD, true>();
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SM];
#ifdef MMST_WG_LOG
  const unsigned long long wg_t0 = wall_clock64();
#endif
  if ((int)blockIdx.x < nblocks) attn_kv_body<HD, CAUSAL, true>(batch, (int)blockIdx.x, smem);
  else attn_q_body<HD, CAUSAL, true>(batch, (int)blockIdx.x - nblocks, smem);
#ifdef MMST_WG_LOG
  ATTN_WGLOG_END(HD == 32 ? 4 : 5);
#endif
}


// ------------------------------------------------------------------------------------------
// Single-pass backward of self-attention without a causal mask (the encoder layers): dQ, dK and dV from ONE
// evaluation of the probabilities (the two-pass form above evaluates exp, the dropout hash and dS twice).
// A workgroup owns one batch element and ALL rows of a column group (one head of 32 with ten waves, or 8 heads of 8
// with two waves per head).  A wave owns 32-key tiles (kt = pass * WPH + part) exactly as in the dK/dV pass: S and dP
// with the key on the lane, dK and dV in accumulators.  All waves walk the query tiles in step.  dS is transposed
// through an LDS region of the wave (16 two-byte writes, two 16-byte reads per tile) and multiplied with the tile's
// K^T fragments, held in registers, to the wave's contribution to dQ of that query tile.  The contributions of a
// head's waves are summed through LDS without atomics (LDS float atomics run at about 0.4 lanes per clock and CU):
// each wave stores its 32 x HD partial into the region (double-buffered by step parity), one barrier, and a fixed
// thread per 16-byte piece adds the partials into the fp32 dQ rows of the current 128-query chunk, which go out as
// bf16 when the chunk is left (added to the first pass's value when a second pass over the key tiles revisits the
// chunk).  Nothing of dQ is reduced through HBM.  delta = dO . O is computed while a chunk is staged.
// ------------------------------------------------------------------------------------------
template <int HD> struct MCfg {
  static constexpr int GCM = HD == 32 ? 32 : 64;    // columns per workgroup
  static constexpr int HG = GCM / HD;               // heads per workgroup
  static constexpr int NW = HD == 32 ? 12 : 8;      // waves
  static constexpr int WPH = NW / HG;               // waves per head
  static constexpr int KS = HD == 32 ? 2 : 1;
  static constexpr int RM_BYTES = KC * GCM * 2;
  static constexpr int TR_BYTES = GCM * TSTR;
  static constexpr int MAXQ = HD == 32 ? 608 : 128; // most tiled rows
  static constexpr int SSTR = 80;                   // bytes per row of the dS^T image
  static constexpr int SLS = (HD + 4) * 4;          // bytes per row of a dQ partial
  static constexpr int REG = 32 * (SSTR > SLS ? SSTR : SLS);   // a wave's region: dS^T image, then its dQ partial
  static constexpr int NPOS = HG * 32 * HD / 4;     // 16-byte pieces of a query tile's dQ over the workgroup's heads
  static constexpr int SMEM = 2 * RM_BYTES + 2 * TR_BYTES + 3 * HG * KC * 4 + 2 * NW * REG + HG * KC * HD * 4;
};

template <int GCM> __device__ __forceinline__ int rmo(int row, int ch) {
  if (GCM == 64) return rm_off(row, ch);
  return row * 64 + ((ch ^ ((row >> 1) & 3)) << 4);   // 64-byte rows
}
template <int GCM>
__device__ __forceinline__ void stage_m(unsigned char* rm, unsigned char* tr, const bf16* src, int ld, int nvalid,
                                        int tid, int nthr) {
  constexpr int CPR = GCM / 8;
  for (int c = tid; c < KC * CPR; c += nthr) {
    const int row = c / CPR, ch = c % CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nvalid) v = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + ch * 8);
    *reinterpret_cast<uint4*>(rm + rmo<GCM>(row, ch)) = v;
    const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<unsigned short*>(tr + (ch * 8 + j) * TSTR + row * 2) = e[j];
  }
}
template <int HD>
__device__ __forceinline__ bf16x8 afrag_m(const unsigned char* img, int row, int s, int h, int hoff) {
  if (HD == 32) return *reinterpret_cast<const bf16x8*>(img + rmo<32>(row, 2 * s + h));
  bf16x8 v = *reinterpret_cast<const bf16x8*>(img + rm_off(row, hoff >> 3));
  return h == 0 ? v : zero8();
}

template <int HD, bool DROP, bool CAUSAL = false>
__global__ __launch_bounds__(MCfg<HD>::NW * 64, 1) void attn_bwd_kernel(const AttnBatch batch) {
  typedef MCfg<HD> C;
  constexpr int NW = C::NW, HG = C::HG, WPH = C::WPH, KS = C::KS, GCM = C::GCM, NTHR = NW * 64;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::SMEM];
  unsigned char* Qimg = smem;
  unsigned char* Dimg = Qimg + C::RM_BYTES;
  unsigned char* Qt = Dimg + C::RM_BYTES;
  unsigned char* Dt = Qt + C::TR_BYTES;
  float* lse_all = reinterpret_cast<float*>(Dt + C::TR_BYTES);   // [HG][KC]
  float* del_all = lse_all + HG * KC;
  uint32_t* rs_all = reinterpret_cast<uint32_t*>(del_all + HG * KC);
  unsigned char* reg_all = reinterpret_cast<unsigned char*>(rs_all + HG * KC);   // [2][NW][REG]
  float* dqc = reinterpret_cast<float*>(reg_all + 2 * NW * C::REG);              // [HG][KC][HD]

  int g = 0;
#pragma unroll
  for (int i = 1; i < ATTN_MAX_GROUPS; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.p[i].block_begin) g = i;
  const AttnProb& pr = batch.p[g];
  const int local = (int)blockIdx.x - pr.block_begin;
  const int ngroups = pr.H / HG;
  const int b = local / ngroups, hg = local % ngroups;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int hl = wave % HG, part = wave / HG, hoff = hl * HD, gcol = hg * GCM;
  const int hh = hg * HG + hl, H = pr.H, S = pr.Sk;
  const int tail = (HD == 8 && !CAUSAL) ? pr.tail_o : 0;
  const int S_main = S - tail;
  const float* lse_s = lse_all + hl * KC;
  const float* del_s = del_all + hl * KC;
  const uint32_t* rs_s = rs_all + hl * KC;
  const float scale = rsqrtf((float)HD);
  const float scale2 = scale * LOG2E;
  const uint32_t thr = drop_threshold(batch.drop_p);
  const float dscale = DROP ? 1.0f / (1.0f - batch.drop_p) : 1.0f;
  const float inv_dscale = DROP ? 1.0f - batch.drop_p : 1.0f;
  constexpr bool dropping = DROP;
  // The tile arithmetic leaves the constant factors out: dS' = P (keep dP - delta / dscale), P' = keep P; the sums are
  // scaled once: dK and dQ by scale * dscale, dV by dscale.
  const float kq_mul = scale * dscale;

  const int nkt = (S_main + 31) >> 5;
  const int npass = (nkt + WPH - 1) / WPH;
  const bool single = S_main <= KC;   // the one chunk is staged once and serves every pass
  const TailCtx tc = {scale, scale2, dscale, thr, dropping};

  // dQ rows of chunk c0 leave the workgroup: bf16, added to what an earlier pass wrote (`add`), plus (head_dim 8,
  // last flush) what the tail keys contribute, which no tile covers; then the chunk array is cleared
  auto flush = [&](int c0, bool add, bool with_tails) {
    const int nvalid = min(KC, S_main - c0);
    constexpr int PPR = HD / 8;   // 16-byte bf16 pieces per row and head
    for (int idx = tid; idx < HG * KC * PPR; idx += NTHR) {
      const int hj = idx / (KC * PPR), q = (idx / PPR) % KC, c = idx % PPR;
      float* src = dqc + (hj * KC + q) * HD + 8 * c;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = src[j] * kq_mul; src[j] = 0.f; }
      if (q >= nvalid) continue;
      const int head = hg * HG + hj;
      bf16* out = pr.dQ + (size_t)(b * S + c0 + q) * pr.lddq + head * HD + 8 * c;
      if (add) {
        const bf16x8 old = ldrow8(out);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)old[j];
      }
      if (HD == 8 && with_tails && tail > 0) {
        const int qg = c0 + q;
        const size_t stat = (size_t)(b * H + head) * S + qg;
        const bf16x8 qrow = ldrow8(pr.Q + (size_t)(b * S + qg) * pr.ldq + head * HD);
        const bf16x8 dorow = ldrow8(pr.dO + (size_t)(b * S + qg) * pr.lddo + head * HD);
        const float dlt = dot8(dorow, ldrow8(pr.O + (size_t)(b * S + qg) * pr.ldo + head * HD));
        const float lse2 = pr.lse[stat];
        const uint32_t rseed = attn_row_seed(pr.seed, (uint32_t)stat);
        for (int t = 0; t < tail; ++t) {
          const int key = S_main + t;
          const bf16x8 kt = ldrow8(pr.K + (size_t)(b * S + key) * pr.ldk + head * HD);
          const bf16x8 vt = ldrow8(pr.V + (size_t)(b * S + key) * pr.ldv + head * HD);
          const float bt = pr.kbias ? pr.kbias[(size_t)b * S + key] * LOG2E : 0.f;
          const float p = __builtin_amdgcn_exp2f(dot8(qrow, kt) * scale2 + bt - lse2);
          float dp = dot8(dorow, vt);
          const bool keep = !dropping || keep_of(rseed, key, thr);
          dp = keep ? dp * dscale : 0.f;
          const float dsv = p * (dp - dlt) * scale;
#pragma unroll
          for (int d = 0; d < 8; ++d) v[d] += dsv * (float)kt[d];
        }
      }
      store_row8(out, v, 1.0f);
    }
  };

  for (int i = tid; i < HG * KC * HD; i += NTHR) dqc[i] = 0.f;
  int step = 0;

  for (int pass = 0; pass < npass; ++pass) {
    const int kt = pass * WPH + part;
    const bool active = kt < nkt;
    const int nact = min(WPH, nkt - pass * WPH);   // waves per head with a key tile in this pass
    const int k0 = kt * 32, key = k0 + r;
    const bool kvalid = active && key < S_main;
    const uint32_t pair_term = (uint32_t)(key >> 1) * ATTN_PAIR_MUL;
    const uint32_t pair_shift = (key & 1) ? 16u : 0u;
    bf16x8 kf[KS], vf[KS], ktf[2];
    load_owner<HD>(kf, pr.K + (size_t)(b * S + (kvalid ? key : 0)) * pr.ldk + hh * HD, h, kvalid);
    load_owner<HD>(vf, pr.V + (size_t)(b * S + (kvalid ? key : 0)) * pr.ldv + hh * HD, h, kvalid);
    float bias2 = -INFINITY;
    if (kvalid) bias2 = pr.kbias ? pr.kbias[(size_t)b * S + key] * LOG2E : 0.f;
    // K^T of the tile as the A operand of dQ^T += K^T dS^T: lane (d, h) holds keys k0 + 16 s + 8 h + j
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = k0 + 16 * s + 8 * h + j;
        const bool ok = active && kk < S_main && (HD == 32 || r < 8);
        ktf[s][j] = ok ? pr.K[(size_t)(b * S + kk) * pr.ldk + hh * HD + (HD == 32 ? r : (r & 7))] : (bf16)0.f;
      }
    f32x16 dkacc = zero16(), dvacc = zero16();

    for (int c0 = 0; c0 < S_main; c0 += KC) {
      const int nvalid = min(KC, S_main - c0);
      if (!(single && pass > 0)) {
        __syncthreads();
        if (c0 > 0) flush(c0 - KC, pass > 0, false);
        stage_m<GCM>(Qimg, Qt, pr.Q + (size_t)(b * S + c0) * pr.ldq + gcol, pr.ldq, nvalid, tid, NTHR);
        stage_m<GCM>(Dimg, Dt, pr.dO + (size_t)(b * S + c0) * pr.lddo + gcol, pr.lddo, nvalid, tid, NTHR);
        for (int j = tid; j < HG * KC; j += NTHR) {
          const int hj = j / KC, i = j % KC, head = hg * HG + hj;
          float lv = 1.0e30f, dv = 0.f;
          if (i < nvalid) {
            lv = pr.lse[(size_t)(b * H + head) * S + c0 + i];
            dv = row_delta<HD>(pr, (size_t)b * S + c0 + i, head);
          }
          lse_all[j] = lv;
          del_all[j] = dv * inv_dscale;
          if (dropping) rs_all[j] = attn_row_seed(pr.seed, (uint32_t)((b * H + head) * S + c0 + i));
        }
        __syncthreads();
      }
      const int ntile = (nvalid + 31) >> 5;
      for (int t = 0; t < ntile; ++t, ++step) {
        const int lq0 = t * 32;
        unsigned char* reg = reg_all + ((step & 1) * NW + wave) * C::REG;
        if (active) {
          f32x16 sacc = zero16(), dpacc = zero16();
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_m<HD>(Qimg, lq0 + r, s, h, hoff), kf[s], sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_m<HD>(Dimg, lq0 + r, s, h, hoff), vf[s], dpacc, 0, 0, 0);
          }
          float pd[16], ds[16];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 ll = *reinterpret_cast<const f32x4*>(lse_s + lq0 + 8 * g4 + 4 * h);
            f32x4 dd = *reinterpret_cast<const f32x4*>(del_s + lq0 + 8 * g4 + 4 * h);
            uint32_t rsv[4] = {0, 0, 0, 0};
            if (dropping) {
              const uint4 rr = *reinterpret_cast<const uint4*>(rs_s + lq0 + 8 * g4 + 4 * h);
              rsv[0] = rr.x; rsv[1] = rr.y; rsv[2] = rr.z; rsv[3] = rr.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int i = 4 * g4 + j;
              float p = __builtin_amdgcn_exp2f(fmaf(sacc[i], scale2, bias2 - ll[j]));
              if (CAUSAL && key > c0 + lq0 + 8 * g4 + 4 * h + j) p = 0.f;   // the key lies behind the query
              float d = dpacc[i];
              float pk = p;
              if (dropping) {
                const uint32_t hb = attn_pair_bits(rsv[j], pair_term);
                const bool keep = ((hb >> pair_shift) & 0xFFFFu) >= thr;
                d = keep ? d : 0.f;
                pk = keep ? p : 0.f;
              }
              pd[i] = pk;
              ds[i] = p * (d - dd[j]);
            }
          }
          const bf16x8 pf0 = pack8(pd), pf1 = pack8(pd + 8), sf0 = pack8(ds), sf1 = pack8(ds + 8);
          dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_tr<HD>(Dt, lq0, 0, lane, hoff), pf0, dvacc, 0, 0, 0);
          dvacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_tr<HD>(Dt, lq0, 1, lane, hoff), pf1, dvacc, 0, 0, 0);
          dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_tr<HD>(Qt, lq0, 0, lane, hoff), sf0, dkacc, 0, 0, 0);
          dkacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag_tr<HD>(Qt, lq0, 1, lane, hoff), sf1, dkacc, 0, 0, 0);
          // dS^T image: row = query of the tile, column = key of the tile
#pragma unroll
          for (int i = 0; i < 16; ++i)
            *reinterpret_cast<bf16*>(reg + acc_row(i, h) * C::SSTR + r * 2) = i < 8 ? sf0[i] : sf1[i - 8];
          __builtin_amdgcn_wave_barrier();
          const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(reg + r * C::SSTR + 16 * h);
          const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(reg + r * C::SSTR + 32 + 16 * h);
          __builtin_amdgcn_wave_barrier();
          f32x16 dqa = zero16();
          dqa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[0], b0, dqa, 0, 0, 0);
          dqa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[1], b1, dqa, 0, 0, 0);
          // the wave's partial: row = query, 4 consecutive d per piece (the image above has been read)
#pragma unroll
          for (int g4 = 0; g4 < (HD == 32 ? 4 : 1); ++g4) {
            const f32x4 o = {dqa[4 * g4], dqa[4 * g4 + 1], dqa[4 * g4 + 2], dqa[4 * g4 + 3]};
            *reinterpret_cast<f32x4*>(reg + r * C::SLS + (8 * g4 + 4 * h) * 4) = o;
          }
        }
        __syncthreads();
        if (tid < C::NPOS) {
          constexpr int PPH = 32 * HD / 4;   // pieces per head
          const int hj = tid / PPH, pp = tid % PPH, q = pp / (HD / 4), f4 = pp % (HD / 4);
          f32x4* acc = reinterpret_cast<f32x4*>(dqc + (hj * KC + lq0 + q) * HD + 4 * f4);
          f32x4 sum = *acc;
          for (int k = 0; k < nact; ++k)
            sum += *reinterpret_cast<const f32x4*>(reg_all + ((step & 1) * NW + hj + HG * k) * C::REG + q * C::SLS + f4 * 16);
          *acc = sum;
        }
      }
    }
    if (kvalid) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { dkacc[i] *= kq_mul; dvacc[i] *= dscale; }
      if (HD == 8 && tail > 0) {
        // tail queries: one score per key lane pair (see attn_kv_kernel)
        const bf16x8 krow = ldrow8(pr.K + (size_t)(b * S + key) * pr.ldk + hh * HD);
        const bf16x8 vrow = ldrow8(pr.V + (size_t)(b * S + key) * pr.ldv + hh * HD);
        for (int t = 0; t < tail; ++t) {
          const int qt = S_main + t;
          const size_t stat = (size_t)(b * H + hh) * S + qt;
          const bf16x8 qrow = ldrow8(pr.Q + (size_t)(b * S + qt) * pr.ldq + hh * HD);
          const bf16x8 dorow = ldrow8(pr.dO + (size_t)(b * S + qt) * pr.lddo + hh * HD);
          const float dlt = dot8(dorow, ldrow8(pr.O + (size_t)(b * S + qt) * pr.ldo + hh * HD));
          const float p = __builtin_amdgcn_exp2f(dot8(krow, qrow) * scale2 + bias2 - pr.lse[stat]);
          float dp = dot8(vrow, dorow), pk = p;
          if (dropping) {
            const bool keep = keep_of(attn_row_seed(pr.seed, (uint32_t)stat), key, thr);
            dp = keep ? dp * dscale : 0.f;
            pk = keep ? p * dscale : 0.f;
          }
          const float dsv = p * (dp - dlt) * scale;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dvacc[j] += pk * (float)(h ? dorow[4 + j] : dorow[j]);
            dkacc[j] += dsv * (float)(h ? qrow[4 + j] : qrow[j]);
          }
        }
      }
      store_rows<HD>(pr.dK + (size_t)(b * S + key) * pr.lddk + hh * HD, dkacc, 1.0f, h);
      store_rows<HD>(pr.dV + (size_t)(b * S + key) * pr.lddv + hh * HD, dvacc, 1.0f, h);
    }
    if (!single || pass == npass - 1) {
      __syncthreads();
      flush(((S_main - 1) / KC) * KC, !single && pass > 0, true);
    }
  }

  if (HD == 8 && tail > 0) {
    // rows outside the tiling: the tail keys' dK/dV and the tail queries' dQ, one wave of the head each
    if (part == 0)
      for (int t = 0; t < tail; ++t) kv_owner_tail<true>(pr, tc, b, hh, S_main + t, lane);
    if (part == WPH - 1)
      for (int t = 0; t < tail; ++t) q_owner_tail<true>(pr, tc, b, hh, S_main + t, lane);
  }
}

// rows of a side that are handled outside the MFMA tiling (see "Tail rows" above)
int tail_of(const AttnBatch& b, int rows) {
  static const int enabled = [] { const char* e = getenv("MMST_ATTN_TAILS"); return e ? atoi(e) : 1; }();
  if (!enabled || b.hd != 8 || b.causal) return 0;
  if (rows <= 2) return rows;   // a side of one or two rows (a cached decode step): no MFMA tile at all on it
  const int r = rows & 31;
  return (rows >= 64 && r >= 1 && r <= 2) ? r : 0;
}

// workgroups per (batch, head group): one per TPB owner tiles
void plan(AttnBatch& b, bool owner_is_key, int (&bpb)[ATTN_MAX_GROUPS]) {
  const int tpb = b.hd == 32 ? Cfg<32>::TPB : Cfg<8>::TPB;
  for (int i = 0; i < b.n; ++i) {
    b.p[i].tail_o = tail_of(b, owner_is_key ? b.p[i].Sk : b.p[i].Tq);
    b.p[i].tail_s = tail_of(b, owner_is_key ? b.p[i].Tq : b.p[i].Sk);
    int rows = (owner_is_key ? b.p[i].Sk : b.p[i].Tq) - b.p[i].tail_o;
    int tiles = (rows + 31) / 32;
    bpb[i] = (tiles + tpb - 1) / tpb;
    if (bpb[i] == 0) bpb[i] = 1;   // every owner row is a tail row: one workgroup per (batch, head group) serves them
  }
}

int check(const AttnBatch& b) {
  if (b.n < 1 || b.n > ATTN_MAX_GROUPS) return MMST_ERR_ARG;
  if (b.hd != 8 && b.hd != 32) return MMST_ERR_ARG;
  for (int i = 0; i < b.n; ++i) {
    const AttnProb& p = b.p[i];
    if (p.B < 1 || p.H < 1 || p.Tq < 1 || p.Sk < 1) return MMST_ERR_ARG;
    if (p.H % (GC / b.hd)) return MMST_ERR_ARG;   // heads come in groups of 128 bytes
    if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3)) return MMST_ERR_ARG;
  }
  return MMST_OK;
}

template <bool BWD>
int launch_q(AttnBatch& b, hipStream_t stream) {
  int bpb[ATTN_MAX_GROUPS];
  plan(b, false, bpb);
  const int hg = GC / b.hd;
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    b.p[i].blocks_per_bh = bpb[i];
    b.p[i].block_begin = total;
    total += bpb[i] * b.p[i].B * (b.p[i].H / hg);
  }
  dim3 grid(total), block(64 * NWAVES);
  if (b.hd == 32) {
    if (b.causal) MMST_LAUNCH((attn_q_kernel<32, true, BWD>), grid, block, 0, stream, b);
    else MMST_LAUNCH((attn_q_kernel<32, false, BWD>), grid, block, 0, stream, b);
  } else {
    if (b.causal) MMST_LAUNCH((attn_q_kernel<8, true, BWD>), grid, block, 0, stream, b);
    else MMST_LAUNCH((attn_q_kernel<8, false, BWD>), grid, block, 0, stream, b);
  }
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}

}  // namespace

int launch_attn_fwd(AttnBatch& b, hipStream_t stream) {
  int rc = check(b);
  if (rc) return rc;
  return launch_q<false>(b, stream);
}

namespace {
int check_bwd(AttnBatch& b) {
  int rc = check(b);
  if (rc) return rc;
  for (int i = 0; i < b.n; ++i) {
    const AttnProb& p = b.p[i];
    if ((p.lddo & 7) || (p.lddq & 3) || (p.lddk & 3) || (p.lddv & 3)) return MMST_ERR_ARG;
  }
  return MMST_OK;
}
}  // namespace

// dQ and delta (the row sums dO . O the dK/dV pass reads)
int launch_attn_bwd_q(AttnBatch& b, hipStream_t stream) {
  int rc = check_bwd(b);
  if (rc) return rc;
  return launch_q<true>(b, stream);
}

namespace {
// The single-pass kernel serves self-attention (causal only with head_dim 8) whose tiled rows fit its dQ array
// (head_dim 8: at most 128 rows beside the tail rows; head_dim 32: at most 608).  Setting (attn_set_bwd_merged, or
// the environment variable MMST_ATTN_BWD_MERGED): bit 0 = use it for head_dim 32, bit 1 = for head_dim 8, bit 2 = for
// head_dim 8 with a causal mask (the decoder's self-attention; the causal form walks every tile and masks inside);
// 0 = the two-pass form everywhere, the default.  Measured at 64 molecules, alone on the GPU: head_dim 32 (582 rows)
// 128 us against 173 us for the two passes; head_dim 8 (3 x 129 + 66 rows) 111 us against 109 us (its workgroups need
// 150 KiB of LDS for eight waves).  Inside a training step, where the attention backward shares the GPU with the
// feed-forward weight-gradient kernel of the second stream, the single-pass kernels lose: their 150-158 KiB of LDS keep
// them off a CU while a weight-gradient workgroup (82 KiB) is resident, so the two run one after the other where the
// two-pass kernels (70 KiB, 128 registers) run beside it: 7.44 ms per step with bit 0 set against 7.38 ms without,
// +0.02 ms for bit 2.  Callers whose backward runs alone (no second stream) gain the difference measured alone.
int g_bwd_merged = -1;
int bwd_merged_mode() {
  if (g_bwd_merged < 0) {
    const char* e = getenv("MMST_ATTN_BWD_MERGED");
    const int v = e ? atoi(e) : 0;
    g_bwd_merged = (v >= 0 && v <= 15) ? v : 0;
  }
  return g_bwd_merged;
}
bool merged_ok(const AttnBatch& b) {
  if (b.causal && b.hd != 8) return false;
  if (!(bwd_merged_mode() & (b.hd == 32 ? 1 : (b.causal ? 4 : 2)))) return false;
  for (int i = 0; i < b.n; ++i) {
    const AttnProb& p = b.p[i];
    if (p.Tq != p.Sk || !p.O || !p.dO || !p.dQ || !p.dK || !p.dV) return false;
    if ((p.ldo & 7) || (p.lddq & 7)) return false;   // 16-byte row pieces
    const int main_rows = p.Sk - tail_of(b, p.Sk);
    if (main_rows < 1) return false;
    if (b.hd == 8 ? main_rows > MCfg<8>::MAXQ : p.Sk > MCfg<32>::MAXQ) return false;
  }
  return true;
}
template <int HD>
int launch_merged(AttnBatch& b, hipStream_t stream) {
  typedef MCfg<HD> C;
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    b.p[i].tail_o = b.p[i].tail_s = tail_of(b, b.p[i].Sk);
    b.p[i].blocks_per_bh = 1;
    b.p[i].block_begin = total;
    total += b.p[i].B * (b.p[i].H / C::HG);
  }
  if (HD == 8 && b.causal) {
    if (b.drop_p > 0.f) MMST_LAUNCH((attn_bwd_kernel<8, true, true>), dim3(total), dim3(C::NW * 64), 0, stream, b);
    else MMST_LAUNCH((attn_bwd_kernel<8, false, true>), dim3(total), dim3(C::NW * 64), 0, stream, b);
  } else if (b.drop_p > 0.f) {
    MMST_LAUNCH((attn_bwd_kernel<HD, true>), dim3(total), dim3(C::NW * 64), 0, stream, b);
  } else {
    MMST_LAUNCH((attn_bwd_kernel<HD, false>), dim3(total), dim3(C::NW * 64), 0, stream, b);
  }
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
}  // namespace

void attn_set_bwd_merged(int mode) { g_bwd_merged = (mode >= 0 && mode <= 15) ? mode : -1; }   // other values: the default
int attn_bwd_merged() { return bwd_merged_mode(); }

int launch_attn_bwd(AttnBatch& b, hipStream_t stream) {
  int rc = check_bwd(b);
  if (rc) return rc;
  if (merged_ok(b)) return b.hd == 32 ? launch_merged<32>(b, stream) : launch_merged<8>(b, stream);
  // self-attention: both passes in one launch (MMST_ATTN_BWD_ONE_LAUNCH=0 restores the two launches)
  static const bool one_launch = [] { const char* e = getenv("MMST_ATTN_BWD_ONE_LAUNCH"); return e ? atoi(e) != 0 : true; }();
  bool self_attn = one_launch;
  for (int i = 0; i < b.n; ++i) self_attn = self_attn && b.p[i].Tq == b.p[i].Sk && (b.p[i].ldo & 7) == 0;
  if (self_attn) {
    int bpb[ATTN_MAX_GROUPS];
    plan(b, false, bpb);   // Tq == Sk: the plan with the keys as owner rows is the same
    const int hg = GC / b.hd;
    int total = 0;
    // short sequences (head_dim 8, not causal, all but the tail rows inside one chunk), bit 3 of the setting: one
    // workgroup per (batch, head group) for both passes.  Opt-in: alone 97 against 105 us for the modality encoders'
    // launch, but its 512 long workgroups fill the gaps beside the weight-gradient stream worse than the 3584 short
    // ones of the kernel below (7.31 against 7.26 ms per training step).
    bool short_seq = (bwd_merged_mode() & 8) && b.hd == 8 && !b.causal;
    for (int i = 0; i < b.n; ++i) {
      const AttnProb& p = b.p[i];
      short_seq = short_seq && p.Tq - p.tail_o <= KC && p.tail_o == p.tail_s &&
                  ((p.ldq | p.ldk | p.ldv | p.lddo) & 7) == 0 &&
                  (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)p.dO) & 15) == 0;
    }
    if (short_seq) {
      for (int i = 0; i < b.n; ++i) {
        b.p[i].blocks_per_bh = 1;
        b.p[i].block_begin = total;
        total += b.p[i].B * (b.p[i].H / hg);
      }
      MMST_LAUNCH((attn_bwd_self_kernel<8>), dim3(total), dim3(64 * NWAVES), 0, stream, b);
      MMST_CHECK_LAUNCH();
      return MMST_OK;
    }
    for (int i = 0; i < b.n; ++i) {
      b.p[i].blocks_per_bh = bpb[i];
      b.p[i].block_begin = total;
      total += bpb[i] * b.p[i].B * (b.p[i].H / hg);
    }
    dim3 grid(2 * total), block(64 * NWAVES);
    if (b.hd == 32) {
      if (b.causal) MMST_LAUNCH((attn_bwd_both_kernel<32, true>), grid, block, 0, stream, b, total);
      else MMST_LAUNCH((attn_bwd_both_kernel<32, false>), grid, block, 0, stream, b, total);
    } else {
      if (b.causal) MMST_LAUNCH((attn_bwd_both_kernel<8, true>), grid, block, 0, stream, b, total);
      else MMST_LAUNCH((attn_bwd_both_kernel<8, false>), grid, block, 0, stream, b, total);
    }
    MMST_CHECK_LAUNCH();
    return MMST_OK;
  }
  // an armed completion event belongs to the LAST kernel of the backward
  hipEvent_t ev = mmst_armed_stop_event;
  mmst_armed_stop_event = nullptr;
  rc = launch_attn_bwd_q(b, stream);
  if (rc) return rc;
  mmst_armed_stop_event = ev;
  return launch_attn_bwd_kv(b, stream);
}

// dK and dV; needs delta from launch_attn_bwd_q
int launch_attn_bwd_kv(AttnBatch& b, hipStream_t stream) {
  int rc = check_bwd(b);
  if (rc) return rc;
  int bpb[ATTN_MAX_GROUPS];
  plan(b, true, bpb);
  const int hg = GC / b.hd;
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    b.p[i].blocks_per_bh = bpb[i];
    b.p[i].block_begin = total;
    total += bpb[i] * b.p[i].B * (b.p[i].H / hg);
  }
  dim3 grid(total), block(64 * NWAVES);
  if (b.hd == 32) {
    if (b.causal) MMST_LAUNCH((attn_kv_kernel<32, true>), grid, block, 0, stream, b);
    else MMST_LAUNCH((attn_kv_kernel<32, false>), grid, block, 0, stream, b);
  } else {
    if (b.causal) MMST_LAUNCH((attn_kv_kernel<8, true>), grid, block, 0, stream, b);
    else MMST_LAUNCH((attn_kv_kernel<8, false>), grid, block, 0, stream, b);
  }
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
