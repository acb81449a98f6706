// Device-side tile helpers shared by the GEMM and the fused feed-forward kernels (gfx950):
// global->register->LDS staging of 16 KiB operand images, MFMA fragment reads, 8-wide epilogue vectors,
// and the LayerNorm forward / backward row epilogues for a wave that owns 32 whole rows of 128 columns.
#pragma once
#include "gemm.h"

namespace {

constexpr int BP = 128, BQ = 128, BK = 64;

__device__ __forceinline__ int fsw(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// ---- global -> registers (one 128x64 / 64x128 bf16 tile = 1024 x 16 B over the workgroup's 1024/NL threads) ----
template <bool T, int NL>
__device__ __forceinline__ void g2r(uint4 (&r)[NL], const bf16* __restrict__ base, int ld, int i0, int iN,
                                    int c0, int cN, int tid) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int c = tid + (1024 / NL) * i;
    int gi, gc;
    size_t off;
    if (!T) {
      gi = i0 + (c >> 3);
      gc = c0 + (c & 7) * 8;
      off = (size_t)gi * ld + gc;
    } else {
      gc = c0 + (c >> 4);
      gi = i0 + (c & 15) * 8;
      off = (size_t)gc * ld + gi;
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gi < iN && gc < cN) v = *reinterpret_cast<const uint4*>(base + off);
    r[i] = v;
  }
}

// ---- registers -> LDS image.  QSW selects the swizzle key of the row-major image:
// the P side is read with natural rows (key row&7), the Q side with interleaved rows.
template <bool T, bool QSW, int NL>
__device__ __forceinline__ void r2s(unsigned char* img, const uint4 (&r)[NL], int tid) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int c = tid + (1024 / NL) * i;
    int off;
    if (!T) {
      int row = c >> 3, ch = c & 7;
      int key = QSW ? fsw(row) : (row & 7);
      off = row * 128 + ((ch ^ key) << 4);
    } else {
      int row = c >> 4, ch = c & 15;
      off = row * 256 + (((((ch >> 1) ^ fsw(row)) << 1) | (ch & 1)) << 4);
    }
    *reinterpret_cast<uint4*>(img + off) = r[i];
  }
}

template <bool QSW>
__device__ __forceinline__ bf16x8 frag_n(const unsigned char* img, int row, int ks, int lane) {
  int chunk = ks * 4 + (lane >> 4);
  int key = QSW ? fsw(row) : (row & 7);
  return *reinterpret_cast<const bf16x8*>(img + row * 128 + ((chunk ^ key) << 4));
}

__device__ __forceinline__ bf16x8 frag_t(const unsigned char* img, int col, int ks, int lane) {
  int krow = ks * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2);
  int off = krow * 256 + (((col >> 4) ^ fsw(krow)) << 5) + (col & 15) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + off));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + off + 4 * 256));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// The same transposing read issued as inline assembly, for loops that keep LDS-DMA (global_load_lds) in flight:
// the compiler orders every LDS read it can see behind a pending LDS-DMA with s_waitcnt vmcnt(0), which would
// drain the DMA pipeline each step.  The caller waits with lds_wait*() before using lo/hi.
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void frag_t_issue(s16x4& lo, s16x4& hi, const unsigned char* img, int col, int ks, int lane) {
  int krow = ks * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2);
  int off = krow * 256 + (((col >> 4) ^ fsw(krow)) << 5) + (col & 15) * 2;
  uint32_t a = (uint32_t)(uintptr_t)(LDS_AS const unsigned char*)(img + off);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a));
}
__device__ __forceinline__ void lds_wait4(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ bf16x8 frag_join(s16x4 lo, s16x4 hi) {
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void load8(float (&o)[8], const bf16* p) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
__device__ __forceinline__ void load8f(float (&o)[8], const float* p) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
  *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ void store8f(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
// dropout over 8 consecutive element indices starting at idx (idx % 8 == 0)
__device__ __forceinline__ void drop8(float (&v)[8], uint32_t seed, uint32_t idx, uint32_t thr, float scale) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    uint32_t h = drop_bits(seed, (idx + e) >> 1);
    v[e] = ((h & 0xFFFFu) >= thr) ? v[e] * scale : 0.f;
    v[e + 1] = ((h >> 16) >= thr) ? v[e + 1] * scale : 0.f;
  }
}

// ---- fp32 partial sums of a row-owning accumulator (4x1 wave layout): slab s holds [rows][128] floats ----
template <int MT>
__device__ __forceinline__ void store_slab(const f32x4 (&acc)[MT][8], float* slab, int rows, int p0, int pw0, int lane) {
  const int q = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
    if (m >= rows) continue;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      float* d = slab + (size_t)m * 128 + tp * 32 + 8 * q;
      *reinterpret_cast<f32x4*>(d) = acc[mt][2 * tp];
      *reinterpret_cast<f32x4*>(d + 4) = acc[mt][2 * tp + 1];
    }
  }
}
// acc += slabs s0, s0 + stride, ... (< n) for the 32 rows at p0 + pw0; all loads of a call are independent
template <int MAXS, int MT>
__device__ __forceinline__ void add_slabs(f32x4 (&acc)[MT][8], const float* slabs, int s0, int stride, int n, int rows,
                                          int p0, int pw0, int lane) {
  const int q = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
    if (m >= rows) continue;
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
      const int s = s0 + k * stride;
      if (s >= n) continue;
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        const float* d = slabs + ((size_t)s * rows + m) * 128 + tp * 32 + 8 * q;
        acc[mt][2 * tp] += *reinterpret_cast<const f32x4*>(d);
        acc[mt][2 * tp + 1] += *reinterpret_cast<const f32x4*>(d + 4);
      }
    }
  }
}

// ---- LayerNorm forward on a row-owning wave layout: acc[mt][nt] holds rows p0+pw0+mt*16+(lane&15), all 128
// columns (MT = 2: 4 waves x 32 rows, MT = 1: 8 waves x 16 rows).
// z = R + drop(acc + bias); C = LN(z) (row remap cS/cL/coff); Z = z; stats = (mean, rstd).
// yf (optional): the output rows as MFMA token-side fragments (yf[mt][tp] = columns 32 tp + 8 q .. + 7 of the lane's row,
// the values stored to C; zero for rows past the end), for a GEMM chained behind the epilogue.
// vec (optional): bias | gamma | beta as 3 x 128 floats the caller has staged (in LDS) in place of pr.bias, pr.gamma, pr.beta.
template <int MT>
__device__ __forceinline__ void epi_ln(f32x4 (&acc)[MT][8], const GemmProb& pr, int p0, int pw0, int lane,
                                       bf16x8 (*yf)[4] = nullptr, const float* vec = nullptr) {
  const int q = lane >> 4;
  const float* bias_v = vec ? vec : pr.bias;
  const float* gamma_v = vec ? vec + 128 : pr.gamma;
  const float* beta_v = vec ? vec + 256 : pr.beta;
  auto crow = [&](int m) -> size_t { return pr.cS > 0 ? (size_t)(m / pr.cS) * pr.cL + pr.coff + (m % pr.cS) : (size_t)m; };
  auto rrow = [&](int m) -> size_t { return pr.rS > 0 ? (size_t)(m / pr.rS) * pr.rL + pr.roff + (m % pr.rS) : (size_t)m; };
  const uint32_t thr = drop_threshold(pr.drop_p);
  const float dscale = pr.drop_p > 0.f ? 1.0f / (1.0f - pr.drop_p) : 1.0f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
    const bool ok = m < pr.Pn;
    float z[4][8];
    float s = 0.f;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      const int n = tp * 32 + 8 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) { z[tp][e] = acc[mt][2 * tp][e]; z[tp][4 + e] = acc[mt][2 * tp + 1][e]; }
      if (bias_v) {
        float b[8];
        load8f(b, bias_v + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[tp][e] += b[e];
      }
      if (pr.drop_p > 0.f) drop8(z[tp], pr.seed, (uint32_t)m * 128u + n, thr, dscale);
      if (ok && pr.R) {
        float r[8];
        load8(r, pr.R + rrow(m) * pr.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[tp][e] += r[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += z[tp][e];
    }
    s = xor32_sum(xor16_sum(s));
    const float mean = s * (1.0f / 128.0f);
    float vs = 0.f;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
      for (int e = 0; e < 8; ++e) { float d = z[tp][e] - mean; vs += d * d; }
    vs = xor32_sum(xor16_sum(vs));
    const float rstd = rsqrtf(vs * (1.0f / 128.0f) + 1e-5f);
    if (ok) {
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        const int n = tp * 32 + 8 * q;
        if (pr.Z) store8(pr.Z + (size_t)m * 128 + n, z[tp]);
        float y[8], gam[8], bet[8];   // scale and shift are read where they are used (cache hits): 64 fewer live registers
        load8f(gam, gamma_v + n);
        load8f(bet, beta_v + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (z[tp][e] - mean) * rstd * gam[e] + bet[e];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)y[e];
        *reinterpret_cast<bf16x8*>((bf16*)pr.C + crow(m) * pr.ldc + n) = o;
        if (yf) yf[mt][tp] = o;
      }
      if (q == 0 && pr.stats) { pr.stats[2 * (size_t)m] = mean; pr.stats[2 * (size_t)m + 1] = rstd; }
    } else if (yf) {
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int e = 0; e < 8; ++e) yf[mt][tp][e] = (bf16)0.f;
    }
  }
}

// ---- LayerNorm backward on the same layout: dy = acc + R; C = dz, C2 = drop(dz); dgamma/dbeta column sums
// (in-wave shuffles, then the NW waves through `smem`, >= NW KiB, which must be free of live data).
// NW = waves of the workgroup that call it (4 or 8: 128-row tile, 1: a single wave with its rows).
template <int NW = 4, int MT>
__device__ __forceinline__ void epi_lnbwd(f32x4 (&acc)[MT][8], const GemmProb& pr, int p0, int pw0, int lane, int wave,
                                          int tid, unsigned char* smem) {
  const int q = lane >> 4;
  auto rrow = [&](int m) -> size_t { return pr.rS > 0 ? (size_t)(m / pr.rS) * pr.rL + pr.roff + (m % pr.rS) : (size_t)m; };
  const uint32_t thr = drop_threshold(pr.drop_p);
  const float dscale = pr.drop_p > 0.f ? 1.0f / (1.0f - pr.drop_p) : 1.0f;
  float gam[4][8];
  float pg[4][8], pb[4][8];
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) {
    load8f(gam[tp], pr.gamma + tp * 32 + 8 * q);
#pragma unroll
    for (int e = 0; e < 8; ++e) { pg[tp][e] = 0.f; pb[tp][e] = 0.f; }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = p0 + pw0 + mt * 16 + (lane & 15);
    const bool ok = m < pr.Pn;
    float mean = 0.f, rstd = 0.f;
    if (ok) { mean = pr.stats[2 * (size_t)m]; rstd = pr.stats[2 * (size_t)m + 1]; }
    float gy[4][8], xh[4][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      const int n = tp * 32 + 8 * q;
      float dy[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { dy[e] = acc[mt][2 * tp][e]; dy[4 + e] = acc[mt][2 * tp + 1][e]; }
      float zz[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) zz[e] = 0.f;
      if (ok) {
        if (pr.R) {
          float r[8];
          load8(r, pr.R + rrow(m) * pr.ldr + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) dy[e] += r[e];
        }
        load8(zz, pr.Z + (size_t)m * 128 + n);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) dy[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = ok ? (zz[e] - mean) * rstd : 0.f;
        xh[tp][e] = x;
        gy[tp][e] = dy[e] * gam[tp][e];
        s1 += gy[tp][e];
        s2 += gy[tp][e] * x;
        pg[tp][e] += dy[e] * x;
        pb[tp][e] += dy[e];
      }
    }
    s1 = xor32_sum(xor16_sum(s1));
    s2 = xor32_sum(xor16_sum(s2));
    s1 *= (1.0f / 128.0f);
    s2 *= (1.0f / 128.0f);
    if (ok) {
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        const int n = tp * 32 + 8 * q;
        float dz[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dz[e] = rstd * (gy[tp][e] - s1 - xh[tp][e] * s2);
        store8((bf16*)pr.C + (size_t)m * pr.ldc + n, dz);
        if (pr.C2) {
          if (pr.drop_p > 0.f) drop8(dz, pr.seed, (uint32_t)m * 128u + n, thr, dscale);
          store8(pr.C2 + (size_t)m * 128 + n, dz);
        }
      }
    }
  }
  // column sums over the tile's rows: 16 lanes (lane&15) in-wave, then the NW waves via LDS
  if (pr.dgamma) {
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pg[tp][e] = row16_sum(pg[tp][e]);
        pb[tp][e] = row16_sum(pb[tp][e]);
      }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [NW waves][2][128]
    if ((lane & 15) == 0) {
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[(wave * 2 + 0) * 128 + tp * 32 + 8 * q + e] = pg[tp][e];
          red[(wave * 2 + 1) * 128 + tp * 32 + 8 * q + e] = pb[tp][e];
        }
    }
    __syncthreads();
    for (int o = tid; o < 256; o += NW * 64) {
      int which = o >> 7, col = o & 127;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(w * 2 + which) * 128 + col];
      atomicAdd((which ? pr.dbeta : pr.dgamma) + col, s);
    }
  }
}

}  // namespace
