// Grouped bf16 MFMA GEMM for gfx950 with fused bias / ReLU / dropout / residual /
// LayerNorm (forward and backward) / split-K atomic epilogues.  See gemm.h.
//
// Tile: 128 (P) x 128 (Q) x 64 (contraction), 256 threads = 4 waves.
//   wave layout 2x2 (64x64 per wave) for the plain epilogues,
//   wave layout 4x1 (32 P-rows x all 128 Q-columns per wave) for the LayerNorm
//   epilogues, so a wave owns whole rows and row statistics need no LDS.
// MFMA: v_mfma_f32_16x16x32_bf16 with the WEIGHT side as the A operand and the token
// side as the B operand: the accumulator then has the token row on the lane
// (lane & 15) and 4 consecutive output columns in its 4 registers.  Two adjacent
// MFMA column tiles are interleaved (row i of tile t maps to column
// 8*(i>>2) + 4*(t&1) + (i&3)) so that every lane owns 8 CONSECUTIVE output columns:
// all epilogue traffic is 16-byte bf16 vectors.
// LDS images (16 KiB each, XOR-swizzled, conflict-free for the reads below):
//   row-major operand      [128 index][64 c]  read with ds_read_b128
//   contraction-major      [64 c][128 index]  read with ds_read_b64_tr_b16
#include "gemm.h"

namespace {

constexpr int BP = 128, BQ = 128, BK = 64;

__device__ __forceinline__ int fsw(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// ---- global -> registers (one 128x64 / 64x128 bf16 tile = 4 x 16 B per thread) ----
template <bool T>
__device__ __forceinline__ void g2r(uint4 (&r)[4], const bf16* __restrict__ base, int ld, int i0, int iN,
                                    int c0, int cN, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + 256 * i;
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
template <bool T, bool QSW>
__device__ __forceinline__ void r2s(unsigned char* img, const uint4 (&r)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + 256 * i;
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

template <int EPI, bool PT, bool QT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmBatch batch) {
  constexpr bool ROWEPI = (EPI == EPI_LN || EPI == EPI_LNBWD);
  constexpr int MT = ROWEPI ? 2 : 4;
  constexpr int NT = ROWEPI ? 8 : 4;
  constexpr int SMEM = (EPI == EPI_ATOMIC) ? 4 * 32 * 65 * 4 : 2 * BP * BK * 2;  // images, or the atomic staging
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  unsigned char* Pimg = smem;
  unsigned char* Qimg = smem + BP * BK * 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g = 0;
#pragma unroll
  for (int i = 1; i < GEMM_MAX_GROUPS; ++i)
    if (i < batch.n && (int)blockIdx.x >= batch.p[i].tile_begin) g = i;
  const GemmProb& pr = batch.p[g];
  int local = blockIdx.x - pr.tile_begin;
  const int qt = local % pr.qtiles;
  local /= pr.qtiles;
  const int ptile = local % pr.ptiles;
  const int ks_id = local / pr.ptiles;
  const int p0 = ptile * BP, q0 = qt * BQ;
  const int pw0 = ROWEPI ? wave * 32 : (wave >> 1) * 64;
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
    uint4 rpA[4], rqA[4], rpB[4], rqB[4];
    g2r<PT>(rpA, pr.P, pr.ldp, p0, pr.Pn, kbeg * BK, pr.K, tid);
    g2r<QT>(rqA, pr.Q, pr.ldq, q0, pr.Qn, kbeg * BK, pr.K, tid);
    if (kbeg + 1 < kend) {
      g2r<PT>(rpB, pr.P, pr.ldp, p0, pr.Pn, (kbeg + 1) * BK, pr.K, tid);
      g2r<QT>(rqB, pr.Q, pr.ldq, q0, pr.Qn, (kbeg + 1) * BK, pr.K, tid);
    }
    for (int kt = kbeg; kt < kend; kt += 2) {
      __syncthreads();
      r2s<PT, false>(Pimg, rpA, tid);
      r2s<QT, true>(Qimg, rqA, tid);
      __syncthreads();
      if (kt + 2 < kend) {
        g2r<PT>(rpA, pr.P, pr.ldp, p0, pr.Pn, (kt + 2) * BK, pr.K, tid);
        g2r<QT>(rqA, pr.Q, pr.ldq, q0, pr.Qn, (kt + 2) * BK, pr.K, tid);
      }
      compute_tile();
      if (kt + 1 < kend) {
        __syncthreads();
        r2s<PT, false>(Pimg, rpB, tid);
        r2s<QT, true>(Qimg, rqB, tid);
        __syncthreads();
        if (kt + 3 < kend) {
          g2r<PT>(rpB, pr.P, pr.ldp, p0, pr.Pn, (kt + 3) * BK, pr.K, tid);
          g2r<QT>(rqB, pr.Q, pr.ldq, q0, pr.Qn, (kt + 3) * BK, pr.K, tid);
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
    // LDS (two passes of 32 rows) and adds whole 64-float row segments.
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 65);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
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
    float gam[4][8], bet[4][8];
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      load8f(gam[tp], pr.gamma + tp * 32 + 8 * q);
      load8f(bet[tp], pr.beta + tp * 32 + 8 * q);
    }
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
        if (pr.bias) {
          float b[8];
          load8f(b, pr.bias + n);
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
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / 128.0f);
      float vs = 0.f;
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = z[tp][e] - mean; vs += d * d; }
      vs += __shfl_xor(vs, 16);
      vs += __shfl_xor(vs, 32);
      const float rstd = rsqrtf(vs * (1.0f / 128.0f) + 1e-5f);
      if (ok) {
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
          const int n = tp * 32 + 8 * q;
          if (pr.Z) store8(pr.Z + (size_t)m * 128 + n, z[tp]);
          float y[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = (z[tp][e] - mean) * rstd * gam[tp][e] + bet[tp][e];
          store8((bf16*)pr.C + crow(m) * pr.ldc + n, y);
        }
        if (q == 0 && pr.stats) { pr.stats[2 * (size_t)m] = mean; pr.stats[2 * (size_t)m + 1] = rstd; }
      }
    }
  }

  if constexpr (EPI == EPI_LNBWD) {
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
      s1 += __shfl_xor(s1, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16);
      s2 += __shfl_xor(s2, 32);
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
    // column sums over the tile's 128 rows: 16 lanes (lane&15) in-wave, then 4 waves via LDS
    if (pr.dgamma) {
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = pg[tp][e], b = pb[tp][e];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
          pg[tp][e] = a;
          pb[tp][e] = b;
        }
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][128]
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
      {
        int which = tid >> 7, col = tid & 127;
        float s = red[(0 * 2 + which) * 128 + col] + red[(1 * 2 + which) * 128 + col] +
                  red[(2 * 2 + which) * 128 + col] + red[(3 * 2 + which) * 128 + col];
        atomicAdd((which ? pr.dbeta : pr.dgamma) + col, s);
      }
    }
  }
}

template <int EPI>
int dispatch(const GemmBatch& b, bool pt, bool qt, int grid, hipStream_t s) {
  if (!pt && !qt) hipLaunchKernelGGL((gemm_kernel<EPI, false, false>), dim3(grid), dim3(256), 0, s, b);
  else if (!pt && qt) hipLaunchKernelGGL((gemm_kernel<EPI, false, true>), dim3(grid), dim3(256), 0, s, b);
  else if (pt && qt) hipLaunchKernelGGL((gemm_kernel<EPI, true, true>), dim3(grid), dim3(256), 0, s, b);
  else return MMST_ERR_ARG;
  return MMST_OK;
}

}  // namespace

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
  const bool timed = tap.epi == epi && tap.pt == (int)pt && tap.qt == (int)qt && tap.count < tap.capacity;
  if (timed) {
    for (int i = 0; i < b.n; ++i) {
      const GemmProb& p = b.p[i];
      const double out_bytes = (epi == EPI_ATOMIC) ? 8.0 : ((p.flags & GF_OUT_F32) ? 4.0 : 2.0);
      tap.flops += 2.0 * p.Pn * (double)p.Qn * p.K;
      tap.bytes += 2.0 * p.K * ((double)p.Pn + p.Qn) + out_bytes * p.Pn * (double)p.Qn +
                   ((epi == EPI_DRELU || epi == EPI_LN) ? 2.0 * p.Pn * (double)p.Qn : 0.0);
    }
    hipEventRecord(tap.start[tap.count], stream);
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
  if (timed) hipEventRecord(tap.stop[tap.count++], stream);
  if (rc != MMST_OK) return rc;
  MMST_CHECK_LAUNCH();
  return MMST_OK;
}
