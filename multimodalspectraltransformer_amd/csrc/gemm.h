// Grouped bf16 MFMA GEMM with fused row epilogues (gfx950).
//
// One problem computes OUT[Pn x Qn] = sum_c Pop[p, c] * Qop[q, c], fp32 accumulate.
// "P" is the activation-like side (token rows), "Q" the weight-like side (output
// columns).  Either operand may be stored contraction-major ([c][index]); the kernel
// then fetches its MFMA fragments with the LDS transposing read, so no transposed
// copies of weights or activations exist anywhere.
#pragma once
#include <stdint.h>

#include "common.h"

enum GemmEpi {
  EPI_BIAS = 0,    // C = act(acc + bias) (+dropout) (+R)          bf16 or fp32 out
  EPI_DRELU = 1,   // C = acc * (R > 0 ? scale : 0)                 grad through ReLU(+dropout)
  EPI_LN = 2,      // z = R + drop(acc + bias); C = LN(z); Z = z; stats
  EPI_LNBWD = 3,   // dy = acc + R; LayerNorm backward -> C (dz), C2 (dropped dz), dgamma, dbeta
  EPI_ATOMIC = 4,  // C(fp32)[p][q] += acc   (split-K weight gradients)
};

enum GemmFlags { GF_RELU = 1, GF_OUT_F32 = 2 };

struct GemmProb {
  const bf16* P; const bf16* Q;
  void* C; const float* bias; const bf16* R;
  const float* gamma; const float* beta;
  bf16* Z; float* stats; bf16* C2;
  float* dgamma; float* dbeta; float* dbias;
  int ldp, ldq, ldc, ldr;
  int Pn, Qn, K;
  int flags;
  float drop_p; uint32_t seed;
  // Optional row remap for C (cS>0) and R (rS>0): logical row m = b*S + s lives at physical
  // row b*L + off + s, i.e. a [B][S] sequence embedded in a longer [B][L] one.
  int cS, cL, coff, rS, rL, roff;
  int ksplit;      // number of K slices (EPI_ATOMIC only), >= 1
  int tile_begin;  // filled by the launcher
  int ptiles, qtiles;
};

#define GEMM_MAX_GROUPS 16
struct GemmBatch {
  int n;
  int lds_pad = 0;   // extra (unused) dynamic LDS per workgroup in bytes: > 14 KiB keeps a split-K launch at one
                     // workgroup per CU, so that kernels of another stream always find free registers next to it
  GemmProb p[GEMM_MAX_GROUPS];
};

int gemm_waves_for(int epi);   // waves per workgroup of the kernel variant that serves this epilogue
void gemm_kernel_name(int epi, bool pt, bool qt, char* out, int cap);

// Diagnostic tap (bench.py): HIP events around every launch of ONE kernel variant, on the launch stream.
constexpr int TAP_FFN_WGRAD = 100;   // tap selector: the recomputing feed-forward weight-gradient kernel (ffn_wgrad.hip)
struct GemmTap {
  int epi, pt, qt;       // variant to time; epi < 0 = off
  int count, capacity;
  hipEvent_t* start; hipEvent_t* stop;
  double flops, bytes;   // algorithmic work of the timed launches
};
extern GemmTap g_gemm_tap;

// PT / QT: operand stored contraction-major.  Returns MMST_OK or an error code.
int launch_gemm(GemmBatch& b, int epi, bool pt, bool qt, hipStream_t stream);
