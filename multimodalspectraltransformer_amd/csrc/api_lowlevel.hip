// Low-level C-ABI: the grouped GEMM and attention launchers, exported for the parity tests
// and for the one-off projections of the host layer (IR embedding, fingerprint head, vocabulary
// projection and their gradients).  Structs are plain C (see gemm.h / attn.h).
#include "attn.h"
#include "gemm.h"

extern "C" {

int mmst_gemm(const GemmBatch* batch, int epi, int p_contraction_major, int q_contraction_major, void* stream) {
  if (!batch) return MMST_ERR_ARG;
  GemmBatch b = *batch;
  return launch_gemm(b, epi, p_contraction_major != 0, q_contraction_major != 0, (hipStream_t)stream);
}

int mmst_attn_fwd(const AttnBatch* batch, void* stream) {
  if (!batch) return MMST_ERR_ARG;
  AttnBatch b = *batch;
  return launch_attn_fwd(b, (hipStream_t)stream);
}

int mmst_attn_bwd(const AttnBatch* batch, void* stream) {
  if (!batch) return MMST_ERR_ARG;
  AttnBatch b = *batch;
  return launch_attn_bwd(b, (hipStream_t)stream);
}

int mmst_sizeof_gemm_batch() { return (int)sizeof(GemmBatch); }
int mmst_sizeof_attn_batch() { return (int)sizeof(AttnBatch); }
int mmst_version() { return 1; }
}
