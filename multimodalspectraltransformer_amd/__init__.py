"""MI355X-native MultimodalTransformer hot path (HIP kernels behind a C ABI)."""
