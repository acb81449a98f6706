#!/bin/bash
# Build libmmst.so (all HIP kernels + C ABI) in-tree for gfx950.
set -e
cd "$(dirname "$0")/multimodalspectraltransformer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o ../libmmst.so *.hip "$@"
