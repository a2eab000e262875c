#!/bin/bash
# Build a variant of the library next to the default one, for A/B runs on the GPU box:
#   tools/build_variant.sh ept64 -DGX_TL_EPT=64      ->  genrich_amd/libgenrich_amd_ept64.so
#   GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_ept64.so python bench.py --no-cpu
# (GX_VARIANT_NOATOMOPT=1: without LLVM's atomic optimizer -- k_sbtile gains 0.01 ms, k_bh_hist's slot counter loses 7 ms at config 5)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
extra=""
[ -n "$GX_VARIANT_NOATOMOPT" ] && extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall \
  -Wno-unused-function $extra "$@" genrich_amd/csrc/gx_api.hip genrich_amd/csrc/gx_emit.cpp -o genrich_amd/libgenrich_amd_$name.so
echo genrich_amd/libgenrich_amd_$name.so
