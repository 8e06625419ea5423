#!/bin/bash
# build tools/ab_<tag>.so from the current sources with extra -D flags (A/B measurements through TFHE_HIP_LIB / tools/ab_full.sh)
# usage: tools/build_variant.sh <tag> [-DNAME=VALUE ...]
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -shared -fPIC "$@" \
  $R/toyfhe.jl_amd/csrc/toyfhe_hip.hip -o $R/tools/ab_$TAG.so 2>&1 | grep -v "hip-link" ; echo "built ab_$TAG.so $*"
