#!/bin/bash
# Builds raytracer_amd/librthip_<name>.so: the f32 kernels recompiled with extra flags (for A/B
# runs with tools/ab_libs.py), everything else taken from the regular build.
#   tools/build_variant.sh NAME [-DFLAG ...]
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/raytracer_amd/csrc
TMP=$(mktemp -d /tmp/rthip_variant_XXXX)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -ffp-contract=fast -fno-slp-vectorize -c -o $TMP/k32.o $SRC/kernels_f32.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/raytracer_amd/librthip_$NAME.so $SRC/_build/api.o $SRC/_build/host.o $TMP/k32.o $SRC/_build/kernels_f64.o $SRC/_build/build_gpu.o $SRC/_build/refpass_gpu.o -ldl
rm -rf $TMP
echo built librthip_$NAME.so
