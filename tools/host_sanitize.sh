#!/bin/bash
# The host half of librthip (scene factories, reference tree, ranks, gates, surface-area build,
# quantisation, the layouts for the device-side build, the records of a refit) under ASan + UBSan.  No GPU needed.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/rthip_san_XXXX)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/raytracer_amd/csrc -I$ROOT/include \
    -o $TMP/drv $ROOT/tools/host_sanitize.cpp $ROOT/raytracer_amd/csrc/host.cpp -lpthread
$TMP/drv
rm -rf $TMP
