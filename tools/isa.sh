#!/bin/bash
# Device assembly of the f32 kernels (/tmp/k32.s) and the register / spill figures of every
# render_kernel instantiation; with an argument, the assembly of the instantiation whose mangled
# name contains it goes to /tmp/kern.s.
#   tools/isa.sh [-DFLAG ...] [-- NAME_PART]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS=(); PART=""
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then PART=$2; shift 2; else FLAGS+=("$1"); shift; fi; done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-command-line-argument "${FLAGS[@]}" -ffp-contract=fast -fno-slp-vectorize --cuda-device-only -S -o /tmp/k32.s $ROOT/raytracer_amd/csrc/kernels_f32.hip
grep -E "^\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count):" /tmp/k32.s | paste - - - - - | grep render_kernel | sed 's/ \+/ /g; s/_ZN5rthip13render_kernelI//; s/EEvNS_10RenderArgsIT_EE//'
if [ -n "$PART" ]; then
  awk -v p="$PART" '$0 ~ "^_ZN5rthip13render_kernelI" p ".*:$" {on=1} on {print} on && /\.end_amdhsa_kernel/ {exit}' /tmp/k32.s > /tmp/kern.s
  wc -l /tmp/kern.s
fi
