#!/bin/bash
# usage: tools/build_variant.sh <source-root> <out.so> [extra hipcc flags...]
# Diagnostic A/B builds of the HIP library (tools/ab_libs.py times them on one box).
set -e
SRC=$1; OUT=$2; shift 2
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -fno-unroll-loops -std=c++17 -ffp-contract=off -fPIC -shared \
  -I"$SRC/include" -I"$SRC/peaksegdisk_amd/csrc" "$@" "$SRC/peaksegdisk_amd/csrc/peakseg_hip.cpp" -o "$OUT"
