#!/bin/bash
# Diagnostic build of conv_wino.hip with per-workgroup start / end stamps (never part of libsqdhip.so).
set -e
cd "$(dirname "$0")/../../squeezedet-pytorch_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-result -DSQD_WINO_STAMP \
  -Xclang -target-feature -Xclang -load-store-opt -shared -o ../../scratch/diag/libwino_stamp.so conv_wino.hip 2> /tmp/wino_stamp.err || { cat /tmp/wino_stamp.err; exit 1; }
echo built
