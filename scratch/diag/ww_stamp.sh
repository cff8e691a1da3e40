#!/bin/bash
# Diagnostic build of the Winograd weight-gradient kernel with s_memtime stamps (never part of libsqdhip.so).
# usage (repo root): bash scratch/diag/ww_stamp.sh  -> scratch/diag/libww_stamp.so
set -e
cd "$(dirname "$0")/../../squeezedet-pytorch_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-result -DSQD_WW_STAMP \
  -Xclang -target-feature -Xclang -load-store-opt -shared -o ../../scratch/diag/libww_stamp.so wino_wgrad.hip ../../scratch/diag/ww_stub.hip 2> /tmp/ww_stamp.err || { cat /tmp/ww_stamp.err; exit 1; }
echo built
