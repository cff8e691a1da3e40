#!/bin/bash
# Ablation builds of the Winograd kernel: one shared object per SQD_WINO_DIAG mask (see conv_wino.hip) -> scratch/diag/libwino_<mask>.so
cd "$(dirname "$0")"
CSRC=../../squeezedet-pytorch_amd/csrc
for M in ${@:-0 1 2 4 8 3 6 12 14}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I$CSRC -DSQD_WINO_DIAG=$M \
     -Xclang -target-feature -Xclang -load-store-opt -shared -o libwino_$M.so $CSRC/conv_wino.hip 2>&1 | grep -v "not a recognized feature" | grep -E "error|warning: v" &
done
wait
ls -la libwino_*.so
