#!/bin/bash
# usage: build_wino_var.sh <name> <-D flags...>  -> scratch/diag/libwino_<name>.so (one experimental build of conv_wino.hip)
cd "$(dirname "$0")"
CSRC=../../squeezedet-pytorch_amd/csrc
NAME=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I$CSRC "$@" \
   -Xclang -target-feature -Xclang -load-store-opt -shared -o libwino_$NAME.so $CSRC/conv_wino.hip 2>&1 | grep -v "not a recognized feature" | grep -E "error|spill"
ls -la libwino_$NAME.so | cut -c25-
