#!/bin/bash
# ablation builds of conv_wino_vs (run here, on the CPU box: hipcc cross-compiles): scratch/diag/libvsdiag.so exports sqd_vs_diag_<mask>
cd "$(dirname "$0")"
SRC=../../squeezedet-pytorch_amd/csrc
OBJS=""
for m in 0 100 1 2 3 4 8 11; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I$SRC -Wno-unused-result -Xclang -target-feature -Xclang -load-store-opt \
    -DSQD_VS_DIAG=$((m % 100)) -DSQD_VS_PRIO=$((1 - m / 100)) -DSQD_VS_ENTRY=sqd_vs_diag_$m -Dconv_wino_vs_kernel=conv_wino_vs_kernel_d$m -c $SRC/conv_wino_vs.hip -o vs_diag_$m.o 2> >(grep -v "recognized feature" >&2) || exit 1
  OBJS="$OBJS vs_diag_$m.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvsdiag.so $OBJS && rm -f $OBJS && echo built scratch/libvsdiag.so
