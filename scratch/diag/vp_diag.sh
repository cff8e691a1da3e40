#!/bin/bash
# ablation builds of conv_wino_vp (cross-compiled here): scratch/libvpdiag.so exports sqd_vp_diag_<mask>
cd "$(dirname "$0")"
SRC=../../squeezedet-pytorch_amd/csrc
VP=conv_wino_vp.hip
OBJS=""
for m in 0 1 2 4 8 16 31; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I$SRC -Wno-unused-result -Xclang -target-feature -Xclang -load-store-opt \
    -DSQD_VP_DIAG=$m -DSQD_VP_ENTRY=sqd_vp_diag_$m -Dconv_wino_vp_kernel=conv_wino_vp_kernel_d$m -c $VP -o vp_diag_$m.o 2> >(grep -v "recognized feature" >&2) || exit 1
  OBJS="$OBJS vp_diag_$m.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvpdiag.so $OBJS && rm -f $OBJS && echo built scratch/libvpdiag.so
