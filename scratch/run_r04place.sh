#!/bin/bash
T=gpurun_out/r04place; mkdir -p $T
cd scratch/diag && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value placement.hip -o placement > ../../$T/build.log 2>&1; echo "build rc $?"
for cfg in "80 512" "64 512" "48 512" "48 768" "16 512" "80 256" "80 1024"; do timeout -k 10 60 ./placement $cfg >> ../../$T/placement.log 2>&1; done; cd ../..; cat $T/placement.log
