#!/bin/bash
T=gpurun_out/r04zz; mkdir -p $T
cd scratch/diag && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value stage_issue.hip -o stage_issue > ../../$T/build.log 2>&1; echo "build rc $?"
timeout -k 10 120 ./stage_issue > ../../$T/stage_issue.log 2>&1; echo "run rc $?"; cd ../..; cat $T/stage_issue.log
