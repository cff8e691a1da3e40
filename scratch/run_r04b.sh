#!/bin/bash
T=gpurun_out/r04b; mkdir -p $T
V="0,1000,2 0,1300,2 0,1700,2 1,1000,2 2,1000,2 3,1000,2 4,1000,2 4,1300,2 4,1700,2 6,1000,2 8,1000,2"
timeout -k 10 300 python tools/sk_bench.py $V > $T/sk_bench.log 2>&1; echo "rc $?"; cat $T/sk_bench.log
echo "---- U_FIRST build"
SQD_HIP_LIBRARY=$PWD/squeezedet-pytorch_amd/csrc/libsqdhip_uf.so timeout -k 10 300 python tools/sk_bench.py $V > $T/sk_bench_uf.log 2>&1; echo "rc $?"; cat $T/sk_bench_uf.log
