#!/bin/bash
T=gpurun_out/r05n; mkdir -p $T
timeout -k 10 400 python scratch/lanes_stress.py > $T/lanes_stress.log 2>&1; echo "stress rc $?"; grep -v amdgpu $T/lanes_stress.log | tail -4


