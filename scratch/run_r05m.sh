#!/bin/bash
T=gpurun_out/r05m; mkdir -p $T
timeout -k 10 300 python scratch/diag/vs_diag.py > $T/vs_variants.log 2>&1; echo rc $?; grep -v amdgpu.ids $T/vs_variants.log
