#!/bin/bash
T=gpurun_out/r05k; mkdir -p $T
timeout -k 10 300 python scratch/vp_check.py > $T/vp_check.log 2>&1; echo rc $?; grep -v amdgpu.ids $T/vp_check.log | tail -6
timeout -k 10 300 python scratch/diag/vp_diag.py > $T/vp_diag.log 2>&1; echo rc $?; grep -v amdgpu.ids $T/vp_diag.log
