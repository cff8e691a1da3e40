#!/bin/bash
T=gpurun_out/r05e; mkdir -p $T
timeout -k 10 300 python scratch/vs_check.py > $T/vs_check.log 2>&1; echo rc $?; grep -v amdgpu.ids $T/vs_check.log
