#!/bin/bash
T=gpurun_out/r04n; mkdir -p $T
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "pool_bridge or plan_refresh" > $T/pytest_new.log 2>&1; echo "pytest new rc $?"; tail -15 $T/pytest_new.log
