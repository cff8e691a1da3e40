#!/bin/bash
T=gpurun_out/r05c; mkdir -p $T
timeout -k 10 300 python -m pytest tests/test_lanes_gpu.py -x -q -m gpu > $T/pytest_lanes.log 2>&1; echo "pytest rc $?"; tail -5 $T/pytest_lanes.log
timeout -k 10 300 python scratch/lanes_probe.py > $T/probe_picked.log 2>&1; echo rc $?; cat $T/probe_picked.log | grep -v amdgpu.ids
