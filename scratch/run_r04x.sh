#!/bin/bash
T=gpurun_out/r04x; mkdir -p $T
timeout -k 10 400 python tools/fuzz_bridge.py 240 1 > $T/fuzz_bridge.log 2>&1; echo "fuzz rc $?"; tail -5 $T/fuzz_bridge.log
