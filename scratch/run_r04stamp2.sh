#!/bin/bash
T=gpurun_out/r04stamp2; mkdir -p $T
bash scratch/diag/wino_stamp.sh > $T/build.log 2>&1; echo "build rc $?"; tail -3 $T/build.log
timeout -k 10 300 python scratch/diag/run_wino_stamp.py > $T/wino_stamp.log 2>&1; echo "run rc $?"; tail -12 $T/wino_stamp.log
