#!/bin/bash
T=gpurun_out/r04t; mkdir -p $T
bash scratch/diag/ww_stamp.sh > $T/build.log 2>&1; echo "build rc $?"
timeout -k 10 300 python scratch/diag/run_ww_stamp.py > $T/ww_stamp.log 2>&1; echo "run rc $?"; cat $T/ww_stamp.log | tail -30
