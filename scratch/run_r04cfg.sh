#!/bin/bash
T=gpurun_out/r04cfg; mkdir -p $T
timeout -k 10 300 python scratch/diag/convdet_cfgs.py > $T/cfgs.log 2>&1; echo "rc $?"; tail -8 $T/cfgs.log
