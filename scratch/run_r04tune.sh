#!/bin/bash
T=gpurun_out/r04tune; mkdir -p $T
MODE=${1:-infer}
timeout -k 10 1080 python tools/tune_insitu.py --mode $MODE --out $T/tuning_insitu_$MODE.json > $T/tune_insitu_$MODE.log 2>&1; echo "tune $MODE rc $?"; tail -6 $T/tune_insitu_$MODE.log
