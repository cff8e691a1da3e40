#!/bin/bash
T=gpurun_out/r04fuzz; mkdir -p $T
timeout -k 10 350 python tools/fuzz_stem.py 240 4 > $T/fuzz_stem.log 2>&1; echo "fuzz_stem rc $?"; tail -2 $T/fuzz_stem.log
