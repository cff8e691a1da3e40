#!/bin/bash
# usage (on the GPU box, repo root): bash scratch/run_profiles.sh <tag>  -- the round's evidence in one call:
# full -m gpu log, default bench line, the serial bench line, kernel traces (serial inference, default inference, training),
# FETCH/WRITE PMC passes of both modes
TAG=${1:-r05z}
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/$TAG/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --inflight 1 --no-pipeline > gpurun_out/$TAG/bench_serial.json 2> gpurun_out/$TAG/bench_serial.err; echo "bench serial rc $?"
bash scratch/prof.sh $TAG _serial --inflight 1 --no-pipeline > gpurun_out/$TAG/prof_infer_serial.log 2>&1; echo "prof serial rc $?"
bash scratch/prof.sh $TAG "" > gpurun_out/$TAG/prof_infer.log 2>&1; echo "prof rc $?"
bash scratch/prof_train.sh $TAG > gpurun_out/$TAG/prof_train.log 2>&1; echo "prof_train rc $?"
SQD_BENCH_ARGS="--gpus 1 --force-dist" bash scratch/prof_train.sh $TAG _dist > gpurun_out/$TAG/prof_train_dist.log 2>&1; echo "prof_train_dist rc $?"
bash scratch/traffic.sh $TAG "infer train" > gpurun_out/$TAG/traffic.log 2>&1; echo "traffic rc $?"
tail -30 gpurun_out/$TAG/traffic.log
