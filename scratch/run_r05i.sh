#!/bin/bash
T=gpurun_out/r05i; mkdir -p $T
PYTHONFAULTHANDLER=1 timeout -k 10 300 python bench.py --gpus 1 --force-dist --mode train --steps 3 --warmup 2 --no-cpu-baseline > $T/b3.json 2> $T/b3.err; echo "dist graph rc $?"; tail -3 $T/b3.err
timeout -k 10 900 python -m pytest tests/test_surface_gpu.py tests/test_data_parallel_gpu.py tests/test_kernels_gpu.py tests/test_inference_gpu.py tests/test_dropout_gpu.py -q -m gpu > $T/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $T/pytest.log
