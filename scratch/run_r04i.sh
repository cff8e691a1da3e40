#!/bin/bash
T=gpurun_out/r04i; mkdir -p $T
timeout -k 10 900 python -m pytest tests/test_training_gpu.py tests/test_checkpoint.py tests/test_surface_gpu.py tests/test_data_parallel_gpu.py tests/test_dropout_gpu.py -q -m gpu -x > $T/pytest_train.log 2>&1; echo "pytest rc $?"; tail -4 $T/pytest_train.log
export TMPDIR=/tmp
OUT=$PWD/$T/prof_train; mkdir -p $OUT
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $OLDPWD/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > $OUT/bench_under_prof.json 2> $OUT/stderr.log); echo "prof rc $?"
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $T/kernel_stats_train.csv
grep -c "at::native" $T/kernel_stats_train.csv; grep "at::native" $T/kernel_stats_train.csv | cut -c1-150 | head
head -12 $T/kernel_stats_train.csv | cut -c1-160
timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-pipeline > $T/bench_train.json 2> $T/bench_train.err; python -c "
import json; d=json.loads(open('$T/bench_train.json').read().strip().splitlines()[-1]); print('train ms', d.get('ms_per_step'), d.get('value'))"
