#!/bin/bash
T=gpurun_out/r05i; mkdir -p $T; rm -f $T/f*.err
for i in $(seq 1 24); do
timeout -k 10 200 python bench.py --gpus 1 --force-dist --mode train --steps 3 --warmup 2 --no-cpu-baseline > $T/f$i.json 2> $T/f$i.err; echo -n "$? "
done
echo; echo "watchdog aborts: $(grep -l 'capturing stream' $T/f*.err | wc -l)"
