#!/bin/bash
# usage: scratch/run_steps.sh <logdir> "<name>|<timeout s>|<command>" ...
# Runs the steps in order, each under `timeout -k 10`, stdout/stderr to gpurun_out/<logdir>/<name>.log.  A step that fails is
# reported and the next one still runs; a step that TIMES OUT or is killed (exit 124 / 137) stops the sequence: no further
# GPU work behind a hung step.
DIR=gpurun_out/$1; shift
mkdir -p $DIR
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "=== [$name] (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 $secs bash -o pipefail -c "$cmd" > $DIR/$name.log 2>&1
  rc=$?
  echo "=== [$name] exit $rc after $(( $(date +%s) - start ))s"
  tail -n 6 $DIR/$name.log | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== [$name] timed out / killed: stopping here"; exit $rc; fi
done
exit 0
