#!/bin/bash
# usage: scratch/fetch_calib.sh <tag>  -> gpurun_out/fetch_calib_<tag>.txt : FETCH_SIZE / WRITE_SIZE per calibration kernel
TAG=${1:-r02}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/fetch_calib_$TAG; mkdir -p $OUT
cd /tmp
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o pmc -- $REPO/scratch/diag/fetch_calib > $OUT/out.txt 2> $OUT/stderr.log
cd $REPO
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
with open("gpurun_out/fetch_calib_$TAG.txt", "w") as o:
    o.write("kernel, launches, FETCH_SIZE raw KB per launch, raw bytes / bytes actually read (1 GiB)\n")
    for k, v in agg.items():
        kb = sum(v) / len(v)
        o.write(f"{k}, {len(v)}, {kb:.0f}, {kb * 1024 / 2**30:.3f}\n")
print(open("gpurun_out/fetch_calib_$TAG.txt").read())
PY
