#!/bin/bash
# usage: scratch/pmc.sh <tag> "<counters>"   -> gpurun_out/pmc_<tag>.csv (per-kernel averages)
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp
timeout -k 10 150 rocprofv3 --pmc $@ --output-format csv -d $OUT -o pmc -- python3 ${SQD_PMC_PROG:-$REPO/bench.py --mode infer --steps 3 --warmup 2 --no-cpu-baseline --no-graph} $SQD_PMC_ARGS > $OUT/out.json 2> $OUT/stderr.log
cd $REPO
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    k = r["Kernel_Name"][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
seen=set()
for r in rows:
    key=(r["Dispatch_Id"]); 
    if key in seen: continue
    seen.add(key); cnt[r["Kernel_Name"][:44]] += 1
names = sorted({c for v in agg.values() for c in v})
with open("gpurun_out/pmc_$TAG.csv","w") as o:
    o.write("kernel,calls," + ",".join(names) + "\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(names[0], 0)):
        o.write(k.replace(",", ";") + f",{cnt[k]}," + ",".join(f"{v.get(n,0)/max(cnt[k],1):.4g}" for n in names) + "\n")
print(open("gpurun_out/pmc_$TAG.csv").read()[:6000])
PY
