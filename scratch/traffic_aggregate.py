"""Aggregate the PMC passes of scratch/traffic.sh (FETCH_SIZE, WRITE_SIZE; per mode) into gpurun_out/traffic_<tag>.json: bytes per
launch per kernel under the names bench.py uses, with the launch set recorded; inference entries at the top level, the training
step's under "train".  usage: python scratch/traffic_aggregate.py <tag> [infer] [train]"""
import sys
TAG = sys.argv[1]
MODES = sys.argv[2:] or ['infer']
import csv, glob, collections, json, re

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from kernel_names import short_name  # noqa: E402


def aggregate(mode):
    res = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": collections.Counter()})
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("gpurun_out/pmc_%s_%s_%s/**/*counter_collection.csv" % (TAG, mode, C), recursive=True)[0]
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != C: continue
            k = short_name(r["Kernel_Name"])
            res[k][C] += float(r["Counter_Value"]); res[k]["n"][C] += 1
    out = {}
    # steps the profiled process launched eagerly (bench.py prints it): launches per step = launches profiled / steps
    line = json.loads([l for l in open("gpurun_out/pmc_%s_%s_FETCH_SIZE/out.json" % (TAG, mode)) if l.startswith("{")][-1])
    steps = line["eager_steps_launched"] if mode == "infer" else line.get("eager_steps_launched", line.get("train", {}).get("eager_steps_launched"))
    for k, v in res.items():
        nf, nw = max(v["n"]["FETCH_SIZE"], 1), max(v["n"]["WRITE_SIZE"], 1)
        fetch_kb, write_kb = v["FETCH_SIZE"] / nf, v["WRITE_SIZE"] / nw
        out[k] = {"launches_profiled": nf, "fetch_size_kb_raw": round(fetch_kb, 1), "write_size_kb": round(write_kb, 1),
                  "hbm_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024),
                  "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); average over all launches of this kernel"}
        if nf % steps == 0:
            out[k]["launches_per_step"] = nf // steps      # the launch set this average was taken over (checked by tests/test_profiles.py)
    what = "inference" if mode == "infer" else "training step: fwd + loss + bwd + clip + SGD"
    out["_meta"] = {"workload": f"python bench.py --mode {mode} --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-pipeline (SqueezeDet bs=20 1248x384 {what})",
                    "eager_steps_profiled": steps,
                    "counters": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KB units; fabric-side L2 request counters: Infinity-Cache hits are included (MI355X_MICROARCH.md, HBM section), so these are L2<->fabric bytes, an upper bound of HBM bytes"}
    print(f"--- {mode}: {steps} eager steps profiled")
    for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1].get("launches_per_step", 0))[:24]:
        print(f'{k[:56]:56s} x{v.get("launches_per_step", "?"):>3} {v["hbm_bytes_per_launch"]/1e6:9.1f} MB/launch  (fetch raw {v["fetch_size_kb_raw"]/1e3:.1f} MB, write {v["write_size_kb"]/1e3:.1f} MB)')
    return out


final = {}
if "infer" in MODES:
    final = aggregate("infer")
else:
    try:
        final = {k: v for k, v in json.load(open("profiles/traffic.json")).items() if k != "train"}
    except OSError:
        final = {}
if "train" in MODES:
    final["train"] = aggregate("train")
else:
    try:
        final["train"] = json.load(open("profiles/traffic.json"))["train"]
    except (OSError, KeyError):
        pass
json.dump(final, open("gpurun_out/traffic_%s.json" % TAG, "w"), indent=1, sort_keys=True)
