"""Aggregate the PMC passes of scratch/traffic.sh (FETCH_SIZE, WRITE_SIZE; per mode) into gpurun_out/traffic_<tag>.json: bytes per
launch per kernel under the names bench.py uses, with the launch set recorded; inference entries at the top level, the training
step's under "train".  usage: python scratch/traffic_aggregate.py <tag> [infer] [train]"""
import sys
TAG = sys.argv[1]
MODES = sys.argv[2:] or ['infer']
import csv, glob, collections, json, re

def short_name(k):
    short = k.split("(")[0].replace("void ", "").strip()
    m = re.match(r"conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), \d+>", short)
    if m: return f"conv_igemm<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)}>"
    # <TAPS, KC, MT, NT, WAVES, MINW, FUSE, WSTAT>: the name bench.py uses ignores MINW / WSTAT, FUSE = fused Fire expand
    # (a ninth parameter, CHAIN = squeeze + expand1x1 in one launch, since round 3)
    m = re.match(r"conv_dma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), \d+, (true|false), (true|false)(, (true|false))?>", short)
    if m:
        if m.group(9) == "true": return f"fire_sq_e1<{m.group(4)}>"
        base = "fire_expand" if m.group(6) == "true" else "conv_dma"
        return f"{base}<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)},{m.group(5)}>"
    if short.startswith("conv_wino_sk_kernel"): return "conv_wino_sk"                  # balanced (stream-K) Winograd kernel, round 4
    m = re.match(r"conv_wino_kernel<(\d+), (\d+)>", short)
    if m: return f"conv_wino<{m.group(1)},{m.group(2)}>"
    m = re.match(r"conv_ws_kernel<(\d+), (\d+), \d+>", short)
    if m: return f"conv_ws<{m.group(1)},{m.group(2)}>"
    m = re.match(r"conv_wino_pipe_kernel<(\d+), (\d+), (true|false)", short)
    if m: return f"conv_wino_{'us' if m.group(3) == 'true' else 'dp'}<{m.group(1)},{m.group(2)}>"
    m = re.match(r"fire_poolbridge16_kernel<\d+, \d+, (true|false)>", short)           # <NSQ, NCH, SAVE>: SAVE = the training form (round 4)
    if m: return "fire_pool_bridge_save" if m.group(1) == "true" else "fire_pool_bridge"
    if short.startswith("fire_poolbridge16_kernel"): return "fire_pool_bridge"
    m = re.match(r"fire_bridge16_kernel<\d+, (\d+), \d+>", short)                      # <NSQ, MODE, NCH>: 0 = plain fused expand, 1 = bridge, 2 = storing bridge
    if m: return {"0": "fire_wino16", "1": "fire_bridge", "2": "fire_bridge_save"}[m.group(1)]
    if short.startswith("fire_bridge_kernel<"): return "fire_bridge"
    m = re.match(r"(maxpool_fwd|maxpool_bwd)_kernel", short)
    if m: return m.group(1)
    m = re.match(r"wino_wgrad_kernel<", short)
    if m: return "conv_wgrad_wino"
    m = re.match(r"conv_wgrad_kernel<(\d+), \d+, \d+, \d+(, (true|false))?>", short)
    if m: return "squeeze_bwd" if m.group(3) == "true" else f"conv_wgrad<{m.group(1)}>"
    m = re.match(r"stem_wave_kernel<\d+, \d+, (true|false), (\d+)>", short)           # the wave-autonomous 3x3 stem (round 3): same bench name as the
    if m:                                                                             # workgroup kernel; SQ > 0 = with the first Fire's squeeze
        return ("stem_pool_sq_train<3>" if m.group(1) == "true" else "stem_pool_sq<3>") if m.group(2) != "0" else "stem_pool<3>"
    if short.startswith("stem_wgrad_gather_kernel<"): return "stem_wgrad_pooled<3>"     # ... and so does the gather form of its weight gradient
    m = re.match(r"stem_wgrad_pooled_kernel<(\d+),", short)
    if m: return f"stem_wgrad_pooled<{m.group(1)}>"
    if short.startswith("wgrad_reduce_batched_kernel"): return "wgrad_reduce_batched"
    m = re.match(r"(stem_pool|stem_conv|stem_wgrad)_kernel<(\d+),", short)
    if m: return f"{m.group(1)}<{m.group(2)}>"
    return {"maxpool_fwd_kernel": "maxpool_fwd", "maxpool_bwd_kernel": "maxpool_bwd", "detect_kernel": "detect", "detect_kernel(DetArgs)": "detect"}.get(short, short)

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
