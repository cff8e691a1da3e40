# usage: bash scratch/run_ab_env.sh "<bench args>" VAR=a VAR=b ...   -- the bench once per environment setting, back to back
ARGS="$1"; shift
mkdir -p gpurun_out/ab
for kv in "$@" "$@"; do
  tag=$(echo $kv | tr '=' '_')
  env $kv timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-pipeline > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { echo "$kv failed"; tail -3 gpurun_out/ab/$tag.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/$tag.json").read().strip().splitlines()[-1])
t=d.get("train") or d
print("$kv", "infer", d.get("ms_per_step") if "train" in d else None, "train", t.get("ms_per_step"), t.get("repeat_window_ms_per_step"))
PY
done
