TAG=${1:-r03zz}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/$TAG/pytest_gpu_final.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/$TAG/pytest_gpu_final.log
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench_default_final.json 2> gpurun_out/$TAG/bench_default_final.err; echo "bench rc $?"
timeout -k 10 400 python bench.py --arch squeezedetplus --batch 16 --no-cpu-baseline --no-pipeline > gpurun_out/$TAG/bench_plus_final.json 2> gpurun_out/$TAG/bench_plus_final.err; echo "bench plus rc $?"
python - <<PY
import json
for f in ("bench_default_final", "bench_plus_final"):
    d=json.loads(open(f"gpurun_out/$TAG/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("train", {}).get("value"), d.get("train", {}).get("ms_per_step"), (d.get("parity") or {}).get("ok"))
PY
