mkdir -p gpurun_out/r03z
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_inference_gpu.py tests/test_surface_gpu.py -x -q -m gpu -k "stem or golden or launch_plan" > gpurun_out/r03z/pytest_sq.log 2>&1 || { tail -30 gpurun_out/r03z/pytest_sq.log; exit 1; }
tail -2 gpurun_out/r03z/pytest_sq.log
for v in sdma1 sdma0; do
  if [ $v = sdma0 ]; then export HSA_ENABLE_SDMA=0; else unset HSA_ENABLE_SDMA; fi
  timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline > gpurun_out/r03z/bench_sq_$v.json 2> gpurun_out/r03z/bench_sq_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03z/bench_sq_$v.json").read().strip().splitlines()[-1])
k=d["kernels_event_profile"]
print("$v", d["value"], d["ms_per_step"], "parity", d["parity"]["ok"], d["parity"]["pred_max_abs_err"], {n:x["ms_per_step"] for n,x in k.items() if "stem" in n or n=="conv_ws<1,4>"}, "pipeline", d["pipeline"]["value"], d["pipeline"]["ms_per_step"], d["pipeline"]["h2d_alone"]["gbs"])
PY
done
