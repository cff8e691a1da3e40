mkdir -p gpurun_out/r03z
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wino or conv_fwd or fire" > gpurun_out/r03z/pytest_pix.log 2>&1 || { tail -5 gpurun_out/r03z/pytest_pix.log; exit 1; }
tail -2 gpurun_out/r03z/pytest_pix.log
for v in pix1 pix0 pix1b pix0b; do
  if [ "${v:0:4}" = pix0 ]; then export SQD_HIP_LIBRARY=$PWD/scratch/ab/libsqdhip_pix0.so; else unset SQD_HIP_LIBRARY; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pipeline > gpurun_out/r03z/bench_$v.json 2> gpurun_out/r03z/bench_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03z/bench_$v.json").read().strip().splitlines()[-1])
k=d["kernels_event_profile"]; t=d["train"]["kernels_event_profile"]
print("$v", d["ms_per_step"], d["train"]["ms_per_step"], "wino24", k["conv_wino<2,4>"]["ms_per_step"], "us24", k["conv_wino_us<2,4>"]["ms_per_step"], "us28", k["conv_wino_us<2,8>"]["ms_per_step"], "| train wino24", t["conv_wino<2,4>"]["ms_per_step"], "us14", t["conv_wino_us<1,4>"]["ms_per_step"], "us28", t["conv_wino_us<2,8>"]["ms_per_step"])
PY
done
