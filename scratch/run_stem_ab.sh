mkdir -p gpurun_out/r03z
timeout -k 10 300 python scratch/stem_wave_check.py > gpurun_out/r03z/stem_wave3.log 2>&1
grep -v amdgpu.ids gpurun_out/r03z/stem_wave3.log | grep -v "   "
for v in 0 2 3 4; do
  SQD_STEM_WAVE=$v timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline > gpurun_out/r03z/bench_infer_sw$v.json 2> gpurun_out/r03z/bench_infer_sw$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03z/bench_infer_sw$v.json").read().strip().splitlines()[-1])
print($v, d["value"], d["ms_per_step"], {k:x["ms_per_step"] for k,x in d["kernels_event_profile"].items() if "stem" in k})
PY
done
