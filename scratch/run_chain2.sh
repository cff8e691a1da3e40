mkdir -p gpurun_out/r03c gpurun_out/r03zz
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fire_sq_e1" > gpurun_out/r03c/pytest_chain_kernel.log 2>&1 || { tail -30 gpurun_out/r03c/pytest_chain_kernel.log; exit 1; }
tail -2 gpurun_out/r03c/pytest_chain_kernel.log
timeout -k 10 500 python scratch/fuzz_stem.py 150 1 > gpurun_out/r03zz/fuzz_stem.log 2>&1 || { tail -5 gpurun_out/r03zz/fuzz_stem.log; exit 1; }
tail -2 gpurun_out/r03zz/fuzz_stem.log
bash scratch/run_ab_env.sh "--mode both" SQD_FUSE_SQ_E1=0 SQD_FUSE_SQ_E1=1
python - <<PY
import json
for v in (0, 1):
    d=json.loads(open(f"gpurun_out/ab/SQD_FUSE_SQ_E1_{v}.json").read().strip().splitlines()[-1])
    print(v, "infer", d["ms_per_step"], {k:(x["ms_per_step"], x["launches_per_step"]) for k,x in d["kernels_event_profile"].items() if k.startswith(("fire_sq","conv_dma<1","conv_ws"))})
    print(v, "train", d["train"]["ms_per_step"], {k:(x["ms_per_step"], x["launches_per_step"]) for k,x in d["train"]["kernels_event_profile"].items() if k.startswith(("fire_sq",))})
PY
