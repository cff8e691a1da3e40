"""rocprofv3 kernel symbol -> the kernel name bench.py / ops.KernelTimer prints (used by scratch/traffic_aggregate.py and
tests/test_profiles.py to tie the committed rocprof summaries to the bench line)."""
import re


def short_name(k):
    short = k.split("(")[0].replace("void ", "").strip()
    m = re.match(r"conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), \d+>", short)
    if m: return f"conv_igemm<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)}>"
    # <TAPS, KC, MT, NT, WAVES, MINW, FUSE, WSTAT>: the name bench.py uses ignores MINW / WSTAT, FUSE = fused Fire expand
    # (rounds 3-4 had a ninth parameter, CHAIN = squeeze + expand1x1 in one launch; retired in round 5)
    m = re.match(r"conv_dma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), \d+, (true|false), (true|false)(, (true|false))?>", short)
    if m:
        if m.group(9) == "true": return f"fire_sq_e1<{m.group(4)}>"
        base = "fire_expand" if m.group(6) == "true" else "conv_dma"
        return f"{base}<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)},{m.group(5)}>"
    if short.startswith("conv_wino_sk_kernel"): return "conv_wino_sk"                  # balanced (stream-K) Winograd kernel, round 4
    if short.startswith("conv_wino_vs_kernel"): return "conv_wino_vs"                  # V-shared kernel for N <= 80 (ConvDet), round 5
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
    if short.startswith("wino_wgrad_group_kernel<"): return "conv_wgrad_wino_group"     # several layers of a backward stage in one launch (round 5)
    if short.startswith("conv_wgrad_group_kernel<"): return "conv_wgrad_group<1>"       # ... and the wide expand1x1 layers of a stage
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

