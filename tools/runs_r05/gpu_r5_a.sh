#!/bin/bash
# round 5, call a: the coarse screen (align_screen2_kernel) for the first time on the GPU -- its tests, the parity of bench.py's own reads,
# then the default line with the new A/B legs
set -u
OUT=gpurun_out/r5a; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_screen.py -x -q > $OUT/tests_screen.log 2>&1; echo "screen tests rc=$?"; tail -5 $OUT/tests_screen.log
timeout 1200 python -m pytest tests/test_gpu_bench_parity.py -x -q -k "through_the_screen" > $OUT/tests_parity.log 2>&1; echo "parity rc=$?"; tail -5 $OUT/tests_parity.log
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --check 2 --leg-steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5a/bench.json") if l.startswith("{")][-1])
    print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"])
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_ms", "frac", "useful_frac", "valu_insts_per_wave_step")})
    print("screen", d["screen"])
    for k, v in d.get("legs", {}).items():
        print(k, v["value"], v["ms_per_step"], v["stage_ms_per_step"], v["screen"], v.get("planted_count_recovered"), v.get("second_round_share"))
    print("check_ok", d["check_ok"], "host", d["host"])
except Exception as e:
    print("no line:", e)
PY
