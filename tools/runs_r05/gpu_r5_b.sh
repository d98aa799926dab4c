#!/bin/bash
# round 5, call b: coarse screen with the relaxed verdict / split heavy windows; what it decides on bench.py's clean and empirical reads
set -u
OUT=gpurun_out/r5b; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_screen.py -x -q > $OUT/tests_screen.log 2>&1; echo "screen tests rc=$?"; tail -5 $OUT/tests_screen.log
STRQ_DEBUG=1 timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --check 2 --leg-steps 3 --no-host-leg > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep -a "coarse screen\|screen verdict\|screen:" $OUT/bench.err | head -40
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5b/bench.json") if l.startswith("{")][-1])
    print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"])
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_ms", "frac", "useful_frac", "valu_insts_per_wave_step")})
    print("screen", d["screen"])
    for k, v in d.get("legs", {}).items():
        print(k, v["value"], v["ms_per_step"], v["stage_ms_per_step"], v["screen"], v.get("planted_count_recovered"), v.get("second_round_share"), v.get("value_no_screen"))
    print("check_ok", d["check_ok"], "host", d["host"])
except Exception as e:
    print("no line:", e)
PY
