#!/bin/bash
# round 5, call z2: read-length threshold of the coarse screen at 28 k samples, cost-model pause thresholds: short reads, the line with legs, screen tests
set -u
OUT=gpurun_out/r5z2; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/minn_probe.py 4096 > $OUT/minn.txt 2>&1; grep -v amdgpu.ids $OUT/minn.txt | grep "default"
timeout 1500 python -m pytest tests/test_gpu_screen.py tests/test_gpu_bench_parity.py -x -q --deselect tests/test_gpu_bench_parity.py::test_bench_four_and_eight_ranks_on_one_gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 1 --leg-steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5z2/bench.json") if l.startswith("{")][-1])
print("value", d["value"], d["stage_ms_per_step"], "fine", d["value_fine_screen"], "none", d["value_no_screen"])
v = d["legs"]["degraded"]
print("degraded", v["value"], v["ms_per_step"], v["stage_ms_per_step"], v["screen"]["mode"], v.get("second_round_share"), v["check"]["all_fields_equal"])
PY
