#!/bin/bash
# round 5, call s: the verdict rules of the coarse screen after r5r (pause when its two looks leave too much), screen + detect tests, the line with legs
set -u
OUT=gpurun_out/r5t; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests/test_gpu_screen.py tests/test_gpu_detect.py tests/test_gpu_bench_parity.py -x -q --deselect tests/test_gpu_bench_parity.py::test_bench_four_and_eight_ranks_on_one_gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
STRQ_DEBUG=1 timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg --check 1 --leg-steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep -a "coarse screen" $OUT/bench.err | tail -4 | cut -c1-330
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5t/bench.json") if l.startswith("{")][-1])
print("value", d["value"], d["stage_ms_per_step"]); print("fine leg", d["legs"]["fine_screen"]["value"], d["legs"]["fine_screen"]["stage_ms_per_step"], d["legs"]["fine_screen"]["roofline"]["kernel"], d["legs"]["fine_screen"]["roofline"]["frac"])
v = d["legs"]["degraded"]
print("degraded", v["value"], v["ms_per_step"], v["stage_ms_per_step"], v["screen"]["mode"], v.get("planted_count_recovered"), v.get("second_round_share"), v.get("value_no_screen"), v["check"]["all_fields_equal"])
PY
