#!/bin/bash
# round 3, GPU call H: general (CSR) Viterbi kernel for models beyond the lane layouts
set -u
OUT=gpurun_out/r3h
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py tests/test_gpu_limits.py -m gpu -q -x --durations=8 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -14 $OUT/tests.log
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3h/bench.json').read().strip().splitlines()[-1])
print(round(d['value'],1), d['stage_ms_per_step'], d['check_ok'])
PY
