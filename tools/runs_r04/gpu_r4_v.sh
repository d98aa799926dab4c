#!/bin/bash
# round 4, GPU call V: the 32-bit screen (one add + one max3 per cell): instruction rates of the packed 16-bit integer ops (why the first
# version was slow), its tests, the probe, a short bench
set -u
OUT=gpurun_out/r4v
mkdir -p $OUT
timeout 120 tools/bin/ubench_pk16 > $OUT/ubench_pk16.log 2>&1; echo "ubench rc=$?"; cat $OUT/ubench_pk16.log
timeout 600 python -m pytest tests/test_gpu_screen.py -m gpu -x -q > $OUT/tests_screen.log 2>&1; echo "screen tests rc=$?"; tail -12 $OUT/tests_screen.log
timeout 300 python tools/screen_probe.py 1024 > $OUT/probe.log 2>&1; echo "probe rc=$?"; tail -8 $OUT/probe.log
( timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
PY
