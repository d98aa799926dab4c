#!/bin/bash
# round 4, GPU call AB: second stream for the bulk launch behind the screen + the last window absorbing extra candidates: screen / detect / bench-parity tests,
# degraded reads (with STRQ_ONE_STREAM=1 as the A/B), a short bench
set -u
OUT=gpurun_out/r4ab
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_screen.py tests/test_gpu_detect.py tests/test_gpu_bench_parity.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -6 $OUT/tests.log
timeout 400 python tools/realism_bench.py --reads 2048 0.0 0.5 1.0 > $OUT/realism_two_streams.md 2> $OUT/realism.err; echo "realism rc=$?"; cat $OUT/realism_two_streams.md
STRQ_ONE_STREAM=1 timeout 400 python tools/realism_bench.py --reads 2048 0.5 1.0 > $OUT/realism_one_stream.md 2> $OUT/realism1.err; echo "realism (one stream) rc=$?"; cat $OUT/realism_one_stream.md
( timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
PY
