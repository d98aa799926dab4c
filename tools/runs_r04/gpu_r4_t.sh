#!/bin/bash
# round 4, GPU call T: first run of the upper-bound screen (csrc/screen_kernels.hip): its own tests, the alignment / detect parity suites
# with the screen in the path, a short bench with and without it
set -u
OUT=gpurun_out/r4t
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_screen.py -m gpu -x -q > $OUT/tests_screen.log 2>&1; echo "screen tests rc=$?"; tail -15 $OUT/tests_screen.log
for v in screen noscreen; do
  if [ $v = noscreen ]; then export STRQ_NO_SCREEN=1; else unset STRQ_NO_SCREEN; fi
  ( timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  echo "bench $v rc=$?"; tail -3 $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
except Exception as e:
    print(sys.argv[2], 'no bench line', e)
PY
done
unset STRQ_NO_SCREEN
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_detect.py -m gpu -x -q > $OUT/tests_align_detect.log 2>&1; echo "align+detect tests rc=$?"; tail -8 $OUT/tests_align_detect.log
