#!/bin/bash
# round 4, GPU call X: the whole GPU suite with the screen in the default path, then the default bench line
set -u
OUT=gpurun_out/r4x
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -12 $OUT/tests.log
( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
r=d['roofline']; print({k: r[k] for k in ('kernel','avg_launch_ms','achieved','frac','useful_frac','with_windows','whole_read','window_columns_over_columns_of_the_reads')}); print(r['exact_pass'])
PY
