#!/bin/bash
# round 3, GPU call AB: the whole -m gpu suite and the default bench line on the committed code
set -u
OUT=gpurun_out/r3ab
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench rc=$?"; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3ab/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok','vs_cpu_baseline')}, d['stage_ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('useful_frac'))
PY
