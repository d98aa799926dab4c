#!/bin/bash
# round 4, GPU call AF: the final committed state -- whole GPU suite, smoke, the default bench line
set -u
OUT=gpurun_out/r4af
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
( timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()}, d.get('host_inclusive_reads_per_s'))
r=d['roofline']; print({k: r.get(k) for k in ('kernel','avg_launch_ms','achieved','frac','useful_frac','with_windows','whole_read','traffic')})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
