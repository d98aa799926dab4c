#!/bin/bash
# round 4, GPU call AA: the committed state -- whole GPU suite, smoke, degraded reads with / without the screen, the other configs, the default bench line
set -u
OUT=gpurun_out/r4aa
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -10 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 400 python tools/realism_bench.py --reads 2048 0.0 0.5 1.0 > $OUT/realism_screen.md 2> $OUT/realism_screen.err; echo "realism rc=$?"; cat $OUT/realism_screen.md
timeout 300 python tools/config_probe.py 4096 > $OUT/config_screen.log 2>&1; grep "configs\[" $OUT/config_screen.log | cut -c1-230
( timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()}, d.get('host_inclusive_reads_per_s'))
r=d['roofline']; print({k: r.get(k) for k in ('kernel','avg_launch_ms','achieved','frac','useful_frac','with_windows','whole_read','traffic')})
print(d.get('cpu_baseline'))
PY
