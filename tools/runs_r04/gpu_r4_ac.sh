#!/bin/bash
# round 4, GPU call AC: the committed state -- whole GPU suite, smoke, degraded reads, kernel stats of the default line, the default bench line
set -u
OUT=gpurun_out/r4ac
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 500 python tools/realism_bench.py --reads 2048 0.0 0.5 1.0 1.5 > $OUT/realism_screen.md 2> $OUT/realism_screen.err; echo "realism rc=$?"; cat $OUT/realism_screen.md
P=gpurun_out/prof_r04s
mkdir -p $P
export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -o r04s -- python3 bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --check 0 > "$P/bench_kt.log" 2>&1; echo "kernel-trace pass rc=$?"
rm -f "$P"/kt/*_kernel_trace.csv "$P"/kt/*.db
( timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()}, d.get('host_inclusive_reads_per_s'))
r=d['roofline']; print({k: r.get(k) for k in ('kernel','avg_launch_ms','achieved','frac','useful_frac','with_windows','whole_read','traffic')})
PY
