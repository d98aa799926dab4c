#!/bin/bash
# round 3, GPU call A: the whole -m gpu suite, the R = 14 / R = 15 A/B of the forward DP on one box, the default bench line
set -u
mkdir -p gpurun_out/r3a
OUT=gpurun_out/r3a
nproc > $OUT/host.txt; free -g >> $OUT/host.txt; lscpu | head -20 >> $OUT/host.txt
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/tests.log 2>&1
echo "tests rc=$?"
tail -5 $OUT/tests.log
for rep in 1 2; do
  STRQ_NO_R14=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 0 > $OUT/bench_r15_$rep.json 2> $OUT/bench_r15_$rep.err
  echo "R15 rc=$?"
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 0 > $OUT/bench_r14_$rep.json 2> $OUT/bench_r14_$rep.err
  echo "R14 rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/bench_r1*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), d['stage_ms_per_step'], d['roofline']['kernel'], d['roofline']['overlap_columns_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench rc=$?"
tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3a/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok','vs_cpu_baseline')})
print(d['cpu_baseline'])
print(d['host'])
PY
