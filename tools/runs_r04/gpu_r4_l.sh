#!/bin/bash
# round 4, GPU call L: modification strings gathered densely before the read-back; CPU quota in the bench line; whole suite
set -u
OUT=gpurun_out/r4l
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 300 python tools/mod_probe.py 4096 > $OUT/mod.log 2>&1; echo "mod rc=$?"; grep "mod=" $OUT/mod.log | cut -c1-260
timeout 300 python tools/fuzz_detect.py 61 40 > $OUT/fuzz_detect.log 2>&1; echo "fuzz_detect rc=$?"; tail -1 $OUT/fuzz_detect.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench rc=$?"; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4l/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok','vs_cpu_baseline')}, d['stage_ms_per_step'], d['roofline']['frac'])
c=d['cpu_baseline']; print(c['value'], c['cores'], c.get('cpu_quota_cores'), [(r['workers'], round(r['reads_per_s'],3)) for r in c['sweep']], c['lut_variant']['value'])
PY
