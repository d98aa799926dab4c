#!/bin/bash
# round 4, GPU call G (last): rocprofv3 evidence (r04), the default bench line as the driver runs it (CPU-baseline sweep included),
# the whole -m gpu suite and smoke on the committed code
set -u
OUT=gpurun_out/r4g
mkdir -p $OUT
bash tools/collect_profiles.sh r04 > $OUT/collect.log 2>&1; echo "collect rc=$?"; tail -4 $OUT/collect.log | cut -c1-300
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench rc=$?"; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4g/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok','vs_cpu_baseline')}, d['stage_ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('useful_frac'))
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['sweep'], d['cpu_baseline']['lut_variant']['value'])
PY
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
