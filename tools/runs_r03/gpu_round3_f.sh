#!/bin/bash
# round 3, GPU call F: the whole -m gpu suite at the consolidated state, rocprofv3 evidence (r03), the default bench line, count from gzip files
set -u
OUT=gpurun_out/r3f
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
bash tools/collect_profiles.sh r03 > $OUT/collect.log 2>&1; echo "collect rc=$?"; tail -12 $OUT/collect.log | cut -c1-300
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench rc=$?"; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3f/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok','vs_cpu_baseline')}, d['stage_ms_per_step'], d['roofline']['kernel'])
PY
timeout 900 python tools/cli_probe.py 8192 50000 --t 16 --compression gzip > $OUT/cli_probe_50k_gzip.log 2>&1; echo "cli_probe gzip rc=$?"; tail -3 $OUT/cli_probe_50k_gzip.log
timeout 900 python tools/cli_probe.py 8192 50000 --t 48 --compression gzip > $OUT/cli_probe_50k_gzip48.log 2>&1; echo "cli_probe gzip t48 rc=$?"; tail -3 $OUT/cli_probe_50k_gzip48.log
timeout 900 python tools/cli_probe.py 16384 50000 --t 8 > $OUT/cli_probe_50k.log 2>&1; echo "cli_probe rc=$?"; tail -3 $OUT/cli_probe_50k.log
