#!/bin/bash
# round 4, GPU call D: two forward-DP kernel bodies (round 3's for launches of 870-row flanks, the switched loop for the rest),
# mark decodes with three emission counters in the payload, second-round accounting, degraded reads; reader-only probe.
set -u
OUT=gpurun_out/r4d
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
( time timeout 600 python bench.py --steps 9 --warmup 3 --no-cpu-baseline ) > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok')}, d['stage_ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])
PY
timeout 300 python tools/fuzz_g2.py 51 40 > $OUT/fuzz_g2.log 2>&1; echo "fuzz_g2 rc=$?"; tail -1 $OUT/fuzz_g2.log
timeout 300 python tools/fuzz_detect.py 53 30 > $OUT/fuzz_detect.log 2>&1; echo "fuzz_detect rc=$?"; tail -1 $OUT/fuzz_detect.log
timeout 400 python tools/flank_sweep.py > $OUT/flank_sweep.md 2> $OUT/flank_sweep.err; echo "flank sweep rc=$?"; cat $OUT/flank_sweep.md
timeout 200 python tools/config_probe.py 4096 > $OUT/config.log 2>&1; echo "config rc=$?"; grep "configs\[" $OUT/config.log | cut -c1-260
timeout 300 python tools/mod_probe.py 4096 > $OUT/mod.log 2>&1; echo "mod rc=$?"; grep "mod=" $OUT/mod.log | cut -c1-260
timeout 900 python tools/realism_bench.py > $OUT/realism.md 2> $OUT/realism.err; echo "realism rc=$?"; cat $OUT/realism.md
timeout 600 python tools/reader_probe.py 8192 50000 > $OUT/reader_probe.log 2>&1; echo "reader probe rc=$?"; cat $OUT/reader_probe.log
