#!/bin/bash
# round 4, GPU call B: (1) the forward DP's steady-state loop switched on the last-row register (every flank length on the fast loop),
# (2) register-resident Viterbi with relayed insert columns (107 instead of 117 / 125 VALU instructions per step outside the sweeps).
# Parity first (alignment, Viterbi, detect suites + fuzzers), then the flank-length sweep, configs, bench.
set -u
OUT=gpurun_out/r4b
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 300 python tools/fuzz_g2.py 41 40 > $OUT/fuzz_g2.log 2>&1; echo "fuzz_g2 rc=$?"; tail -2 $OUT/fuzz_g2.log
timeout 300 python tools/fuzz_align.py 43 300 > $OUT/fuzz_align.log 2>&1; echo "fuzz_align rc=$?"; tail -1 $OUT/fuzz_align.log
timeout 300 python tools/fuzz_detect.py 47 40 > $OUT/fuzz_detect.log 2>&1; echo "fuzz_detect rc=$?"; tail -1 $OUT/fuzz_detect.log
timeout 400 python tools/flank_sweep.py > $OUT/flank_sweep.md 2> $OUT/flank_sweep.err; echo "flank sweep rc=$?"; cat $OUT/flank_sweep.md
timeout 200 python tools/config_probe.py 4096 > $OUT/config.log 2>&1; echo "config rc=$?"; grep "configs\[" $OUT/config.log | cut -c1-260
timeout 300 python tools/mod_probe.py 4096 > $OUT/mod.log 2>&1; echo "mod rc=$?"; grep "mod=" $OUT/mod.log | cut -c1-260
( time timeout 600 python bench.py --steps 9 --warmup 3 --no-cpu-baseline ) > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok')}, d['stage_ms_per_step'], d['roofline']['frac'])
PY
