#!/bin/bash
# round 4, GPU call A: parity-agnostic register-resident Viterbi (one launch for mixed C9orf72 / FMR1 / HTT sub-batches), the
# round's host-side changes (one HIP runtime per process, validated edge lists, final gather) -- whole -m gpu suite, configs[1]/[3]/[4],
# bench under both HIP runtimes
set -u
OUT=gpurun_out/r4a
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 200 python tools/config_probe.py 4096 > $OUT/config.log 2>&1; echo "config rc=$?"; grep "configs\[" $OUT/config.log | cut -c1-260
timeout 300 python tools/mod_probe.py 4096 > $OUT/mod.log 2>&1; echo "mod rc=$?"; grep "mod=" $OUT/mod.log | cut -c1-260
for rt in auto system; do
  ( time STRQ_HIP_RUNTIME=$rt timeout 600 python bench.py --steps 9 --warmup 3 --no-cpu-baseline ) > $OUT/bench_$rt.json 2> $OUT/bench_$rt.err
  echo "bench $rt rc=$?"
  python - $OUT/bench_$rt.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok')}, d['stage_ms_per_step'], d['roofline']['frac'])
PY
done
