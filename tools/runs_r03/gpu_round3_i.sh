#!/bin/bash
# round 3, GPU call I: Viterbi time loop specialised once per window (fast / general emissions) -- parity and A/B on one box
set -u
OUT=gpurun_out/r3i
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for rep in 1 2 3; do
  STRQ_LIB=$PWD/tools/bin/lib_head.so timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 1 > $OUT/bench_head_$rep.json 2> $OUT/bench_head_$rep.err; echo "head rc=$?"
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_new_$rep.json 2> $OUT/bench_new_$rep.err; echo "new rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 900 python tools/mod_probe.py 4096 > $OUT/mod_probe.log 2>&1; echo "mod_probe rc=$?"; tail -3 $OUT/mod_probe.log
