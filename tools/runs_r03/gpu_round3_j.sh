#!/bin/bash
# round 3, GPU call J: 96-bit cell stores, no count arithmetic on silent states (experiment) -- parity and A/B on one box
set -u
OUT=gpurun_out/r3j
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for rep in 1 2 3; do
for v in head default nosc; do
  if [ $v = default ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 1 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
unset STRQ_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
