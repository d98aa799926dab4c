#!/bin/bash
# round 3, GPU call K: one LDS wait per gather instead of one per read (Viterbi step) -- A/B on one box
set -u
OUT=gpurun_out/r3k
mkdir -p $OUT
for rep in 1 2 3; do
for v in head onewait; do
  export STRQ_LIB=$PWD/tools/bin/lib_$v.so
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3k/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
