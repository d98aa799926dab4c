#!/bin/bash
# round 3, GPU call AF: the register-resident Viterbi compiled with the default scheduler instead of max-ilp -- A/B
set -u
OUT=gpurun_out/r3af
mkdir -p $OUT
for rep in 1 2; do
for v in new g2def; do
  if [ $v = new ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 300 python bench.py --steps 5 --warmup 1 --batches 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3af/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), round(d['stage_ms_per_step']['viterbi'],2), d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
