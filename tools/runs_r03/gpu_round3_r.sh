#!/bin/bash
# round 3, GPU call R: what a Viterbi time step waits for -- redundant LDS stores / LDS reads / float64 adds per step (A/B),
# and three waves per SIMD with 128-bit reads
set -u
OUT=gpurun_out/r3r
mkdir -p $OUT
for rep in 1 2; do
for v in new xst6 xst12 xrd8 xrd16 xva32 w12 w12p4; do
  if [ $v = new ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3r/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
