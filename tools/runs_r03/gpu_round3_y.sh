#!/bin/bash
# round 3, GPU call Y: code-generation variants of the forward DP (same source, other scheduler / alignment options) -- A/B on one box
set -u
OUT=gpurun_out/r3y
mkdir -p $OUT
for v in base dpa dpb dpc dpe dpf dpg dph dpj dpk dpd dpi base2; do
  if [ $v = base ] || [ $v = base2 ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --batches 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}.json 2> $OUT/bench_${v}.err; echo "$v rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3y/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
