#!/bin/bash
# round 3, GPU call Z: where the steady-state loop of the forward DP starts relative to a 64-byte boundary (16 placements) -- A/B on one box
set -u
OUT=gpurun_out/r3z
mkdir -p $OUT
for v in base pad0 pad1 pad2 pad3 pad4 pad5 pad6 pad7 pad8 pad9 pad10 pad11 pad12 pad13 pad14 pad15 base2; do
  if [ $v = base ] || [ $v = base2 ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --batches 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}.json 2> $OUT/bench_${v}.err; echo "$v rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3z/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), round(d['stage_ms_per_step']['forward_dp'],2), d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
