#!/bin/bash
# round 3, GPU call X: forward DP with the DPP zero fill for the free top row (two v_mov less per wave-step) -- parity, A/B
set -u
OUT=gpurun_out/r3x
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_bench_parity.py tests/test_gpu_shim.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
for rep in 1 2 3; do
for v in nozfill new; do
  if [ $v = new ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3x/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
