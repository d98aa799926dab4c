#!/bin/bash
# round 3, GPU call AD: trace pass with compare + add-with-carry trace codes -- parity (alignment, detect, shim), A/B against the select form
set -u
OUT=gpurun_out/r3ad
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_gpu_shim.py tests/test_gpu_detect.py tests/test_gpu_bench_parity.py tests/test_gpu_limits.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 600 python tools/fuzz_align.py 101 60 > $OUT/fuzz_align.log 2>&1; echo "fuzz_align rc=$?"; tail -1 $OUT/fuzz_align.log
for rep in 1 2; do
for v in trsel new; do
  if [ $v = new ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3ad/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
