#!/bin/bash
# round 3, GPU call Q: 96-bit cell reads; three waves per SIMD (167 VGPRs) with and without priority for the longest windows
set -u
OUT=gpurun_out/r3q
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_bench_parity.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
STRQ_LIB=$PWD/tools/bin/lib_w12p4.so timeout 900 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_bench_parity.py -m gpu -q -x > $OUT/tests_w12.log 2>&1
echo "tests w12p4 rc=$?"; tail -2 $OUT/tests_w12.log
for rep in 1 2; do
for v in rd128 new w12 w12p4 w12p8; do
  if [ $v = new ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3q/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
