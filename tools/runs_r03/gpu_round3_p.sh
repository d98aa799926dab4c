#!/bin/bash
# round 3, GPU call P: Viterbi chain hops without the silent count increment (lane shift folded into the payload select),
# and wave priority for the longest windows -- parity, then A/B on one box
set -u
OUT=gpurun_out/r3p
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py tests/test_gpu_bench_parity.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 600 python tools/fuzz_viterbi.py 77 30 > $OUT/fuzz_viterbi.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_viterbi.log
for rep in 1 2; do
for v in head new prio8 prio4; do
  if [ $v = new ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3p/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
