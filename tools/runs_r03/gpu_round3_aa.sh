#!/bin/bash
# round 3, GPU call AA: register-resident Viterbi, odd slots first so that their LDS exchange is in flight under the even tournaments -- parity, A/B
set -u
OUT=gpurun_out/r3aa
mkdir -p $OUT
STRQ_LIB=$PWD/tools/bin/lib_g2early.so timeout 900 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_bench_parity.py -m gpu -q -x > $OUT/tests_early.log 2>&1
echo "tests early rc=$?"; tail -2 $OUT/tests_early.log
STRQ_LIB=$PWD/tools/bin/lib_g2early.so timeout 600 python tools/fuzz_g2.py 11 30 > $OUT/fuzz_g2.log 2>&1; echo "fuzz_g2 rc=$?"; tail -1 $OUT/fuzz_g2.log
for rep in 1 2 3; do
for v in new g2early; do
  if [ $v = new ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3aa/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), round(d['stage_ms_per_step']['viterbi'],2), round(d['stage_ms_per_step']['forward_dp'],2), d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
