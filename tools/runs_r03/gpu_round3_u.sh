#!/bin/bash
# round 3, GPU call U: register-resident Viterbi, LDS exchange level 2 (skip-edge neighbour and broadcast sources through LDS too) -- parity, A/B
set -u
OUT=gpurun_out/r3u
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py tests/test_gpu_bench_parity.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
STRQ_VIT_G2_LDS=2 timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py -m gpu -q -x > $OUT/tests_lx2.log 2>&1
echo "tests lx2 rc=$?"; tail -2 $OUT/tests_lx2.log
STRQ_VIT_G2_LDS=2 timeout 600 python tools/fuzz_detect.py 93 20 > $OUT/fuzz_detect.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_detect.log
for rep in 1 2 3; do
for v in lx1 lx2; do
  unset STRQ_VIT_NO_G2 STRQ_VIT_G2_WAVES STRQ_VIT_G2_LDS
  case $v in lx1) export STRQ_VIT_G2_LDS=1;; lx2) export STRQ_VIT_G2_LDS=2;; esac
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3u/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
