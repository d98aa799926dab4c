#!/bin/bash
# round 3, GPU call T: register-resident Viterbi with the three odd-slot neighbours through LDS -- parity, then A/B
set -u
OUT=gpurun_out/r3t
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py tests/test_gpu_bench_parity.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout 600 python tools/fuzz_detect.py 92 20 > $OUT/fuzz_detect.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_detect.log
for rep in 1 2; do
for v in lane g2 g2lx g2lx12; do
  unset STRQ_VIT_NO_G2 STRQ_VIT_G2_WAVES STRQ_VIT_G2_LDS
  case $v in lane) export STRQ_VIT_NO_G2=1;; g2) export STRQ_VIT_G2_LDS=0;; g2lx) export STRQ_VIT_G2_LDS=1;; g2lx12) export STRQ_VIT_G2_LDS=1 STRQ_VIT_G2_WAVES=12;; esac
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3t/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
