#!/bin/bash
# round 3, GPU call S: register-resident Viterbi (profile chain, two positions per lane, no LDS) -- parity, then A/B against the lane layout
set -u
OUT=gpurun_out/r3s
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py tests/test_gpu_bench_parity.py tests/test_gpu_limits.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
timeout 600 python tools/fuzz_detect.py 91 30 > $OUT/fuzz_detect.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_detect.log
for rep in 1 2; do
for v in lane g2w8 g2w12 g2w4; do
  unset STRQ_VIT_NO_G2 STRQ_VIT_G2_WAVES
  case $v in lane) export STRQ_VIT_NO_G2=1;; g2w8) export STRQ_VIT_G2_WAVES=8;; g2w12) export STRQ_VIT_G2_WAVES=12;; g2w4) export STRQ_VIT_G2_WAVES=4;; esac
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3s/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
