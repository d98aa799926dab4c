#!/bin/bash
# round 3, GPU call W: fuzz of the register-resident Viterbi over random flanked models; waves per CU at exchange level 2
set -u
OUT=gpurun_out/r3w
mkdir -p $OUT
timeout 1200 python tools/fuzz_g2.py 7 60 > $OUT/fuzz_g2.log 2>&1; echo "fuzz_g2 rc=$?"; tail -3 $OUT/fuzz_g2.log
timeout 600 python tools/fuzz_viterbi.py 78 30 > $OUT/fuzz_viterbi.log 2>&1; echo "fuzz_viterbi rc=$?"; tail -1 $OUT/fuzz_viterbi.log
for rep in 1 2; do
for v in w8 w4 w12; do
  unset STRQ_VIT_G2_WAVES
  case $v in w4) export STRQ_VIT_G2_WAVES=4;; w12) export STRQ_VIT_G2_WAVES=12;; esac
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3w/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
