#!/bin/bash
# round 4, GPU call S: the register-resident Viterbi's LDS round trips taken off the critical path -- lane - 1's odd delete slot fetched at
# the head of the step (-DSTRQ_G2_LATE_DO), the odd slots + even insert first and the even match behind the exchange (-DSTRQ_G2_EARLY2,
# with / without scheduling barriers), ds_read_b96 (-DSTRQ_G2_LOAD96).  A/B on one box + parity of the candidates.
set -u
OUT=gpurun_out/r4s
mkdir -p $OUT
for v in intree latedo late96 base96 early2 early2b e2b96 intree; do
  if [ $v = intree ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  ( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
PY
done
for v in e2b96 late96; do
  export STRQ_LIB=$PWD/tools/bin/lib_$v.so
  timeout 300 python tools/fuzz_g2.py 91 100 > $OUT/fuzz_g2_$v.log 2>&1; echo "fuzz_g2 $v rc=$?"; tail -1 $OUT/fuzz_g2_$v.log
  timeout 600 python -m pytest tests/test_gpu_viterbi.py -m gpu -q > $OUT/tests_$v.log 2>&1; echo "viterbi tests ($v) rc=$?"; tail -1 $OUT/tests_$v.log
  timeout 200 python tools/config_probe.py 4096 > $OUT/config_$v.log 2>&1; grep "configs\[3" $OUT/config_$v.log | cut -c1-230
done
