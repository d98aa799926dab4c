#!/bin/bash
# round 4, GPU call P: two unconditional chain sweeps in tree; two sweeps per convergence test in the loop behind them (-DSTRQ_G2_LOOP2), A/B;
# Viterbi / detect parity suites and fuzzers on the in-tree build
set -u
OUT=gpurun_out/r4p
mkdir -p $OUT
for v in intree vit_p2l2 vit_p1l2 intree; do
  if [ $v = intree ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  ( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
PY
done
unset STRQ_LIB
timeout 900 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py tests/test_g2_layout.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 300 python tools/fuzz_g2.py 81 100 > $OUT/fuzz_g2.log 2>&1; echo "fuzz_g2 rc=$?"; tail -1 $OUT/fuzz_g2.log
timeout 300 python tools/mod_probe.py 4096 > $OUT/mod.log 2>&1; grep "mod=" $OUT/mod.log | cut -c1-200
