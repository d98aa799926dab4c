#!/bin/bash
# round 4, GPU call O: the first 1 / 2 / 3 chain sweeps of a Viterbi time step without the convergence test (-DSTRQ_G2_PRESWEEPS), A/B on one box
set -u
OUT=gpurun_out/r4o
mkdir -p $OUT
for v in intree vit_pre1 vit_pre2 vit_pre3 intree; do
  if [ $v = intree ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  ( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
PY
done
export STRQ_LIB=$PWD/tools/bin/lib_vit_pre2.so
timeout 300 python tools/fuzz_g2.py 71 60 > $OUT/fuzz_g2_pre2.log 2>&1; echo "fuzz_g2 pre2 rc=$?"; tail -1 $OUT/fuzz_g2_pre2.log
timeout 200 python tools/config_probe.py 4096 > $OUT/config_pre2.log 2>&1; grep "configs\[3" $OUT/config_pre2.log | cut -c1-230
