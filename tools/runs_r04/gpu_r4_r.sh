#!/bin/bash
# round 4, GPU call R: leave the sweep loop right behind the unconditional sweeps when the last of them changed nothing (-DSTRQ_G2_WHILE), A/B
set -u
OUT=gpurun_out/r4r
mkdir -p $OUT
for v in intree vit_wh2 vit_wh3 intree; do
  if [ $v = intree ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  ( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
PY
done
export STRQ_LIB=$PWD/tools/bin/lib_vit_wh2.so
timeout 300 python tools/fuzz_g2.py 95 60 > $OUT/fuzz_g2_wh2.log 2>&1; echo "fuzz_g2 wh2 rc=$?"; tail -1 $OUT/fuzz_g2_wh2.log
