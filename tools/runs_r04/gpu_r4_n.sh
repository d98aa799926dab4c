#!/bin/bash
# round 4, GPU call N: scheduler strategies for viterbi_kernels.hip with the round-4 kernel (in tree: max-ilp), A/B on one box
set -u
OUT=gpurun_out/r4n
mkdir -p $OUT
for v in intree vit_default vit_memclause vit_minreg intree; do
  if [ $v = intree ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  ( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
PY
done
