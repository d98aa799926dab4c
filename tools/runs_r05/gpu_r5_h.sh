#!/bin/bash
# round 5, call h: A/B of the coarse screen's variants (STRQ_SCREEN2_VARIANT: 0 committed, 1 additions hoisted at 4 waves per SIMD, 2 default body at 4, 3 hoisted at 5)
set -u
OUT=gpurun_out/r5h; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 0 1 2 3 0; do
  STRQ_SCREEN2_VARIANT=$v timeout 600 python bench.py --steps 4 --warmup 2 --batches 1 --no-cpu-baseline --no-host-leg --no-legs --check 1 > $OUT/bench_v$v.json 2> $OUT/bench_v$v.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_v$v.json") if l.startswith("{")][-1])
print("variant $v: value %.0f ms %.1f screen %.2f fwd %.2f check %s" % (d["value"], d["ms_per_step"], d["screen"]["ms_per_step"], d["stage_ms_per_step"]["forward_dp"], d["check_ok"]))
PY
done
