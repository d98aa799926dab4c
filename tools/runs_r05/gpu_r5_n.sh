#!/bin/bash
# round 5, call n: A/B of the side streams behind the coarse screen, and of its first-look settings (margin, candidate cap)
set -u
OUT=gpurun_out/r5n; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --batches 1 --no-cpu-baseline --no-host-leg --no-legs --check 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1])
print("$tag: value %.0f ms %.1f screen %.2f fwd %.2f trace %.2f check %s second-round %.3f window cols %.4f" % (d["value"], d["ms_per_step"], d["screen"]["ms_per_step"], d["stage_ms_per_step"]["forward_dp"], d["stage_ms_per_step"]["trace"], d["check_ok"], 0.0, d["screen"]["window_columns_over_columns_of_the_reads"]))
PY
}
run side1 STRQ_SIDE_STREAMS=1
run side3 STRQ_SIDE_STREAMS=3
run side1b STRQ_SIDE_STREAMS=1
run cand8 STRQ_SCREEN2_MAX_CAND=8
run cand32 STRQ_SCREEN2_MAX_CAND=32
run margin300 STRQ_SCREEN2_MARGIN=300
run margin700 STRQ_SCREEN2_MARGIN=700
run cand4m250 STRQ_SCREEN2_MAX_CAND=4 STRQ_SCREEN2_MARGIN=250
