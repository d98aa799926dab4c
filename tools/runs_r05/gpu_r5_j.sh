#!/bin/bash
# round 5, call j: the coarse screen with 2, 3 and 6 flank rows per DP row (STRQ_SCREEN2_MERGE): tests, then the line for each
set -u
OUT=gpurun_out/r5j; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_screen.py -x -q -k coarse > $OUT/tests_screen.log 2>&1; echo "screen tests rc=$?"; tail -4 $OUT/tests_screen.log
for v in 2 3 6 2; do
  STRQ_DEBUG=1 STRQ_SCREEN2_MERGE=$v timeout 600 python bench.py --steps 4 --warmup 2 --batches 1 --no-cpu-baseline --no-host-leg --no-legs --check 1 > $OUT/bench_m$v.json 2> $OUT/bench_m$v.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_m$v.json") if l.startswith("{")][-1])
print("merge $v: value %.0f ms %.1f screen %.2f fwd %.2f trace %.2f check %s cand/alignment %.1f window cols %.4f" % (d["value"], d["ms_per_step"], d["screen"]["ms_per_step"], d["stage_ms_per_step"]["forward_dp"], d["stage_ms_per_step"]["trace"], d["check_ok"], d["screen"]["candidate_chunks_per_alignment"], d["screen"]["window_columns_over_columns_of_the_reads"]))
PY
  grep -a "second look" $OUT/bench_m$v.err | tail -1
done
