#!/bin/bash
# round 5, call u: reads per sub-batch against the coarse screen's rounds (6 workgroups per CU x 256 CUs = 1536 reads per round)
set -u
OUT=gpurun_out/r5u; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  tag=$1; reads=$2; shift; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --batches 1 --reads $reads --no-cpu-baseline --no-host-leg --no-legs --check 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1])
print("$tag: value %.0f ms %.1f screen %.2f fwd %.2f trace %.2f vit %.2f check %s" % (d["value"], d["ms_per_step"], d["screen"]["ms_per_step"], d["stage_ms_per_step"]["forward_dp"], d["stage_ms_per_step"]["trace"], d["stage_ms_per_step"]["viterbi"], d["check_ok"]))
PY
}
run r4096 4096 STRQ_SUBBATCH_READS=4096
run r4608 4608 STRQ_SUBBATCH_READS=4608
run r3072 3072 STRQ_SUBBATCH_READS=3072
run r6144 6144 STRQ_SUBBATCH_READS=6144
run r4096g5 4096 STRQ_SUBBATCH_READS=4096 STRQ_SCREEN2_GROUPS=5
run r3840g5 3840 STRQ_SUBBATCH_READS=3840 STRQ_SCREEN2_GROUPS=5
