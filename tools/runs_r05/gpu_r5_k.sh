#!/bin/bash
# round 5, call k: (1) operand-class microbenchmark (verdict item 5), (2) Viterbi co-resident with the screen (verdict item 4),
# (3) the default line with merge 3 + one trace launch
set -u
OUT=gpurun_out/r5k; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_operands.hip -o /tmp/ubench_operands 2>/dev/null || cp tools/bin/ubench_operands /tmp/ubench_operands
timeout 300 /tmp/ubench_operands > $OUT/ubench_operands.txt 2>&1; cat $OUT/ubench_operands.txt
timeout 900 python tools/coresident_probe.py 2048 6 > $OUT/coresident.txt 2>&1; tail -5 $OUT/coresident.txt
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --check 2 --leg-steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5k/bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"], "host-inclusive", d.get("host_inclusive_reads_per_s"))
for k, v in d.get("legs", {}).items():
    print(k, v["value"], v["ms_per_step"], v["stage_ms_per_step"], v["screen"]["mode"], v.get("planted_count_recovered"), v.get("second_round_share"), v.get("value_no_screen"))
print("check_ok", d["check_ok"], d["host"]["peak_host_rss_gb_per_rank"])
PY
