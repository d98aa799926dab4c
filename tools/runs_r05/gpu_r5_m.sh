#!/bin/bash
# round 5, call m: fine screen without register copies, forward launches on up to three side streams: tests + the line
set -u
OUT=gpurun_out/r5m; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests/test_gpu_screen.py tests/test_gpu_bench_parity.py tests/test_gpu_align.py -x -q --deselect tests/test_gpu_bench_parity.py::test_bench_four_and_eight_ranks_on_one_gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --check 2 --leg-steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5m/bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"], "host-inclusive", d.get("host_inclusive_reads_per_s"))
print({k: d["roofline"].get(k) for k in ("kernel", "frac", "useful_frac", "valu_insts_per_wave_step", "flank_rows_per_dp_row")}, d["roofline_viterbi"]["frac"], d["roofline_viterbi"]["frac_of_float64_issue"])
for k, v in d.get("legs", {}).items():
    print(k, v["value"], v["ms_per_step"], v["stage_ms_per_step"], v["screen"]["mode"], v.get("planted_count_recovered"), v.get("second_round_share"), v.get("value_no_screen"))
print("check_ok", d["check_ok"], d["host"]["peak_host_rss_gb_per_rank"])
PY
