#!/bin/bash
# round 5, call g: coarse screen with the bounded first look + the second look in groups of pieces: tests, default line with legs, kernel trace
set -u
OUT=gpurun_out/r5g; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_screen.py tests/test_gpu_bench_parity.py -x -q -k "screen or through" > $OUT/tests_screen.log 2>&1; echo "screen tests rc=$?"; tail -5 $OUT/tests_screen.log
STRQ_DEBUG=1 timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --check 2 --leg-steps 3 --no-host-leg > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep -a "coarse screen\|screen verdict" $OUT/bench.err | sort | uniq -c | sort -rn | head -12
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5g/bench.json") if l.startswith("{")][-1])
    print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"])
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_ms", "frac", "useful_frac", "valu_insts_per_wave_step")})
    print("screen", d["screen"])
    for k, v in d.get("legs", {}).items():
        print(k, v["value"], v["ms_per_step"], v["stage_ms_per_step"], v["screen"], v.get("planted_count_recovered"), v.get("second_round_share"), v.get("value_no_screen"))
    print("check_ok", d["check_ok"])
except Exception as e:
    print("no line:", e)
PY
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r5g -- python3 bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --no-cpu-baseline --no-host-leg --no-legs --check 0 > $OUT/bench_kt.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r5g/kt/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:9]:
    print("%-100s calls %4s avg %10.3f ms total %10.2f ms  %s%%" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"]) / 1e6, row["Percentage"]))
PY
rm -f $OUT/kt/*_kernel_trace.csv $OUT/kt/*/*_kernel_trace.csv
