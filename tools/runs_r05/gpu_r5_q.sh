#!/bin/bash
# round 5, call q: SQ counters of the fine screen's round-5 body (STRQ_SCREEN_MODE=fine) -> valu_insts_per_wave_step of align_screen_kernel
set -u
P=gpurun_out/prof_r05f
mkdir -p $P
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 STRQ_SCREEN_MODE=fine
BENCH_PMC="bench.py --steps 1 --warmup 0 --reads 1024 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$P/pmc_sq" -o r05f -- python3 $BENCH_PMC > "$P/bench_sq.log" 2>&1; echo "SQ pass rc=$?"
python3 - <<'PY'
import csv, glob, json, collections
f = glob.glob("gpurun_out/prof_r05f/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float)
for row in csv.DictReader(open(f)):
    if "align_screen_kernel" in row["Kernel_Name"] and row["Counter_Name"] == "SQ_INSTS_VALU":
        acc["v"] += float(row["Counter_Value"])
d = json.loads([l for l in open("gpurun_out/prof_r05f/bench_sq.log") if l.startswith("{")][-1])
steps = d["roofline"]["wave_steps_per_launch"] * d["roofline"]["launches_per_step"]
print("align_screen_kernel: %.4g VALU over %.4g wave-steps = %.2f per wave-step; screen %.2f ms" % (acc["v"], steps, acc["v"] / steps, d["screen"]["ms_per_step"]))
PY
