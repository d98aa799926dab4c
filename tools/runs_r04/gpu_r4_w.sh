#!/bin/bash
# round 4, GPU call W: SQ counters and kernel stats of the screen kernel (what its step waits for)
set -u
OUT=gpurun_out/r4w
mkdir -p $OUT
export TMPDIR=/tmp
BENCH_PMC="bench.py --steps 1 --warmup 0 --reads 1024 --batches 1 --synth-workers 4 --no-cpu-baseline --no-host-leg --check 0"
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o r4w -- python3 $BENCH_PMC > "$OUT/bench_sq.log" 2>&1
echo "SQ pass rc=$?"
timeout 500 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_sq2" -o r4w -- python3 $BENCH_PMC > "$OUT/bench_sq2.log" 2>&1
echo "SQ pass 2 rc=$?"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for sub in ("pmc_sq", "pmc_sq2"):
    files = glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(sub, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        if "screen" in k or "align_forward" in k:
            print(sub, k[:70], {c: "%.4g" % x for c, x in sorted(v.items())})
PY
