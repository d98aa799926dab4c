#!/bin/bash
# round 5, call d: where the 95 ms behind the coarse screen go (kernel trace of the default line)
set -u
OUT=gpurun_out/r5d; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r5d -- python3 bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --no-cpu-baseline --no-host-leg --no-legs --check 0 > $OUT/bench_kt.log 2>&1
echo "kt rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r5d/kt/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    print("%-110s calls %4s avg %10.3f ms total %10.2f ms  %s%%" % (row["Name"][:110], row["Calls"], float(row["AverageNs"]) / 1e6, float(row["TotalDurationNs"]) / 1e6, row["Percentage"]))
PY
rm -f $OUT/kt/*/*_kernel_trace.csv $OUT/kt/*/*.db
tail -1 $OUT/bench_kt.log | cut -c1-300
