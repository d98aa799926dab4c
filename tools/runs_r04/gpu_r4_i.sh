#!/bin/bash
# round 4, GPU call I: where the reader threads' time goes (inflate time measured inside the library), kernel times of a --mod_model run
set -u
OUT=gpurun_out/r4i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/reader_probe.py 8192 50000 > $OUT/reader_probe.log 2>&1; echo "reader probe rc=$?"; grep "libdeflate one" $OUT/reader_probe.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mod -o mod -- python3 tools/mod_probe.py 4096 > $OUT/mod_prof.log 2>&1; echo "mod profile rc=$?"; grep "mod=" $OUT/mod_prof.log | cut -c1-200
rm -f $OUT/prof_mod/*/*_kernel_trace.csv $OUT/prof_mod/*/*.db 2>/dev/null
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/r4i/prof_mod/**/*kernel_stats.csv", recursive=True))
if f:
    for r in list(csv.DictReader(open(f[-1])))[:16]:
        print("%-90s calls %4s avg %9.3f ms total %9.1f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
