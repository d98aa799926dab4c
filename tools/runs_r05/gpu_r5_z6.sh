#!/bin/bash
# empirical-noise reads: is the two-rows-per-DP-row screen (merge 2) tight enough where the three-row one is not?
mkdir -p gpurun_out/r5z6
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z6/build.log 2>&1
B="bench.py --steps 6 --warmup 3 --reads 2048 --batches 2 --workload empirical --no-cpu-baseline --no-host-leg --no-legs --check 0"
for m in 2 3; do
  STRQ_SCREEN_MODE=coarse STRQ_SCREEN_ALWAYS=1 STRQ_SCREEN2_MERGE=$m timeout 600 python $B > gpurun_out/r5z6/emp_merge$m.json 2> gpurun_out/r5z6/emp_merge$m.err; echo "merge $m rc=$?"
done
STRQ_SCREEN_MODE=fine timeout 600 python $B > gpurun_out/r5z6/emp_fine.json 2> gpurun_out/r5z6/emp_fine.err; echo "fine rc=$?"
python - <<'PY'
import json
for n in ("merge2","merge3","fine"):
    try:
        d=json.loads(open("gpurun_out/r5z6/emp_%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), d["ms_per_step"], d.get("stage_ms"), d.get("screen"), d.get("second_round"))
    except Exception as e: print(n, "failed", e)
PY
