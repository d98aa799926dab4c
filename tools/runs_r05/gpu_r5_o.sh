#!/bin/bash
# round 5, call o: the whole GPU suite (4- and 8-rank bench runs and the self-launch test included), smoke, the default line as the driver runs it
set -u
OUT=gpurun_out/r5o; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -m gpu -q -x > $OUT/tests_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -6 $OUT/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
/usr/bin/time -v -o $OUT/bench_time.txt timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; grep -a "Elapsed\|Maximum resident" $OUT/bench_time.txt
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5o/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"], "host-inclusive", d.get("host_inclusive_reads_per_s"))
print("no_screen", d["value_no_screen"], "fine", d["value_fine_screen"], "degraded", d["value_degraded"], d["legs"]["degraded"]["planted_count_recovered"], d["legs"]["degraded"]["screen"]["mode"])
cb = d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["per_core_reads_per_s"], cb["extrapolated_physical_cores"]["reads_per_s"], [(r["workers"], round(r["reads_per_s"], 3)) for r in cb["sweep"]], d["vs_cpu_baseline"])
print("check_ok", d["check_ok"], d["host"])
PY
