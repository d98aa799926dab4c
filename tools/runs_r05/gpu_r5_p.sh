#!/bin/bash
# round 5, call p: the default line exactly as the driver runs it (wall time of the whole command)
set -u
OUT=gpurun_out/r5p; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5p/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_step"], "host-inclusive", d.get("host_inclusive_reads_per_s"))
print("no_screen", d["value_no_screen"], "fine", d["value_fine_screen"], "degraded", d["value_degraded"], d["legs"]["degraded"]["planted_count_recovered"], d["legs"]["degraded"]["screen"]["mode"], d["legs"]["degraded"]["value_no_screen"])
cb = d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["per_core_reads_per_s"], cb["extrapolated_physical_cores"]["reads_per_s"], [(r["workers"], round(r["reads_per_s"], 3)) for r in cb["sweep"]], d["vs_cpu_baseline"])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "useful_frac", "achieved", "valu_insts_per_wave_step", "traffic")})
print("viterbi", {k: d["roofline_viterbi"].get(k) for k in ("frac", "frac_of_float64_issue", "achieved", "ms_per_step")})
print("check_ok", d["check_ok"], d["host"])
PY
