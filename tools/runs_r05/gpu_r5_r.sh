#!/bin/bash
# round 5, call r: the other BASELINE configs with the coarse screen (configs[1] 10 kb, configs[3] mixed targets, configs[4] --mod_model)
set -u
OUT=gpurun_out/r5r; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/config_probe.py 4096 > $OUT/config.log 2>&1; grep "configs\[" $OUT/config.log | cut -c1-260
STRQ_SCREEN_MODE=fine timeout 600 python tools/config_probe.py 4096 > $OUT/config_fine.log 2>&1; grep "configs\[" $OUT/config_fine.log | cut -c1-260
timeout 600 python tools/mod_probe.py 4096 > $OUT/mod.log 2>&1; grep "mod=" $OUT/mod.log
STRQ_DEBUG=1 timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg --check 1 --leg-steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep -a "coarse screen" $OUT/bench.err | tail -8 | cut -c1-330
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5r/bench.json") if l.startswith("{")][-1])
print("value", d["value"], d["stage_ms_per_step"])
v = d["legs"]["degraded"]
print("degraded", v["value"], v["ms_per_step"], v["stage_ms_per_step"], v["screen"], v.get("planted_count_recovered"), v.get("second_round_share"), v.get("value_no_screen"), v["check"])
PY
