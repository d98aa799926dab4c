#!/bin/bash
# round 5, call x: the RCCL -> gloo fallback of bench.py's gather (two ranks on ONE device make RCCL refuse: "duplicate GPU"), launched plainly
set -u
OUT=gpurun_out/r5x; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
env -u RANK -u WORLD_SIZE timeout 900 python bench.py --gpus 2 --share-device --backend nccl --reads 64 --steps 2 --warmup 1 --no-cpu-baseline --check 1 --synth-workers 2 > $OUT/bench_nccl2.json 2> $OUT/bench_nccl2.err; echo "rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5x/bench_nccl2.json") if l.startswith("{")][-1])
    print("n_gpus", d["n_gpus"], "world seen", d["world_size_seen_by_the_collective"], json.dumps(d["collective"])[:900], "check_ok", d["check_ok"])
except Exception as e:
    print("no line", e)
PY
tail -5 $OUT/bench_nccl2.err | cut -c1-300
timeout 300 python tools/nccl_smoke.py > $OUT/nccl_smoke.log 2>&1; tail -2 $OUT/nccl_smoke.log
