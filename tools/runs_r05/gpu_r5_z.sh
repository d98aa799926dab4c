#!/bin/bash
# round 5, call z: from which read length the screens pay (STRQ_SCREEN_MIN_N)
set -u
OUT=gpurun_out/r5z; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/minn_probe.py 4096 > $OUT/minn.txt 2>&1; grep -v amdgpu.ids $OUT/minn.txt
