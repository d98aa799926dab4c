#!/bin/bash
# round 5, call v: the Viterbi's tail on reads whose flanks are found in the wrong place; raised priority for the windows the launch waits for
set -u
OUT=gpurun_out/r5v; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 240 python tools/vit_tail_probe.py 4096 empirical > $OUT/tail_empirical.txt 2>&1; grep -v amdgpu.ids $OUT/tail_empirical.txt
timeout 240 python tools/vit_tail_probe.py 4096 clean > $OUT/tail_clean.txt 2>&1; grep -v amdgpu.ids $OUT/tail_clean.txt

