#!/bin/bash
# round 5, call w: randomised parity runs -- detect through every screen setting, align_overlap, Viterbi (register-resident variants)
set -u
OUT=gpurun_out/r5w; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/fuzz_detect.py 501 120 screen > $OUT/fuzz_detect_screen.log 2>&1; tail -2 $OUT/fuzz_detect_screen.log; grep -c MISMATCH $OUT/fuzz_detect_screen.log
timeout 600 python tools/fuzz_detect.py 502 80 > $OUT/fuzz_detect.log 2>&1; tail -1 $OUT/fuzz_detect.log
timeout 400 python tools/fuzz_align.py 503 600 > $OUT/fuzz_align.log 2>&1; tail -1 $OUT/fuzz_align.log
timeout 400 python tools/fuzz_g2.py 504 40 > $OUT/fuzz_g2.log 2>&1; tail -1 $OUT/fuzz_g2.log
