#!/bin/bash
# round 4, GPU call H: long randomised parity runs and the soak run on the committed code
set -u
OUT=gpurun_out/r4h
mkdir -p $OUT
timeout 900 python tools/fuzz_align.py 401 1200 > $OUT/fuzz_align.log 2>&1; echo "fuzz_align rc=$?"; tail -1 $OUT/fuzz_align.log
timeout 900 python tools/fuzz_viterbi.py 402 1200 > $OUT/fuzz_viterbi.log 2>&1; echo "fuzz_viterbi rc=$?"; tail -1 $OUT/fuzz_viterbi.log
timeout 900 python tools/fuzz_g2.py 403 300 > $OUT/fuzz_g2.log 2>&1; echo "fuzz_g2 rc=$?"; tail -1 $OUT/fuzz_g2.log
timeout 1500 python tools/fuzz_detect.py 404 150 > $OUT/fuzz_detect.log 2>&1; echo "fuzz_detect rc=$?"; tail -1 $OUT/fuzz_detect.log
timeout 600 python tools/soak.py --calls 20 > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -2 $OUT/soak.log
