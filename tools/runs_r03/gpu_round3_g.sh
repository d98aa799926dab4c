#!/bin/bash
# round 3, GPU call G: where a Viterbi step's cycles go, longer fuzz runs at the consolidated state, the other BASELINE configs, soak
set -u
OUT=gpurun_out/r3g
mkdir -p $OUT
STRQ_LIB=$PWD/tools/bin/lib_vittiming.so timeout 600 python tools/vit_timing.py 512 50000 > $OUT/vit_timing.log 2>&1; echo "vit_timing rc=$?"; cat $OUT/vit_timing.log | tail -5
timeout 900 python tools/fuzz_align.py 31 1200 > $OUT/fuzz_align.log 2>&1; echo "fuzz_align rc=$?"; tail -2 $OUT/fuzz_align.log
timeout 900 python tools/fuzz_viterbi.py 32 1200 > $OUT/fuzz_viterbi.log 2>&1; echo "fuzz_viterbi rc=$?"; tail -2 $OUT/fuzz_viterbi.log
timeout 1500 python tools/fuzz_detect.py 33 150 > $OUT/fuzz_detect.log 2>&1; echo "fuzz_detect rc=$?"; tail -2 $OUT/fuzz_detect.log
timeout 900 python tools/config_probe.py 4096 > $OUT/config_probe.log 2>&1; echo "config_probe rc=$?"; tail -6 $OUT/config_probe.log
timeout 900 python tools/mod_probe.py 4096 > $OUT/mod_probe.log 2>&1; echo "mod_probe rc=$?"; tail -3 $OUT/mod_probe.log
timeout 900 python tools/soak.py --reads 2048 --calls 20 > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -4 $OUT/soak.log
