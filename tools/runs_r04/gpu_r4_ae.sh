#!/bin/bash
# round 4, GPU call AE: degraded reads with the pause rule for heavy alignments; screen tests
set -u
OUT=gpurun_out/r4ae
mkdir -p $OUT
timeout 500 python tools/realism_bench.py --reads 2048 0.0 0.5 1.0 1.5 > $OUT/realism_screen.md 2> $OUT/realism_screen.err; echo "realism rc=$?"; cat $OUT/realism_screen.md
timeout 600 python -m pytest tests/test_gpu_screen.py tests/test_gpu_detect.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
