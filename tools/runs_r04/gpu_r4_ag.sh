#!/bin/bash
# round 4, GPU call AG: the certificate's safety net forced (STRQ_SCREEN_TEST_RAISE), strq_align_batch counting its second round: screen + alignment suites
set -u
OUT=gpurun_out/r4ag
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_screen.py tests/test_gpu_align.py tests/test_gpu_shim.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
