#!/bin/bash
# round 5, call i: the whole GPU suite on the coarse-screen state
set -u
OUT=gpurun_out/r5i; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_parity.py::test_bench_four_and_eight_ranks_on_one_gpu > $OUT/tests_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -15 $OUT/tests_gpu.log
