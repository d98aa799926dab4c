#!/bin/bash
# fuzz of detect against the oracle on the final tree (int16 and float64 reads, ties), the caller's-own-statistics test
mkdir -p gpurun_out/r5z12
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z12/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_detect.py -m gpu -q -x -k "callers_own or float64" > gpurun_out/r5z12/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r5z12/tests.log
timeout 1500 python tools/fuzz_detect.py 11 60 > gpurun_out/r5z12/fuzz_detect.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r5z12/fuzz_detect.txt
timeout 1200 python tools/fuzz_detect.py 12 40 screen > gpurun_out/r5z12/fuzz_detect_screen.txt 2>&1; echo "fuzz screen rc=$?"; tail -3 gpurun_out/r5z12/fuzz_detect_screen.txt
