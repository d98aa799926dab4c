#!/bin/bash
mkdir -p gpurun_out/r5z10
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z10/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_detect.py -m gpu -q -x -k "float64 or odd_signals or conditioning or tiny or without_tails or reference_scenarios" > gpurun_out/r5z10/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5z10/tests.log
timeout 600 python tools/f64_probe.py 512 > gpurun_out/r5z10/f64_probe.txt 2>&1; echo "probe rc=$?"; tail -4 gpurun_out/r5z10/f64_probe.txt
