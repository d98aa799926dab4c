#!/bin/bash
# float64 order statistics on the GPU: the detect tests (float64 reads against the oracle, the statistics against numpy), throughput
mkdir -p gpurun_out/r5z9
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z9/build.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_detect.py -m gpu -q -x > gpurun_out/r5z9/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r5z9/tests.log
timeout 600 python tools/f64_probe.py 512 > gpurun_out/r5z9/f64_probe.txt 2>&1; echo "probe rc=$?"; tail -4 gpurun_out/r5z9/f64_probe.txt
