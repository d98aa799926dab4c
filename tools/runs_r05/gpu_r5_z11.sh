#!/bin/bash
# the tree with the float64 statistics on the GPU: whole GPU suite, smoke, the default bench line
mkdir -p gpurun_out/r5z11
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z11/build.log 2>&1
timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r5z11/tests_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -22 gpurun_out/r5z11/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z11/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5z11/smoke.log
T0=$(date +%s)
timeout 1200 python bench.py > gpurun_out/r5z11/bench_default.json 2> gpurun_out/r5z11/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
