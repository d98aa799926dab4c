#!/bin/bash
# the last tree of the round: whole GPU suite, smoke, the default bench line
mkdir -p gpurun_out/r5z7
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z7/build.log 2>&1
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r5z7/tests_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -3 gpurun_out/r5z7/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z7/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5z7/smoke.log
