#!/bin/bash
# two contexts on the empirical-noise workload: does the second context's forward stage hide the Viterbi tail?
mkdir -p gpurun_out/r5z3
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z3/build.log 2>&1
timeout 900 python tools/coresident_probe.py 2048 6 empirical default > gpurun_out/r5z3/co_emp_default.txt 2>&1; echo "rc=$?"
tail -5 gpurun_out/r5z3/co_emp_default.txt
timeout 900 python tools/coresident_probe.py 2048 6 empirical coresident > gpurun_out/r5z3/co_emp_cores.txt 2>&1; echo "rc=$?"
tail -5 gpurun_out/r5z3/co_emp_cores.txt
