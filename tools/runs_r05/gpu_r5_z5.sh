#!/bin/bash
# two contexts taking turns in the forward stage (STRQ_FORWARD_TOKEN): empirical-noise reads, then clean ones
mkdir -p gpurun_out/r5z5
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z5/build.log 2>&1
STRQ_SCREEN_MODE=fine timeout 900 python tools/coresident_probe.py 2048 6 empirical token > gpurun_out/r5z5/co_emp_token.txt 2>&1; echo "rc=$?"
tail -4 gpurun_out/r5z5/co_emp_token.txt
timeout 900 python tools/coresident_probe.py 2048 6 clean token > gpurun_out/r5z5/co_clean_token.txt 2>&1; echo "rc=$?"
tail -4 gpurun_out/r5z5/co_clean_token.txt
STRQ_SCREEN_MODE=fine timeout 900 python tools/coresident_probe.py 4096 4 empirical token > gpurun_out/r5z5/co_emp_token_4096.txt 2>&1; echo "rc=$?"
tail -4 gpurun_out/r5z5/co_emp_token_4096.txt
