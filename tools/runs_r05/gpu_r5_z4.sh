#!/bin/bash
# timeline of two contexts on the empirical-noise workload (fine screen pinned: no retries of the coarse one inside the loops):
# do kernels of the two contexts overlap at all?
mkdir -p gpurun_out/r5z4
export TMPDIR=/tmp STRQ_SCREEN_MODE=fine
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z4/build.log 2>&1
timeout 900 python tools/coresident_probe.py 2048 6 empirical default > gpurun_out/r5z4/co_plain.txt 2>&1; echo "rc=$?"
tail -4 gpurun_out/r5z4/co_plain.txt
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5z4/kt -o co -- python3 tools/coresident_probe.py 2048 3 empirical default > gpurun_out/r5z4/co_traced.txt 2>&1; echo "rc=$?"
tail -4 gpurun_out/r5z4/co_traced.txt
find gpurun_out/r5z4/kt -name "*.db" -delete
ls -la gpurun_out/r5z4/kt/* | head
