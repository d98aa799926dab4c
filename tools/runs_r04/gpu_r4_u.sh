#!/bin/bash
# round 4, GPU call U: where the time of a screened batch goes (tools/screen_probe.py, rocprofv3 kernel stats), screen tests again
set -u
OUT=gpurun_out/r4u
mkdir -p $OUT
timeout 600 python tools/screen_probe.py 2048 > $OUT/probe.log 2>&1; echo "probe rc=$?"; cat $OUT/probe.log | tail -12
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o screen -- python3 $GRAFT_REPO_ROOT/tools/screen_probe.py 1024 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); echo $f; head -12 $f | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_screen.py -m gpu -x -q > $OUT/tests_screen.log 2>&1; echo "screen tests rc=$?"; tail -12 $OUT/tests_screen.log
