#!/bin/bash
# round 4, GPU call AD: what a screened batch of degraded reads spends (tools/screen_probe.py at realism 1.0 and 0.5)
set -u
OUT=gpurun_out/r4ad
mkdir -p $OUT
timeout 300 python tools/screen_probe.py 1024 50000 1.0 > $OUT/probe_r10.log 2>&1; echo "probe rc=$?"; tail -7 $OUT/probe_r10.log | cut -c1-400
STRQ_NO_SCREEN=1 timeout 300 python tools/screen_probe.py 1024 50000 1.0 > $OUT/probe_r10_noscreen.log 2>&1; echo "probe rc=$?"; tail -4 $OUT/probe_r10_noscreen.log | cut -c1-300
