#!/bin/bash
# round 3, GPU call AG: sub-batches whose targets need both parities of the register-resident Viterbi keep the lane layout -- parity, configs[3]
set -u
OUT=gpurun_out/r3ag
mkdir -p $OUT
timeout 260 python -m pytest tests/test_gpu_detect.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 150 python tools/config_probe.py 4096 > $OUT/config.log 2>&1; echo "config rc=$?"; grep "configs\[" $OUT/config.log | cut -c1-260
