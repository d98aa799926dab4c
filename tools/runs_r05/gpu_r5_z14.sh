#!/bin/bash
mkdir -p gpurun_out/r5z14
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z14/build.log 2>&1
timeout 900 python tools/batch_size_probe.py 50000 > gpurun_out/r5z14/batch_size_50k.txt 2>&1; echo "rc=$?"; tail -9 gpurun_out/r5z14/batch_size_50k.txt
timeout 900 python tools/batch_size_probe.py 20000 > gpurun_out/r5z14/batch_size_20k.txt 2>&1; echo "rc=$?"; tail -9 gpurun_out/r5z14/batch_size_20k.txt
