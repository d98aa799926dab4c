#!/bin/bash
# the screens' adaptive state machine under a changing workload, rows against a context that never screens
mkdir -p gpurun_out/r5z13
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z13/build.log 2>&1
timeout 1500 python tools/soak_screen.py 60 1 > gpurun_out/r5z13/soak1.txt 2>&1; echo "soak 1 rc=$?"; tail -3 gpurun_out/r5z13/soak1.txt
timeout 1500 python tools/soak_screen.py 60 2 > gpurun_out/r5z13/soak2.txt 2>&1; echo "soak 2 rc=$?"; tail -3 gpurun_out/r5z13/soak2.txt
