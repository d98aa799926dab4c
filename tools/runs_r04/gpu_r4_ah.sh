#!/bin/bash
# round 4, GPU call AH: the whole GPU suite and smoke on the round's last commit
set -u
OUT=gpurun_out/r4ah
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
