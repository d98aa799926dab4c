#!/bin/bash
# round 4, GPU call M: VBZ-compressed fast5 (what MinKNOW writes) decoded by the library in one call per read -- `count` end to end
# against the per-chunk Python decoder; file-based GPU tests
set -u
OUT=gpurun_out/r4m
mkdir -p $OUT
rm -rf /tmp/strq_cli_* /tmp/strq_rd_* 2>/dev/null
timeout 600 python -m pytest tests/test_cli_end_to_end.py tests/test_bundled_read.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 900 python tools/cli_probe.py 32768 50000 --t 16 --compression vbz > $OUT/cli_vbz_32k_t16.log 2>&1; echo "cli vbz 32k rc=$?"; grep "count pass" $OUT/cli_vbz_32k_t16.log | tail -2
STRQ_H5_PYTHON=1 timeout 900 python tools/cli_probe.py 8192 50000 --t 16 --compression vbz > $OUT/cli_vbz_8k_python.log 2>&1; echo "cli vbz 8k, Python decoder rc=$?"; grep "count pass" $OUT/cli_vbz_8k_python.log | tail -1
timeout 900 python tools/cli_probe.py 8192 50000 --t 16 --compression vbz > $OUT/cli_vbz_8k.log 2>&1; echo "cli vbz 8k rc=$?"; grep "count pass" $OUT/cli_vbz_8k.log | tail -1
