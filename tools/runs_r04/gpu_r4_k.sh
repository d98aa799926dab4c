#!/bin/bash
# round 4, GPU call K: how many CPUs the job really gets; `count` from contiguous fast5 files with the native locator
set -u
OUT=gpurun_out/r4k
mkdir -p $OUT
rm -rf /tmp/strq_cli_* /tmp/strq_rd_* 2>/dev/null
df -h /tmp | tail -1
timeout 300 python tools/cpu_scaling_probe.py > $OUT/cpu_scaling.log 2>&1; echo "cpu scaling rc=$?"; cat $OUT/cpu_scaling.log
timeout 900 python tools/cli_probe.py 32768 50000 --t 16 > $OUT/cli_contig_32k_t16.log 2>&1; echo "cli contiguous 32k rc=$?"; grep "count pass" $OUT/cli_contig_32k_t16.log | tail -2
STRQ_H5_PYTHON=1 timeout 900 python tools/cli_probe.py 32768 50000 --t 16 > $OUT/cli_contig_32k_t16_python.log 2>&1; echo "cli contiguous 32k, Python locate rc=$?"; grep "count pass" $OUT/cli_contig_32k_t16_python.log | tail -1
timeout 900 python tools/cli_probe.py 32768 10000 --t 16 > $OUT/cli_contig_10kb.log 2>&1; echo "cli contiguous 10 kb rc=$?"; grep "count pass" $OUT/cli_contig_10kb.log | tail -1
