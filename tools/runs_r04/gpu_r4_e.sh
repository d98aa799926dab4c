#!/bin/bash
# round 4, GPU call E: only launches of 870-row flanks on round 3's kernel body, retuned realism model, one inflate call per
# reader task; whole suite, flank sweep, realism bench, reader probe, `count` from gzip files at steady state (32 768 reads)
set -u
OUT=gpurun_out/r4e
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout 400 python tools/flank_sweep.py > $OUT/flank_sweep.md 2> $OUT/flank_sweep.err; echo "flank sweep rc=$?"; cat $OUT/flank_sweep.md
timeout 900 python tools/realism_bench.py > $OUT/realism.md 2> $OUT/realism.err; echo "realism rc=$?"; cat $OUT/realism.md
timeout 600 python tools/reader_probe.py 8192 50000 > $OUT/reader_probe.log 2>&1; echo "reader probe rc=$?"; grep -v "^wrote" $OUT/reader_probe.log | head -32
for t in 16 24; do
  timeout 900 python tools/cli_probe.py 32768 50000 --t $t --compression gzip > $OUT/cli_gzip_32k_t$t.log 2>&1; echo "cli gzip 32k t=$t rc=$?"; grep "count pass" $OUT/cli_gzip_32k_t$t.log | tail -2
done
STRQ_READ_ONE_BY_ONE=1 timeout 900 python tools/cli_probe.py 32768 50000 --t 16 --compression gzip > $OUT/cli_gzip_32k_t16_onebyone.log 2>&1; echo "cli gzip 32k one by one rc=$?"; grep "count pass" $OUT/cli_gzip_32k_t16_onebyone.log | tail -1
timeout 900 python tools/cli_probe.py 32768 50000 --t 16 > $OUT/cli_contig_32k_t16.log 2>&1; echo "cli contiguous 32k rc=$?"; grep "count pass" $OUT/cli_contig_32k_t16.log | tail -1
