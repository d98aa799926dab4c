#!/bin/bash
# round 4, GPU call J: datasets located by the library (strq_h5_locate) instead of ~50 us of Python per read -- reader probe, `count`
# from gzip-compressed and contiguous fast5 files at steady state, the file-based GPU tests
set -u
OUT=gpurun_out/r4j
mkdir -p $OUT
timeout 600 python -m pytest tests/test_cli_end_to_end.py tests/test_bundled_read.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout 600 python tools/reader_probe.py 8192 50000 > $OUT/reader_probe.log 2>&1; echo "reader probe rc=$?"; grep "libdeflate one" $OUT/reader_probe.log
for t in 16 24 32; do
  timeout 900 python tools/cli_probe.py 32768 50000 --t $t --compression gzip > $OUT/cli_gzip_32k_t$t.log 2>&1; echo "cli gzip 32k t=$t rc=$?"; grep "count pass" $OUT/cli_gzip_32k_t$t.log | tail -2
done
STRQ_H5_PYTHON=1 timeout 900 python tools/cli_probe.py 32768 50000 --t 24 --compression gzip > $OUT/cli_gzip_32k_t24_python.log 2>&1; echo "cli gzip 32k t=24, Python locate rc=$?"; grep "count pass" $OUT/cli_gzip_32k_t24_python.log | tail -1
timeout 900 python tools/cli_probe.py 32768 50000 --t 16 > $OUT/cli_contig_32k_t16.log 2>&1; echo "cli contiguous 32k rc=$?"; grep "count pass" $OUT/cli_contig_32k_t16.log | tail -2
timeout 900 python tools/cli_probe.py 32768 10000 --t 16 > $OUT/cli_contig_10kb.log 2>&1; echo "cli contiguous 10 kb rc=$?"; grep "count pass" $OUT/cli_contig_10kb.log | tail -1
