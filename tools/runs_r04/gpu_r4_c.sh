#!/bin/bash
# round 4, GPU call C: the 870-row instance of round 3 kept beside the switched loop; 4- and 8-rank bench on one GPU at 50 kb;
# libdeflate in the fast5 reader + a leaner Python side of get_raw: `count` from gzip-compressed files with 16 / 32 / 48 reader threads
set -u
OUT=gpurun_out/r4c
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
( time timeout 600 python bench.py --steps 9 --warmup 3 --no-cpu-baseline ) > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','host_inclusive_reads_per_s','check_ok')}, d['stage_ms_per_step'], d['roofline']['frac'])
PY
timeout 200 python tools/config_probe.py 4096 > $OUT/config.log 2>&1; echo "config rc=$?"; grep "configs\[" $OUT/config.log | cut -c1-260
for t in 16 32 48; do
  timeout 600 python tools/cli_probe.py 8192 50000 --t $t --compression gzip > $OUT/cli_gzip_t$t.log 2>&1; echo "cli gzip t=$t rc=$?"; grep "count pass" $OUT/cli_gzip_t$t.log | tail -2
done
STRQ_NO_LIBDEFLATE=1 timeout 600 python tools/cli_probe.py 8192 50000 --t 32 --compression gzip > $OUT/cli_gzip_zlib_t32.log 2>&1; echo "cli gzip zlib t=32 rc=$?"; grep "count pass" $OUT/cli_gzip_zlib_t32.log | tail -1
timeout 600 python tools/cli_probe.py 16384 50000 --t 16 > $OUT/cli_contig_t16.log 2>&1; echo "cli contiguous rc=$?"; grep "count pass" $OUT/cli_contig_t16.log | tail -1
