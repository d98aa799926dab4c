#!/bin/bash
# round 4, GPU call Z: the screen kernel compiled for 4 / 5 / 6 / 8 waves per SIMD and with the max-ILP scheduler, A/B on one box (tools/screen_probe.py, 1024 reads);
# degraded reads with the class-demotion fix
set -u
OUT=gpurun_out/r4z
mkdir -p $OUT
run() {  # name tables lib
  if [ -n "$3" ]; then export STRQ_LIB=$PWD/tools/bin/lib_$3.so; else unset STRQ_LIB; fi
  STRQ_SCREEN_TABLES=$2 timeout 200 python tools/screen_probe.py 1024 > $OUT/probe_$1.log 2>&1
  echo "$1: $(grep 'pass 2' $OUT/probe_$1.log | cut -c1-120)  screen ms $(grep -A1 'pass 2' $OUT/probe_$1.log | grep -o "'ms': [0-9.]*")"
}
run intree_t6 6 ""
run w5_t5 5 scr_w5
run w4_t4 4 scr_w4
run w8_t8 8 scr_w8
run ilp_t6 6 scr_ilp
run intree_t6b 6 ""
unset STRQ_LIB
timeout 400 python tools/realism_bench.py --reads 2048 0.5 1.0 > $OUT/realism_screen.md 2> $OUT/realism_screen.err; echo "realism rc=$?"; cat $OUT/realism_screen.md
