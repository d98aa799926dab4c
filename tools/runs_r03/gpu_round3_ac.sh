#!/bin/bash
# round 3, GPU call AC: three waves per SIMD for the register-resident Viterbi where it does not spill (odd variant, exchange level 1) vs two;
# a longer count-from-files probe (65 536 reads)
set -u
OUT=gpurun_out/r3ac
mkdir -p $OUT
for v in lx1w8 lx1w12 lx2w8 lx2w12; do
  unset STRQ_VIT_G2_WAVES STRQ_VIT_G2_LDS
  case $v in lx1w8) export STRQ_VIT_G2_LDS=1 STRQ_VIT_G2_WAVES=8;; lx1w12) export STRQ_VIT_G2_LDS=1 STRQ_VIT_G2_WAVES=12;; lx2w8) export STRQ_VIT_G2_LDS=2 STRQ_VIT_G2_WAVES=8;; lx2w12) export STRQ_VIT_G2_LDS=2 STRQ_VIT_G2_WAVES=12;; esac
  timeout 600 python tools/config_probe.py 4096 > $OUT/config_$v.log 2>&1; echo "$v rc=$?"; grep "configs\[3\]" $OUT/config_$v.log | cut -c1-260
done
unset STRQ_VIT_G2_WAVES STRQ_VIT_G2_LDS
timeout 900 python tools/cli_probe.py 65536 50000 --t 8 > $OUT/cli_probe_64k.log 2>&1; echo "cli_probe rc=$?"; tail -3 $OUT/cli_probe_64k.log
