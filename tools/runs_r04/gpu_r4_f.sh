#!/bin/bash
# round 4, GPU call F: Viterbi cost per time step on the real read against synthetic ones; three Viterbi waves per SIMD now that the
# kernel needs 164 -- 168 VGPRs; more mapped files kept open by the reader (32 instead of 4): reader probe and `count` from gzip files
set -u
OUT=gpurun_out/r4f
mkdir -p $OUT
timeout 600 python tools/vit_real_probe.py 2048 > $OUT/vit_real.md 2> $OUT/vit_real.err; echo "vit real rc=$?"; cat $OUT/vit_real.md
for w in 8 12; do for lx in 2 1; do
  STRQ_VIT_G2_WAVES=$w STRQ_VIT_G2_LDS=$lx timeout 300 python tools/config_probe.py 4096 > $OUT/config_w${w}_lx$lx.log 2>&1; echo "waves $w lds $lx rc=$?"; grep "configs\[3" $OUT/config_w${w}_lx$lx.log | cut -c1-230
done; done
for w in 8 12; do
  ( STRQ_VIT_G2_WAVES=$w timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench_w$w.json 2> $OUT/bench_w$w.err
  python - $OUT/bench_w$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k:d.get(k) for k in ('value','ms_per_step','check_ok')}, d['stage_ms_per_step'])
PY
done
timeout 600 python tools/reader_probe.py 8192 50000 > $OUT/reader_probe.log 2>&1; echo "reader probe rc=$?"; grep "libdeflate one" $OUT/reader_probe.log
for t in 24 32 48; do
  timeout 900 python tools/cli_probe.py 32768 50000 --t $t --compression gzip > $OUT/cli_gzip_32k_t$t.log 2>&1; echo "cli gzip 32k t=$t rc=$?"; grep "count pass" $OUT/cli_gzip_32k_t$t.log | tail -1
done
