#!/bin/bash
# round 4, GPU call Y: the screen's own tests again (pause test rewritten), what it does on degraded reads and on 10 kb reads, and the rocprofv3
# evidence of the default line with the screen (kernel stats, HBM counters, SQ counters; tag r04s)
set -u
OUT=gpurun_out/r4y
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_screen.py tests/test_gpu_bench_parity.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "screen + bench-parity tests rc=$?"; tail -5 $OUT/tests.log
timeout 400 python tools/realism_bench.py --reads 2048 > $OUT/realism_screen.md 2> $OUT/realism_screen.err; echo "realism rc=$?"; cat $OUT/realism_screen.md
STRQ_NO_SCREEN=1 timeout 400 python tools/realism_bench.py --reads 2048 > $OUT/realism_noscreen.md 2> $OUT/realism_noscreen.err; echo "realism (no screen) rc=$?"; cat $OUT/realism_noscreen.md
timeout 300 python tools/config_probe.py 4096 > $OUT/config_screen.log 2>&1; grep "configs\[" $OUT/config_screen.log | cut -c1-230
STRQ_NO_SCREEN=1 timeout 300 python tools/config_probe.py 4096 > $OUT/config_noscreen.log 2>&1; grep "configs\[" $OUT/config_noscreen.log | cut -c1-230
P=gpurun_out/prof_r04s
mkdir -p $P
export TMPDIR=/tmp
BENCH_KT="bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --check 0"
BENCH_PMC="bench.py --steps 1 --warmup 0 --reads 1024 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --check 0"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -o r04s -- python3 $BENCH_KT > "$P/bench_kt.log" 2>&1; echo "kernel-trace pass rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/pmc_fetch" -o r04s -- python3 $BENCH_PMC > "$P/bench_fetch.log" 2>&1; echo "FETCH_SIZE pass rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/pmc_write" -o r04s -- python3 $BENCH_PMC > "$P/bench_write.log" 2>&1; echo "WRITE_SIZE pass rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$P/pmc_sq" -o r04s -- python3 $BENCH_PMC > "$P/bench_sq.log" 2>&1; echo "SQ pass rc=$?"
rm -f "$P"/kt/*_kernel_trace.csv "$P"/kt/*.db
ls -la $P/*
