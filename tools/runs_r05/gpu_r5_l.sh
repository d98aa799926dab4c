#!/bin/bash
# round 5, call l: the rocprofv3 evidence of the default line (coarse screen, three flank rows per DP row): kernel stats, HBM counters, SQ counters; tag r05
set -u
P=gpurun_out/prof_r05
mkdir -p $P
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH_KT="bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0"
BENCH_PMC="bench.py --steps 1 --warmup 0 --reads 1024 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -o r05 -- python3 $BENCH_KT > "$P/bench_kt.log" 2>&1; echo "kernel-trace pass rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/pmc_fetch" -o r05 -- python3 $BENCH_PMC > "$P/bench_fetch.log" 2>&1; echo "FETCH_SIZE pass rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/pmc_write" -o r05 -- python3 $BENCH_PMC > "$P/bench_write.log" 2>&1; echo "WRITE_SIZE pass rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$P/pmc_sq" -o r05 -- python3 $BENCH_PMC > "$P/bench_sq.log" 2>&1; echo "SQ pass rc=$?"
rm -f "$P"/kt/*_kernel_trace.csv "$P"/kt/*.db "$P"/kt/*/*_kernel_trace.csv
ls -la $P/*
