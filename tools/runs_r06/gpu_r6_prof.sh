#!/bin/bash
# round 6: the rocprofv3 evidence of the default line (two sub-batches in flight: the Viterbi launch of step k under the screen of step k + 1):
# kernel stats of the pipelined run, HBM counters and SQ counters (separate passes, counters only); tag r06.
# bench.py prints a compact line now: the full record of every pass goes to $P/bench_<pass>.json (--detail).
set -u
P=gpurun_out/prof_r06
mkdir -p $P
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
KT="bench.py --steps 4 --warmup 1 --reads 4096 --batches 2 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0"
PMC="bench.py --steps 1 --warmup 0 --reads 1024 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0"
echo "$KT" > $P/cmd_kt.txt; echo "$PMC" > $P/cmd_pmc.txt
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -o r06 -- python3 $KT --detail $P/bench_kt.json > "$P/bench_kt.log" 2>&1; echo "kernel-trace pass rc=$?"
python3 tools/overlap_report.py $(find $P/kt -name "*kernel_trace.csv" | head -1) 400 > $P/overlap.txt 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/pmc_fetch" -o r06 -- python3 $PMC --detail $P/bench_fetch.json > "$P/bench_fetch.log" 2>&1; echo "FETCH_SIZE pass rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/pmc_write" -o r06 -- python3 $PMC --detail $P/bench_write.json > "$P/bench_write.log" 2>&1; echo "WRITE_SIZE pass rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$P/pmc_sq" -o r06 -- python3 $PMC --detail $P/bench_sq.json > "$P/bench_sq.log" 2>&1; echo "SQ pass rc=$?"
STRQ_SCREEN_MODE=fine timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$P/pmc_sq_fine" -o r06 -- python3 $PMC --detail $P/bench_sq_fine.json > "$P/bench_sq_fine.log" 2>&1; echo "SQ (fine screen) pass rc=$?"
rm -f "$P"/kt/*_kernel_trace.csv "$P"/kt/*.db "$P"/kt/*/*_kernel_trace.csv
ls -la $P/*
