#!/bin/bash
# round 5, call y: the final state -- rocprofv3 evidence (tag r05), the default command as the driver runs it, the whole GPU suite, smoke
set -u
OUT=gpurun_out/r5y; mkdir -p $OUT
P=gpurun_out/prof_r05
rm -rf $P; mkdir -p $P
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH_KT="bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0"
BENCH_PMC="bench.py --steps 1 --warmup 0 --reads 1024 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -o r05 -- python3 $BENCH_KT > "$P/bench_kt.log" 2>&1; echo "kernel-trace pass rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/pmc_fetch" -o r05 -- python3 $BENCH_PMC > "$P/bench_fetch.log" 2>&1; echo "FETCH_SIZE pass rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/pmc_write" -o r05 -- python3 $BENCH_PMC > "$P/bench_write.log" 2>&1; echo "WRITE_SIZE pass rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$P/pmc_sq" -o r05 -- python3 $BENCH_PMC > "$P/bench_sq.log" 2>&1; echo "SQ pass rc=$?"
STRQ_SCREEN_MODE=fine timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$P/pmc_sq_fine" -o r05 -- python3 $BENCH_PMC > "$P/bench_sq_fine.log" 2>&1; echo "SQ pass (fine screen) rc=$?"
rm -f "$P"/kt/*_kernel_trace.csv "$P"/kt/*.db "$P"/kt/*/*_kernel_trace.csv
T0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
timeout 3000 python -m pytest tests -m gpu -q -x > $OUT/tests_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -3 $OUT/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
