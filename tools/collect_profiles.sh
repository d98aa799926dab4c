#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# Separate passes, as MI355X_MICROARCH.md prescribes: kernel trace + stats, then one TCC counter per
# pass and one pass of eight SQ counters (never combined with a trace domain).  Raw output lands in gpurun_out/prof_<tag>/;
# tools/summarize_profiles.py turns it into the committed profiles/<tag>_*.{md,csv,json}.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_KT="bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --synth-workers 1 --no-cpu-baseline --no-host-leg --check 0"
BENCH_PMC="bench.py --steps 1 --warmup 0 --reads 1024 --batches 1 --synth-workers 1 --no-cpu-baseline --no-host-leg --check 0"

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o "$TAG" -- python3 $BENCH_KT > "$OUT/bench_kt.log" 2>&1
echo "kernel-trace pass rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o "$TAG" -- python3 $BENCH_PMC > "$OUT/bench_fetch.log" 2>&1
echo "FETCH_SIZE pass rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o "$TAG" -- python3 $BENCH_PMC > "$OUT/bench_write.log" 2>&1
echo "WRITE_SIZE pass rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o "$TAG" -- python3 $BENCH_PMC > "$OUT/bench_sq.log" 2>&1
echo "SQ pass rc=$?"
# the same counters for the one-wave-per-table 24-bit geometry of round 1 (comparison row + the constant for packed tables)
export STRQ_SEG=1 STRQ_PACK=1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_sq_packed" -o "$TAG" -- python3 $BENCH_PMC > "$OUT/bench_sq_packed.log" 2>&1
echo "SQ pass (24-bit tables, one wave per table) rc=$?"
unset STRQ_SEG STRQ_PACK
# the kernel trace itself is large; keep stats + counters only
rm -f "$OUT"/kt/*_kernel_trace.csv "$OUT"/kt/*.db
tail -1 "$OUT/bench_kt.log" | cut -c1-400
ls -la "$OUT"/*
