#!/bin/bash
# round 3, GPU call C: ping-pong forward DP parity, Viterbi variants A/B on one box, count-from-files throughput
set -u
OUT=gpurun_out/r3c
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_align.py tests/test_gpu_bench_parity.py tests/test_cli_end_to_end.py tests/test_gpu_shim.py -m gpu -q -x --durations=5 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
for rep in 1 2; do
for v in default head_pipe prevvit pipe_f0 pipe_f1 pipe_f2 nopipe; do
  if [ $v = default ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 1 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err; echo "$v rc=$?"
done
done
unset STRQ_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 900 python tools/cli_probe.py 16384 50000 --t 8 > $OUT/cli_probe_50k.log 2>&1; echo "cli_probe 50k rc=$?"; tail -5 $OUT/cli_probe_50k.log
timeout 900 python tools/cli_probe.py 16384 50000 --t 16 > $OUT/cli_probe_50k_t16.log 2>&1; echo "cli_probe 50k t16 rc=$?"; tail -4 $OUT/cli_probe_50k_t16.log
timeout 900 python tools/cli_probe.py 8192 50000 --t 16 --compression gzip > $OUT/cli_probe_50k_gzip.log 2>&1; echo "cli_probe gzip rc=$?"; tail -4 $OUT/cli_probe_50k_gzip.log
timeout 600 python tools/cli_probe.py 16384 10000 --t 8 > $OUT/cli_probe_10k.log 2>&1; echo "cli_probe 10k rc=$?"; tail -4 $OUT/cli_probe_10k.log
