#!/bin/bash
# round 3, GPU call B: pipelined Viterbi step (A/B against the previous library on one box), NaN windows, 64 strips, count-from-files profile
set -u
OUT=gpurun_out/r3b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_viterbi.py tests/test_gpu_detect.py tests/test_gpu_limits.py -m gpu -q -x --durations=8 > $OUT/tests_vit.log 2>&1
echo "viterbi/detect tests rc=$?"; tail -4 $OUT/tests_vit.log
timeout 900 python tools/fuzz_detect.py 303 60 > $OUT/fuzz_detect.log 2>&1; echo "fuzz_detect rc=$?"; tail -2 $OUT/fuzz_detect.log
for rep in 1 2; do
  STRQ_LIB=$PWD/tools/bin/lib_prevvit.so timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 0 > $OUT/bench_prev_$rep.json 2> $OUT/bench_prev_$rep.err; echo "prev rc=$?"
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_new_$rep.json 2> $OUT/bench_new_$rep.err; echo "new rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 900 python -m pytest tests/test_gpu_align.py -m gpu -q -x -k "long_flanks or strip_limit or flank_shapes" > $OUT/tests_strips.log 2>&1
echo "strip tests rc=$?"; tail -3 $OUT/tests_strips.log
timeout 900 python tools/cli_probe.py 4096 50000 --t 8 --profile > $OUT/cli_probe.log 2>&1; echo "cli_probe rc=$?"; tail -40 $OUT/cli_probe.log
