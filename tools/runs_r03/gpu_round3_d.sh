#!/bin/bash
# round 3, GPU call D: two waves per Viterbi window (parity, A/B against the one-wave kernel on one box), native chunk inflate in `count`
set -u
OUT=gpurun_out/r3d
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_detect.py tests/test_gpu_viterbi.py tests/test_cli_end_to_end.py tests/test_gpu_bench_parity.py -m gpu -q -x --durations=5 > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
for rep in 1 2; do
  STRQ_VIT_PAIR=0 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 1 > $OUT/bench_single_$rep.json 2> $OUT/bench_single_$rep.err; echo "single rc=$?"
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_pair_$rep.json 2> $OUT/bench_pair_$rep.err; echo "pair rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 900 python tools/fuzz_detect.py 404 40 > $OUT/fuzz_detect.log 2>&1; echo "fuzz_detect rc=$?"; tail -2 $OUT/fuzz_detect.log
timeout 900 python tools/mod_probe.py 4096 > $OUT/mod_probe.log 2>&1; echo "mod_probe rc=$?"; tail -4 $OUT/mod_probe.log
timeout 900 python tools/cli_probe.py 8192 50000 --t 16 --compression gzip > $OUT/cli_probe_50k_gzip.log 2>&1; echo "cli_probe gzip rc=$?"; tail -4 $OUT/cli_probe_50k_gzip.log
timeout 900 python tools/cli_probe.py 8192 50000 --t 32 --compression gzip > $OUT/cli_probe_50k_gzip32.log 2>&1; echo "cli_probe gzip t32 rc=$?"; tail -4 $OUT/cli_probe_50k_gzip32.log
