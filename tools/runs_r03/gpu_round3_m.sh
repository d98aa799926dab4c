#!/bin/bash
# round 3, GPU call M: 16-byte-per-lane median filter + histogram kernel -- parity (conditioning and detect suites) and A/B
set -u
OUT=gpurun_out/r3m
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_detect.py tests/test_gpu_limits.py tests/test_cli_end_to_end.py tests/test_bundled_read.py -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for rep in 1 2; do
  STRQ_COND_SCALAR=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 1 > $OUT/bench_scalar_$rep.json 2> $OUT/bench_scalar_$rep.err; echo "scalar rc=$?"
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --check 2 > $OUT/bench_vec_$rep.json 2> $OUT/bench_vec_$rep.err; echo "vec rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3m/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check_ok'), round(d.get('host_inclusive_reads_per_s',0)))
    except Exception as e:
        print(f, 'ERR', e)
PY
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o m -- python3 bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --synth-workers 1 --no-cpu-baseline --no-host-leg --check 0 > $OUT/kt.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r3m/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('medfilt','quant','hist')): print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
rm -f $OUT/kt/*/*.db $OUT/kt/*/*kernel_trace.csv 2>/dev/null; true
