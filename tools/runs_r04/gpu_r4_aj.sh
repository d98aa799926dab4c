#!/bin/bash
# round 4, GPU call AJ: the windows kernel as one wave per alignment: screen tests, the screened bench reads, a short bench
set -u
OUT=gpurun_out/r4aj
mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_screen.py "tests/test_gpu_bench_parity.py::test_benchmarked_reads_through_the_screen_all_fields" -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
( timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()}, d['roofline']['avg_launch_ms'], d['roofline']['exact_pass']['ms_per_step'])
PY
