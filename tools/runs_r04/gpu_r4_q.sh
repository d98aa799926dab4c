#!/bin/bash
# round 4, GPU call Q: test-first loop behind the unconditional chain sweeps (-DSTRQ_G2_TESTFIRST, 1 / 2 / 3 pre-sweeps), A/B on one box + parity of the candidate
set -u
OUT=gpurun_out/r4q
mkdir -p $OUT
for v in intree vit_tf1 vit_tf vit_tf3 intree; do
  if [ $v = intree ]; then unset STRQ_LIB; else export STRQ_LIB=$PWD/tools/bin/lib_$v.so; fi
  ( timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg ) > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), d['check_ok'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()})
PY
done
export STRQ_LIB=$PWD/tools/bin/lib_vit_tf.so
timeout 300 python tools/fuzz_g2.py 91 100 > $OUT/fuzz_g2_tf.log 2>&1; echo "fuzz_g2 tf rc=$?"; tail -1 $OUT/fuzz_g2_tf.log
timeout 600 python -m pytest tests/test_gpu_viterbi.py -m gpu -q > $OUT/tests_tf.log 2>&1; echo "viterbi tests (tf) rc=$?"; tail -1 $OUT/tests_tf.log
timeout 200 python tools/config_probe.py 4096 > $OUT/config_tf.log 2>&1; grep "configs\[3" $OUT/config_tf.log | cut -c1-230
