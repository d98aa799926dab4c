#!/bin/bash
# which launch of the exact pass is which: the launch plan (STRQ_DEBUG) next to a kernel trace of the same steps
mkdir -p gpurun_out/r5z8
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5z8/build.log 2>&1
B="bench.py --steps 2 --warmup 2 --reads 4096 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0"
STRQ_DEBUG=1 timeout 600 python $B > gpurun_out/r5z8/dbg.json 2> gpurun_out/r5z8/dbg.err; echo "rc=$?"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5z8/kt -o x -- python3 $B > gpurun_out/r5z8/kt.log 2>&1; echo "rc=$?"
find gpurun_out/r5z8/kt -name "*.db" -delete
grep -v "^W2026\|^E2026" gpurun_out/r5z8/dbg.err | tail -60
