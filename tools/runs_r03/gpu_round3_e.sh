#!/bin/bash
# round 3, GPU call E: why is the two-wave Viterbi kernel slow?  blocks per CU sweep + kernel trace (occupancy fields)
set -u
OUT=gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
for b in 1 2 4 8 16; do
  STRQ_VIT_PAIR_BLOCKS=$b timeout 300 python bench.py --reads 1024 --steps 2 --warmup 1 --batches 1 --no-cpu-baseline --no-host-leg --check 0 > $OUT/bench_b$b.json 2> $OUT/bench_b$b.err; echo "blocks $b rc=$?"
done
STRQ_VIT_PAIR=0 timeout 300 python bench.py --reads 1024 --steps 2 --warmup 1 --batches 1 --no-cpu-baseline --no-host-leg --check 0 > $OUT/bench_single.json 2> $OUT/bench_single.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r3e -- python3 bench.py --reads 1024 --steps 2 --warmup 1 --batches 1 --no-cpu-baseline --no-host-leg --check 0 > $OUT/kt.log 2>&1
echo "kt rc=$?"
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r3e/kt/**/*kernel_trace.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(rows[0].keys())
    seen=set()
    for r in rows:
        n=r['Kernel_Name'][:50]
        if ('viterbi' in n or 'align_forward_seg' in n) and n not in seen:
            seen.add(n)
            print({k:r[k] for k in r if k in ('Kernel_Name','LDS_Block_Size','Scratch_Size','VGPR_Count','Accum_VGPR_Count','SGPR_Count','Workgroup_Size','Grid_Size','Private_Segment_Size','Group_Segment_Size','Start_Timestamp','End_Timestamp')})
PY
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o r3e -- python3 bench.py --reads 1024 --steps 1 --warmup 0 --batches 1 --no-cpu-baseline --no-host-leg --check 0 > $OUT/pmc.log 2>&1
echo "pmc rc=$?"
python - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/r3e/pmc/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'viterbi' in n: acc[(n[:40], r['Counter_Name'])]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()): print(k, v)
PY
rm -f $OUT/kt/*/*.db $OUT/kt/*/*kernel_trace.csv 2>/dev/null; true
