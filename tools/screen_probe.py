#!/usr/bin/env python3
"""What the upper-bound screen does on a batch of the benchmark's reads: stage times, strq_last_screen, the geometry of the
exact pass, second-round alignments.   python tools/screen_probe.py [reads] [read_nt] [realism]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench          # noqa: E402


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    read_nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    realism = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    pm, cfg = bench.load_inputs()
    table = synth.KmerTable(pm)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    sigs, strands = [], []
    for i in range(reads):
        s, strand = synth.make_read(table, 3, i, read_nt, (repeat, prefix, suffix), bench.REPEAT_SWEEP[i % len(bench.REPEAT_SWEEP)], realism=realism)
        sigs.append(s); strands.append(strand)
    counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    counter.add_target("c9orf72", repeat, prefix, suffix)
    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    tids = [counter._classifier_for("c9orf72", s).target_id for s in strands]
    ctx = counter.ctx
    ctx.batch_upload(np.concatenate(sigs), off, tids)
    for rep in range(3):
        ctx.device_synchronize(); t0 = time.time()
        ctx.batch_run_range(0, reads)
        res = ctx.batch_fetch()[:reads]
        ctx.device_synchronize(); dt = time.time() - t0
        tm = ctx.last_timing(); sc = ctx.last_screen(); geo = ctx.last_geometry(); sr = ctx.last_second_round(); cn = ctx.last_counters()
        print("pass %d: %.1f ms  stages(ms) lut %.2f fwd %.2f trace %.2f cond %.2f vit %.2f  launches %d" % (rep, dt * 1e3, tm[0], tm[1], tm[2], tm[5], tm[6], int(tm[7])))
        print("   screen:", {k: round(v, 2) for k, v in sc.items()})
        print("   exact pass: wave-steps %.3g columns %.3g; geometry %s; second round %s" % (cn[0], cn[1], geo, sr))
    print("counts ok:", int(np.sum(np.abs(res["count"] - np.array([bench.REPEAT_SWEEP[i % len(bench.REPEAT_SWEEP)] for i in range(reads)])) <= 2)), "of", reads)


if __name__ == "__main__":
    main()
