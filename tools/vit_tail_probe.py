#!/usr/bin/env python3
"""How long the flanked-HMM Viterbi of a sub-batch lasts against its work: the window lengths (suffix_end - prefix_begin) of a batch of clean or
empirical-noise reads, the Viterbi stage time with and without the raised wave priority for the windows the launch waits for (STRQ_VIT_NO_PRIO).

    python tools/vit_tail_probe.py [reads=4096] [clean|empirical]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    workload = sys.argv[2] if len(sys.argv) > 2 else "empirical"
    pm, cfg = bench.load_inputs()
    from strique_amd.counter import repeatCounter
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    rc.add_target("c9orf72", repeat, prefix, suffix)
    sigs, strands, nreps = bench.make_batches_parallel(n, 50000, 0, 16, workload)
    off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    tids = [rc._classifier_for("c9orf72", s).target_id for s in strands]
    ctx = rc.ctx
    ctx.batch_upload(np.concatenate(sigs), off, tids)
    for _ in range(4):
        ctx.batch_run()
    res = ctx.batch_fetch()
    T = (res["suffix_end"] - res["prefix_begin"]).astype(np.int64)
    T = np.where((res["score_prefix"] > 0) & (res["score_suffix"] > 0) & (T > 0), T, 0)
    slots = 2048
    print("%s reads: %d windows, time steps total %.3e, per wave slot %.0f; longest windows %s; median %d" % (workload, int((T > 0).sum()), T.sum(), T.sum() / slots, np.sort(T)[-6:][::-1].tolist(), int(np.median(T[T > 0]))))
    for tag, val in (("priority for the longest windows", None), ("STRQ_VIT_NO_PRIO", "1"), ("priority for the longest windows", None)):
        ctx.set_option("STRQ_VIT_NO_PRIO", val)
        ctx.batch_run()
        t0 = time.time()
        for _ in range(3):
            ctx.batch_run()
        dt = (time.time() - t0) / 3
        tm = ctx.last_timing()
        print("  %-34s %.1f ms per pass, Viterbi %.1f ms (work bound at 0.67 us per step and slot: %.1f ms, longest window alone: %.1f ms)" % (tag, dt * 1e3, tm[6], T.sum() / slots * 0.67e-3, T.max() * 0.67e-3))


if __name__ == "__main__":
    main()
