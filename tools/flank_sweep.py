#!/usr/bin/env python3
"""Forward-DP rate against the flank length (GPU box): does every flank of the 14-rows-per-lane shape run the same loop?

    python tools/flank_sweep.py [--reads 512] [--read-nt 50000] [--from 134] [--to 154]

One resident batch of synthetic 50 kb reads, one target per flank length L (prefix and suffix of L nt: m = 6 (L - 5) flank
rows, 774 ... 894 for L = 134 ... 154), the worst-case overlap pinned (STRQ_OVERLAP=0: one round, no dependence on whether
the random flanks are found).  Prints the forward-DP time per million wave-steps (strq_last_counters[0]: one wave-step = two DP
columns of every flank row of a piece) and its ratio to STRique's own L = 150 (m = 870)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=512)
    ap.add_argument("--read-nt", type=int, default=50000)
    ap.add_argument("--from", dest="lo", type=int, default=134)
    ap.add_argument("--to", dest="hi", type=int, default=154)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    os.environ["STRQ_OVERLAP"] = "0"
    from strique_amd.counter import repeatCounter
    pm, cfg = bench.load_inputs()
    sigs, strands, nreps = bench.make_batches_parallel(a.reads, a.read_nt, 0, 8)
    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    flat = np.concatenate(sigs)
    rng = np.random.default_rng(7)
    nt = lambda n: "".join(rng.choice(list("ACGT"), n))
    counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rows = []
    for L in list(range(a.lo, a.hi + 1)) + [150]:
        name = "f%d_%d" % (L, len(rows))
        counter.add_target(name, "GGCCCC", nt(L), nt(L))
        tids = [counter._classifier_for(name, s).target_id for s in strands]
        ctx = counter.ctx
        ctx.batch_upload(flat, off, tids)
        ctx.batch_run()
        fwd = 0.0; steps = 0.0
        for _ in range(a.steps):
            ctx.batch_run(); fwd += float(ctx.last_timing()[1]); steps += float(ctx.last_counters()[0])
        geo = ctx.last_geometry()
        rows.append((L, 6 * (L - 5), (6 * (L - 5) - 1) % max(1, geo["rows_per_lane"]), geo["rows_per_lane"], geo["waves_per_alignment"], fwd / a.steps, fwd / steps * 1e6))
    ref = [r for r in rows if r[0] == 150][-1][6]
    print("| flank nt | rows m | (m-1) %% R | R | waves per alignment | forward DP ms | ms per 10^6 wave-steps | against m = 870 |")
    print("|---|---|---|---|---|---|---|---|")
    for L, m, rm, R, seg, ms, rate in rows:
        print("| %d | %d | %d | %d | %d | %.2f | %.4f | %+.1f %% |" % (L, m, rm, R, seg, ms, rate, (rate / ref - 1) * 100))


if __name__ == "__main__":
    main()
