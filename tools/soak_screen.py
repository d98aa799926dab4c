#!/usr/bin/env python3
"""The screens' adaptive state machine under a changing workload: one context with the default policy (coarse screen -> fine screen
-> none, pauses that double, retries, cold-start transients) sees a random sequence of batches -- clean reads and reads with the
empirical noise of the bundled real read, 20 / 30 / 50 kb, 64 ... 768 reads -- and every batch's rows must equal, byte for byte,
those of a second context that never screens (STRQ_NO_SCREEN=1).  Prints the mode the first context chose per batch.

    python tools/soak_screen.py [batches=40] [seed=1]
"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from strique_amd.counter import repeatCounter
    pm, cfg = bench.load_inputs()
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    ctxs = []
    for opts in ({}, {"STRQ_NO_SCREEN": "1"}):
        rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
        rc.add_target("c9orf72", repeat, prefix, suffix)
        for k, v in opts.items():
            rc.ctx.set_option(k, v)
        ctxs.append(rc)
    rng = np.random.default_rng(seed)
    modes = collections.Counter(); bad = 0; first_read = 0; t_a = t_b = 0.0
    for bi in range(n_batches):
        workload = "empirical" if rng.random() < 0.45 else "clean"
        nt = int(rng.choice([20000, 30000, 50000], p=[0.2, 0.3, 0.5]))
        n = int(rng.choice([64, 128, 256, 512, 768]))
        sigs, strands, _ = bench.make_batches_parallel(n, nt, first_read, 16, workload)
        first_read += n
        off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
        flat = np.concatenate(sigs)
        tids = [ctxs[0]._classifier_for("c9orf72", s).target_id for s in strands]
        t0 = time.time(); got = ctxs[0].ctx.detect_batch(flat, off, tids); t1 = time.time(); want = ctxs[1].ctx.detect_batch(flat, off, tids); t2 = time.time()
        t_a += t1 - t0; t_b += t2 - t1
        ls = ctxs[0].ctx.last_screen()
        mode = "%s%s" % (ls.get("mode"), "" if ls.get("mode") != "coarse" else "/%s" % ls.get("merge"))
        modes[(workload, mode)] += 1
        same = got.tobytes() == want.tobytes()
        bad += not same
        print("batch %2d: %-9s %5d nt %4d reads  screen %-9s pauses coarse %3s fine %3s  second round %s  %s" % (
            bi, workload, nt, n, mode, ls.get("coarse_pause"), ls.get("fine_pause"), ctxs[0].ctx.last_second_round(), "same rows" if same else "ROWS DIFFER"), flush=True)
    print("modes:", dict(modes))
    print("%d batches, %d with different rows; adaptive %.1f s, no screen %.1f s" % (n_batches, bad, t_a, t_b))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
