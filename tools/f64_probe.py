#!/usr/bin/env python3
"""float64 reads against int16 reads of the same signals through repeatCounter.detect_batch (host buffers, one per read):
the float64 path takes its order statistics with f64_stats_kernel (radix selection + numpy's summation tree on the GPU).

    python tools/f64_probe.py [reads=512] [read_nt=50000]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    from strique_amd.counter import repeatCounter
    pm, cfg = bench.load_inputs()
    sigs, strands, nreps = bench.make_batches_parallel(n, nt, 0, 16)
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    rc.add_target("c9orf72", repeat, prefix, suffix)
    rows = {}
    for name, cast in (("int16", np.int16), ("float64", np.float64)):
        items = [("c9orf72", s.astype(cast), st) for s, st in zip(sigs, strands)]
        rc.detect_batch(items)
        t0 = time.time(); out = rc.detect_batch(items); dt = time.time() - t0
        tm = rc.ctx.last_timing()
        rows[name] = out
        print("%s: %d reads %.3f s  %.0f reads/s  stages(ms) cond %.1f fwd %.1f vit %.1f; counts within 2: %d" % (
            name, n, dt, n / dt, tm[5], tm[1], tm[6], sum(abs(o[0] - k) <= 2 for o, k in zip(out, nreps))), flush=True)
    print("same rows:", rows["int16"] == rows["float64"])


if __name__ == "__main__":
    main()
