#!/usr/bin/env python3
"""Longer randomised run of strq_align_overlap against the oracle than the test suite holds (GPU box):
    python tools/fuzz_align.py SEED TRIALS
flank sizes 1 ... 1024 classes, every `samples` run length, collapsed and general affine gap parameters, level
spacings that give narrow and wide score bands, reads from 1 to 60 000 samples (several column segments), the flank
planted or not; score bits, end / start column, per-row record and both index lists must equal the oracle's."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import strique_oracle as orc      # noqa: E402  (checker)
from strique_amd import ffi                    # noqa: E402


def toy(rng, n, k, s, scale, plant):
    cls = rng.uniform(60, 120, k).astype(np.float32)
    flank = np.repeat(cls, s)
    lval = (40 + scale * np.arange(256)).astype(np.float32)
    lv = np.repeat(rng.integers(30, 200, n // 3 + 1), rng.integers(3, 10, n // 3 + 1))[:n].astype(np.uint8)
    if plant and n > 4:
        emb = np.repeat(np.clip(np.round((cls - 40) / scale), 0, 255).astype(np.uint8), rng.integers(max(1, s - 1), s + 4, k))
        pos = int(rng.integers(0, max(1, n - len(emb))))
        emb = emb[:max(0, n - pos)]
        lv[pos:pos + len(emb)] = emb[:len(lv) - pos]
    return lval[lv], flank


def main():
    seed, trials = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    orc.lib()
    ctx = ffi.Context(0)
    bad = 0; t0 = time.time()
    for it in range(trials):
        s = int(rng.choice([6, 6, 6, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 17]))
        kmax = 1024 if s != 17 else 60
        k = int(rng.choice([1, 2, 7, 33, 64, 128, 145, 158, 159, 200, 300, 512, 1024, int(rng.integers(1, 1025))]))
        k = min(k, kmax)
        n = int(rng.choice([int(rng.integers(1, 400)), int(rng.integers(400, 8000)), int(rng.integers(8000, 60000))]))
        scale = float(rng.choice([0.45, 0.3, 0.12]))
        if rng.random() < 0.5:
            e_h = -float(rng.integers(1, 4)); e_v = -float(rng.integers(2, 20)); params = [e_h, e_h, e_v, e_v]
        else:
            params = [-float(rng.integers(1, 6)), -float(rng.integers(1, 4)), -float(rng.integers(4, 24)), -float(rng.integers(1, 18))]
        params += [float(rng.choice([8.0, 16.0, 12.5])), float(rng.choice([0.0, 0.0, -2.0, -16.0]))]
        ctx.set_align_params(*params)
        a, flank = toy(rng, n, k, s, scale, rng.random() < 0.8)
        o = orc.align_overlap(a, flank, np.array(params, np.float32), want_idx=True)
        g = ctx.align_overlap(a, flank, want_idx=True)
        ok = (np.float32(o[0]).tobytes() == np.float32(g[0]).tobytes() and o[4] == g[4] and o[5] == g[5] and np.array_equal(o[3], g[3])
              and np.array_equal(o[1], g[1]) and np.array_equal(o[2], g[2]))
        if not ok:
            bad += 1
            print("MISMATCH it=%d s=%d k=%d n=%d scale=%g params=%s: oracle %r %r %r  gpu %r %r %r" % (it, s, k, n, scale, params, o[0], o[4], o[5], g[0], g[4], g[5]), flush=True)
    print("seed %d: %d alignments, %d mismatches, %.0f s" % (seed, trials, bad, time.time() - t0))


if __name__ == "__main__":
    main()
