#!/usr/bin/env python3
"""Longer randomised run of strq_viterbi against the oracle than the test suite holds (GPU box):
    python tools/fuzz_viterbi.py SEED TRIALS
Random baked models (1 ... 500 emitting, 2 ... 250 silent states, one silent chain plus optional silent edges outside
it, counts on random states) decoded on random windows: log-probability bits, carried count and the full state path
(back-pointer build) must equal the oracle's."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import strique_oracle as orc      # noqa: E402  (checker)
from strique_amd import ffi                    # noqa: E402
from test_gpu_viterbi import _random_model     # noqa: E402


def main():
    seed, trials = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    orc.lib()
    ctx = ffi.Context(0)
    bad = 0; t0 = time.time()
    for it in range(trials):
        ne = int(rng.choice([int(rng.integers(1, 65)), int(rng.integers(65, 257)), int(rng.integers(257, 501))]))
        ns = int(rng.choice([int(rng.integers(2, 65)), int(rng.integers(65, 129)), int(rng.integers(129, 251))]))
        multi = bool(rng.random() < 0.5)
        try:
            baked = _random_model(rng, ne, ns, multi)
            mid = ctx.model_create(baked)
        except Exception as e:
            print("trial %d (%d, %d): %s" % (it, ne, ns, str(e)[:70])); continue
        for T in (1, int(rng.integers(2, 40)), int(rng.integers(100, 3000))):
            x = rng.uniform(55, 125, T)
            if rng.random() < 0.3:
                x = np.round(x)
            lo, po, co = orc.viterbi(baked, x)
            lg, cg, sg, pg = ctx.viterbi(mid, x, want_path=True)
            lg2, cg2, sg2, _ = ctx.viterbi(mid, x, want_path=False)
            if po is None:
                ok = sg == 1 and sg2 == 1
            else:
                ok = (np.float64(lo).tobytes() == np.float64(lg).tobytes() == np.float64(lg2).tobytes() and co == cg == cg2
                      and np.array_equal(po, pg) and sg == 0)
            if not ok:
                bad += 1
                print("MISMATCH trial %d ne=%d ns=%d multi=%s T=%d: oracle %r %r  gpu %r %r / %r %r" % (it, ne, ns, multi, T, lo, co, lg, cg, lg2, cg2), flush=True)
    print("seed %d: %d models x 3 windows, %d mismatches, %.0f s" % (seed, trials, bad, time.time() - t0))


if __name__ == "__main__":
    main()
