#!/usr/bin/env python3
"""Randomised run of the register-resident Viterbi kernel (viterbi_g2_kernel) against the oracle (GPU box):
    python tools/fuzz_g2.py SEED TRIALS
Random flanked-repeat models (repeat units of 1 ... 14 nt, flanks of 12 ... 50 nt, random HMM transition settings within
the reference's ranges), windows with deletions, insertions, clipped stretches, missing observations and pure noise, decoded
count-only (the register-resident image, all exchange levels) and with the state path (lane layout): log-probability bits and
count must equal the oracle's everywhere, and the path build must agree with the count build."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import strique_oracle as orc      # noqa: E402  (checker)
from strique_amd import ffi, hmm               # noqa: E402
import bench                                   # noqa: E402  (pore model tables)


def main():
    seed, trials = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    orc.lib()
    ctx = ffi.Context(0)
    pm, cfg = bench.load_inputs()
    bad = 0; g2 = 0; t0 = time.time(); windows = 0
    for it in range(trials):
        rl = int(rng.integers(1, 15)); repeat = "".join(rng.choice(list("ACGT"), rl))
        if len(set(repeat)) == 1 and rl > 1:
            repeat = repeat[:-1] + ("A" if repeat[0] != "A" else "C")
        pl, sl = int(rng.integers(12, 51)), int(rng.integers(12, 51))
        prefix = "".join(rng.choice(list("ACGT"), pl)); suffix = "".join(rng.choice(list("ACGT"), sl))
        conf = dict(cfg["HMM"])
        if rng.random() < 0.5:
            conf.update({"rep_std_scale": float(rng.uniform(0.8, 2.0)), "seq_std_scale": float(rng.uniform(0.8, 2.0)), "e1_ratio": float(rng.uniform(0.02, 0.5))})
        try:
            fm = hmm.FlankedRepeatModel(repeat, prefix, suffix, pm, conf)
            mid = ctx.model_create(fm.baked)
        except Exception as e:
            print("trial %d (%s): %s" % (it, repeat, str(e)[:90])); continue
        g2 += ctx.last_positions_rc == 0
        seqs = []
        for k in range(6):
            nrep = int(rng.integers(0, 40))
            seq = prefix + repeat * nrep + suffix
            if rng.random() < 0.5 and len(seq) > 30:
                a = int(rng.integers(5, len(seq) - 12)); seq = seq[:a] + seq[a + int(rng.integers(1, 9)):]
            if rng.random() < 0.3:
                a = int(rng.integers(5, len(seq) - 6)); seq = seq[:a] + "".join(rng.choice(list("ACGT"), int(rng.integers(1, 7)))) + seq[a:]
            if len(seq) < pm.kmer + 1:
                continue
            x = pm.generate_signal(seq, samples=int(rng.integers(3, 10)), noise=True, rng=rng)
            x = np.clip(x, pm.model_min + .5, pm.model_max - .5)
            r = rng.random()
            if r < 0.15:
                x = np.where(rng.random(len(x)) < 0.2, np.nan, x)
            elif r < 0.25:
                x = rng.uniform(pm.model_min - 5, pm.model_max + 5, len(x))      # leaves the uniform supports: the general emission code
            seqs.append(x)
        seqs.append(np.full(int(rng.integers(1, 90)), np.nan))
        want = [orc.viterbi(fm.baked, s, want_path=False) for s in seqs]
        for env in ({"STRQ_VIT_G2_LDS": "2"}, {"STRQ_VIT_G2_LDS": "1"}, {"STRQ_VIT_G2_LDS": "0"}, {"STRQ_VIT_NO_G2": "1"}):
            for k in ("STRQ_VIT_G2_LDS", "STRQ_VIT_NO_G2"):
                os.environ.pop(k, None)
            os.environ.update(env)
            lg, cg, sg, _ = ctx.viterbi_batch(mid, seqs)
            for i, (lo, _, co) in enumerate(want):
                windows += 1
                if not np.isfinite(lo):
                    ok = not np.isfinite(lg[i]) or sg[i] == 1
                else:
                    ok = np.float64(lo).tobytes() == np.float64(lg[i]).tobytes() and co == cg[i] and sg[i] == 0
                if not ok:
                    bad += 1
                    print("MISMATCH trial %d repeat %s flanks %d/%d window %d (T=%d) %s: oracle %r %r  gpu %r %r status %d" % (it, repeat, pl, sl, i, len(seqs[i]), env, lo, co, lg[i], cg[i], sg[i]), flush=True)
        for k in ("STRQ_VIT_G2_LDS", "STRQ_VIT_NO_G2"):
            os.environ.pop(k, None)
    print("seed %d: %d models (%d with a register-resident image), %d window decodes, %d mismatches, %.0f s" % (seed, trials, g2, windows, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
