#!/usr/bin/env python3
"""Viterbi kernel throughput by model size (which kernel shape / waves per SIMD it lands on):
    python tools/vit_probe.py [--windows 4096] [--T 40000]
Flanked-repeat models with HMM flanks of 50 nt (the configured size, shape (4,2): 2 waves per SIMD) and of 25 nt
(about half the states: shape (2,2), 4 waves per SIMD), same observation windows."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=4096)
    ap.add_argument("--T", type=int, default=40000)
    a = ap.parse_args()
    from strique_amd import ffi, hmm
    from strique_amd.pore_model import pore_model
    t = np.load(os.path.join(ROOT, "tests", "golden", "pore_tables.npz"))
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "config.json")))
    pm = pore_model(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
    chrom, b, e, rep, pre, suf = cfg["repeat"]["c9orf72"]
    ctx = ffi.Context(0)
    rng = np.random.default_rng(5)
    for flank in (50, 25, 12):
        m = hmm.FlankedRepeatModel(rep.upper(), pre.upper()[-flank:], suf.upper()[:flank], pm, cfg["HMM"])
        mid = ctx.model_create(m.baked)
        seqs = [rng.normal(90, 12, a.T) for _ in range(8)]
        seqs = [seqs[i % 8] for i in range(a.windows)]
        ctx.viterbi_batch(mid, seqs[:64])
        t0 = time.time()
        out = ctx.viterbi_batch(mid, seqs)
        dt = time.time() - t0
        tm = ctx.last_timing()
        ne = m.baked.silent_start; ns = m.baked.n_states - ne
        print("flank %2d nt: %4d emitting + %3d silent states | %d windows x %d steps: kernel %.1f ms (call %.2f s) | %.3f us per step per window-slot of 2048"
              % (flank, ne, ns, a.windows, a.T, tm[0], dt, tm[0] * 1e3 / (a.windows * a.T / 2048.0)), flush=True)


if __name__ == "__main__":
    main()
