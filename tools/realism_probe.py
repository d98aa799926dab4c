#!/usr/bin/env python3
"""Where do the flank scores of synthetic reads sit against the real read the reference bundles?  (CPU, oracle only.)

    python tools/realism_probe.py [realism ...]

For every realism level (strique_amd/synth.py: make_signal) a few 10 kb C9orf72 reads go through the oracle's conditioning and
flank alignment; printed is the raw best score of the two alignments as a fraction of the maximum (rows x dist_offset) --
the quantity the library plans its column-segment overlap with -- next to the same fraction for data/c9orf72.fast5
(tests/golden/bundled_read.npz) and the recovered repeat counts."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import strique_oracle as orc          # noqa: E402
from strique_amd import synth                     # noqa: E402
from strique_amd.pore_model import pore_model     # noqa: E402


def fractions(sig, tc, opm, params, dist_offset):
    flt, u8, morph, fltn = orc.condition(np.asarray(sig), opm)
    out = []
    for key in ("prefix_ext", "suffix_ext"):
        score = orc.align_overlap(morph, tc[key], params, want_idx=False)[0]
        out.append(float(score) / (len(tc[key]) * dist_offset))
    return out


def main():
    levels = [float(v) for v in sys.argv[1:]] or [0.0, 0.25, 0.5, 0.75, 1.0]
    t = np.load(os.path.join(ROOT, "tests", "golden", "pore_tables.npz"))
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "config.json")))
    pm = pore_model(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
    opm = orc.PoreModel(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
    params = orc.align_params(cfg["align"])
    dist_offset = float(cfg["align"]["dist_offset"])
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    table = synth.KmerTable(pm)
    real = np.load(os.path.join(ROOT, "tests", "golden", "bundled_read.npz"))
    sig = real[[k for k in real.files if "sig" in k.lower() or "raw" in k.lower()][0]]
    tc = orc.classifier(repeat, prefix, suffix, "-", opm, None, cfg["HMM"])
    fr = fractions(sig, tc, opm, params, dist_offset)
    print("| input | flank score / maximum (prefix, suffix) | counts recovered |")
    print("|---|---|---|")
    print("| data/c9orf72.fast5 (real, - strand) | %.3f, %.3f | %s |" % (fr[0], fr[1], orc.detect(sig, tc, opm, params)[0][0]))
    for r in levels:
        fs, cnt = [], []
        for i in range(6):
            s, strand = synth.make_read(table, 21, i, 10000, (repeat, prefix, suffix), 40 + 10 * i, realism=r)
            tc = orc.classifier(repeat, prefix, suffix, strand, opm, None, cfg["HMM"])
            fs += fractions(s, tc, opm, params, dist_offset)
            cnt.append("%d/%d" % (orc.detect(s, tc, opm, params)[0][0], 40 + 10 * i))
        print("| synthetic, realism %.2f | median %.3f (min %.3f, max %.3f) | %s |" % (r, np.median(fs), min(fs), max(fs), " ".join(cnt)))


if __name__ == "__main__":
    main()
