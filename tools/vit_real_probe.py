#!/usr/bin/env python3
"""Viterbi cost per time step on the real read the reference bundles against synthetic reads (GPU box).

    python tools/vit_real_probe.py [copies]

The flanked-model Viterbi's chain sweeps run until nothing changes: how long that takes depends on the signal.
`copies` copies of data/c9orf72.fast5 (tests/golden/bundled_read.npz; 284 k samples, 733 repeats) go through the resident
pipeline next to synthetic reads of the same length and repeat count at realism 0 / 0.5 / 1; printed is the Viterbi stage time
divided by the time steps decoded (strq_last_counters[7]) and by the 2048 wave slots of the chip."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    pm, cfg = bench.load_inputs()
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    real = np.load(os.path.join(ROOT, "tests", "golden", "bundled_read.npz"))["signal"].astype(np.int16)
    table = synth.KmerTable(pm)
    sets = [("data/c9orf72.fast5 x %d" % copies, [real] * copies, ["-"] * copies)]
    nt = int(len(real) / 7.5)
    for r in (0.0, 0.5, 1.0):
        sigs, strands = [], []
        for i in range(copies):
            s, st = synth.make_read(table, 51, i % 64, nt, (repeat, prefix, suffix), 733, realism=r)      # 64 distinct reads, repeated
            sigs.append(s); strands.append(st)
        sets.append(("synthetic, realism %.1f, %d nt, 733 repeats" % (r, nt), sigs, strands))
    print("| input | reads | Viterbi ms | time steps | ns per time step and wave slot | forward DP ms | counts |")
    print("|---|---|---|---|---|---|---|")
    for name, sigs, strands in sets:
        counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
        counter.add_target("c9orf72", repeat, prefix, suffix)
        off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
        tids = [counter._classifier_for("c9orf72", s).target_id for s in strands]
        ctx = counter.ctx
        ctx.batch_upload(np.concatenate(sigs), off, tids)
        ctx.batch_run(); ctx.batch_run()
        tm = ctx.last_timing(); cn = ctx.last_counters(); res = ctx.batch_fetch()
        counts = sorted(set(int(c) for c in res["count"]))
        print("| %s | %d | %.1f | %.3g | %.0f | %.1f | %s |" % (name, len(sigs), tm[6], cn[7], tm[6] * 1e6 / max(1.0, cn[7] / 2048.0), tm[1],
              (str(counts[0]) if len(counts) == 1 else "%d ... %d" % (counts[0], counts[-1]))), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
