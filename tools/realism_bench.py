#!/usr/bin/env python3
"""Throughput on degraded reads (GPU box): what do lower flank scores cost?

    python tools/realism_bench.py [--reads 4096] [--read-nt 50000] [realism ...]

bench.py quotes reads/s on the SURVEY.md 8d recipe -- the easiest input the overlap heuristic of the forward DP will ever see
(flank scores at 0.79 of the maximum: short overlap, no second round).  Here the same pipeline runs resident batches synthesised
at several `realism` levels (strique_amd/synth.py: make_signal; 1.0 = the flank scores of the real read the reference bundles)
and reports, per level: reads/s, the overlap the column segments were cut with after adapting to the previous pass, the share
of alignments that needed the second round, the stage times and the share of planted counts recovered."""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _chunk(args):
    first, count, read_nt, realism = args
    from strique_amd import synth
    pm, cfg = bench.load_inputs()
    table = synth.KmerTable(pm)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    out = []
    for i in range(first, first + count):
        nrep = bench.REPEAT_SWEEP[i % len(bench.REPEAT_SWEEP)]
        s, strand = synth.make_read(table, 31, i, read_nt, (repeat, prefix, suffix), nrep, realism=realism)
        out.append((s, strand, nrep))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4096)
    ap.add_argument("--read-nt", type=int, default=50000)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("levels", nargs="*", type=float, default=[0.0, 0.5, 1.0, 1.5])
    a = ap.parse_args()
    batches = {}
    with mp.get_context("spawn").Pool(a.workers) as pool:          # before anything touches the GPU
        per = (a.reads + a.workers - 1) // a.workers
        for r in a.levels:
            parts = pool.map(_chunk, [(k * per, min(per, a.reads - k * per), a.read_nt, r) for k in range(a.workers) if k * per < a.reads])
            batches[r] = [x for p in parts for x in p]
    from strique_amd.counter import repeatCounter
    pm, cfg = bench.load_inputs()
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    print("| realism | median flank score / maximum | reads/s | ms per pass | forward DP ms | Viterbi ms | overlap columns (first round; worst case) | second-round alignments | counts within 2 | screen: alignments with windows / screened / whole read (last pass) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in a.levels:
        counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
        counter.add_target("c9orf72", repeat, prefix, suffix)
        sigs = [x[0] for x in batches[r]]; strands = [x[1] for x in batches[r]]; nreps = [x[2] for x in batches[r]]
        off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
        tids = [counter._classifier_for("c9orf72", s).target_id for s in strands]
        ctx = counter.ctx
        ctx.batch_upload(np.concatenate(sigs), off, tids)
        ctx.batch_run(); ctx.batch_run()                      # sizes the buffers; the second pass already runs at the adapted overlap
        t0 = time.time(); tm = np.zeros(8); redo = 0; total = 0
        for _ in range(3):
            ctx.batch_run(); tm += ctx.last_timing()
            a_, b_ = ctx.last_second_round(); redo += a_; total += b_
        ctx.device_synchronize()
        dt = (time.time() - t0) / 3
        tm /= 3
        res = ctx.batch_fetch(); geo = ctx.last_geometry()
        m_rows = 6 * (len(prefix) - 5)
        fr = np.concatenate([res["score_prefix"], res["score_suffix"]])          # normalised scores; the raw fraction is what the library planned with
        ok = int(sum(abs(int(x["count"]) - w) <= 2 for x, w in zip(res, nreps)))
        scr = ctx.last_screen()
        print("| %.2f | %s | %.0f | %.1f | %.1f | %.1f | %d; %d | %d of %d (%.2f %%) | %d / %d | %d / %d / %d |" % (
            r, os.environ.get("STRQ_FRACTION_NOTE", "see tools/realism_probe.py"), len(sigs) / dt, dt * 1e3, tm[1], tm[6], geo["overlap_first"], geo["overlap_worst"],
            redo, total, 100.0 * redo / max(1, total), ok, len(sigs), scr["windowed"], scr["screened"], scr["whole_read"]), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
