#!/usr/bin/env python3
"""Forward-DP launch geometry sweep on one resident batch (BASELINE configs[2] reads): column segments
per alignment (STRQ_SEG), score tables per CU (STRQ_TABLES), float32 vs 24-bit tables (STRQ_NO_PACK).
    python tools/dp_sweep.py [--reads 4096] [--read-nt 50000] "SEG=2,TABLES=8" "SEG=2,TABLES=6,NO_PACK=1" ...
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4096)
    ap.add_argument("--read-nt", type=int, default=50000)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--nrep", type=int, default=0, help="fixed repeat count (default: the configs[2] sweep)")
    ap.add_argument("configs", nargs="*", default=["SEG=1,TABLES=8", "SEG=2,TABLES=8"])
    a = ap.parse_args()
    from strique_amd.counter import repeatCounter
    if a.nrep:
        bench.REPEAT_SWEEP = (a.nrep,)
    pm, cfg = bench.load_inputs()
    counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    counter.add_target("c9orf72", repeat, prefix, suffix)
    sigs, strands, nreps = bench.make_batch(pm, cfg, a.reads, a.read_nt, 0)
    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    tids = [counter._classifier_for("c9orf72", s).target_id for s in strands]
    ctx = counter.ctx
    ctx.batch_upload(np.concatenate(sigs), off, tids)
    ref = None
    for conf in a.configs:
        for k in ("STRQ_SEG", "STRQ_TABLES", "STRQ_NO_PACK", "STRQ_PACK", "STRQ_MAX_WAVES"):
            os.environ.pop(k, None)
        for kv in conf.split(","):
            if kv:
                k, v = kv.split("=")
                os.environ["STRQ_" + k] = v
        ctx.batch_run()
        t0 = time.time(); tm = np.zeros(8)
        for _ in range(a.steps):
            ctx.batch_run(); tm += ctx.last_timing()
        ctx.device_synchronize()
        dt = (time.time() - t0) / a.steps
        res = ctx.batch_fetch()
        key = (res["count"].tolist(), res["score_prefix"].tolist(), res["score_suffix"].tolist(), res["offset"].tolist(), res["ticks"].tolist(), res["log_p"].tolist())
        if ref is None:
            ref = key
        tm /= a.steps
        print("%-34s step %7.1f ms | fwd %7.1f trace %5.1f vit %6.1f cond %5.1f | launches %d | %s" % (
            conf, dt * 1e3, tm[1], tm[2], tm[6], tm[5], int(tm[7]), "same results" if key == ref else "RESULTS DIFFER"), flush=True)


if __name__ == "__main__":
    main()
