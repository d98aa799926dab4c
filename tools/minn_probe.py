#!/usr/bin/env python3
"""From which read length do the screens pay?  Resident reads of a few lengths through the default path with the screens' length threshold
(STRQ_SCREEN_MIN_N, default 65 536 samples) at 0 and at its default.    python tools/minn_probe.py [reads=4096]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    pm, cfg = bench.load_inputs()
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    table = synth.KmerTable(pm)
    for nt in (3000, 5000, 7000, 9000):
        rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
        rc.add_target("c9orf72", repeat, prefix, suffix)
        sigs, tids = [], []
        for i in range(n):
            s, strand = synth.make_read(table, 5, i, nt, (repeat, prefix, suffix), 30)
            sigs.append(s); tids.append(rc._classifier_for("c9orf72", strand).target_id)
        off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
        ctx = rc.ctx
        ctx.batch_upload(np.concatenate(sigs), off, tids)
        ref = None
        for min_n in ("", "0", "16384", "32768"):
            ctx.set_option("STRQ_SCREEN_MIN_N", min_n if min_n else None)
            for _ in range(2):
                ctx.batch_run()
            t0 = time.time()
            for _ in range(3):
                ctx.batch_run()
            dt = (time.time() - t0) / 3
            tm = ctx.last_timing(); scr = ctx.last_screen(); res = ctx.batch_fetch()
            if ref is None:
                ref = res.copy()
            print("%5d nt (N~%6d): MIN_N %-7s %.1f ms per %d reads = %7.0f reads/s  forward %.1f ms, screen %s (%.1f ms), rows equal %s"
                  % (nt, off[-1] // n, min_n or "default", dt * 1e3, n, n / dt, tm[1], scr["mode"], scr["ms"], bool(np.array_equal(res, ref))), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
