#!/usr/bin/env python3
"""From how many reads per sub-batch on do the screens pay?  A screen runs one wave per read for the whole read, the float32 DP four
waves per alignment over a quarter each: on a batch that does not fill the GPU the DP over whole reads has the shorter critical path.
Clean 50 kb reads (and 20 kb ones), batches of 64 ... 4096 reads: ms per batch with the coarse screen, the fine one, and none.

    python tools/batch_size_probe.py [read_nt=50000]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    from strique_amd.counter import repeatCounter
    pm, cfg = bench.load_inputs()
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    sigs, strands, _ = bench.make_batches_parallel(4096, nt, 0, 16)
    rows = {}
    for name, opts in (("coarse", {"STRQ_SCREEN_MODE": "coarse"}), ("fine", {"STRQ_SCREEN_MODE": "fine"}), ("none", {"STRQ_NO_SCREEN": "1"})):
        rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
        rc.add_target("c9orf72", repeat, prefix, suffix)
        for k, v in opts.items():
            rc.ctx.set_option(k, v)
        for n in (64, 128, 256, 512, 1024, 2048, 4096):
            off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs[:n]])
            tids = [rc._classifier_for("c9orf72", s).target_id for s in strands[:n]]
            rc.ctx.batch_upload(np.concatenate(sigs[:n]), off, tids)
            for _ in range(2):
                rc.ctx.batch_run()
            t0 = time.time()
            for _ in range(3):
                rc.ctx.batch_run()
            dt = (time.time() - t0) / 3
            tm = rc.ctx.last_timing()
            rows[(name, n)] = (dt, tm[1], tm[6])
        rc.ctx.close()
    print("reads of %d nt; ms per batch (forward stage, Viterbi)" % nt)
    print("%6s %28s %28s %28s" % ("reads", "coarse", "fine", "none"))
    for n in (64, 128, 256, 512, 1024, 2048, 4096):
        print("%6d " % n + " ".join("%8.1f (%6.1f, %6.1f)    " % (rows[(m, n)][0] * 1e3, rows[(m, n)][1], rows[(m, n)][2]) for m in ("coarse", "fine", "none")))


if __name__ == "__main__":
    main()
