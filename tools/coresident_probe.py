#!/usr/bin/env python3
"""Round-4 verdict item 4: the flanked-HMM Viterbi of one half-batch CO-RESIDENT with the screen of the other.

Two contexts on one GPU, one host thread each (ctypes releases the interpreter lock for the whole native call), each with its own
resident half-batch of configs[2] reads; thread B starts half a step behind thread A, so that A's Viterbi (persistent workgroups,
STRQ_VIT_G2_WAVES waves per CU) runs while B's screen kernel (STRQ_SCREEN2_GROUPS workgroups of four waves per CU) does, and the
other way round.  Register budget per SIMD: one Viterbi wave of <= 192 VGPRs + three screen waves of <= 85 = 447 of 512.
Prints reads/s of (a) one context over the whole batch, default geometry, (b) one context with the co-residency geometry,
(c) the two threads together.

    python tools/coresident_probe.py [reads_per_half=2048] [steps=6] [clean|empirical] [default|coresident|token geometry for (c); token = default geometry + STRQ_FORWARD_TOKEN=1: the contexts take turns in the forward stage]

With "empirical" reads (noise resampled from the bundled real read) the Viterbi launch lasts as long as its longest, mislocated window
(profiles/r05_viterbi_tail.txt) while most of the GPU idles: the second context's forward stage can run under that tail.
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench


def make_ctx(pm, cfg, options):
    from strique_amd.counter import repeatCounter
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    rc.add_target("c9orf72", repeat, prefix, suffix)
    for k, v in options.items():
        rc.ctx.set_option(k, v)
    return rc


def upload(rc, sigs, strands):
    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    tids = [rc._classifier_for("c9orf72", s).target_id for s in strands]
    rc.ctx.batch_upload(np.concatenate(sigs), off, tids)


def loop(rc, steps, out, delay=0.0):
    time.sleep(delay)
    t0 = time.time()
    for _ in range(steps):
        rc.ctx.batch_run(); rc.ctx.batch_fetch()
    out.append((t0, time.time()))


def main():
    half = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    workload = sys.argv[3] if len(sys.argv) > 3 else "clean"
    geo_c = sys.argv[4] if len(sys.argv) > 4 else "coresident"
    pm, cfg = bench.load_inputs()
    sigs, strands, _ = bench.make_batches_parallel(2 * half, 50000, 0, 16, workload)
    geo = {"STRQ_SCREEN2_GROUPS": "3", "STRQ_VIT_G2_WAVES": "4"}
    print("workload %s, %d reads, mean %d samples" % (workload, 2 * half, int(np.mean([len(s) for s in sigs]))))
    # (a) one context, default geometry, the whole batch
    rc = make_ctx(pm, cfg, {})
    upload(rc, sigs, strands)
    for _ in range(4):
        rc.ctx.batch_run()
    t0 = time.time()
    for _ in range(steps):
        rc.ctx.batch_run(); rc.ctx.batch_fetch()
    dt = time.time() - t0
    tm = rc.ctx.last_timing()
    print("(a) one context, default geometry: %.0f reads/s (%.1f ms per %d reads; forward %.1f, Viterbi %.1f ms)" % (2 * half * steps / dt, dt / steps * 1e3, 2 * half, tm[1], tm[6]))
    for k, v in geo.items():
        rc.ctx.set_option(k, v)
    for _ in range(2):
        rc.ctx.batch_run()
    t0 = time.time()
    for _ in range(steps):
        rc.ctx.batch_run(); rc.ctx.batch_fetch()
    dt = time.time() - t0
    tm = rc.ctx.last_timing()
    print("(b) one context, 3 screen groups + 4 Viterbi waves per CU: %.0f reads/s (forward %.1f, Viterbi %.1f ms)" % (2 * half * steps / dt, tm[1], tm[6]))
    step_s = dt / steps
    rc.ctx.close()
    # (c) two contexts, half a batch each, half a step apart
    gc = geo if geo_c == "coresident" else ({"STRQ_FORWARD_TOKEN": "1"} if geo_c == "token" else {})
    a = make_ctx(pm, cfg, gc); b = make_ctx(pm, cfg, gc)
    upload(a, sigs[:half], strands[:half]); upload(b, sigs[half:], strands[half:])
    for rc2 in (a, b):
        for _ in range(4):
            rc2.ctx.batch_run()
    oa, ob = [], []
    ta = threading.Thread(target=loop, args=(a, 2 * steps, oa)); tb = threading.Thread(target=loop, args=(b, 2 * steps, ob, step_s / 4))
    ta.start(); tb.start(); ta.join(); tb.join()
    span = max(oa[0][1], ob[0][1]) - min(oa[0][0], ob[0][0])
    print("(c) two contexts of %d reads (%s geometry), co-resident kernels: %.0f reads/s (%.1f ms per %d reads)" % (half, geo_c, 2 * half * 2 * steps / span, span / (2 * steps) * 1e3, 2 * half))
    ra = a.ctx.batch_fetch(); rb = b.ctx.batch_fetch()
    print("rows:", len(ra), len(rb))


if __name__ == "__main__":
    main()
