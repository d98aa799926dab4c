#!/usr/bin/env python3
"""Repeated host-buffer batches on one context: results identical from call to call, device memory flat.
    python tools/soak.py [--reads 2048] [--calls 25]"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def free_bytes():
    hip = ctypes.CDLL("libamdhip64.so.7")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipDeviceSynchronize(); hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return free.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2048)
    ap.add_argument("--calls", type=int, default=25)
    a = ap.parse_args()
    from strique_amd.counter import repeatCounter
    pm, cfg = bench.load_inputs()
    counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    counter.add_target("c9orf72", *cfg["repeat"]["c9orf72"][3:6])
    sigs, strands, nreps = bench.make_batch(pm, cfg, a.reads, 50000, 0)
    # ragged on purpose: every call sees the reads in another order and a different number of them
    rng = np.random.default_rng(1)
    ref = None; f0 = None
    for call in range(a.calls):
        order = rng.permutation(a.reads)[: a.reads - int(rng.integers(0, a.reads // 4))]
        flat = np.concatenate([sigs[i] for i in order])
        off = np.zeros(len(order) + 1, np.int64); off[1:] = np.cumsum([len(sigs[i]) for i in order])
        tids = [counter._classifier_for("c9orf72", strands[i]).target_id for i in order]
        t0 = time.time()
        res = counter.ctx.detect_batch(flat, off, tids)
        dt = time.time() - t0
        by_read = {int(i): tuple(res[k].tolist()) for k, i in enumerate(order)}
        if ref is None:
            ref = dict(by_read)
        for i, r in by_read.items():
            if i in ref:
                assert ref[i] == r, (call, i, ref[i], r)
            else:
                ref[i] = r
        fb = free_bytes()
        if call == min(5, a.calls - 1):      # the grow-only device buffers have reached their high-water mark by then
            f0 = fb
        print("call %2d: %4d reads %.2f s (%.0f reads/s), device memory free %.2f GB" % (call, len(order), dt, len(order) / dt, fb / 2**30), flush=True)
    ok = sum(1 for i in range(a.reads) if i in ref and abs(ref[i][0] - nreps[i]) <= 2)
    print("planted counts recovered (+-2): %d / %d; free memory drift since call 5: %.1f MB" % (ok, len(ref), (f0 - fb) / 2**20))


if __name__ == "__main__":
    main()
