#!/usr/bin/env python3
"""Host -> HBM upload rate of strq_batch_upload for the pageable path (STRQ_UPLOAD_THREADS=0) and the
pinned staging ring with 1..N copy threads.  usage (GPU box): python tools/upload_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strique_amd import ffi  # noqa: E402

ctx = ffi.Context(0)
ctx.set_pore_stats(60.0, 120.0, 40.0, 140.0)
n_reads, n = 2048, 375000
sig = (np.arange(n_reads * n, dtype=np.int64) % 977).astype(np.int16)
off = np.arange(n_reads + 1, dtype=np.int64) * n
fl = np.repeat(np.linspace(60, 120, 145).astype(np.float32), 6)
from strique_amd import hmm, pore_model  # noqa: E402,F401
# a target is needed only for validation of ids; uploads do not touch it
import json
t = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pore_tables.npz"))
pm = pore_model.pore_model(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
cfg = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "config.json")))
from strique_amd.counter import repeatCounter  # noqa: E402
rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], context=ctx)
rc.add_target("c9orf72", *cfg["repeat"]["c9orf72"][3:6])
tids = np.zeros(n_reads, np.int32)
for th in sys.argv[1:] or ["0", "1", "2", "4", "6", "12", "24"]:
    os.environ["STRQ_UPLOAD_THREADS"] = th
    ctx.batch_upload(sig, off, tids)
    t0 = time.time(); ctx.batch_upload(sig, off, tids); dt = time.time() - t0
    print("threads %2s: %.3f s  %.1f GB/s" % (th, dt, sig.nbytes / dt / 1e9), flush=True)
