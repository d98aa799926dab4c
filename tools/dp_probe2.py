"""Forward-DP throughput against waves per CU (STRQ_WPB=4..8, STRQ_MIN_ROUNDS=0): 8192 alignments of
40 k columns against a 100-class and a 145-class flank.  Measured: 100 classes (4.7 k-float tables, up to 8
waves fit) 41.8 / 38.0 / 31.9 / 30.0 / 27.1 ms at 4 / 5 / 6 / 7 / 8 waves; 145 classes (6.8 k floats) stop
at 6 waves: 39.6 ms.  usage (GPU box): STRQ_WPB=8 STRQ_MIN_ROUNDS=0 python tools/dp_probe2.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strique_amd import ffi
na = 8192; n = 40000
ctx = ffi.Context(0)
ctx.set_align_params(-1, -1, -16, -16, 16, 0)
rng = np.random.default_rng(3)
nreads = 64
levels = rng.integers(60, 200, (nreads, n)).astype(np.uint8)
lval = np.tile((50.0 + 0.45 * np.arange(256)).astype(np.float32), (nreads, 1))
off = np.arange(nreads + 1, dtype=np.int64) * n
for k in (100, 145):
    flank = np.repeat(rng.uniform(60, 120, k).astype(np.float32), 6)
    fl = np.tile(flank, na); foff = np.arange(na + 1, dtype=np.int64) * len(flank)
    ar = (np.arange(na) % nreads).astype(np.int32)
    ctx.align_batch(levels.ravel(), off, lval, ar, fl, foff, want_rec=False)
    ctx.align_batch(levels.ravel(), off, lval, ar, fl, foff, want_rec=False)
    t = ctx.last_timing()
    print("WPB=%s k=%d fwd %.1f ms launches %d" % (os.environ.get("STRQ_WPB"), k, t[1], int(t[7])), flush=True)
