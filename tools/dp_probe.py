"""Forward-DP time per step as a function of rows per lane (R) and waves per CU.
usage (GPU box): python tools/dp_probe.py [n_align] [n_cols]"""
import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from strique_amd import ffi
na = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ctx = ffi.Context(0)
ctx.set_align_params(-1, -1, -16, -16, 16, 0)
rng = np.random.default_rng(3)
nreads = 64
levels = rng.integers(60, 200, (nreads, n)).astype(np.uint8)
lval = np.tile((50.0 + 0.32 * np.arange(256)).astype(np.float32), (nreads, 1))
off = np.arange(nreads + 1, dtype=np.int64) * n
for k in (64, 74, 85, 128, 145, 158):
    flank = np.repeat(rng.uniform(60, 120, k).astype(np.float32), 6)
    fl = np.tile(flank, na); foff = np.arange(na + 1, dtype=np.int64) * len(flank)
    ar = (np.arange(na) % nreads).astype(np.int32)
    ctx.align_batch(levels.ravel(), off, lval, ar, fl, foff, want_rec=False)
    ctx.align_batch(levels.ravel(), off, lval, ar, fl, foff, want_rec=False)
    t = ctx.last_timing()
    steps = (n + 1) // 2 + 63
    print("k=%3d rows=%3d  fwd %.2f ms  launches %d  -> %.1f ns/step if one round" % (k, 6 * k, t[1], int(t[7]), t[1] * 1e6 / steps), flush=True)
