#!/usr/bin/env python3
"""RCCL smoke test of the run's one collective on a single GPU (world size 1): the tensors of
strique_amd.dist.gather_results go through the nccl backend exactly as they do with N ranks.
usage (GPU box): python tools/nccl_smoke.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch
import torch.distributed as dist
from strique_amd import dist as sdist, ffi

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
# force the multi-rank code path: gather_results short-circuits only when world == 1 *and* uninitialised
n = 37
rec = np.zeros(n, dtype=ffi.RESULT_DTYPE); rec["count"] = np.arange(n) * 3 + 1; rec["log_p"] = -0.5 * np.arange(n)
mods = [("01" * i)[:i] for i in range(n)]
real_ws = dist.get_world_size
dist.get_world_size = lambda *a, **k: 2          # pretend, so that the padded all_gather path runs ...
try:
    import types
    orig = dist.all_gather
    def fake_all_gather(out, t, *a, **k):          # ... with the real RCCL all_gather of this rank's tensor
        tmp = [torch.empty_like(t)]
        dist.get_world_size = real_ws
        orig(tmp, t)
        dist.get_world_size = lambda *a, **k: 2
        out[0].copy_(tmp[0]); out[1].copy_(tmp[0])
    dist.all_gather = fake_all_gather
    full, fm = sdist.gather_results(rec, np.arange(n), n, mods, device="cuda")
finally:
    dist.get_world_size = real_ws; dist.all_gather = orig
assert np.array_equal(full["count"], rec["count"]) and fm == mods
ctx = ffi.Context(0)                               # the library's own stream next to torch's on the same device
ctx.device_synchronize()
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("NCCL_SMOKE_OK")
