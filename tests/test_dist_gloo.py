"""The N > 1 path on CPU: two gloo ranks shard the reads and gather the result records."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from strique_amd import dist as sdist, ffi
rank, world, local = sdist.init_process_group(backend="gloo")
n = 23
cost = np.arange(n)[::-1] * 7 %% 11 + 1
mine = sdist.shard_indices(n, rank, world, cost)
rec = np.zeros(len(mine), dtype=ffi.RESULT_DTYPE)
rec["count"] = mine * 3 + 1          # stand-in for detect results of my shard
rec["log_p"] = -1.5 * mine
full = sdist.gather_records(rec, mine, n, device="cpu")
if rank == 0:
    assert full is not None and np.array_equal(full["count"], np.arange(n) * 3 + 1)
    assert np.array_equal(full["log_p"], -1.5 * np.arange(n))
    print("GATHER_OK")
else:
    assert full is None
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
''' % ROOT


def test_two_rank_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]
