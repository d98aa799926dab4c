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
full2, mods = sdist.gather_results(rec, mine, n, [("m" + str(i)) * (i %% 4) for i in mine], device="cpu")
if rank == 0:
    assert full is not None and np.array_equal(full["count"], np.arange(n) * 3 + 1)
    assert np.array_equal(full["log_p"], -1.5 * np.arange(n))
    assert np.array_equal(full2, full) and mods == [("m" + str(i)) * (i %% 4) for i in range(n)]
    print("GATHER_OK")
else:
    assert full is None and full2 is None and mods is None
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


CLI_WORKER = r'''
import io, json, os, sys
import numpy as np
sys.path.insert(0, %r)
from strique_amd import cli, dist as sdist
rank, world, local = sdist.init_process_group(backend="gloo")
cfg = json.load(open(os.path.join(%r, "tests", "golden", "config.json")))
loci = {}
for name, (chrom, b, e, *_r) in cfg["repeat"].items():
    loci.setdefault(chrom, []).append((name, b, e))
lines = ["@HD\tVN:1.0"]
for i in range(37):
    chrom, pos = ("chr9", 27570000) if i %% 3 else ("chrX", 146990000)
    if i %% 7 == 6:
        chrom = "chr1"                                   # no locus: skipped by every rank alike
    lines.append("\t".join(["read%%d" %% i, "16" if i %% 2 else "0", chrom, str(pos), "60", "5S8000M3S", "*", "0", "0", "ACGT", "*"]))

class FakeCounter(object):                               # stands in for the GPU engine: rows depend on the inputs only
    def detect_batch(self, items):
        return [(len(raw) %% 97, 1.5, 2.5, -3.0 * len(t), int(raw[0]), 7, "01"[len(raw) %% 2] * (len(raw) %% 5)) for t, raw, s in items]

def get_raw(qname):
    i = int(qname[4:])
    return None if i == 11 else np.arange(100 + i, 300 + 2 * i)     # read11 has no fast5

log = cli.Log("error")
stats = {}
mine = cli.run_count(iter(lines), loci, get_raw, FakeCounter(), log, 5, rank, world, stats=stats)
import torch.distributed as dist
merged = cli.gather_rows(mine, stats["items"], sdist)            # the run's one collective (records + byte pool)
if rank == 0:
    buf = io.StringIO(); cli.write_rows(buf, merged)
    one = io.StringIO(); cli.run_count(iter(lines), loci, get_raw, FakeCounter(), log, 5, 0, 1, one)
    assert buf.getvalue() == one.getvalue(), (buf.getvalue(), one.getvalue())
    assert len(buf.getvalue().splitlines()) == 1 + 31                  # 37 records - 5 off-target - 1 without fast5
    assert len(mine) in (15, 16)                                       # dealt by read length: equal shares
    print("CLI_SHARD_OK")
else:
    assert merged is None
dist.barrier(); dist.destroy_process_group()
''' % (ROOT, ROOT)


def test_count_rows_shard_and_merge(tmp_path):
    """`count` under torchrun: ranks take the accepted (read, target) pairs round-robin and rank 0 merges
    the rows into input order -- identical to the single-process output."""
    script = tmp_path / "cli_worker.py"
    script.write_text(CLI_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "CLI_SHARD_OK" in outs[0]


FAULT_WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
from strique_amd import cli, dist as sdist, ffi
rank, world, local = sdist.init_process_group(backend="gloo")
cfg = json.load(open(os.path.join(%r, "tests", "golden", "config.json")))
loci = {}
for name, (chrom, b, e, *_r) in cfg["repeat"].items():
    loci.setdefault(chrom, []).append((name, b, e))
lines = ["\t".join(["read%%d" %% i, "0", "chr9", "27570000", "60", "8000M", "*", "0", "0", "ACGT", "*"]) for i in range(12)]

class Counter(object):
    def detect_batch(self, items):
        if rank == 1:
            raise ffi.StriqueHipError(ffi.STRQ_ERR_DEVICE, "hipErrorLaunchFailure (simulated)")
        return [(1, 1.5, 2.5, -3.0, 4, 7, "-") for _ in items]
    detect = None

fault = 0
try:
    cli.run_count(iter(lines), loci, lambda q: np.arange(200), Counter(), cli.Log("error"), 4, rank, world, stats={})
except cli.DeviceFault:
    fault = 1
assert fault == (1 if rank == 1 else 0)
agreed = sdist.any_rank(fault, device="cpu")          # what `count` does before the gather: nobody is left waiting
assert agreed
print("FAULT_AGREED")
import torch.distributed as dist
dist.destroy_process_group()
''' % (ROOT, ROOT)


def test_device_fault_on_one_rank_does_not_leave_the_others_waiting(tmp_path):
    """A device error on one rank of `count`: every rank learns about it through one tiny all_reduce placed in
    front of the gather, so all of them can exit with status 3 instead of the healthy ones blocking in all_gather."""
    script = tmp_path / "fault_worker.py"
    script.write_text(FAULT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29521", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("FAULT_AGREED" in o for o in outs), outs
