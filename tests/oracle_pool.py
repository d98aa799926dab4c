"""The CPU oracle on several host cores (test infrastructure): worker processes that import nothing but
`oracle/` and the recorded pore-model tables, so that parity tests at the benchmark's own size -- a 50 kb
read costs the oracle ~5 s -- finish in a few minutes.  Spawned, never forked: the calling test process
holds a HIP context."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_STATE = {}


def _setup():
    if not _STATE:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        from oracle import strique_oracle as orc
        t = np.load(os.path.join(ROOT, "tests", "golden", "pore_tables.npz"))
        _STATE["orc"] = orc
        _STATE["opm"] = orc.PoreModel(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
        _STATE["cfg"] = json.load(open(os.path.join(ROOT, "tests", "golden", "config.json")))
        _STATE["tc"] = {}
    return _STATE


def detect_one(args):
    """(signal, strand, (repeat, prefix, suffix)) -> the oracle's detect() tuple, with the oracle's own model,
    templates and un-baked HMMs (tests/golden/config.json parameters)."""
    sig, strand, target = args
    st = _setup()
    orc, opm, cfg = st["orc"], st["opm"], st["cfg"]
    key = (strand,) + tuple(target)
    if key not in st["tc"]:
        st["tc"][key] = orc.classifier(target[0], target[1], target[2], strand, opm, None, cfg["HMM"])
    res, _ = orc.detect(sig, st["tc"][key], opm, orc.align_params(cfg["align"]))
    return tuple(res)


def detect_many(jobs, workers=None):
    """jobs: list of (signal, strand, target).  A few worker processes (each holds the full DP matrix of one
    alignment: ~2 GB at 50 kb); falls back to this process for a single job."""
    if len(jobs) <= 1:
        return [detect_one(j) for j in jobs]
    if workers is None:
        workers = max(1, min(12, (os.cpu_count() or 1) // 2, len(jobs)))
    with mp.get_context("spawn").Pool(workers) as pool:
        return pool.map(detect_one, jobs, chunksize=1)
