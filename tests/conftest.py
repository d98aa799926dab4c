import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tables():
    return np.load(os.path.join(GOLDEN, "pore_tables.npz"))


@pytest.fixture(scope="session")
def pm(tables):
    from strique_amd.pore_model import pore_model
    return pore_model(table=(tables["base_kmer"], tables["base_mean"], tables["base_stdv"]))


@pytest.fixture(scope="session")
def pm_mod(tables):
    from strique_amd.pore_model import pore_model
    return pore_model(table=(tables["mod_kmer"], tables["mod_mean"], tables["mod_stdv"]))


@pytest.fixture(scope="session")
def cfg():
    return json.load(open(os.path.join(GOLDEN, "config.json")))


@pytest.fixture(scope="session")
def orc():
    from oracle import strique_oracle
    strique_oracle.lib()
    return strique_oracle


@pytest.fixture(scope="session")
def opm(orc, tables):
    """The oracle's own pore model, from the recorded k-mer table (not from the product's object)."""
    return orc.PoreModel(table=(tables["base_kmer"], tables["base_mean"], tables["base_stdv"]))


@pytest.fixture(scope="session")
def opm_mod(orc, tables):
    return orc.PoreModel(table=(tables["mod_kmer"], tables["mod_mean"], tables["mod_stdv"]))


@pytest.fixture(scope="session")
def targets(cfg):
    """name -> (repeat, prefix, suffix): the bundled loci plus the build-authored HTT/CAG row
    (tests/golden/repeat_config_htt.tsv; BASELINE configs[3])."""
    from strique_amd.cli import parse_config
    out = {name: tuple(v[3:6]) for name, v in cfg["repeat"].items()}
    extra = parse_config(os.path.join(GOLDEN, "repeat_config_htt.tsv"))["repeat"]
    out.update({name: tuple(v[3:6]) for name, v in extra.items()})
    return out


@pytest.fixture(scope="session")
def gpu_counter(pm, cfg, targets):
    """repeatCounter on cuda:0 with both bundled targets; fails loudly when there is no GPU."""
    from strique_amd.counter import repeatCounter
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    for name, (repeat, prefix, suffix) in targets.items():
        rc.add_target(name, repeat, prefix, suffix)
    return rc


_TC_CACHE = {}


def oracle_tc(orc, opm, targets, name, strand, hmm_cfg, opm_mod=None):
    """The oracle's own classifier of one strand of a target (templates + un-baked HMMs, built in
    oracle/ from the sequences; nothing comes from the product)."""
    key = (name, strand, opm_mod is not None, json.dumps(hmm_cfg, sort_keys=True))
    if key not in _TC_CACHE:
        repeat, prefix, suffix = targets[name]
        _TC_CACHE[key] = orc.classifier(repeat, prefix, suffix, strand, opm, opm_mod, hmm_cfg)
    return _TC_CACHE[key]


def oracle_map(fn, jobs, workers=None):
    """fn(job) for every job, on a few threads: the oracle's DP and Viterbi are ctypes calls into oracle/liboracle.so, which release
    the interpreter lock -- the expected values of a test with dozens of 100 k-column alignments come in seconds instead of a
    minute (the GPU suite has a time limit).  Results in job order; the first exception is re-raised."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    jobs = list(jobs)
    if workers is None:
        workers = max(1, min(8, len(jobs), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))
    if workers <= 1:
        return [fn(j) for j in jobs]
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, jobs))
