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
def opm(orc, pm):
    o = orc.PoreModel.__new__(orc.PoreModel)
    o.means = pm._means; o.model_min = pm.model_min; o.model_max = pm.model_max
    return o


@pytest.fixture(scope="session")
def gpu_counter(pm, cfg):
    """repeatCounter on cuda:0 with both bundled targets; fails loudly when there is no GPU."""
    from strique_amd.counter import repeatCounter
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    for name, (chrom, b, e, repeat, prefix, suffix) in cfg["repeat"].items():
        rc.add_target(name, repeat, prefix, suffix)
    return rc


def oracle_tc(counter, name, strand):
    tc = counter._classifier_for(name, strand)
    return dict(prefix=tc.prefix, suffix=tc.suffix, prefix_ext=tc.prefix_ext, suffix_ext=tc.suffix_ext, hmm=tc.repeatHMM)
