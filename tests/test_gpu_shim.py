"""The `pyseqan.align_raw` shim (integration/pyseqan.py, the class INTEGRATION.md section 1 describes)
and the full align_overlap domain of the C ABI: float-valued `a` with thousands of distinct values,
flanks without run structure and of any length -- reference src/pyalign.cpp:47-61, src/align_raw.h:84-114."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "integration"))


@pytest.fixture(scope="module")
def aligner():
    import pyseqan
    return pyseqan.align_raw()


def _set(al, params):
    al.gap_open_h, al.gap_extension_h, al.gap_open_v, al.gap_extension_v, al.dist_offset, al.dist_min = [float(v) for v in params]


def _check(al, orc, a, b, params):
    score, a_idx, b_idx = al.align_overlap(a, b)
    o = orc.align_overlap(np.asarray(a, np.float64), np.asarray(b, np.float64), np.asarray(params, np.float32))
    assert isinstance(score, float) and isinstance(a_idx, list) and isinstance(b_idx, list)
    assert np.float32(score).tobytes() == np.float32(o[0]).tobytes()
    assert a_idx == o[1].tolist() and b_idx == o[2].tolist()


def test_properties_mirror_the_binding(aligner):
    al = aligner
    # defaults of align_raw_settings (src/align_raw.h:51-60)
    assert (al.gap_open_h, al.gap_open_v, al.gap_extension_h, al.gap_extension_v, al.dist_offset, al.dist_min) == (-2.0, -2.0, -8.0, -8.0, 8.0, -16.0)
    al.gap_open = -3.0              # set_gap_open writes both directions (src/align_raw.h:84)
    assert (al.gap_open, al.gap_open_h, al.gap_open_v) == (-3.0, -3.0, -3.0)
    al.gap_extension = -0.5
    assert (al.gap_extension, al.gap_extension_h, al.gap_extension_v) == (-0.5, -0.5, -0.5)
    al.gap_open_v = -7.0
    assert al.gap_open == -3.0 and al.gap_open_v == -7.0
    assert np.allclose(al._ctx.get_align_params(), [-3.0, -0.5, -7.0, -0.5, 8.0, -16.0])


@pytest.mark.parametrize("m", [1, 5, 100, 869, 1025, 2500])
def test_float_signal_and_arbitrary_flank(aligner, orc, m):
    """`a`: a normalised float signal (thousands of distinct values); `b`: no run structure, lengths that are
    not multiples of 6, more than one 1024-row strip."""
    rng = np.random.default_rng(40 + m)
    params = orc.align_params(None)
    _set(aligner, params)
    n = 6000
    level = np.repeat(rng.uniform(60, 120, n // 4 + 1), rng.integers(5, 10, n // 4 + 1))[:n]
    a = level + rng.normal(0, 1.5, n)
    start = int(rng.integers(0, n - min(m, n - 1)))
    b = a[start:start + m][:m] + rng.normal(0, 0.7, len(a[start:start + m][:m]))
    if len(b) < m:
        b = np.concatenate([b, rng.uniform(60, 120, m - len(b))])
    _check(aligner, orc, a, b, params)


def test_general_affine_parameters_and_degenerate_inputs(aligner, orc):
    rng = np.random.default_rng(77)
    a = rng.uniform(50, 130, 3000)
    b = rng.uniform(50, 130, 333)
    for params in ([-2, -8, -2, -8, 8, -16], [-3, -1, -20, -4, 16, 0], [-1, -1, -16, -16, 16, 0]):
        _set(aligner, params)
        _check(aligner, orc, a, b, params)
    params = [-1, -1, -16, -16, 16, 0]
    _set(aligner, params)
    _check(aligner, orc, a[:5], b, params)                    # read shorter than the flank
    _check(aligner, orc, np.zeros(0), b[:40], params)         # empty read
    _check(aligner, orc, np.full(500, 91.25), np.full(60, 91.25), params)      # constant inputs: every tie rule at once
    a2 = a.copy(); a2[::97] = -0.0; a2[5::101] = 0.0
    _check(aligner, orc, a2, b, params)                       # both zeros present


def test_detect_like_inputs_take_the_table_kernels(aligner, orc, monkeypatch):
    """An 8-bit signal against a 6-run template goes through the LDS-table kernels; forcing the generic
    kernel on the same input must give the same answer."""
    rng = np.random.default_rng(5)
    params = orc.align_params(None)
    _set(aligner, params)
    cls = rng.uniform(60, 120, 145)
    flank = np.repeat(cls, 6)
    lval = 40 + 0.45 * np.arange(256)
    lv = np.repeat(rng.integers(30, 200, 4001), rng.integers(3, 10, 4001))[:20000]
    a = lval[lv]
    fast = aligner.align_overlap(a, flank)
    assert aligner._ctx.last_timing()[1] > 0          # forward-DP kernel time of the table path
    monkeypatch.setenv("STRQ_GENERIC_ALIGN", "1")
    slow = aligner.align_overlap(a, flank)
    assert fast == slow
    _check(aligner, orc, a, flank, params)
