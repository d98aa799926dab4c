"""The arithmetic behind the upper-bound screen (csrc/screen_kernels.hip), restated in numpy -- no GPU.

The kernel stores T = S * sc + i |e_v| sc + j |e_h| sc with scores rounded up to multiples of 1 / sc and runs
T[i][j] = max3(T[i-1][j-1] + s'', T[i][j-1], T[i-1][j]).  Claims checked here against the float64 DP of the same scores
(S = max(diag + s, left + e_h, up + e_v), free top row, column 0 = i e_v, as src/align_raw.h:106-158 with open == extend):
the decoded last row is never below the exact one and less than m / sc above it, rows of score 0 <= s'' under the flank copy the
last flank row, and every stored value stays below 2^31 for a 2 M-sample read."""
import numpy as np
import pytest


def _exact(vals, flank, off=16.0, dmin=0.0, eh=-1.0, ev=-16.0):
    n = len(vals)
    j = np.arange(n + 1, dtype=np.float64)
    prev = np.zeros(n + 1)
    rows = []
    for i, f in enumerate(flank, 1):
        s = np.maximum(off - np.abs(vals - f) ** 1.2, dmin)
        cur = np.empty(n + 1)
        cur[0] = i * ev
        cur[1:] = np.maximum(prev[:-1] + s, prev[1:] + ev)
        cur = np.maximum.accumulate(cur - eh * j) + eh * j
        prev = cur
        rows.append(s)
    return prev, rows


def _screen(rows, n, sc, eh=-1.0, ev=-16.0, pad_rows=0):
    hh, v = int(-eh * sc), int(-ev * sc)
    T = np.arange(n + 1, dtype=np.int64) * hh            # top row: S = 0
    m = len(rows)
    for i in range(1, m + pad_rows + 1):
        if i <= m:
            s2 = np.ceil(rows[i - 1] * sc).astype(np.int64) + hh + v
        else:
            s2 = np.zeros(n, np.int64)                    # rows under the flank
        cur = np.empty(n + 1, np.int64)
        cur[0] = 0                                        # S[i][0] = i e_v
        cur[1:] = np.maximum(T[:-1] + s2, T[1:])
        T = np.maximum.accumulate(cur)                    # the horizontal move is free in T
    return T


@pytest.mark.parametrize("sc", [1024, 8])
def test_integer_potential_dp_bounds_the_exact_last_row(sc):
    rng = np.random.default_rng(5)
    k, n = 20, 4000
    flank = np.repeat(rng.uniform(60, 120, k), 6)
    m = len(flank)
    vals = np.repeat(rng.uniform(50, 130, n // 5 + 1), rng.integers(3, 10, n // 5 + 1))[:n]
    vals[1500:1500 + m] = flank + rng.normal(0, 1.5, m)
    exact, rows = _exact(vals, flank)
    T = _screen(rows, n, sc)
    j = np.arange(n + 1)
    ub = (T - j * sc - m * 16 * sc) / sc                  # decode: S = (T - j |e_h| sc - m |e_v| sc) / sc
    assert np.all(ub >= exact - 1e-9)
    assert np.all(ub < exact + m / sc + 1e-9)
    assert exact.argmax() == ub.argmax() or abs(exact.max() - exact[ub.argmax()]) < m / sc
    # rows below the flank copy its last row
    Tp = _screen(rows, n, sc, pad_rows=9)
    assert np.array_equal(Tp, T)


def _screen_merged(rows, n, sc, eh=-1.0, ev=-16.0):
    """The coarse screen's recurrence (align_screen2_kernel): rows 2 i - 1, 2 i of a class are one DP row whose diagonal step gains
    both rows' entries at one column."""
    hh, v = int(-eh * sc), int(-ev * sc)
    T = np.arange(n + 1, dtype=np.int64) * hh
    m = len(rows)
    assert m % 2 == 0
    for i in range(0, m, 2):
        assert np.array_equal(rows[i], rows[i + 1])       # the two rows of a pair are the same k-mer class
        s2 = 2 * (np.ceil(rows[i] * sc).astype(np.int64) + hh + v)
        cur = np.empty(n + 1, np.int64)
        cur[0] = 0
        cur[1:] = np.maximum(T[:-1] + s2, T[1:])
        T = np.maximum.accumulate(cur)
    return T


@pytest.mark.parametrize("sc", [512, 16])
def test_merged_row_dp_bounds_the_exact_last_row(sc):
    """Every last-row value of the merged-row DP is an upper bound of the exact one (the coarse screen prunes by it), on signals with
    long events, short events (where a merged row gets away with one column for two rows: the bound is loose there) and noise."""
    rng = np.random.default_rng(11)
    k, n = 24, 5000
    flank = np.repeat(rng.uniform(60, 120, k), 6)
    m = len(flank)
    for dwell in ((6, 10), (1, 4), (2, 12)):
        vals = np.repeat(rng.uniform(50, 130, n), rng.integers(dwell[0], dwell[1], n))[:n]
        emb = np.repeat(flank[::6] + rng.normal(0, 1.0, k), rng.integers(dwell[0], dwell[1], k))
        vals[2000:2000 + len(emb)] = emb
        exact, rows = _exact(vals, flank)
        T = _screen_merged(rows, n, sc)
        j = np.arange(n + 1)
        ub = (T - j * sc - m * 16 * sc) / sc
        assert np.all(ub >= exact - 1e-9), dwell
        fine = (_screen(rows, n, sc) - j * sc - m * 16 * sc) / sc
        assert np.all(ub >= fine - 1e-9)                    # and never below the fine screen's bound
        if dwell[0] >= 6:
            assert ub.max() - exact.max() < 0.05 * exact.max()   # events at least as long as a class: nearly tight
    # the largest table entry of the coarse screen fits 16 bits at STRique's parameters
    assert 2 * (16 * 512 + 512 + 16 * 512) < 65536


def test_stored_values_fit_31_bits_for_the_longest_reads():
    sc, rows_max, n_max = 1024, 896, 1_900_000
    top = rows_max * (16 * sc + 16 * sc) + (n_max + 256) * sc + 0x00800000
    assert top < 2 ** 31 - 2 ** 24            # below 0x7f800000: every bit pattern is a finite positive float32 (v_max3_f32 orders them)
    assert 16 * sc + 16 * sc + sc < 65536     # a table entry fits 16 bits


def test_screen_plan_of_the_library(monkeypatch):
    """The frame the library plans for (host code, no GPU): STRique's parameters run at a scale of 1024 with 16 score units of float32
    slack; reads beyond ~1.9 M samples at 512; general affine gaps, a negative dist_min or other samples-per-k-mer get no screen."""
    import ctypes
    from strique_amd import build
    lib = ctypes.CDLL(build.build_lib())
    lib.strq_debug_screen_plan.restype = ctypes.c_int
    out = (ctypes.c_int32 * 6)()

    def plan(params, samples=6, max_n=400000):
        arr = (ctypes.c_float * 6)(*params)
        return lib.strq_debug_screen_plan(arr, ctypes.c_int32(samples), ctypes.c_int32(max_n), out), list(out)

    ok, o = plan([-1, -1, -16, -16, 16, 0])
    assert ok == 1 and o[:4] == [1024, 1024, 16384, 17408] and o[4] == 16 * 1024
    assert o[2] + o[1] + 16 * o[0] < 65536                      # the largest table entry fits 16 bits
    ok, o = plan([-1, -1, -16, -16, 16, 0], max_n=3_000_000)
    assert ok == 1 and o[0] == 512
    assert plan([-3, -1, -20, -4, 16, 0])[0] == 0                 # open != extend
    assert plan([-1, -1, -16, -16, 16, -4])[0] == 0               # dist_min < 0
    assert plan([-1, -1, -16, -16, 16, 0], samples=5)[0] == 0
    ok, o = plan([-0.5, -0.5, -8, -8, 8, 0])
    assert ok == 1 and o[1] * 2 == o[0] and o[2] == 8 * o[0]
    monkeypatch.setenv("STRQ_NO_SCREEN", "1")
    assert plan([-1, -1, -16, -16, 16, 0])[0] == 0
