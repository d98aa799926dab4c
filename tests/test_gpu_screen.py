"""The upper-bound screen of the flank alignment (csrc/screen_kernels.hip) through the C ABI.

Two things are checked.  (1) What the screen claims: every chunk value it writes is an upper bound of the exact last-row
values of the chunk's columns (computed here in float64 from the same scores), and a tight one (within m / 1024).  (2) What it is used for:
alignments that ran only over the screen's windows return the oracle's score bits, end / start column and whole path --
for planted flanks at piece seams, twice in one read, five and six times (more separate candidates than the exact launch has
pieces: the last window takes the rest), and for reads the screen cannot prune (whole read, then the screen pauses)."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 14          # rows per lane of the screen


@pytest.fixture(scope="module")
def ctx():
    from strique_amd import ffi
    return ffi.Context(0)


def _exact_last_row(vals, flank, params):
    """Last row of the collapsed semi-global DP (src/align_raw.h:106-158 with open == extend), float64, one row at a time."""
    open_h, ext_h, open_v, ext_v, off, dmin = [float(v) for v in params]
    assert open_h == ext_h and open_v == ext_v
    n = len(vals)
    j = np.arange(n + 1, dtype=np.float64)
    prev = np.zeros(n + 1)
    vals = vals.astype(np.float32)
    for i, f in enumerate(flank.astype(np.float32), 1):
        d = np.abs(vals - f).astype(np.float32)
        s = np.float32(off) - np.power(d.astype(np.float64), 1.2).astype(np.float32)
        s = np.maximum(s, np.float32(dmin)).astype(np.float64)
        cur = np.empty(n + 1)
        cur[0] = i * ext_v
        cur[1:] = np.maximum(prev[:-1] + s, prev[1:] + ext_v)
        cur = np.maximum.accumulate(cur - ext_h * j) + ext_h * j
        prev = cur
    return prev


def _read_dump(path):
    raw = open(path, "rb").read()
    ng, sc, hh, v, delta, seg, slack, _ = struct.unpack_from("8i", raw, 0)
    pos = 32
    groups = []
    for _g in range(ng):
        a, b, bound, lane_last = struct.unpack_from("4i", raw, pos); pos += 16
        pieces = []
        for _w in range(seg):
            col_off, n, m, nch = struct.unpack_from("4i", raw, pos); pos += 16
            vals = np.frombuffer(raw, np.int32, nch, pos).copy(); pos += 4 * nch
            pieces.append(dict(col_off=col_off, n=n, m=m, vals=vals))
        f = struct.unpack_from("9i2fi", raw, pos); pos += 48
        win = dict(n_win=f[0], lo=f[1:5], hi=f[5:9], lower=f[9], upper=f[10], n_cand=f[11])
        groups.append(dict(a=a, bound=bound, pieces=pieces, win=win, lane_last=lane_last))
    return dict(sc=sc, hh=hh, v=v, mode=delta, slack=slack, groups=groups)


def _planted(rng, n, k, plants, scale=0.45):
    cls = rng.uniform(60, 120, k).astype(np.float32)
    flank = np.repeat(cls, 6)
    lval = (40 + scale * np.arange(256)).astype(np.float32)
    lv = np.repeat(rng.integers(30, 200, n // 5 + 1), rng.integers(3, 10, n // 5 + 1))[:n].astype(np.uint8)
    emb = np.repeat(np.clip(np.round((cls - 40) / scale), 0, 255).astype(np.uint8), rng.integers(6, 10, k))
    for p in plants:
        p = max(0, min(n - len(emb), p))
        lv[p:p + len(emb)] = emb
    return lv, lval, flank


def test_chunk_values_bound_the_exact_last_row(ctx, orc, monkeypatch, tmp_path):
    rng = np.random.default_rng(77)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "0")
    dump = str(tmp_path / "screen.bin")
    monkeypatch.setenv("STRQ_SCREEN_DUMP", dump)
    n, k = 60000, 145
    reads = [_planted(rng, n, k, [20000]), _planted(rng, n, 100, [41000, 5000]), _planted(rng, n, k, [])]
    levels = np.concatenate([r[0] for r in reads])
    flanks = np.concatenate([r[2] for r in reads])
    foff = np.cumsum([0] + [len(r[2]) for r in reads]).astype(np.int64)
    got = ctx.align_batch(levels, np.arange(len(reads) + 1, dtype=np.int64) * n, np.stack([r[1] for r in reads]),
                          np.arange(len(reads), dtype=np.int32), flanks, foff)
    d = _read_dump(dump)
    assert d["sc"] == 1024 and d["mode"] == 1 and len(d["groups"]) == len(reads)
    checked = 0
    for g in d["groups"]:
        lv, lval, flank = reads[g["a"]]
        m = len(flank)
        lM = (m - 1) // R
        assert g["lane_last"] == lM
        shift = -m * d["v"]
        exact = _exact_last_row(lval[lv], flank, params)
        o = orc.align_overlap(lval[lv], flank, params, want_idx=False)
        assert abs(exact.max() - float(o[0])) < 0.5          # the float64 restatement above and the float32 oracle agree
        for pc in g["pieces"]:
            if pc["n"] <= 0:
                continue
            for c, x in enumerate(pc["vals"]):
                lo, hi = 128 * c - 2 * lM + 1, 128 * c - 2 * lM + 128
                lo, hi = max(lo, 1), min(hi, pc["n"])
                if hi < lo:
                    continue
                ub = (int(x) + shift) / d["sc"]
                ex = exact[pc["col_off"] + lo:pc["col_off"] + hi + 1].max()
                # cold-started pieces: a bound of the whole matrix only above the score their overlap was sized for, and only
                # behind their overlap zone (8192 columns in the first call of a context)
                if ex * d["sc"] >= g["bound"] and (pc["col_off"] == 0 or lo > 8192):
                    assert ub >= ex - 1e-3, (g["a"], pc["col_off"], c, ub, ex)
                    checked += 1
                    # ... and a tight one: less than one rounding per row above (when the chunk's best path lies inside the piece)
                    if pc["col_off"] == 0:
                        assert ub <= ex + m / d["sc"] + 0.05, (g["a"], c, ub, ex)
        w = g["win"]
        assert w["lower"] <= float(o[0]) <= w["upper"], (w, float(o[0]))
    assert checked > 500
    for i, (lv, lval, flank) in enumerate(reads):
        o = orc.align_overlap(lval[lv], flank, params, want_idx=False)
        assert np.float32(o[0]).tobytes() == np.float32(got[0][i]).tobytes()
        assert (o[4], o[5]) == (int(got[1][i]), int(got[2][i]))
        assert np.array_equal(o[3], got[3][foff[i]:foff[i + 1]])


@pytest.mark.parametrize("k,n", [(145, 100000), (40, 40000), (149, 70000)])
def test_windowed_alignments_equal_the_oracle(ctx, orc, monkeypatch, k, n):
    rng = np.random.default_rng(5 * k + n)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "0")
    m = 6 * k
    ov = int(m + m * 16 * 1.01 + 64.0) + 1
    seams = [n // 4, n // 2, 3 * n // 4, 8192, n - 8192]
    plants = [[p + d] for p in seams for d in (-700, -1, 0, 1, 130)]
    plants += [[3000, n - 4000], [n // 3, n // 3 + 2000], [100, 9000, 20000, 30000, n - 2000], [], []]
    plants += [[0], [n]]                                       # at the very ends of the read
    reads = []
    for pl in plants:
        lv, lval, flank = _planted(rng, n, k, pl)
        reads.append(lv)
    # the same flank for all (one table shape), own level values per read
    _, lval, flank = _planted(np.random.default_rng(1), n, k, [])
    # re-plant with this flank's embedding so that the occurrences belong to it
    cls = flank[::6]
    emb_rng = np.random.default_rng(2)
    for lv, pl in zip(reads, plants):
        emb = np.repeat(np.clip(np.round((cls - 40) / 0.45), 0, 255).astype(np.uint8), emb_rng.integers(6, 10, k))
        for p in pl:
            p = max(0, min(n - len(emb), p))
            lv[p:p + len(emb)] = emb
    na = len(reads)
    got = ctx.align_batch(np.concatenate(reads), np.arange(na + 1, dtype=np.int64) * n, np.tile(lval, (na, 1)),
                          np.arange(na, dtype=np.int32), np.tile(flank, na), np.arange(na + 1, dtype=np.int64) * m)
    s = ctx.last_screen()
    assert s["screened"] == na and s["windowed"] >= na - 6 and s["scale"] == 1024, s
    assert s["window_columns"] < 0.2 * na * n, s
    from conftest import oracle_map
    _want = oracle_map(lambda lv: orc.align_overlap(lval[lv], flank, params, want_idx=False), reads)
    for i, lv in enumerate(reads):
        o = _want[i]
        assert np.float32(o[0]).tobytes() == np.float32(got[0][i]).tobytes(), (i, plants[i])
        assert (o[4], o[5]) == (int(got[1][i]), int(got[2][i])), (i, plants[i])
        assert np.array_equal(o[3], got[3][i * m:(i + 1) * m]), (i, plants[i])
    # the same batch without the screen: the same bytes
    monkeypatch.setenv("STRQ_NO_SCREEN", "1")
    ref = ctx.align_batch(np.concatenate(reads), np.arange(na + 1, dtype=np.int64) * n, np.tile(lval, (na, 1)),
                          np.arange(na, dtype=np.int32), np.tile(flank, na), np.arange(na + 1, dtype=np.int64) * m)
    assert ctx.last_screen()["screened"] == 0
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_other_parameters_and_short_reads_skip_the_screen(ctx, orc, monkeypatch):
    rng = np.random.default_rng(9)
    lv, lval, flank = _planted(rng, 30000, 145, [12000])
    # general affine parameters: no collapsed recurrence, no screen
    ctx.set_align_params(-3.0, -1.0, -20.0, -4.0, 16.0, 0.0)
    g = ctx.align_batch(lv, [0, len(lv)], lval[None, :], [0], flank, [0, len(flank)])
    assert ctx.last_screen()["screened"] == 0
    o = orc.align_overlap(lval[lv], flank, np.array([-3, -1, -20, -4, 16, 0], np.float32), want_idx=False)
    assert np.float32(o[0]).tobytes() == np.float32(g[0][0]).tobytes()
    # STRique's parameters, a read shorter than the screen's threshold
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "16384")
    g = ctx.align_batch(lv[:9000], [0, 9000], lval[None, :], [0], flank, [0, len(flank)])
    assert ctx.last_screen()["screened"] == 0
    g = ctx.align_batch(lv, [0, len(lv)], lval[None, :], [0], flank, [0, len(flank)])
    assert ctx.last_screen()["screened"] == 1
    o = orc.align_overlap(lval[lv], flank, params, want_idx=False)
    assert np.float32(o[0]).tobytes() == np.float32(g[0][0]).tobytes() and (o[4], o[5]) == (int(g[1][0]), int(g[2][0]))


def test_screen_pauses_after_a_batch_it_cannot_prune(orc, monkeypatch):
    """A sub-batch of reads without the flank, screened in pieces whose cold start is sized for a score none of them reaches
    (STRQ_OVERLAP pins a short overlap), gets no windows: the whole reads run, results as ever, and the context skips the screen
    for its next sub-batches instead of paying for it again."""
    from strique_amd import ffi
    c = ffi.Context(0)
    params = orc.align_params(None)
    c.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "0")
    monkeypatch.setenv("STRQ_OVERLAP", "1500")
    rng = np.random.default_rng(31)
    n, k, na = 30000, 60, 70
    m = 6 * k
    reads = [_planted(rng, n, k, [])[0] for _ in range(na)]
    _, lval, flank = _planted(np.random.default_rng(3), n, k, [])
    args = (np.concatenate(reads), np.arange(na + 1, dtype=np.int64) * n, np.tile(lval, (na, 1)), np.arange(na, dtype=np.int32),
            np.tile(flank, na), np.arange(na + 1, dtype=np.int64) * m)
    got = c.align_batch(*args)
    s = c.last_screen()
    assert s["screened"] == na and s["windowed"] < 0.9 * na, s
    again = c.align_batch(*args)
    s2 = c.last_screen()
    assert s2["screened"] == 0 and s2["whole_read"] == na, s2
    for a, b in zip(got, again):
        assert np.array_equal(a, b)
    for i in (0, 1, na - 1):
        o = orc.align_overlap(lval[reads[i]], flank, params, want_idx=False)
        assert np.float32(o[0]).tobytes() == np.float32(got[0][i]).tobytes() and (o[4], o[5]) == (int(got[1][i]), int(got[2][i]))
    monkeypatch.setenv("STRQ_SCREEN_ALWAYS", "1")
    c.align_batch(*args)
    assert c.last_screen()["screened"] == na
    c.close()


def test_more_occurrences_than_windows(ctx, orc, monkeypatch):
    """Six identical occurrences of the flank, far apart: more separate candidates than the exact launch has pieces -- the last
    window takes the rest of the read, and the leftmost occurrence wins as in the oracle."""
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "0")
    rng = np.random.default_rng(41)
    n, k = 60000, 145
    m = 6 * k
    lv, lval, flank = _planted(rng, n, k, [])
    emb = np.repeat(np.clip(np.round((flank[::6] - 40) / 0.45), 0, 255).astype(np.uint8), rng.integers(6, 10, k))
    for p in (2000, 11000, 20000, 29000, 38000, 47000):
        lv[p:p + len(emb)] = emb
    got = ctx.align_batch(lv, [0, n], lval[None, :], [0], flank, [0, m])
    s = ctx.last_screen()
    assert s["windowed"] == 1 and s["window_columns"] > 15000, s
    o = orc.align_overlap(lval[lv], flank, params, want_idx=False)
    assert np.float32(o[0]).tobytes() == np.float32(got[0][0]).tobytes() and (o[4], o[5]) == (int(got[1][0]), int(got[2][0]))
    assert np.array_equal(o[3], got[3]) and 1900 < int(got[2][0]) < 2000 + len(emb)


def test_a_lower_bound_nobody_reaches_sends_everything_through_the_second_round(orc, monkeypatch):
    """The safety net under the windows: the exact pass must find a score that reaches the screen's lower bound, or the alignment
    runs its whole read in the second round.  With the bound raised artificially (STRQ_SCREEN_TEST_RAISE) every alignment takes
    that route -- and returns the oracle's result."""
    from strique_amd import ffi
    c = ffi.Context(0)
    params = orc.align_params(None)
    c.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "0")
    monkeypatch.setenv("STRQ_SCREEN_TEST_RAISE", "500")
    rng = np.random.default_rng(51)
    n, k = 50000, 145
    m = 6 * k
    reads = []
    _, lval, flank = _planted(np.random.default_rng(6), n, k, [])
    for p in (3000, 20000, 44000):
        lv = _planted(rng, n, k, [])[0]
        emb = np.repeat(np.clip(np.round((flank[::6] - 40) / 0.45), 0, 255).astype(np.uint8), rng.integers(6, 10, k))
        lv[p:p + len(emb)] = emb
        reads.append(lv)
    na = len(reads)
    got = c.align_batch(np.concatenate(reads), np.arange(na + 1, dtype=np.int64) * n, np.tile(lval, (na, 1)), np.arange(na, dtype=np.int32),
                        np.tile(flank, na), np.arange(na + 1, dtype=np.int64) * m)
    assert c.last_screen()["windowed"] == na and c.last_second_round()[0] + c.last_screen()["second_look"] == na, (c.last_screen(), c.last_second_round())
    from conftest import oracle_map
    _want = oracle_map(lambda lv: orc.align_overlap(lval[lv], flank, params, want_idx=False), reads)
    for i, lv in enumerate(reads):
        o = _want[i]
        assert np.float32(o[0]).tobytes() == np.float32(got[0][i]).tobytes() and (o[4], o[5]) == (int(got[1][i]), int(got[2][i]))
        assert np.array_equal(o[3], got[3][i * m:(i + 1) * m])
    c.close()


# ---- the coarse screen (align_screen2_kernel): two flank rows per DP row, both flank alignments of a read in one wave

def _pair_reads(rng, n, ka, kb, plants_a, plants_b):
    """Reads with two flanks each (the prefix / suffix alignment of detect): level streams, one level-value table, flank A and B."""
    cls_a = rng.uniform(60, 120, ka).astype(np.float32); cls_b = rng.uniform(60, 120, kb).astype(np.float32)
    lval = (40 + 0.45 * np.arange(256)).astype(np.float32)
    reads = []
    for pa, pb in zip(plants_a, plants_b):
        lv = np.repeat(rng.integers(30, 200, n // 5 + 1), rng.integers(3, 10, n // 5 + 1))[:n].astype(np.uint8)
        for cls, pl in ((cls_a, pa), (cls_b, pb)):
            emb = np.repeat(np.clip(np.round((cls - 40) / 0.45), 0, 255).astype(np.uint8), rng.integers(6, 10, len(cls)))
            for p in pl:
                p = max(0, min(n - len(emb), p))
                lv[p:p + len(emb)] = emb
        reads.append(lv)
    return reads, lval, np.repeat(cls_a, 6), np.repeat(cls_b, 6)


def _align_pairs(ctx, reads, lval, fa, fb):
    nr, n = len(reads), len(reads[0])
    flanks = np.concatenate([np.concatenate([fa, fb])] * nr)
    foff = np.zeros(2 * nr + 1, np.int64)
    foff[1:] = np.cumsum([len(fa), len(fb)] * nr)
    got = ctx.align_batch(np.concatenate(reads), np.arange(nr + 1, dtype=np.int64) * n, np.tile(lval, (nr, 1)),
                          np.repeat(np.arange(nr, dtype=np.int32), 2), flanks, foff)
    return got, foff


def _check_pairs_against_the_oracle(orc, params, reads, lval, fa, fb, got, foff):
    from conftest import oracle_map
    want = oracle_map(lambda job: orc.align_overlap(lval[reads[job[0]]], (fa, fb)[job[1]], params, want_idx=False), [(i, f) for i in range(len(reads)) for f in range(2)])
    for i, lv in enumerate(reads):
        for f, flank in enumerate((fa, fb)):
            a = 2 * i + f
            o = want[a]
            assert np.float32(o[0]).tobytes() == np.float32(got[0][a]).tobytes(), (i, f)
            assert (o[4], o[5]) == (int(got[1][a]), int(got[2][a])), (i, f)
            assert np.array_equal(o[3], got[3][foff[a]:foff[a + 1]]), (i, f)


@pytest.mark.parametrize("ka,kb", [(145, 145), (100, 140)])
def test_coarse_chunk_values_bound_the_exact_last_row(ctx, orc, monkeypatch, tmp_path, ka, kb):
    """What the coarse screen claims: every chunk value is an upper bound of the exact last row of its columns, for both flanks of a
    read (lanes 0 .. 28 / 32 .. 60 of the wave), flanks shorter than 145 classes included; and what it is used for: the alignments
    return the oracle's bits through its windows."""
    rng = np.random.default_rng(1000 + ka)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "0")
    monkeypatch.setenv("STRQ_SCREEN_MODE", "coarse")
    scale = 512          # three flank rows per DP row (the one merge factor left: DESIGN.md 4.2e)
    dump = str(tmp_path / "screen2.bin")
    monkeypatch.setenv("STRQ_SCREEN_DUMP", dump)
    n = 60000
    reads, lval, fa, fb = _pair_reads(rng, n, ka, kb, [[20000], [41000, 5000], []], [[30000], [12000], [50000]])
    got, foff = _align_pairs(ctx, reads, lval, fa, fb)
    s = ctx.last_screen()
    assert s["mode"] == "coarse" and s["scale"] == scale and s["screened"] == 6, s
    d = _read_dump(dump)
    assert d["sc"] == scale and d["mode"] == 2 and len(d["groups"]) == 6
    checked = 0
    for g in d["groups"]:
        lv = reads[g["a"] // 2]; flank = (fa, fb)[g["a"] % 2]
        m = len(flank)
        lM = g["lane_last"]
        assert lM == (g["a"] % 2) * 32 + (m // 6 - 1) // 5
        shift = -m * d["v"]
        exact = _exact_last_row(lval[lv], flank, params)
        for pc in g["pieces"]:
            if pc["n"] <= 0:
                continue
            for c, x in enumerate(pc["vals"]):
                lo, hi = 128 * c - 2 * lM + 1, 128 * c - 2 * lM + 128
                lo, hi = max(lo, 1), min(hi, pc["n"])
                if hi < lo:
                    continue
                ub = (int(x) + shift) / d["sc"]
                ex = exact[pc["col_off"] + lo:pc["col_off"] + hi + 1].max()
                if ex * d["sc"] >= g["bound"] and (pc["col_off"] == 0 or lo > 8192):
                    assert ub >= ex - 1e-3, (g["a"], pc["col_off"], c, ub, ex)
                    checked += 1
        assert exact.max() <= g["win"]["upper"]
    assert checked > 1000
    _check_pairs_against_the_oracle(orc, params, reads, lval, fa, fb, got, foff)


def test_coarse_windowed_alignments_equal_the_oracle(ctx, orc, monkeypatch):
    """Planted flanks at the seams of the coarse screen's pieces, in their overlap zones, at the ends of the read, twice, five times
    (more candidates than pieces) and not at all: score bits, end / start column and whole path equal the oracle's for both
    alignments of every read; without the screen the same bytes."""
    rng = np.random.default_rng(2024)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "0")
    monkeypatch.setenv("STRQ_SCREEN_MODE", "coarse")
    n, k = 90000, 145
    seams = [n // 4, n // 2, 3 * n // 4, 8192, n - 8192]
    pa = [[p + d] for p in seams for d in (-700, -1, 0, 1, 130)] + [[3000, n - 4000], [100, 9000, 20000, 30000, n - 2000], [], [0], [n]]
    pb = [[(p[0] + 37000) % n] if p else [n // 2] for p in pa]
    pb[-3] = []
    reads, lval, fa, fb = _pair_reads(rng, n, k, k, pa, pb)
    got, foff = _align_pairs(ctx, reads, lval, fa, fb)
    s = ctx.last_screen()
    na = 2 * len(reads)
    assert s["mode"] == "coarse" and s["screened"] == na and s["windowed"] >= na - 8, s
    assert s["window_columns"] < 0.25 * na * n, s
    _check_pairs_against_the_oracle(orc, params, reads, lval, fa, fb, got, foff)
    monkeypatch.setenv("STRQ_NO_SCREEN", "1")
    ref, _ = _align_pairs(ctx, reads, lval, fa, fb)
    assert ctx.last_screen()["screened"] == 0
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_coarse_margin_too_small_ends_in_the_second_round(orc, monkeypatch):
    """The coarse bound is loose: with a candidate margin of one score unit the lower bound it hands on lies above what the exact
    pass can find, the certificate fails, and the alignments run their whole reads in the second round -- the oracle's results."""
    from strique_amd import ffi
    c = ffi.Context(0)
    params = orc.align_params(None)
    c.set_align_params(*[float(v) for v in params])
    c.set_option("STRQ_SCREEN_MIN_N", "0"); c.set_option("STRQ_SCREEN_MODE", "coarse"); c.set_option("STRQ_SCREEN2_MARGIN", "1")
    assert c.get_option("STRQ_SCREEN_MODE") == "coarse"
    rng = np.random.default_rng(77)
    n, k = 50000, 145
    reads, lval, fa, fb = _pair_reads(rng, n, k, k, [[], [], []], [[], [], []])
    # the flanks planted with three samples per k-mer: a merged row needs one column for its two flank rows and matches them all,
    # the exact DP has to skip every other flank row -- the coarse bound lies far above the exact score there
    for lv, pa, pb in zip(reads, (3000, 20000, 44000), (30000, 5000, 10000)):
        for flank, p in ((fa, pa), (fb, pb)):
            emb = np.repeat(np.clip(np.round((flank[::6] - 40) / 0.45), 0, 255).astype(np.uint8), 3)
            lv[p:p + len(emb)] = emb
    got, foff = _align_pairs(c, reads, lval, fa, fb)
    assert c.last_screen()["mode"] == "coarse" and c.last_screen()["second_look"] + c.last_second_round()[0] >= 1, (c.last_screen(), c.last_second_round())
    _check_pairs_against_the_oracle(orc, params, reads, lval, fa, fb, got, foff)
    c.close()


def test_options_per_context_override_the_environment(orc, monkeypatch):
    """strq_set_option: a switch set on a context wins over the environment, "" unsets it for that context, None hands it back."""
    from strique_amd import ffi
    c = ffi.Context(0)
    monkeypatch.setenv("STRQ_NO_SCREEN", "1")
    assert c.get_option("STRQ_NO_SCREEN") == "1"
    c.set_option("STRQ_NO_SCREEN", "")
    assert c.get_option("STRQ_NO_SCREEN") == ""
    params = orc.align_params(None)
    c.set_align_params(*[float(v) for v in params])
    c.set_option("STRQ_SCREEN_MIN_N", "0")
    rng = np.random.default_rng(3)
    lv, lval, flank = _planted(rng, 40000, 145, [12000])
    c.align_batch(lv, [0, len(lv)], lval[None, :], [0], flank, [0, len(flank)])
    assert c.last_screen()["screened"] == 1          # the context's "" beat the environment's STRQ_NO_SCREEN
    c.set_option("STRQ_NO_SCREEN", None)
    c.align_batch(lv, [0, len(lv)], lval[None, :], [0], flank, [0, len(flank)])
    assert c.last_screen()["screened"] == 0
    c.close()


def test_fine_screen_with_both_flanks_per_wave_is_tight(ctx, orc, monkeypatch, tmp_path):
    """The fine screen on a sub-batch that holds both alignments of its reads (every detect call): the two-flanks-per-wave kernel without
    row merging (align_screen1_kernel) -- the same bound as align_screen_kernel: never below the exact last row, less than m / 1024
    above it -- and the fine screen's rules (no margin, no candidate cap).  Results: the oracle's bits."""
    rng = np.random.default_rng(4711)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SCREEN_MIN_N", "0")
    monkeypatch.setenv("STRQ_SCREEN_MODE", "fine")
    dump = str(tmp_path / "screen1.bin")
    monkeypatch.setenv("STRQ_SCREEN_DUMP", dump)
    n = 60000
    reads, lval, fa, fb = _pair_reads(rng, n, 145, 130, [[20000], [41000, 5000], []], [[30000], [12000], [50000]])
    got, foff = _align_pairs(ctx, reads, lval, fa, fb)
    s = ctx.last_screen()
    assert s["mode"] == "fine" and s["merge"] == 1 and s["scale"] == 1024 and s["screened"] == 6, s
    d = _read_dump(dump)
    assert d["sc"] == 1024 and len(d["groups"]) == 6
    checked = 0
    for g in d["groups"]:
        lv = reads[g["a"] // 2]; flank = (fa, fb)[g["a"] % 2]
        m = len(flank); lM = g["lane_last"]; shift = -m * d["v"]
        exact = _exact_last_row(lval[lv], flank, params)
        for pc in g["pieces"]:
            if pc["n"] <= 0:
                continue
            for c, x in enumerate(pc["vals"]):
                lo, hi = max(128 * c - 2 * lM + 1, 1), min(128 * c - 2 * lM + 128, pc["n"])
                if hi < lo:
                    continue
                ub = (int(x) + shift) / d["sc"]
                ex = exact[pc["col_off"] + lo:pc["col_off"] + hi + 1].max()
                if ex * d["sc"] >= g["bound"] and (pc["col_off"] == 0 or lo > 8192):
                    assert ub >= ex - 1e-3, (g["a"], pc["col_off"], c, ub, ex)
                    checked += 1
                    if pc["col_off"] == 0:
                        assert ub <= ex + m / d["sc"] + 0.05, (g["a"], c, ub, ex)
        assert g["win"]["lower"] <= exact.max() + 1e-3 <= g["win"]["upper"] + 1e-3
    assert checked > 1000
    _check_pairs_against_the_oracle(orc, params, reads, lval, fa, fb, got, foff)
