"""Internal consistency of the C oracles (no GPU): memoised vs per-cell pow, view positions vs the
compact record, Viterbi vs a brute-force pure-python decode on tiny models."""
import itertools
import math

import numpy as np


def _toy(rng, n, k=20, s=6):
    cls = rng.uniform(60, 120, k).astype(np.float32)
    flank = np.repeat(cls, s)
    lval = (40 + 0.45 * np.arange(256)).astype(np.float32)
    lv = np.repeat(rng.integers(30, 200, n // 5 + 1), rng.integers(3, 9, n // 5 + 1))[:n].astype(np.uint8)
    emb = np.repeat(np.clip(np.round((cls - 40) / 0.45), 0, 255).astype(np.uint8), rng.integers(5, 10, k))
    pos = int(rng.integers(0, max(1, n - len(emb))))
    emb = emb[:max(0, n - pos)]
    lv[pos:pos + len(emb)] = emb
    return lval[lv], flank


def test_align_lut_equals_pow_and_positions(orc):
    rng = np.random.default_rng(2)
    params = orc.align_params(None)
    for n in (1, 7, 60, 300, 1500):
        a, b = _toy(rng, n)
        r0 = orc.align_overlap(a, b, params, use_lut=False)
        r1 = orc.align_overlap(a, b, params, use_lut=True)
        assert np.float32(r0[0]).tobytes() == np.float32(r1[0]).tobytes()
        for x, y in zip(r0[1:], r1[1:]):
            assert np.array_equal(x, y)
        score, a_idx, b_idx, rec, j_end, j0 = r1
        # view positions are strictly increasing and consistent with the record
        assert np.all(np.diff(a_idx.astype(np.int64)) > 0) and np.all(np.diff(b_idx.astype(np.int64)) > 0)
        rows = list(range(len(b)))
        want = [int(np.abs(a_idx.astype(np.int64) - int(b_idx[k])).argmin()) for k in rows]
        assert orc.positions_from_rec(rec, j0, j_end, n, rows) == want


def test_align_affine_general_parameters(orc):
    rng = np.random.default_rng(4)
    a, b = _toy(rng, 400)
    for p in ([-2, -8, -2, -8, 8, -16], [-3, -1, -20, -4, 16, 0], [-1, -1, -16, -16, 16, -2]):
        params = np.array(p, np.float32)
        r0 = orc.align_overlap(a, b, params, use_lut=False)
        r1 = orc.align_overlap(a, b, params, use_lut=True)
        assert r0[0] == r1[0] and np.array_equal(r0[3], r1[3])


def _brute_viterbi(baked, x):
    """Enumerate all state paths of a tiny model (emitting sequences x silent closures) -- exponential."""
    m, ne = baked.n_states, baked.silent_start
    edges = {}
    for l in range(m):
        for e in range(baked.in_ptr[l], baked.in_ptr[l + 1]):
            edges.setdefault(int(baked.in_src[e]), []).append((l, float(baked.in_logp[e])))

    def emis(l, v):
        if baked.emis_kind[l] == 1:
            d = v - baked.emis_a[l]
            return baked.emis_c[l] - (d * d) * baked.emis_b[l]
        return baked.emis_c[l] if baked.emis_a[l] <= v <= baked.emis_b[l] else -math.inf
    best = [-math.inf]

    def go(state, t, lp, depth):
        if lp == -math.inf or depth > 4 * (len(x) + m):
            return
        if t == len(x) and state == baked.end:
            best[0] = max(best[0], lp)
        for nxt, a in edges.get(state, []):
            if nxt < ne:
                if t < len(x):
                    go(nxt, t + 1, lp + a + emis(nxt, x[t]), depth + 1)
            else:
                go(nxt, t, lp + a, depth + 1)
    go(baked.start, 0, 0.0, 0)
    return best[0]


def test_viterbi_against_brute_force(orc, pm):
    from strique_amd import hmm
    g = hmm.Graph()
    prof = hmm.add_profile(g, "ACGTACGT", pm, None, "x")         # 3 k-mers, with silent deletes
    g.add_transition(g.start, prof.s1, 0.3); g.add_transition(g.start, prof.s2, 0.7)
    g.add_transition(prof.e1, g.end, 1); g.add_transition(prof.e2, g.end, 1)
    baked = hmm.bake(g)
    rng = np.random.default_rng(6)
    for T in (1, 2, 4, 6):
        x = rng.uniform(70, 110, T)
        lp, path, _ = orc.viterbi(baked, x)
        assert abs(lp - _brute_viterbi(baked, x)) < 1e-9 * max(1.0, abs(lp))
        assert path is not None and len(path) == T and np.all(path < baked.silent_start)


def test_viterbi_missing_observations(orc, pm, cfg):
    """A NaN observation has log-probability 0 under every emission (pomegranate >= 0.9 missing-value support,
    [recalled]; the reference reaches it at scripts/STRique.py:603 when normalize2model returns NaNs): decoding NaNs
    equals decoding the same model with every emission replaced by Uniform(0, 1) -- log 1 = 0 -- at x = 0.5, for
    all-NaN windows and for NaNs mixed with numbers."""
    from strique_amd import hmm
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    fm = hmm.FlankedRepeatModel(repeat, prefix[-50:], suffix[:50], pm, cfg["HMM"])
    ne = fm.baked.silent_start
    flat = fm.baked._replace(emis_kind=np.full(ne, 2, np.int32), emis_a=np.zeros(ne), emis_b=np.ones(ne), emis_c=np.zeros(ne))
    for T in (1, 40, 700):
        want = orc.viterbi(flat, np.full(T, 0.5))
        got = orc.viterbi(fm.baked, np.full(T, np.nan))
        assert want[0] == got[0] and np.array_equal(want[1], got[1]) and want[2] == got[2] and (np.isfinite(got[0]) or T == 1)
    # mixed: NaN only where the mask says so
    rng = np.random.default_rng(5)
    x = np.clip(pm.generate_signal(prefix[-50:] + repeat * 12 + suffix[:50], samples=8, noise=True, rng=rng), pm.model_min + .5, pm.model_max - .5)
    mask = rng.random(len(x)) < 0.3
    xn = np.where(mask, np.nan, x)
    lp, path, cnt = orc.viterbi(fm.baked, xn)
    assert path is not None and np.isfinite(lp)
    # score of that path recomputed by hand: transitions + emissions of the observed samples only
    total = 0.0
    bk = fm.baked
    for t, l in enumerate(path):
        if not mask[t]:
            if bk.emis_kind[l] == 1:
                total += bk.emis_c[l] - (x[t] - bk.emis_a[l]) ** 2 * bk.emis_b[l]
            else:
                total += bk.emis_c[l]
    lp_clean = orc.viterbi(bk, x)[0]
    assert lp > lp_clean          # dropping evidence can only raise the best path's log-probability (emissions are <= 0 here)
    assert total > lp - 1e-6 * abs(lp) or total <= 0.0
