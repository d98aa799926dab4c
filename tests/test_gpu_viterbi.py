"""HIP Viterbi vs the CPU oracle through the C ABI: identical log-probability bits, counts, paths.

Both sides of THIS file decode the same baked arrays (the product's `hmm.bake()` output, or random models): it pins
the kernel against the oracle's decode, not the bake.  That `hmm.bake()` itself equals the reference's model is the
business of tests/test_oracle_independent.py (three-way decode: graph recorded from the reference's classes, the
oracle's own un-baked construction, the product's baked arrays) and of every detect-level GPU test, whose
expectations are built inside oracle/ from sequences, never from product objects."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from strique_amd import ffi
    return ffi.Context(0)


def _signal(pm, rng, seq, noise=True):
    s = pm.generate_signal(seq, samples=8, noise=noise, rng=rng)
    return np.clip(s, pm.model_min + .5, pm.model_max - .5)


@pytest.mark.parametrize("name", ["c9orf72", "fmr1"])
def test_flanked_model_parity(ctx, orc, pm, cfg, name):
    from strique_amd import hmm
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"][name]
    fm = hmm.FlankedRepeatModel(repeat, prefix[-50:], suffix[:50], pm, cfg["HMM"])
    mid = ctx.model_create(fm.baked)
    rng = np.random.default_rng(3)
    for nrep, noise in ((3, False), (17, True), (120, True)):
        x = _signal(pm, rng, prefix[-50:] + repeat * nrep + suffix[:50], noise)
        lo, po, co = orc.viterbi(fm.baked, x)
        lg, cg, sg, pg = ctx.viterbi(mid, x, want_path=True)
        assert np.float64(lo).tobytes() == np.float64(lg).tobytes() and co == cg and sg == 0
        assert np.array_equal(po, pg)
        assert co + fm.count_bias == nrep
        lg2, cg2, _, _ = ctx.viterbi(mid, x, want_path=False)      # count-only kernel variant
        assert lg2 == lg and cg2 == cg


def test_register_resident_kernel_variants_agree(ctx, orc, pm, cfg, monkeypatch):
    """Count-only decodes of a flanked model run on viterbi_g2_kernel (strq_model_set_positions accepted the chain): its
    three exchange levels (all DPP / odd-slot neighbours through LDS / skip and broadcast sources too) and the lane-layout
    kernel must give the oracle's log-probability and count on the same windows -- even and odd repeat profiles, ragged
    lengths, a window of missing observations."""
    from strique_amd import hmm
    rng = np.random.default_rng(77)
    for name, rep_override in (("c9orf72", None), ("fmr1", None), ("c9orf72", "CAGCA")):
        chrom, b, e, repeat, prefix, suffix = cfg["repeat"][name]
        repeat = rep_override or repeat
        fm = hmm.FlankedRepeatModel(repeat, prefix[-50:], suffix[:50], pm, cfg["HMM"])
        mid = ctx.model_create(fm.baked)
        assert ctx.last_positions_rc == 0
        seqs = [_signal(pm, rng, prefix[-50:] + repeat * int(k) + suffix[:50]) for k in (1, 2, 7, 33, 90)]
        seqs.append(np.full(57, np.nan)); seqs.append(seqs[2][:1])
        want = [orc.viterbi(fm.baked, s, want_path=False) for s in seqs]
        for env in ({"STRQ_VIT_G2_LDS": "0"}, {"STRQ_VIT_G2_LDS": "1"}, {"STRQ_VIT_G2_LDS": "2"}, {"STRQ_VIT_NO_G2": "1"}, {"STRQ_VIT_G2_WAVES": "4"}):
            with monkeypatch.context() as mp:
                for k, v in env.items():
                    mp.setenv(k, v)
                lg, cg, sg, _ = ctx.viterbi_batch(mid, seqs)
            for i, (lo, _, co) in enumerate(want):
                if len(seqs[i]) == 1 and not np.isfinite(lo):
                    assert not np.isfinite(lg[i]); continue
                assert np.float64(lo).tobytes() == np.float64(lg[i]).tobytes() and co == cg[i], (name, env, i)


def test_register_resident_kernel_breaks_ties_like_the_oracle(ctx, orc, pm, cfg, monkeypatch):
    """Integer log-probabilities and constant emissions (tests/test_g2_layout.py::tie_prone): equal candidates at almost every
    state and time step, so the count depends on every tie going to the first in-edge in ascending source order -- including
    the ties that the relay flag of the even insert slot settles (VitG2::hub_mask; the crafted seed-0 model reaches them at
    every time step after the first round of the loop).  Every exchange level of viterbi_g2_kernel and the lane layout."""
    from strique_amd import hmm
    from test_g2_layout import tie_prone
    for name, rep_override in (("c9orf72", None), ("fmr1", None), ("c9orf72", "CAGCA")):
        chrom, b, e, repeat, prefix, suffix = cfg["repeat"][name]
        repeat = rep_override or repeat
        bk0 = hmm.FlankedRepeatModel(repeat, prefix[-30:], suffix[:30], pm, cfg["HMM"]).baked
        for seed in range(4):
            rng = np.random.default_rng(100 + seed)
            bk = tie_prone(bk0, rng if seed else None)
            mid = ctx.model_create(bk)
            assert ctx.last_positions_rc == 0
            seqs = [rng.uniform(10.0, 150.0, T) for T in (40, 75, 130, 333)]
            want = [orc.viterbi(bk, s, want_path=False) for s in seqs]
            for env in ({"STRQ_VIT_G2_LDS": "2"}, {"STRQ_VIT_G2_LDS": "1"}, {"STRQ_VIT_G2_LDS": "0"}, {"STRQ_VIT_NO_G2": "1"}):
                with monkeypatch.context() as mp:
                    for k, v in env.items():
                        mp.setenv(k, v)
                    lg, cg, sg, _ = ctx.viterbi_batch(mid, seqs)
                for i, (lo, _, co) in enumerate(want):
                    assert lo == lg[i] and co == cg[i] and sg[i] == 0, (name, seed, env, i, lo, lg[i], co, cg[i])


def test_flanked_model_with_counted_silent_states(ctx, orc, pm, cfg):
    """The kernels picked for STRique's flanked models carry a silent state's payload on unchanged (STRique counts the
    emitting dummy states only, scripts/STRique.py:341-342,375-377).  The same model with some delete states counted must
    run on a kernel that adds their increments: count, log-probability and path against the oracle."""
    from strique_amd import hmm
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    fm = hmm.FlankedRepeatModel(repeat, prefix[-50:], suffix[:50], pm, cfg["HMM"])
    rng = np.random.default_rng(31)
    inc = fm.baked.count_inc.copy()
    silent = np.arange(fm.baked.silent_start, fm.baked.n_states)
    inc[silent[::2]] = 1
    baked = fm.baked._replace(count_inc=inc)
    mid = ctx.model_create(baked)
    mid0 = ctx.model_create(fm.baked)
    differs = False
    for nrep, k in ((5, 0), (40, 6), (90, 13)):
        seq = prefix[-50:] + repeat * nrep + suffix[:50]
        seq = seq[:30] + seq[30 + k:]          # a deletion in the prefix: the best path goes through delete states
        x = _signal(pm, rng, seq)
        lo, po, co = orc.viterbi(baked, x)
        lg, cg, sg, pg = ctx.viterbi(mid, x, want_path=True)
        assert np.float64(lo).tobytes() == np.float64(lg).tobytes() and co == cg and sg == 0
        assert np.array_equal(po, pg)
        lg2, cg2, _, _ = ctx.viterbi(mid, x, want_path=False)
        assert lg2 == lg and cg2 == cg
        l0, c0, _, _ = ctx.viterbi(mid0, x, want_path=False)
        assert l0 == lg and c0 == orc.viterbi(fm.baked, x, want_path=False)[2]
        differs = differs or c0 != cg
    assert differs, "the counted silent states never lay on a best path: the test checks nothing"


def test_mod_model_and_edge_cases(ctx, orc, pm, pm_mod, cfg):
    from strique_amd import hmm
    mm = hmm.RepeatModModel("GGCCCC", pm, pm_mod, cfg["HMM"])
    mid = ctx.model_create(mm.baked)
    rng = np.random.default_rng(4)
    x = np.clip(_signal(pm, rng, "GGCCCC" * 40 + "GGCCC"), mm.model_min, mm.model_max)
    lo, po, co = orc.viterbi(mm.baked, x)
    lg, cg, sg, pg = ctx.viterbi(mid, x, want_path=True)
    assert lo == lg and np.array_equal(po, pg)
    # no path: observation outside every emission support
    lg, cg, sg, pg = ctx.viterbi(mid, np.full(20, 1e6), want_path=True)
    assert sg == 1 and lg == -np.inf
    assert orc.viterbi(mm.baked, np.full(20, 1e6))[1] is None
    # a single observation
    lo, po, co = orc.viterbi(mm.baked, x[:1])
    lg, cg, sg, pg = ctx.viterbi(mid, x[:1], want_path=True)
    assert (lo == lg or (np.isinf(lo) and np.isinf(lg)))


def test_ragged_batch(ctx, orc, pm, cfg):
    from strique_amd import hmm
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    fm = hmm.FlankedRepeatModel(repeat, prefix[-50:], suffix[:50], pm, cfg["HMM"])
    mid = ctx.model_create(fm.baked)
    rng = np.random.default_rng(8)
    seqs = [_signal(pm, rng, prefix[-50:] + repeat * int(k) + suffix[:50]) for k in rng.integers(1, 80, 70)]
    lg, cg, sg, _ = ctx.viterbi_batch(mid, seqs)
    for i, s in enumerate(seqs):
        lo, _, co = orc.viterbi(fm.baked, s, want_path=False)
        assert lo == lg[i] and co == cg[i] and sg[i] == 0


def _random_model(rng, ne, ns, extra_silent_edges):
    """A random baked model: `ne` emitting states, `ns` silent ones (first = start, last = end) in
    topological order, in-edges sorted by source, every state with at least one in-edge."""
    import math
    from strique_amd.hmm import BakedHMM
    n = ne + ns
    start, end = ne, n - 1
    ins = [set() for _ in range(n)]
    for l in range(ne):                                      # emitting: from emitting states and silent non-end states
        for k in rng.choice(ne, size=int(rng.integers(1, 5)), replace=False):
            ins[l].add(int(k))
        if rng.random() < 0.5:
            ins[l].add(int(rng.integers(ne, n - 1)))
    for l in range(ne + 1, n):                               # silent (not start): from emitting and lower silent states
        for k in rng.choice(ne, size=int(rng.integers(1, 4)), replace=False):
            ins[l].add(int(k))
        ins[l].add(l - 1)                                    # a chain through the silent states
        if extra_silent_edges and l - ne >= 3 and rng.random() < 0.5:
            ins[l].add(int(rng.integers(ne, l - 1)))         # silent predecessor outside the chain: multi-stage model
    in_ptr = np.zeros(n + 1, np.int32); src, lp = [], []
    for l in range(n):
        ks = sorted(ins[l])
        in_ptr[l + 1] = in_ptr[l] + len(ks)
        src += ks; lp += [math.log(rng.uniform(0.05, 0.9)) for _ in ks]
    kind = rng.integers(1, 3, ne).astype(np.int32)
    mu = rng.uniform(60, 120, ne); sigma = rng.uniform(1.0, 4.0, ne)
    ea = np.where(kind == 1, mu, 40.0); eb = np.where(kind == 1, 1.0 / (2 * sigma ** 2), 140.0)
    ec = np.where(kind == 1, -np.log(sigma * 2.50662827463), -math.log(100.0))
    count_inc = (rng.random(n) < 0.1).astype(np.int32)
    return BakedHMM(n, ne, start, end, in_ptr, np.array(src, np.int32), np.array(lp), kind, ea, eb, ec, count_inc,
                    np.zeros(n, np.int32), ["s%d" % i for i in range(n)], np.arange(n, dtype=np.int32),
                    np.full(n, -1, np.int32), np.full(n, -1, np.int32))


@pytest.mark.parametrize("ne,ns,multi", [(5, 3, False), (40, 10, False), (64, 20, True), (130, 40, True), (200, 70, False), (260, 130, True)])
def test_random_models_generic_kernel_shapes(ctx, orc, ne, ns, multi):
    """Models without layout hints and with arbitrary topology go through the generic kernel shapes
    (degree-sorted ownership, chain decomposition, multi-stage silent phase): log-probability bits,
    state path and carried counts must equal the oracle's."""
    rng = np.random.default_rng(1000 + ne)
    baked = _random_model(rng, ne, ns, multi)
    mid = ctx.model_create(baked)
    for T in (1, 7, 150):
        x = rng.uniform(55, 125, T)
        lo, po, co = orc.viterbi(baked, x)
        lg, cg, sg, pg = ctx.viterbi(mid, x, want_path=True)
        if po is None:
            assert sg == 1 and lg == -np.inf
            continue
        assert np.float64(lo).tobytes() == np.float64(lg).tobytes() and sg == 0
        lg2, cg2, _, _ = ctx.viterbi(mid, x, want_path=False)
        assert lg2 == lg and cg2 == cg
        # ties between equal-probability paths may be broken differently only if the value is the same;
        # a path is accepted when it is the oracle's, or scores the same log-probability
        if not np.array_equal(po, pg):
            pytest.fail("state path differs from the oracle's")
        assert co == cg


def test_missing_observations(ctx, orc, pm, cfg):
    """NaN observations (pomegranate's missing-value rule, [recalled]: log-probability 0 under every distribution):
    all-NaN windows and NaNs mixed with numbers through strq_viterbi, count and path included."""
    from strique_amd import hmm
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    fm = hmm.FlankedRepeatModel(repeat, prefix[-50:], suffix[:50], pm, cfg["HMM"])
    mid = ctx.model_create(fm.baked)
    rng = np.random.default_rng(13)
    x = _signal(pm, rng, prefix[-50:] + repeat * 25 + suffix[:50])
    cases = [np.full(12, np.nan), np.full(333, np.nan), np.where(rng.random(len(x)) < 0.25, np.nan, x), np.where(np.arange(len(x)) % 2 == 0, np.nan, x)]
    for xn in cases:
        lo, po, co = orc.viterbi(fm.baked, xn)
        lg, cg, sg, pg = ctx.viterbi(mid, xn, want_path=True)
        assert np.float64(lo).tobytes() == np.float64(lg).tobytes() and co == cg and sg == 0 and np.isfinite(lg)
        assert np.array_equal(po, pg)
        lg2, cg2, _, _ = ctx.viterbi(mid, xn, want_path=False)
        assert lg2 == lg and cg2 == cg


@pytest.mark.parametrize("ne,ns,multi,fan", [(600, 300, True, 4), (1500, 700, False, 4), (40, 12, True, 14), (2600, 1400, True, 5)])
def test_models_beyond_the_lane_layouts_run_on_the_general_kernel(ctx, orc, ne, ns, multi, fan):
    """More than 512 emitting / 256 silent states, or more than eight in-edges per state: no lane layout exists and the
    model goes to viterbi_csr_kernel (one workgroup per window, silent states level by level).  Log-probability bits,
    counts and full state paths equal the oracle's; a model beyond 4096 states is refused when it is registered."""
    from strique_amd import ffi
    rng = np.random.default_rng(7000 + ne)
    baked = _random_model(rng, ne, ns, multi)
    if fan > 8:                                   # widen the in-degree of some emitting states beyond 8
        in_ptr = [0]; src = []; lp = []
        for l in range(baked.n_states):
            ks = list(baked.in_src[baked.in_ptr[l]:baked.in_ptr[l + 1]]); ps = list(baked.in_logp[baked.in_ptr[l]:baked.in_ptr[l + 1]])
            if l < ne and l % 3 == 0:
                extra = [int(k) for k in rng.choice(ne, size=fan, replace=False) if int(k) not in ks]
                ks += extra; ps += [float(np.log(rng.uniform(0.05, 0.9))) for _ in extra]
                order = np.argsort(ks); ks = [ks[i] for i in order]; ps = [ps[i] for i in order]
            src += ks; lp += ps; in_ptr.append(len(src))
        baked = baked._replace(in_ptr=np.array(in_ptr, np.int32), in_src=np.array(src, np.int32), in_logp=np.array(lp))
    mid = ctx.model_create(baked)
    for T in (1, 9, 120):
        x = rng.uniform(55, 125, T)
        if T == 120:
            x[::7] = np.nan                       # missing observations as well
        lo, po, co = orc.viterbi(baked, x)
        lg, cg, sg, pg = ctx.viterbi(mid, x, want_path=True)
        if po is None:
            assert sg == 1 and lg == -np.inf
            continue
        assert np.float64(lo).tobytes() == np.float64(lg).tobytes() and sg == 0 and co == cg
        assert np.array_equal(po, pg)
        lg2, cg2, _, _ = ctx.viterbi(mid, x, want_path=False)
        assert lg2 == lg and cg2 == cg
    if ne == 2600:
        with pytest.raises(ffi.StriqueHipError, match="more than 4096 states"):
            ctx.model_create(_random_model(rng, 3000, 1200, False))
    # a model of this size with a broken edge list is refused as a bad argument before any builder indexes with it
    # (round-3 advisor finding: the general kernel's builder used to receive such models unvalidated)
    if ne >= 600:
        big = baked
        e0 = int(big.in_ptr[big.silent_start + 5])                # first in-edge of a silent state
        assert big.in_ptr[big.silent_start + 6] > e0
        for what, breaker in (("edge source out of range", lambda a: a.__setitem__(e0, big.n_states + 3)),
                              ("topological order", lambda a: a.__setitem__(e0, big.n_states - 1))):
            src = big.in_src.copy(); breaker(src)
            with pytest.raises(ffi.StriqueHipError, match=what) as ei:
                ctx.model_create(big._replace(in_src=src))
            assert ei.value.code == ffi.STRQ_ERR_ARG
        deg = np.diff(big.in_ptr); l2 = int(np.argmax(deg >= 2))      # a state with two in-edges: swap them
        src = big.in_src.copy(); a0 = int(big.in_ptr[l2]); src[a0], src[a0 + 1] = src[a0 + 1], src[a0]
        with pytest.raises(ffi.StriqueHipError, match="sorted by source") as ei:
            ctx.model_create(big._replace(in_src=src))
        assert ei.value.code == ffi.STRQ_ERR_ARG
        ptr = big.in_ptr.copy(); ptr[10] = ptr[11] + 1
        with pytest.raises(ffi.StriqueHipError, match="in_ptr") as ei:
            ctx.model_create(big._replace(in_ptr=ptr))
        assert ei.value.code == ffi.STRQ_ERR_ARG
        kinds = big.emis_kind.copy(); kinds[3] = 0
        with pytest.raises(ffi.StriqueHipError, match="emission kind") as ei:
            ctx.model_create(big._replace(emis_kind=kinds))
        assert ei.value.code == ffi.STRQ_ERR_ARG
