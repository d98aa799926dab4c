"""HIP Viterbi vs the CPU oracle through the C ABI: identical log-probability bits, counts, paths."""
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
