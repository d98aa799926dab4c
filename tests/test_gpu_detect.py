"""repeatCounter.detect on the GPU vs the CPU oracle: every output field identical."""
import numpy as np
import pytest

from conftest import oracle_tc

pytestmark = pytest.mark.gpu


_TABLE = []


def _read(pm, targets, name, strand, total_nt, nrep, idx, as_int16=True):
    from strique_amd import synth
    if not _TABLE:
        _TABLE.append(synth.KmerTable(pm))
    return synth.make_read(_TABLE[0], 9, idx, total_nt, targets[name], nrep, strand=strand, as_int16=as_int16)[0]


@pytest.fixture
def want(orc, opm, targets, cfg):
    """The oracle's detect() for (name, signal, strand), with the oracle's own classifier."""
    params = orc.align_params(cfg["align"])
    return lambda name, sig, strand: orc.detect(sig, oracle_tc(orc, opm, targets, name, strand, cfg["HMM"]), opm, params)[0]


def _check(gpu_counter, want, items):
    got = gpu_counter.detect_batch([(n, s, st) for n, s, st in items])
    for (name, sig, strand), g in zip(items, got):
        w = want(name, sig, strand)
        assert tuple(g[:6]) == tuple(w[:6]), (name, strand, g, w)
    return got


def test_int16_and_float64_reads(gpu_counter, want, pm, targets):
    rng = np.random.default_rng(2)
    items = []
    for k in range(10):
        name = ["c9orf72", "fmr1"][k % 2]; strand = "+-"[(k // 2) % 2]
        items.append((name, _read(pm, targets, name, strand, int(rng.integers(3000, 8000)), int(rng.integers(4, 90)), k, as_int16=(k % 3 != 0)), strand))
    ints = [it for it in items if it[1].dtype == np.int16]
    flts = [it for it in items if it[1].dtype != np.int16]
    _check(gpu_counter, want, ints)
    _check(gpu_counter, want, flts)


def test_reference_scenarios(pm, pm_mod, cfg):
    """All four tests of the reference's scripts/STRique_test.py through the drop-in `repeatCounter`, with their
    own loops and assertions (`n == i`): test_Detection (:43-62), test_Interpolation with its own flanks
    (:66-82), test_Normalization (:85-100), test_Modification (:103-124).  Class defaults, no JSON config,
    float64 signals from generate_signal -- exactly how the reference's tests call it (backbone and noise seeded here)."""
    import random
    from strique_amd.counter import repeatCounter
    from test_oracle_golden import _INTERP_PREFIX, _INTERP_SUFFIX
    rnd = random.Random(20260102)
    backbone = ''.join(rnd.choice('ACTG') for _ in range(2000))
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    dt = repeatCounter(pm, device=0)
    dt.add_target('c9orf72', repeat, prefix, suffix)
    for i in range(100, 301, 100):
        seq = backbone[:1000] + prefix + repeat * i + suffix + backbone[-1000:]
        assert dt.detect('c9orf72', pm.generate_signal(seq, samples=8), '+')[0] == i
    for i in range(10, 100, 10):
        assert dt.detect('c9orf72', pm.generate_signal(prefix + repeat * i + suffix, samples=8), '+')[0] == i
    dt.add_target('fmr1', 'GCG', _INTERP_PREFIX, _INTERP_SUFFIX)
    for i in range(100, 301, 100):
        seq = backbone[:1000] + _INTERP_PREFIX + 'GCG' * i + _INTERP_SUFFIX + backbone[-1000:]
        assert dt.detect('fmr1', pm.generate_signal(seq, samples=8), '+')[0] == i
    dm = repeatCounter(pm, mod_model_file=pm_mod, device=0)
    dm.add_target('c9orf72', repeat, prefix, suffix)
    rng = np.random.default_rng(20260103)
    for i in range(100, 301, 100):
        seq = backbone[:1000] + prefix + repeat * i + suffix + backbone[-1000:]
        dm.detect('c9orf72', pm.generate_signal(seq, samples=8, noise=True, rng=rng), '+')
        n = dm.detect('c9orf72', pm_mod.generate_signal(seq, samples=8, noise=True, rng=rng), '+')[0]
        assert n == i


def test_bad_reads_do_not_kill_the_batch(gpu_counter, want, pm, targets):
    rng = np.random.default_rng(6)
    good = _read(pm, targets, "c9orf72", "+", 4000, 12, 77)
    const = np.full(3000, 500, np.int16)                       # normalisation undefined
    noise = rng.integers(300, 900, 5000).astype(np.int16)      # no locus inside
    got = gpu_counter.detect_batch([("c9orf72", const, "+"), ("c9orf72", good, "+"), ("fmr1", noise, "-")])
    assert got[0][0] == 0 and got[0][6] == "-"
    assert tuple(got[1][:6]) == tuple(want("c9orf72", good, "+")[:6]) and got[1][0] == 12
    assert tuple(got[2][:6]) == tuple(want("fmr1", noise, "-")[:6])


def test_errors_match_the_reference(gpu_counter):
    with pytest.raises(ValueError):
        gpu_counter.detect("nope", np.zeros(100, np.int16), "+")           # STRique.py:618
    with pytest.raises(ValueError):
        gpu_counter.detect("c9orf72", np.zeros(100, np.int16), "x")        # STRique.py:589
    with pytest.raises(ValueError):
        gpu_counter.add_target("c9orf72", "GGCCCC", "A" * 150, "C" * 150)  # STRique.py:579


def test_conditioning_stage(gpu_counter, orc, opm, pm, targets):
    """Steps 1-6 of detect (STRique.py:590-597): 8-bit morphology levels and their values."""
    sigs = [_read(pm, targets, "c9orf72", "+", 3000 + 500 * i, 10 + i, 200 + i) for i in range(3)]
    gpu_counter.detect_batch([("c9orf72", s, "+") for s in sigs])
    for i, s in enumerate(sigs):
        lv, lval, sc = gpu_counter.ctx.debug_conditioning(i, len(s))
        flt, u8, morph, fltn = orc.condition(s, opm)
        assert np.array_equal(lv, u8)
        uq, first = np.unique(u8, return_index=True)
        assert np.array_equal(lval[uq], morph[first].astype(np.float32))
        assert sc[0] == np.median(flt) and sc[1] == orc.mad(flt)


def test_full_size_reads_properties(gpu_counter, want, pm, targets):
    """BASELINE config 3 size (50 kb reads).  One read is compared field by field with the oracle;
    for the rest: determinism, planted count recovered, geometry consistent with the planted locus."""
    plan = [(200, "+"), (1000, "-"), (2000, "+"), (500, "-")]
    sigs = [_read(pm, targets, "c9orf72", st, 50000, n, 300 + i) for i, (n, st) in enumerate(plan)]
    items = [("c9orf72", s, st) for s, (n, st) in zip(sigs, plan)]
    a = gpu_counter.detect_batch(items)
    b = gpu_counter.detect_batch(items)
    assert a == b
    for (n, st), r, s in zip(plan, a, sigs):
        assert abs(r[0] - n) <= 2
        assert 0 < r[4] < len(s) and 6 * 6 * n * 0.9 < r[5] < 9 * 6 * n * 1.1      # ticks ~ 6 nt x n x dwell
    assert tuple(a[0][:6]) == tuple(want("c9orf72", sigs[0], "+")[:6])


def test_modification_pass(pm, pm_mod, cfg, orc, opm, opm_mod, targets):
    """detect steps 12-14 (STRique.py:605-609) and repeatModHMM.mod_repeats (:492-500): the
    pattern string must equal the oracle's, for signals drawn from the base and from the mCpG model
    (the reference's own test_Modification, scripts/STRique_test.py:104-124, only asserts the count)."""
    from strique_amd.counter import repeatCounter
    from strique_amd import hmm
    rc = repeatCounter(pm, mod_model_file=pm_mod, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    rc.add_target("c9orf72", repeat, prefix, suffix)
    rng = np.random.default_rng(12)
    backbone = "".join(rng.choice(list("ACTG"), 2000))
    items, truth = [], []
    for i, model in ((40, pm), (40, pm_mod), (110, pm_mod), (7, pm)):
        seq = backbone[:1000] + prefix + repeat * i + suffix + backbone[-1000:]
        sig = model.generate_signal(seq, samples=8, noise=True, rng=rng)
        items.append(("c9orf72", sig, "+")); truth.append(i)
        items.append(("c9orf72", np.round(sig * (8192 / 1400.0) - 10).astype(np.int16), "+")); truth.append(i)
    flt_items = [it for it in items if it[1].dtype != np.int16]
    int_items = [it for it in items if it[1].dtype == np.int16]
    tc = oracle_tc(orc, opm, targets, "c9orf72", "+", cfg["HMM"], opm_mod)
    params = orc.align_params(cfg["align"])
    for group, planted in ((flt_items, truth[0::2]), (int_items, truth[1::2])):
        got = rc.detect_batch(group)
        for (name, sig, strand), g, n_true in zip(group, got, planted):
            want, _ = orc.detect(sig, tc, opm, params, pm_mod=opm_mod)
            assert tuple(g) == tuple(want), (g, want)
            assert set(g[6]) <= set("01") and abs(len(g[6]) - g[0]) <= 3
            assert abs(g[0] - n_true) <= 1, (g[0], n_true)
    # sub-batches of a modification target are in flight two at a time as well (their MARK-mode Viterbi launches under the next one's
    # alignments, their second pass when the rows are taken): the same rows and patterns in pieces, and in the serial order
    whole = rc.detect_batch(int_items + int_items[::-1])
    rc.ctx.set_option("STRQ_SUBBATCH_READS", "3")
    pieces = rc.detect_batch(int_items + int_items[::-1])
    rc.ctx.set_option("STRQ_SERIAL", "1")
    serial = rc.detect_batch(int_items + int_items[::-1])
    assert whole == pieces == serial and all(set(g[6]) <= set("01") and g[0] > 0 for g in whole)


def test_empty_batch_and_tiny_reads(gpu_counter, want, pm, targets):
    """Empty batch, empty read, reads far shorter than the flank or the morphology window: nothing
    crashes, such reads come back as the reference's failed-gate row (n = 0, mod '-'), and the real
    reads that share the batch are untouched (bit-equal to the oracle)."""
    assert gpu_counter.detect_batch([]) == []
    rng = np.random.default_rng(8)
    good = _read(pm, targets, "c9orf72", "-", 3500, 9, 91)
    tiny = [rng.integers(300, 900, n).astype(np.int16) for n in (0, 1, 2, 3, 7, 8, 9, 20, 869, 870, 871)]
    items = [("c9orf72", t, "+") for t in tiny[:6]] + [("c9orf72", good, "-")] + [("fmr1", t, "-") for t in tiny[6:]]
    got = gpu_counter.detect_batch(items)
    assert len(got) == len(items)
    for (name, sig, strand), g in zip(items, got):
        if sig is good:
            assert tuple(g[:6]) == tuple(want("c9orf72", good, "-")[:6]) and g[0] == 9
        else:
            assert g[0] == 0 and g[6] == "-", (len(sig), g)
    # float64 input path as well
    got = gpu_counter.detect_batch([("c9orf72", np.zeros(0), "+"), ("c9orf72", np.array([80.0, 90.0, 100.0]), "+")])
    assert [g[0] for g in got] == [0, 0]


def test_float64_order_statistics_equal_numpy(pm, pm_mod, cfg, targets):
    """float64 reads have no exact histogram: their median, MAD and the two 'minmax' maps (STRique.py:142-143, 152-160, 590-592)
    come from a radix selection and from numpy's own summation tree on the GPU (cond_kernels.hip: f64_stats_kernel).  Bit for bit
    against scipy medfilt(3), np.median, np.mean(|x - median|), np.percentile([1, 99]) and the medians of the tails, at sizes
    around every blocking boundary of numpy's pairwise summation (8, 128, 8192) and with ties, zeros of both signs, two-valued
    and constant reads."""
    import warnings
    import scipy.signal
    from strique_amd.counter import repeatCounter
    rc = repeatCounter(pm, mod_model_file=pm_mod, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    for name, (repeat, prefix, suffix) in targets.items():
        rc.add_target(name, repeat, prefix, suffix)
    rng = np.random.default_rng(99)
    sizes = [1, 2, 3, 7, 8, 9, 100, 127, 128, 129, 1000, 8191, 8192, 8193, 8200, 16384, 16385, 24000, 40000, 100003, 284184]
    sigs = []
    for k, n in enumerate(sizes):
        if k % 3 == 0:
            s = rng.normal(90, 12, n)
        elif k % 3 == 1:
            s = np.round(rng.normal(90, 12, n) * 4) / 4            # many ties
        else:
            s = rng.normal(90, 12, n); s[rng.integers(0, n, max(1, n // 50))] = 300.0
        sigs.append(s)
    sigs.append(np.full(500, 42.0))                                 # constant: empty tails
    sigs.append(np.where(np.arange(9000) % 2, 1.0, 2.0))            # two values
    z = rng.normal(0, 1, 20000); z[rng.integers(0, 20000, 3000)] = 0.0; z[rng.integers(0, 20000, 3000)] = -0.0
    sigs.append(z)                                                  # negative values, zeros of both signs
    sigs.append(-np.abs(rng.normal(500, 100, 12345)))               # all negative

    def tails(x):
        q_lo, q_hi = np.percentile(x, [1, 99])
        m_lo = np.median(x[x < q_lo]); m_hi = np.median(x[x > q_hi])
        return m_lo + (m_hi - m_lo) / 2, (m_hi - m_lo) / 2
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s in sigs:
            rc.detect_batch([("c9orf72", s, "+")])
            sc = rc.ctx.debug_conditioning(0, 1)[2]
            flt = scipy.signal.medfilt(s, 3)
            med = np.median(flt)
            want = [med, np.mean(np.absolute(np.subtract(flt, med)))] + list(tails(flt)) + list(tails(s))
            got = [sc[0], sc[1], sc[2], sc[3], sc[6], sc[7]]
            assert np.array_equal(np.asarray(want), np.asarray(got), equal_nan=True), (len(s), want, got)
    # several reads in one batch: every read its own workgroups and its own scratch
    batch = [("c9orf72", s, "+") for s in sigs[8:16]]
    rc.detect_batch(batch)
    for i, (_, s, _) in enumerate(batch):
        flt = scipy.signal.medfilt(s, 3); med = np.median(flt)
        sc = rc.ctx.debug_conditioning(i, 1)[2]
        assert sc[0] == med and sc[1] == np.mean(np.absolute(np.subtract(flt, med))), (i, len(s))


def test_float64_reads_with_the_callers_own_statistics(gpu_counter, pm, targets):
    """strq_detect_batch(..., host_stats): a caller may hand over numpy's six scalars per float64 read (here: strq_host_stats, the
    host-side helper pinned against numpy on CPU); the rows equal those of the library's own statistics from the GPU."""
    from strique_amd import ffi
    rng = np.random.default_rng(5)
    sigs, tids = [], []
    for k in range(6):
        name = ["c9orf72", "fmr1"][k % 2]; strand = "+-"[(k // 2) % 2]
        sigs.append(_read(pm, targets, name, strand, int(rng.integers(3000, 9000)), int(rng.integers(5, 70)), 700 + k, as_int16=False))
        tids.append(gpu_counter._classifier_for(name, strand).target_id)
    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    flat = np.concatenate(sigs)
    own = gpu_counter.ctx.detect_batch(flat, off, tids)
    given = gpu_counter.ctx.detect_batch(flat, off, tids, host_stats=ffi.host_stats(flat, off))
    assert own.tobytes() == given.tobytes()
    assert (own["count"] > 0).sum() >= 5


def test_sub_batches_give_the_same_results(gpu_counter, want, pm, targets, monkeypatch):
    """A batch larger than one sub-batch is processed in pieces (strq_batch_run); the pieces must not
    see each other: results equal those of the one-piece run, in input order."""
    rng = np.random.default_rng(21)
    items = []
    for k in range(11):
        name = ["c9orf72", "fmr1"][k % 2]; strand = "+-"[(k // 2) % 2]
        items.append((name, _read(pm, targets, name, strand, int(rng.integers(2500, 6000)), int(rng.integers(4, 60)), 400 + k), strand))
    whole = gpu_counter.detect_batch(items)
    monkeypatch.setenv("STRQ_SUBBATCH_READS", "3")
    pieces = gpu_counter.detect_batch(items)
    assert pieces == whole
    # two sub-batches in flight (the default: a sub-batch's Viterbi launches run under the next one's flank alignments, rows are
    # taken one sub-batch late) against everything on one stream (STRQ_SERIAL, the order of rounds 1-5): the same rows
    monkeypatch.setenv("STRQ_SERIAL", "1")
    serial = gpu_counter.detect_batch(items)
    assert serial == whole
    monkeypatch.delenv("STRQ_SERIAL")
    # an oracle row at either end of the batch (the last sub-batch is the one still in flight when the run call returns)
    for i in (0, len(items) - 1):
        name, sig, strand = items[i]
        w = want(name, sig, strand)
        assert tuple(pieces[i][:6]) == tuple(w[:6]), (i, pieces[i], w)


def test_rows_of_a_range_while_the_next_is_in_flight(gpu_counter, pm, targets, monkeypatch):
    """strq_batch_run_range returns with the Viterbi launches of its last sub-batch queued; strq_batch_fetch_range waits only for
    the sub-batches that hold the rows asked for.  A resident batch of 12 reads run as four ranges, each fetched after the next has
    been queued (bench.py's step loop), then again in one call and serially: the same rows every way -- and fetching a range that
    was never run gives empty rows, not someone else's."""
    rng = np.random.default_rng(5)
    items = []
    for k in range(12):
        name = ["c9orf72", "fmr1"][k % 2]; strand = "+-"[(k // 2) % 2]
        items.append((name, _read(pm, targets, name, strand, int(rng.integers(2500, 7000)), int(rng.integers(4, 60)), 4400 + k), strand))
    ctx = gpu_counter.ctx
    sigs = [np.ascontiguousarray(it[1]) for it in items]
    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(x) for x in sigs])
    tids = [gpu_counter._classifier_for(it[0], it[2]).target_id for it in items]
    monkeypatch.setenv("STRQ_SUBBATCH_READS", "2")
    ctx.batch_upload(np.concatenate(sigs), off, tids)
    ctx.batch_run(); one_call = ctx.batch_fetch().copy()
    ctx.batch_upload(np.concatenate(sigs), off, tids)
    assert not ctx.batch_fetch_range(0, 12)["count"].any()          # nothing has run on this upload
    got = np.zeros_like(one_call); prev = None
    for lo in (0, 3, 6, 9):
        ctx.batch_run_range(lo, lo + 3)
        if prev is not None:
            got[prev:prev + 3] = ctx.batch_fetch_range(prev, prev + 3)
        prev = lo
    got[prev:prev + 3] = ctx.batch_fetch_range(prev, prev + 3)
    assert got.tobytes() == one_call.tobytes()
    monkeypatch.setenv("STRQ_SERIAL", "1")
    ctx.batch_upload(np.concatenate(sigs), off, tids)
    ctx.batch_run()
    assert ctx.batch_fetch().tobytes() == one_call.tobytes()
    assert (one_call["count"] > 0).sum() >= 10
    with pytest.raises(Exception):
        ctx.batch_fetch_range(5, 13)


def test_config1_ten_kb_thirty_repeats(gpu_counter, want, pm, targets):
    """configs[1]: 10 kb reads, 30 x GGGGCC (C9orf72), both strands."""
    items = [("c9orf72", _read(pm, targets, "c9orf72", st, 10000, 30, 500 + i), st) for i, st in enumerate("+-")]
    got = _check(gpu_counter, want, items)
    assert [g[0] for g in got] == [30, 30]


def test_config3_three_targets_mixed(gpu_counter, want, pm, targets, monkeypatch):
    """configs[3]: C9orf72 / FMR1(CGG) / HTT(CAG) targets from repeat_config.tsv mixed in one batch,
    both strands, n ~ U{30..1000} scaled to short reads; every field equals the oracle's."""
    rng = np.random.default_rng(33)
    items = []
    for k in range(12):
        name = ["c9orf72", "fmr1", "htt"][k % 3]; strand = "+-"[(k // 3) % 2]
        unit = len(targets[name][0])
        nt = int(rng.integers(4000, 8000))
        nrep = int(rng.integers(30, (nt - 2400) // unit))
        items.append((name, _read(pm, targets, name, strand, nt, nrep, 600 + k), strand))
    got = _check(gpu_counter, want, items)
    assert all(g[0] > 0 for g in got)
    # GGGGCC has a repeat profile of even length, CGG and CAG of odd length: both parities of the register-resident Viterbi
    # in ONE launch (round 3 ran such a sub-batch on the lane layout); the lane layout must agree record by record
    assert gpu_counter.ctx.last_viterbi_launches() == {"launches": 1, "register_resident": 1, "lane_layout": 0, "general": 0}
    monkeypatch.setenv("STRQ_VIT_NO_G2", "1")
    again = gpu_counter.detect_batch([(n, s, st) for n, s, st in items])
    assert gpu_counter.ctx.last_viterbi_launches()["register_resident"] == 0 and again == got


@pytest.mark.parametrize("name,nrep", [("c9orf72", 1000), ("fmr1", 1000), ("htt", 700)])
def test_config3_full_size_read_per_target(gpu_counter, want, pm, targets, name, nrep):
    """configs[2]/[3] at full size: one 50 kb read per target against the oracle, field by field."""
    strand = "-" if name == "fmr1" else "+"
    sig = _read(pm, targets, name, strand, 50000, nrep, 700 + len(name))
    got = _check(gpu_counter, want, [(name, sig, strand)])
    assert abs(got[0][0] - nrep) <= 2


def test_config4_modification_pass_full_size(pm, pm_mod, cfg, orc, opm, opm_mod, targets):
    """configs[4]: a 50 kb int16 read drawn from the mCpG model through the dual-HMM pass; the whole
    tuple, including the per-unit modification string, equals the oracle's."""
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    rc = repeatCounter(pm, mod_model_file=pm_mod, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rc.add_target("c9orf72", *targets["c9orf72"])
    sig_mod = synth.make_read(synth.KmerTable(pm_mod), 5, 1, 50000, targets["c9orf72"], 800, strand="+")[0]
    sig_base = synth.make_read(synth.KmerTable(pm), 5, 2, 50000, targets["c9orf72"], 300, strand="-")[0]
    got = rc.detect_batch([("c9orf72", sig_mod, "+"), ("c9orf72", sig_base, "-")])
    params = orc.align_params(cfg["align"])
    for g, sig, strand, planted in zip(got, (sig_mod, sig_base), "+-", (800, 300)):
        w, _ = orc.detect(sig, oracle_tc(orc, opm, targets, "c9orf72", strand, cfg["HMM"], opm_mod), opm, params, pm_mod=opm_mod)
        assert tuple(g) == tuple(w)
        assert abs(g[0] - planted) <= 2 and abs(len(g[6]) - g[0]) <= 3
    # the mCpG read is called mostly modified, the base read mostly unmodified
    assert got[0][6].count("1") > 0.8 * len(got[0][6]) and got[1][6].count("0") > 0.8 * len(got[1][6])


def test_modification_pass_hub_records_equal_backpointers(pm, pm_mod, cfg, targets, monkeypatch):
    """The modification model is decoded with hub records (one hop per repeat unit); the general
    back-pointer + traceback path (STRQ_MOD_BACKPOINTERS=1, also the fallback for models without the hub
    structure) must give the same strings."""
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    rc = repeatCounter(pm, mod_model_file=pm_mod, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    for name in ("c9orf72", "fmr1"):
        rc.add_target(name, *targets[name])
    items = []
    for i in range(6):
        name = ["c9orf72", "fmr1"][i % 2]; strand = "+-"[(i // 2) % 2]
        table = synth.KmerTable(pm_mod if i % 3 else pm)
        items.append((name, synth.make_read(table, 6, i, 5000 + 700 * i, targets[name], 12 + 9 * i, strand=strand)[0], strand))
    a = rc.detect_batch(items)
    monkeypatch.setenv("STRQ_MOD_BACKPOINTERS", "1")
    b = rc.detect_batch(items)
    assert a == b and all(set(x[6]) <= set("01") and len(x[6]) > 0 for x in a)


def test_mixed_read_lengths_run_as_length_classes(gpu_counter, want, pm, targets, monkeypatch):
    """Reads of very different lengths in one batch: the forward DP runs them as separate launches with 1, 2
    and 4 waves per alignment (STRQ_CLASS_MIN=1 keeps even tiny classes apart); every field equals the oracle's."""
    monkeypatch.setenv("STRQ_CLASS_MIN", "1")
    plan = [("c9orf72", "+", 3000, 12), ("fmr1", "-", 3500, 20), ("c9orf72", "-", 5000, 25), ("htt", "+", 6000, 40),
            ("c9orf72", "+", 12000, 60), ("fmr1", "+", 30000, 300), ("c9orf72", "-", 60000, 500)]
    items = [(name, _read(pm, targets, name, st, nt, nrep, 900 + i), st) for i, (name, st, nt, nrep) in enumerate(plan)]
    # with the upper-bound screen the long reads run over their windows only (one wave per window); the length classes are
    # what the whole-read passes are cut into
    got = _check(gpu_counter, want, items)
    assert gpu_counter.ctx.last_screen()["windowed"] >= 4
    monkeypatch.setenv("STRQ_NO_SCREEN", "1")
    got = _check(gpu_counter, want, items)
    assert [abs(g[0] - p[3]) <= 2 for g, p in zip(got, plan)] == [True] * len(plan)
    assert gpu_counter.ctx.last_timing()[7] >= 3            # at least three forward-DP launches


def test_odd_signals_against_the_oracle(gpu_counter, want, pm, targets):
    """Signals a sequencer would never produce on purpose: saturated samples, long constant stretches, ramps,
    spikes, pure noise, a locus cut in half -- whatever comes out (mostly failed gates), every field equals the
    oracle's."""
    rng = np.random.default_rng(4711)
    names = ["c9orf72", "fmr1", "htt"]
    items = []
    for k in range(24):
        name = names[k % 3]; strand = "+-"[(k // 3) % 2]
        n = int(rng.integers(900, 6000))
        kind = k % 8
        if kind == 0:
            sig = rng.integers(-32768, 32768, n)                                   # full-range noise
        elif kind == 1:
            sig = np.repeat(rng.integers(300, 900, n // 40 + 1), 40)[:n]            # long plateaus
        elif kind == 2:
            sig = np.linspace(200, 1000, n) + rng.normal(0, 3, n)                   # a ramp
        elif kind == 3:
            sig = rng.normal(600, 40, n); sig[rng.integers(0, n, 30)] = 32767; sig[rng.integers(0, n, 30)] = -32768      # saturated spikes
        elif kind == 4:
            sig = _read(pm, targets, name, strand, 4000, 15, 1200 + k).astype(np.float64); sig = sig[:len(sig) // 2]     # the locus cut in half
        elif kind == 5:
            sig = np.where(np.arange(n) % 2, 650, 640)                              # two alternating values
        elif kind == 6:
            base = _read(pm, targets, name, strand, 3500, 8, 1300 + k).astype(np.float64); sig = base[::-1]              # a read played backwards
        else:
            sig = _read(pm, targets, name, strand, 3000 + 100 * k, 5 + k, 1400 + k).astype(np.float64) * 1.7 - 300       # strongly rescaled
        items.append((name, np.clip(np.round(sig), -32768, 32767).astype(np.int16), strand))
    _check(gpu_counter, want, items)


def test_targets_with_other_flank_lengths(pm, cfg, orc, opm):
    """repeat_config.tsv rows are the user's: flanks shorter than the 50 nt the HMM takes (no trim), much longer
    than the bundled 150 nt (a flank of more than 960 samples runs as strips of 768 rows, up to 64 of them), a two-letter
    and a twelve-letter repeat; repeat units of 70 and 120 nt, whose HMMs exceed every lane layout and run on the general
    kernel (viterbi_csr_kernel).  Every field equals the oracle's on both strands."""
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    rng = np.random.default_rng(31337)
    nt = lambda n: "".join(rng.choice(list("ACGT"), n))
    custom = {"short": ("CAG", nt(30), nt(44)), "long": ("GGCCTG", nt(230), nt(201)), "uneven": ("CA", nt(64), nt(170)),
              "dodeca": ("CCCCGCCCCGCG", nt(120), nt(98)), "longer": ("CTG", nt(300), nt(415)), "longest": ("GAA", nt(1029), nt(163)),
              "vntr33": (nt(33), nt(150), nt(150)), "vntr48": (nt(48), nt(150), nt(150)),      # larger HMMs: other kernel shapes
              "kb3": ("CAG", nt(3000), nt(260)),                                               # 24 strips of 768 flank rows
              "vntr70": (nt(70), nt(150), nt(150)), "vntr120": (nt(120), nt(150), nt(150))}    # 600 / 800 emitting states: no lane layout, the general kernel
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    for name, t in custom.items():
        rc.add_target(name, *t)
    table = synth.KmerTable(pm)
    params = orc.align_params(cfg["align"])
    items = []
    for k, name in enumerate(custom):
        for strand in "+-":
            unit, pre, suf = custom[name]
            total = 6500 + 700 * k
            nrep = min(12 + 9 * k, (total - len(pre) - len(suf) - 2100) // len(unit))       # leave 2 kb of backbone around the locus
            sig = synth.make_read(table, 9, 7000 + 2 * k + (strand == "-"), total, custom[name], nrep, strand=strand)[0]
            items.append((name, sig, strand))
    got = rc.detect_batch(items)
    from conftest import oracle_map
    tcs = {(name, strand): oracle_tc(orc, opm, custom, name, strand, cfg["HMM"]) for name, _, strand in items}
    want = oracle_map(lambda it: orc.detect(it[1], tcs[(it[0], it[2])], opm, params)[0], items)
    for (name, sig, strand), g, w in zip(items, got, want):
        assert tuple(g[:6]) == tuple(w[:6]), (name, strand, g, w)
        assert g[0] > 0, (name, strand, g)
    with pytest.raises(Exception, match="flank shape"):
        rc.add_target("too_long", "CAG", nt(8198), nt(100))
    with pytest.raises(Exception, match="more than 4096 states"):      # beyond the LDS of the general kernel
        rc.add_target("vntr2000", nt(2000), nt(150), nt(150))


def test_degraded_reads_equal_the_oracle(gpu_counter, pm, targets, monkeypatch):
    """Reads synthesised with `realism` (strique_amd/synth.py): long-tailed dwell times, skipped k-mers, level jitter, baseline
    drift, spikes -- at realism 1 the flank scores of the real read the reference bundles (0.67 of the maximum), at 1.5 below it.
    Lower scores are what moves the column segments' overlap and sends alignments into the second forward round; whatever the
    heuristic does, all six fields equal the oracle's.  Sixteen 10 kb reads at both levels, both strands and targets, first at
    the initial overlap, then again after the library adapted to them, then with a short overlap pinned (second round for
    most of them: strq_last_second_round counts it); two 50 kb reads at realism 1."""
    import oracle_pool
    from strique_amd import synth
    table = synth.KmerTable(pm)
    items = []
    for i in range(16):
        name = ("c9orf72", "fmr1")[i % 2]; strand = "+-"[(i // 2) % 2]
        sig = synth.make_read(table, 41, i, 10000, targets[name], 20 + 3 * i, strand=strand, realism=(1.0, 1.5)[i // 8])[0]
        items.append((name, sig, strand))
    for i in range(2):
        items.append(("c9orf72", synth.make_read(table, 41, 100 + i, 50000, targets["c9orf72"], 400 + 300 * i, strand="+-"[i], realism=1.0)[0], "+-"[i]))
    want = oracle_pool.detect_many([(sig, strand, targets[name]) for name, sig, strand in items])
    first = gpu_counter.detect_batch(items)
    again = gpu_counter.detect_batch(items)
    monkeypatch.setenv("STRQ_OVERLAP", "1200"); monkeypatch.setenv("STRQ_SEG", "4")
    pinned = gpu_counter.detect_batch(items)
    redo, total = gpu_counter.ctx.last_second_round()
    assert total == 2 * len(items) and redo > 0
    for (name, sig, strand), w, a, b, c in zip(items, want, first, again, pinned):
        assert tuple(a[:6]) == tuple(w[:6]) and tuple(b[:6]) == tuple(w[:6]) and tuple(c[:6]) == tuple(w[:6]), (name, strand, w, a, b, c)


def test_empirical_noise_reads_equal_the_oracle(pm, cfg, targets):
    """Sixteen 50 kb reads whose dwell times, level offsets and sample residuals are resampled from the one real read the reference
    bundles (strique_amd.synth.EmpiricalNoise; docs/installation/test.md:15-16 is that read's row) -- flank scores at the real
    read's 0.67 ... 0.70 of the maximum, next to the background's own.  Through the default path (the coarse screen runs its first
    and second look on them and decides for itself whether to go on), with the coarse screen forced, with the fine screen only and
    without any screen: all six fields equal the oracle's every time."""
    import oracle_pool
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    table = synth.KmerTable(pm)
    noise = synth.EmpiricalNoise()
    items = []
    for i in range(16):
        strand = "+-"[i % 2]
        sig = synth.make_read(table, 7, 500 + i, 50000, targets["c9orf72"], (200, 500, 1000, 1500)[i % 4], strand=strand, noise=noise)[0]
        items.append(("c9orf72", sig, strand))
    want = oracle_pool.detect_many([(sig, strand, targets[name]) for name, sig, strand in items])
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rc.add_target("c9orf72", *targets["c9orf72"])
    runs = {}
    runs["default"] = rc.detect_batch(items); modes = [rc.ctx.last_screen()["mode"]]
    rc.ctx.set_option("STRQ_SCREEN_MODE", "coarse")
    runs["coarse"] = rc.detect_batch(items); modes.append(rc.ctx.last_screen()["mode"]); redo = [rc.ctx.last_screen()["second_look"] + rc.ctx.last_second_round()[0]]
    rc.ctx.set_option("STRQ_SCREEN_MODE", "fine")
    runs["fine"] = rc.detect_batch(items); modes.append(rc.ctx.last_screen()["mode"])
    rc.ctx.set_option("STRQ_NO_SCREEN", "1")
    runs["none"] = rc.detect_batch(items); modes.append(rc.ctx.last_screen()["mode"])
    rc.ctx.close()
    assert modes == ["coarse", "coarse", "fine", None], modes
    assert redo[0] >= 4, redo          # on such reads the coarse bound is loose: a good part of the alignments needs the second look
    for key, got in runs.items():
        for (name, sig, strand), w, a in zip(items, want, got):
            assert tuple(a[:6]) == tuple(w[:6]), (key, strand, w, a)


def test_a_hopeless_coarse_attempt_starts_over_with_the_fine_screen(pm, cfg, targets):
    """Forty empirical-noise reads (80 alignments: the pause rules apply from 64 on).  The coarse screen's first look certifies hardly
    any of them; instead of a second look over most of their columns the sub-batch starts over with the fine screen (align_core) and
    the coarse screen pauses.  The rows are those of the fine screen alone and of a run with that rule switched off
    (STRQ_SCREEN2_NO_BAIL), byte for byte, and the oracle's on six of them."""
    import oracle_pool
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    table = synth.KmerTable(pm)
    noise = synth.EmpiricalNoise()
    items = []
    for i in range(40):
        strand = "+-"[i % 2]
        sig = synth.make_read(table, 7, 9100 + i, 50000, targets["c9orf72"], (200, 500, 1000, 1500)[i % 4], strand=strand, noise=noise)[0]
        items.append(("c9orf72", sig, strand))
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rc.add_target("c9orf72", *targets["c9orf72"])
    got = rc.detect_batch(items); scr = rc.ctx.last_screen()
    assert scr["mode"] == "fine" and scr["coarse_pause"] >= 7 and scr["screened"] == 80, scr          # started over: the fine screen's figures only
    rc.ctx.set_option("STRQ_SCREEN_MODE", "fine")
    fine = rc.detect_batch(items)
    rc.ctx.close()
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rc.add_target("c9orf72", *targets["c9orf72"])
    rc.ctx.set_option("STRQ_SCREEN2_NO_BAIL", "1")
    through = rc.detect_batch(items); scr2 = rc.ctx.last_screen()
    rc.ctx.close()
    assert scr2["mode"] == "coarse" and scr2["coarse_pause"] >= 7, scr2
    assert got == fine == through
    want = oracle_pool.detect_many([(sig, strand, targets[name]) for name, sig, strand in items[:6]])
    for (name, sig, strand), w, a in zip(items[:6], want, got[:6]):
        assert tuple(a[:6]) == tuple(w[:6]), (strand, w, a)


def test_every_flank_length_of_the_fourteen_row_shape(pm, cfg, orc, opm, monkeypatch):
    """Flanks of 134 ... 154 nt (129 ... 149 k-mer classes, 774 ... 894 flank rows) all run at 14 rows per lane, and the last
    flank row sits in register (m - 1) % 14 of its lane -- 1, 3, ..., 13 over this range.  Round 3 knew that register at
    compile time for STRique's 870 rows only; now the steady-state loop of align_forward_seg_kernel is compiled once per
    register and picked by a scalar branch (forward_one, RMSW).  One target per flank length, prefix and suffix of different
    lengths, both strands, four waves per alignment on reads long enough for every piece to run full 64-step chunks: all
    six fields against the oracle, and the geometry the library reports."""
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    rng = np.random.default_rng(4141)
    nt = lambda n: "".join(rng.choice(list("ACGT"), n))
    lengths = list(range(134, 155))
    custom = {"f%d" % L: ("GGCCCC" if L % 2 else "CAG", nt(L), nt(lengths[(i * 7 + 3) % len(lengths)])) for i, L in enumerate(lengths)}
    monkeypatch.setenv("STRQ_SEG", "4"); monkeypatch.setenv("STRQ_CLASS_MIN", "1")
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    for name, t in custom.items():
        rc.add_target(name, *t)
    table = synth.KmerTable(pm)
    params = orc.align_params(cfg["align"])
    items = []
    for k, name in enumerate(custom):
        strand = "+-"[k % 2]
        sig = synth.make_read(table, 9, 8100 + k, 4200, custom[name], 20 + k, strand=strand)[0]      # ~31 k samples: pieces of ~8 k + overlap columns
        items.append((name, sig, strand))
    got = rc.detect_batch(items)
    geo = rc.ctx.last_geometry()
    assert geo["rows_per_lane"] == 14 and geo["waves_per_alignment"] == 4, geo
    import oracle_pool          # the oracle on a few worker processes (31 k samples x 2 flanks cost it ~0.5 s each)
    want = oracle_pool.detect_many([(sig, strand, custom[name]) for name, sig, strand in items])
    for (name, sig, strand), g, w in zip(items, got, want):
        assert tuple(g[:6]) == tuple(w[:6]), (name, strand, g, w)
        assert g[0] > 0, (name, strand, g)


@pytest.mark.parametrize("samples", [4, 8, 9, 12])
def test_other_samples_setting(pm, cfg, orc, opm, targets, samples):
    """`"samples"` in the `align` block of the JSON config changes the flank templates (STRique.py:562-565)."""
    from strique_amd.counter import repeatCounter
    acfg = dict(cfg["align"], samples=samples)
    rc = repeatCounter(pm, align_config=acfg, HMM_config=cfg["HMM"], device=0)
    rc.add_target("c9orf72", *targets["c9orf72"])
    params = orc.align_params(acfg)
    items = [("c9orf72", _read(pm, targets, "c9orf72", st, 7000 + 500 * k, 25 + 10 * k, 3100 + k), st) for k, st in enumerate("+-+")]
    got = rc.detect_batch(items)
    for (name, sig, strand), g in zip(items, got):
        tc = orc.classifier(*targets[name], strand, opm, None, cfg["HMM"], samples=samples)
        w = orc.detect(sig, tc, opm, params)[0]
        assert tuple(g[:6]) == tuple(w[:6]), (samples, strand, g, w)
    rc.ctx.close()


def test_randomised_configurations(pm, pm_mod, cfg, orc, opm, opm_mod):
    """Sixteen random configurations -- gap parameters (collapsed and general affine), dist_offset / dist_min,
    `samples`, every HMM probability and std scale, flank lengths, repeat units, with and without the modification
    model -- three reads each (both strands, int16 and float64): the whole tuple equals the oracle's."""
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    rng = np.random.default_rng(20261002)
    nt = lambda n: "".join(rng.choice(list("ACGT"), n))
    for trial in range(16):
        collapsed = trial % 2 == 0
        gh = -float(rng.integers(1, 4)); gv = -float(rng.integers(4, 20))
        acfg = dict(gap_open_h=gh, gap_extension_h=gh if collapsed else -float(rng.integers(1, 4)),
                    gap_open_v=gv, gap_extension_v=gv if collapsed else -float(rng.integers(4, 20)),
                    dist_offset=float(rng.choice([8.0, 12.0, 16.0, 20.0])), dist_min=float(rng.choice([0.0, 0.0, -1.0, -4.0])),
                    samples=int(rng.choice([6, 6, 6, 5, 8, 12])))
        hcfg = dict(cfg["HMM"])
        for key in ("match_loop", "match_match", "match_insert", "match_delete"):
            hcfg[key] = float(hcfg[key] * rng.uniform(0.6, 1.4))
        hcfg.update(leave_repeat=float(rng.uniform(0.0005, 0.01)), e1_ratio=float(rng.uniform(0.05, 0.5)),
                    seq_std_scale=float(rng.uniform(0.8, 1.5)), rep_std_scale=float(rng.uniform(0.8, 1.6)),
                    rep_std_offset=float(rng.choice([0.0, 0.1])), delete_delete=float(rng.uniform(0.001, 0.05)))
        with_mod = trial % 4 == 3
        unit = [nt(int(rng.integers(2, 9))), "CGG", "GGCCCC", "CAG"][trial % 4]
        if with_mod:
            unit = "GGCCCC"                                    # CpG in every unit
        target = (unit, nt(int(rng.integers(40, 260))), nt(int(rng.integers(40, 260))))
        rc = repeatCounter(pm, mod_model_file=pm_mod if with_mod else None, align_config=acfg, HMM_config=hcfg, device=0)
        rc.add_target("t", *target)
        table = synth.KmerTable(pm_mod if with_mod and trial % 8 == 7 else pm)
        params = orc.align_params(acfg)
        items = []
        for k in range(3):
            strand = "+-"[(trial + k) % 2]
            sig = synth.make_read(table, 11, 100 * trial + k, int(rng.integers(3000, 9000)), target, int(rng.integers(5, 60)),
                                  strand=strand, as_int16=(k != 1))[0]
            items.append(("t", sig, strand))
        got = rc.detect_batch(items)
        for (name, sig, strand), g in zip(items, got):
            tc = orc.classifier(*target, strand, opm, opm_mod if with_mod else None, hcfg, samples=acfg["samples"])
            w = orc.detect(sig, tc, opm, params, pm_mod=opm_mod if with_mod else None)[0]
            assert tuple(g) == tuple(w), (trial, acfg, strand, g, w)
        rc.ctx.close()


def test_overlap_chosen_from_the_previous_batch_changes_nothing(pm, cfg, targets, monkeypatch):
    """The column segments of a sub-batch are cut with the overlap that was cheapest for the previous sub-batch's
    scores: the first call of a context runs at the initial 8192 columns, the second at the adapted one, a third context
    always at the worst case (STRQ_OVERLAP=0) -- the rows must not differ, whatever the read contains."""
    from strique_amd.counter import repeatCounter
    items = []
    for k in range(48):
        name = ["c9orf72", "fmr1", "htt"][k % 3]; strand = "+-"[(k // 3) % 2]
        if k % 8 == 7:      # no locus at all: these alignments never certify a short overlap and run twice
            rng = np.random.default_rng(k)
            items.append((name, rng.normal(600, 80, 90000).astype(np.int16), strand))
        else:
            items.append((name, _read(pm, targets, name, strand, 11000 + 500 * (k % 5), 20 + k, 5200 + k), strand))

    def fresh():
        rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
        for name, t in targets.items():
            rc.add_target(name, *t)
        return rc
    rc = fresh()
    first = rc.detect_batch(items)
    second = rc.detect_batch(items)          # overlap adapted to the first call's scores
    third = rc.detect_batch(list(reversed(items)))[::-1]
    rc.ctx.close()
    monkeypatch.setenv("STRQ_OVERLAP", "0")
    rc = fresh()
    worst = rc.detect_batch(items)
    rc.ctx.close()
    assert first == second == third == worst
    assert sum(1 for r in first if r[0] > 0) >= 40


def test_filtered_signal_without_tails_is_decoded_as_missing_values(gpu_counter, want, orc, opm, pm, pm_mod, cfg, targets, opm_mod):
    """A read whose lowest and highest 1 % of filtered samples sit on two saturated values (runs of three samples on a
    floor / ceiling: they survive the median filter, the 1 x 8 opening / closing removes them): np.percentile puts the
    1st / 99th percentile ON those values, the strict tails are empty, normalize2model returns NaNs for the filtered
    signal (STRique.py:155-160,597) -- while the morphology signal normalises, the flanks are found and the gate passes.
    The reference then decodes a window of NaNs (STRique.py:603), which pomegranate scores as missing values."""
    from strique_amd.counter import repeatCounter
    rng = np.random.default_rng(5)
    items = []
    for k, (name, strand) in enumerate((("c9orf72", "+"), ("fmr1", "-"), ("c9orf72", "-"))):
        s = _read(pm, targets, name, strand, 6000 + 500 * k, 20 + 5 * k, 4242 + k).copy()
        floor, ceil_ = int(s.min()) - 40, int(s.max()) + 40
        n = len(s)
        for pos in rng.choice(np.arange(10, n - 10, 12), size=n // 100, replace=False):
            s[pos:pos + 3] = floor
        for pos in rng.choice(np.arange(16, n - 10, 12), size=n // 100, replace=False):
            s[pos:pos + 3] = ceil_
        flt, u8, morph, fltn = orc.condition(s, opm)
        assert np.isnan(fltn).all() and np.isfinite(morph).all()
        items.append((name, s, strand))
        items.append((name, s.astype(np.float64) * 0.17 + 3.0, strand))          # the float64 input path as well
    got = _check(gpu_counter, want, [it for it in items if it[1].dtype == np.int16])
    got += _check(gpu_counter, want, [it for it in items if it[1].dtype != np.int16])
    assert all(g[0] > 0 and g[3] < 0 for g in got), got          # a path exists: count and log-probability of the transitions alone
    # with the modification model: the second Viterbi runs on the (finite) raw normalisation of the marked stretch
    rc = repeatCounter(pm, mod_model_file=pm_mod, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rc.add_target("c9orf72", *targets["c9orf72"])
    name, s, strand = items[0]
    g = rc.detect(name, s, strand)
    w, _ = orc.detect(s, oracle_tc(orc, opm, targets, name, strand, cfg["HMM"], opm_mod), opm, orc.align_params(cfg["align"]), pm_mod=opm_mod)
    assert tuple(g) == tuple(w), (g, w)
    rc.ctx.close()



def test_large_repeat_unit_with_the_modification_model(pm, pm_mod, cfg, orc, opm, opm_mod):
    """A 70-nt repeat unit with CpGs: the flanked HMM (600 emitting states) runs on the general kernel in MARK mode
    (the stretch decoded as `repeat*` is carried along the best path there as well), the dual base / mCpG model of
    such a unit on the lane kernels with back-pointers or hub records.  Whole tuple, pattern string included."""
    from strique_amd import synth
    from strique_amd.counter import repeatCounter
    rng = np.random.default_rng(707)
    unit = "".join(rng.choice(list("ACGT"), 70))
    unit = unit[:10] + "CG" + unit[12:40] + "CG" + unit[42:]
    target = (unit, "".join(rng.choice(list("ACGT"), 150)), "".join(rng.choice(list("ACGT"), 150)))
    rc = repeatCounter(pm, mod_model_file=pm_mod, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rc.add_target("big", *target)
    params = orc.align_params(cfg["align"])
    items = []
    for k, (table, strand) in enumerate(((synth.KmerTable(pm), "+"), (synth.KmerTable(pm_mod), "-"))):
        items.append(("big", synth.make_read(table, 12, 70 + k, 9000, target, 40 + 15 * k, strand=strand)[0], strand))
    got = rc.detect_batch(items)
    for (name, sig, strand), g in zip(items, got):
        tc = orc.classifier(*target, strand, opm, opm_mod, cfg["HMM"])
        w = orc.detect(sig, tc, opm, params, pm_mod=opm_mod)[0]
        assert tuple(g) == tuple(w), (strand, g, w)
        assert g[0] > 0 and set(g[6]) <= set("01") and len(g[6]) > 0
    rc.ctx.close()


def test_overlap_counters_say_what_ran_beside_what(gpu_counter, pm, targets, monkeypatch):
    """strq_last_overlap: with two sub-batches in flight the Viterbi launches of a sub-batch lie under the alignment stage of the one that
    follows (counted when the following run call takes the rows); serially nothing overlaps; either way the rows are the same."""
    items = []
    for k in range(8):
        strand = "+-"[k % 2]
        items.append(("c9orf72", _read(pm, targets, "c9orf72", strand, 30000, 150 + 40 * k, 5200 + k), strand))
    ctx = gpu_counter.ctx
    sigs = [np.ascontiguousarray(it[1]) for it in items]
    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(x) for x in sigs])
    tids = [gpu_counter._classifier_for(it[0], it[2]).target_id for it in items]
    monkeypatch.setenv("STRQ_SUBBATCH_READS", "4")
    ctx.batch_upload(np.concatenate(sigs), off, tids)
    ctx.batch_run()
    ov = ctx.last_overlap()          # the first sub-batch's rows were taken by the second one's run; the second one is still in flight
    assert ov["sub_batches"] == 1 and ov["viterbi_ms"] > 0 and 0 < ov["under_alignment_stage_ms"] <= ov["viterbi_ms"] * 1.001, ov
    rows = ctx.batch_fetch().copy()
    assert ctx.last_overlap()["viterbi_ms"] > ov["viterbi_ms"]          # ... and taken by the fetch: alone, not counted as overlapped
    assert ctx.last_overlap()["sub_batches"] == 1
    monkeypatch.setenv("STRQ_SERIAL", "1")
    ctx.batch_run()
    ov = ctx.last_overlap()
    assert ov["sub_batches"] == 0 and ov["under_alignment_stage_ms"] == 0 and ov["viterbi_ms"] > 0, ov
    assert ctx.batch_fetch().tobytes() == rows.tobytes() and (rows["count"] > 100).all()
