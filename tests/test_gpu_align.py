"""HIP flank alignment vs the CPU oracle through the C ABI: bit-exact score, identical path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from strique_amd import ffi
    return ffi.Context(0)


def _toy(rng, n, k=145, s=6, scale=0.45):
    cls = rng.uniform(60, 120, k).astype(np.float32)
    flank = np.repeat(cls, s)
    lval = (40 + scale * np.arange(256)).astype(np.float32)
    lv = np.repeat(rng.integers(30, 200, n // 5 + 1), rng.integers(3, 10, n // 5 + 1))[:n].astype(np.uint8)
    emb = np.repeat(np.clip(np.round((cls - 40) / scale), 0, 255).astype(np.uint8), rng.integers(6, 10, k))
    pos = int(rng.integers(0, max(1, n - len(emb))))
    emb = emb[:max(0, n - pos)]
    lv[pos:pos + len(emb)] = emb
    return lv, lval, flank


def _same(o, g):
    assert np.float32(o[0]).tobytes() == np.float32(g[0]).tobytes()
    assert o[4] == g[4] and o[5] == g[5]
    assert np.array_equal(o[3], g[3])
    if o[1] is not None and g[1] is not None:
        assert np.array_equal(o[1], g[1]) and np.array_equal(o[2], g[2])


@pytest.mark.parametrize("n", [1, 2, 10, 63, 127, 128, 129, 511, 512, 513, 1000, 1023, 1024, 1025, 2049, 20001])
def test_align_overlap_lengths(ctx, orc, n):
    """Includes reads shorter than the flank (vertical run at column 0), the 64-step wavefront
    boundaries and the 1024-column checkpoint boundaries."""
    rng = np.random.default_rng(100 + n)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    lv, lval, flank = _toy(rng, n)
    a = lval[lv]
    _same(orc.align_overlap(a, flank, params), ctx.align_overlap(a, flank))


@pytest.mark.parametrize("params", [[-2, -8, -2, -8, 8, -16], [-3, -1, -20, -4, 16, 0], [-1, -1, -12, -16, 16, -2],
                                    [-2, -1, -16, -16, 16, 0]])
def test_general_affine_parameters(ctx, orc, params):
    """open != extend in one or both directions runs the non-collapsed kernels; the first row is
    align_raw's own defaults (src/align_raw.h:51-60)."""
    rng = np.random.default_rng(7)
    p = np.array(params, np.float32)
    ctx.set_align_params(*[float(v) for v in p])
    for n in (300, 5000):
        lv, lval, flank = _toy(rng, n)
        a = lval[lv]
        _same(orc.align_overlap(a, flank, p), ctx.align_overlap(a, flank))
    ctx.set_align_params(*[float(v) for v in orc.align_params(None)])


@pytest.mark.parametrize("k", [1, 40, 64, 100, 128, 129, 140, 145, 149, 150, 158])
def test_flank_shapes(ctx, orc, k):
    """6 / 7 / 8 / 12 / 14 / 15 rows per lane (k = 129 ... 149 classes: 14 or 15, whichever keeps the last flank row in a
    register known at compile time; STRique's 145-class flanks are the k = 145 case at 14), last flank row in an
    interior register or not."""
    rng = np.random.default_rng(k)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    lv, lval, flank = _toy(rng, 4000, k=k)
    a = lval[lv]
    _same(orc.align_overlap(a, flank, params), ctx.align_overlap(a, flank))


def test_wide_bands_and_table_classes(ctx, orc):
    """Level spacing decides the band width: 64-, 128- and full-width table classes."""
    rng = np.random.default_rng(11)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    for scale in (0.45, 0.2, 0.05):
        lv, lval, flank = _toy(rng, 3000, scale=scale)
        a = lval[lv]
        _same(orc.align_overlap(a, flank, params, want_idx=False), ctx.align_overlap(a, flank, want_idx=False))


def test_batch_ragged(ctx, orc):
    rng = np.random.default_rng(5)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    reads, lvals, flanks = [], [], []
    for n in (50, 700, 1300, 9000, 2500, 64, 4097):
        lv, lval, flank = _toy(rng, n)
        reads.append(lv); lvals.append(lval); flanks.append(flank)
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    foff = np.concatenate([[0], np.cumsum([len(f) for f in flanks])])
    sc, je, j0, rec = ctx.align_batch(np.concatenate(reads), off, np.stack(lvals), np.arange(len(reads)), np.concatenate(flanks), foff)
    for i in range(len(reads)):
        o = orc.align_overlap(lvals[i][reads[i]], flanks[i], params, want_idx=False)
        assert np.float32(o[0]).tobytes() == sc[i].tobytes() and o[4] == je[i] and o[5] == j0[i]
        assert np.array_equal(o[3], rec[foff[i]:foff[i + 1]])
    assert ctx.align_batch(np.zeros(0, np.uint8), [0], np.zeros((0, 256), np.float32), [], np.zeros(0, np.float32), [0])[0].size == 0


@pytest.mark.parametrize("seg", [None, "4", "2"])
def test_full_size_read(ctx, orc, monkeypatch, seg):
    """BASELINE config 3 size: N = 375 000 columns.  One alignment does not fill the chip and would run on one
    wave; STRQ_SEG=4 puts it on the benchmark's geometry (four waves sharing the score table, column segments)."""
    if seg:
        monkeypatch.setenv("STRQ_SEG", seg)
    rng = np.random.default_rng(9)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    lv, lval, flank = _toy(rng, 375000)
    a = lval[lv]
    _same(orc.align_overlap(a, flank, params, want_idx=False), ctx.align_overlap(a, flank, want_idx=False))
    if seg:
        g = ctx.last_geometry()
        assert g["waves_per_alignment"] == int(seg) and g["rows_per_lane"] in (14, 15) and g["wpe"] >= 3, g


@pytest.mark.parametrize("segs", [2, 4])
@pytest.mark.parametrize("kind", ["permuted", "fine"])
def test_large_tables_with_several_waves_per_alignment(ctx, orc, monkeypatch, segs, kind):
    """Score tables that leave room for only one or two per CU -- a host-rebuilt full-width table (148 KB: level
    values that are not monotone) or wide bands from a fine level spacing -- under column segments: such launches
    have fewer resident waves than any compiled waves-per-SIMD cap and must still run (and equal the oracle)."""
    monkeypatch.setenv("STRQ_SEG", str(segs))
    rng = np.random.default_rng(500 + segs)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    if kind == "permuted":
        lv, lval, flank = _toy(rng, 100000)
        perm = rng.permutation(256)
        lval_p = np.empty_like(lval); lval_p[perm] = lval
        lv, lval = perm[lv].astype(np.uint8), lval_p
    else:
        # every class within 10 pA of every level: all bands are full width (145 x 256 floats)
        n, k = 100000, 145
        cls = rng.uniform(84, 90, k).astype(np.float32)
        flank = np.repeat(cls, 6)
        lval = (80 + 0.05 * np.arange(256)).astype(np.float32)
        lv = np.repeat(rng.integers(0, 256, n // 5 + 1), rng.integers(3, 10, n // 5 + 1))[:n].astype(np.uint8)
        emb = np.repeat(np.clip(np.round((cls - 80) / 0.05), 0, 255).astype(np.uint8), rng.integers(6, 10, k))
        pos = int(rng.integers(0, n - len(emb)))
        lv[pos:pos + len(emb)] = emb
    off = np.array([0, len(lv)]); foff = np.array([0, len(flank)])
    want = orc.align_overlap(lval[lv], flank, params, want_idx=False)
    sc, je, j0, rec = ctx.align_batch(lv, off, lval[None, :], [0], flank, foff)
    g = ctx.last_geometry()
    assert g["waves_per_alignment"] == segs and g["tables_per_cu"] <= 2, g
    assert np.float32(want[0]).tobytes() == sc[0].tobytes() and want[4] == je[0] and want[5] == j0[0]
    assert np.array_equal(want[3], rec)


def test_dist_min_above_dist_offset_keeps_segments_exact(ctx, orc, monkeypatch):
    """`dist_min` > `dist_offset` (a user's `align` block): every cell scores dist_min, so a path gains dist_min per
    diagonal step, not dist_offset -- the span bound the pieces are cut with must use the larger of the two."""
    monkeypatch.setenv("STRQ_SEG", "4")
    rng = np.random.default_rng(77)
    params = np.array([-1, -1, -16, -16, 4, 6], np.float32)
    ctx.set_align_params(*[float(v) for v in params])
    try:
        lv, lval, flank = _toy(rng, 30000, k=40)
        a = lval[lv]
        _same(orc.align_overlap(a, flank, params, want_idx=False), ctx.align_overlap(a, flank, want_idx=False))
    finally:
        ctx.set_align_params(*[float(v) for v in orc.align_params(None)])


def test_formerly_unsupported_inputs_now_match_the_oracle(ctx, orc):
    """More than 256 distinct values in `a`, a flank without runs of 6: the generic kernel covers them
    (reference src/pyalign.cpp:59-61 takes any two float lists); bad arguments still raise."""
    from strique_amd import ffi
    rng = np.random.default_rng(1)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    a = rng.normal(90, 10, 5000); b = np.repeat(rng.uniform(60, 120, 10), 6)
    _same(orc.align_overlap(a, b, params), ctx.align_overlap(a, b))
    a = np.ones(100); b = rng.uniform(60, 120, 61)
    _same(orc.align_overlap(a, b, params), ctx.align_overlap(a, b))
    with pytest.raises(ffi.StriqueHipError) as e:
        ctx.align_overlap(a, np.zeros(0))                      # an empty flank is not an alignment
    assert e.value.code == ffi.STRQ_ERR_ARG


def test_two_strip_path(orc, monkeypatch):
    """Flank rows cut into two strips (bottom row streamed through HBM, trace crossing the strip
    boundary): same results as the oracle.  Forced through STRQ_STRIPS=2."""
    from strique_amd import ffi
    monkeypatch.setenv("STRQ_STRIPS", "2")
    ctx2 = ffi.Context(0)
    rng = np.random.default_rng(21)
    for params in ([-1, -1, -16, -16, 16, 0], [-2, -8, -2, -8, 8, -16]):
        p = np.array(params, np.float32)
        ctx2.set_align_params(*[float(v) for v in p])
        for n, k in ((40, 145), (3000, 145), (20011, 145), (5000, 100), (2500, 158)):
            lv, lval, flank = _toy(rng, n, k=k)
            a = lval[lv]
            _same(orc.align_overlap(a, flank, p), ctx2.align_overlap(a, flank))
    ctx2.close()


def test_non_monotone_level_values_take_the_host_path(ctx, orc):
    """The table kernel finds each class's band by bisection, which needs level values that do not
    decrease with the level; any other level -> value map is handed to the host (full-width tables).
    Same alignment, levels relabelled by a random permutation: identical results."""
    rng = np.random.default_rng(17)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    lv, lval, flank = _toy(rng, 2500)
    perm = rng.permutation(256)                      # new label of old level q is perm[q]
    lval_p = np.empty_like(lval); lval_p[perm] = lval
    lv_p = perm[lv].astype(np.uint8)
    assert np.array_equal(lval_p[lv_p], lval[lv])
    off = np.array([0, len(lv)]); foff = np.array([0, len(flank)])
    want = orc.align_overlap(lval[lv], flank, params, want_idx=False)
    for levels, table in ((lv, lval), (lv_p, lval_p)):
        sc, je, j0, rec = ctx.align_batch(levels, off, table[None, :], [0], flank, foff)
        assert np.float32(want[0]).tobytes() == sc[0].tobytes() and want[4] == je[0] and want[5] == j0[0]
        assert np.array_equal(want[3], rec)


def test_randomised_sweep(ctx, orc):
    """Seeded random sweep over read length, flank size, level spacing and gap / distance parameters
    (collapsed and general affine recurrences): score bits, end column, start column and the per-row
    record must equal the oracle's every time."""
    rng = np.random.default_rng(2026)
    ks = [1, 2, 7, 33, 64, 65, 96, 127, 128, 145, 158]
    for it in range(48):
        k = int(rng.choice(ks)); n = int(rng.integers(1, 6000)); scale = float(rng.choice([0.45, 0.3, 0.12]))
        if it % 3 == 0:
            e_h = -float(rng.integers(1, 4)); e_v = -float(rng.integers(2, 20)); params = [e_h, e_h, e_v, e_v]
        else:
            params = [-float(rng.integers(1, 6)), -float(rng.integers(1, 4)), -float(rng.integers(4, 24)), -float(rng.integers(1, 18))]
        params += [float(rng.choice([8.0, 16.0, 12.5])), float(rng.choice([0.0, -2.0, -16.0]))]
        ctx.set_align_params(*params)
        lv, lval, flank = _toy(rng, n, k=k, scale=scale)
        a = lval[lv]
        _same(orc.align_overlap(a, flank, np.array(params, np.float32), want_idx=True), ctx.align_overlap(a, flank, want_idx=True))


def _overlap(m, params):
    # align_segment_overlap (align_kernels.hip): longest column span of a path that scores >= 0
    open_h, ext_h, open_v, ext_v, off, dmin = [float(v) for v in params]
    return int(m + m * off / -max(open_h, ext_h) * 1.01 + 64.0) + 1


def _seams(n, segs, ov):
    use = segs
    while use > 1 and n < (use + 1) * ov:
        use -= 1
    if use < 2:
        return [], []
    ln = (n + (use - 1) * ov + use - 1) // use
    own_end = [ln + j * (ln - ov) for j in range(use - 1)]            # last column owned by piece j
    return own_end, [e - ov for e in own_end]                         # ... and the cold-start column of piece j + 1


@pytest.mark.parametrize("segs", [2, 4])
@pytest.mark.parametrize("k,n,ov_env", [(40, 30000, None), (145, 100000, None), (145, 100000, "0"), (145, 60000, "1500")])
def test_column_segments_seam_adversarial(ctx, orc, monkeypatch, segs, k, n, ov_env):
    """Several waves per alignment (column segments with a cold-started overlap): the flank
    is planted so that its best path ends exactly on / next to every piece boundary, starts exactly at a
    piece's cold-start column, lies inside an overlap zone, or occurs twice with identical samples in
    two different pieces (tie: the leftmost must win).  Score bits, end column, start column and the
    whole path must equal the single-matrix oracle's.  The pieces are first cut with a short overlap
    (STRQ_OVERLAP, default 8192 columns) and alignments whose best score does not certify it run again
    with the worst-case overlap: "0" forces the worst case everywhere, "1500" sends the weak alignments
    through the second round."""
    rng = np.random.default_rng(1000 * segs + k)
    params = orc.align_params(None)
    ctx.set_align_params(*[float(v) for v in params])
    monkeypatch.setenv("STRQ_SEG", str(segs))
    if ov_env is not None:
        monkeypatch.setenv("STRQ_OVERLAP", ov_env)
    scale = 0.45
    cls = rng.uniform(60, 120, k).astype(np.float32)
    flank = np.repeat(cls, 6)
    m = len(flank)
    lval = (40 + scale * np.arange(256)).astype(np.float32)
    ov = _overlap(m, params)
    ov_fast = ov if ov_env == "0" else min(ov, int(ov_env) if ov_env else 8192)
    emb = np.repeat(np.clip(np.round((cls - 40) / scale), 0, 255).astype(np.uint8), rng.integers(6, 10, k))
    w = len(emb)

    def background():
        return np.repeat(rng.integers(30, 200, n // 5 + 1), rng.integers(3, 10, n // 5 + 1))[:n].astype(np.uint8)

    cases = []
    for o in sorted({ov, ov_fast}):
        own_end, cold = _seams(n, segs, o)
        for e, c0 in zip(own_end, cold):
            for end in (e - 1, e, e + 1, e + 2, e + w // 2):               # path ends around the boundary
                cases.append([end - w])
            for start in (c0 - 1, c0, c0 + 1, c0 + o // 2):                # path starts at the cold column / inside the overlap
                cases.append([start])
            cases.append([c0 - w - 50, e + 50])                            # identical occurrences left and right of the seam
            cases.append([e + 50, c0 - w - 50])
    cases.append([])                                                       # no occurrence at all: a weak best score
    cases.append([])
    assert len(cases) > 4
    levels, offs = [], [0]
    for plant in cases:
        lv = background()
        for p in plant:
            p = max(0, min(n - w, p))
            lv[p:p + w] = emb
        levels.append(lv); offs.append(offs[-1] + n)
    na = len(cases)
    got = ctx.align_batch(np.concatenate(levels), np.array(offs, np.int64), np.tile(lval, (na, 1)), np.arange(na, dtype=np.int32),
                          np.tile(flank, na), np.arange(na + 1, dtype=np.int64) * m)
    assert ctx.last_timing()[7] >= 1
    from conftest import oracle_map
    want = oracle_map(lambda lv: orc.align_overlap(lval[lv], flank, params, want_idx=False), levels)
    for i, o in enumerate(want):
        assert np.float32(o[0]).tobytes() == np.float32(got[0][i]).tobytes(), (i, cases[i])
        assert (o[4], o[5]) == (int(got[1][i]), int(got[2][i])), (i, cases[i])
        assert np.array_equal(o[3], got[3][i * m:(i + 1) * m]), (i, cases[i])


@pytest.mark.parametrize("k", [159, 160, 200, 256, 257, 300, 384, 385, 512, 1024, 1025, 2500])
@pytest.mark.parametrize("params", [None, [-3, -1, -20, -4, 16, 0]])
def test_long_flanks_run_as_strips_with_their_own_tables(ctx, orc, k, params):
    """Flanks of more than 158 k-mer classes (948 samples): strips of 768 rows, each with its own score table,
    the boundary row handed on through HBM, the traceback climbing from strip to strip.  STRique's collapsed
    parameters and a general affine set; reads that contain the flank, and one shorter than it."""
    rng = np.random.default_rng(900 + k)
    p = orc.align_params(None) if params is None else np.array(params, np.float32)
    ctx.set_align_params(*[float(v) for v in p])
    try:
        for n in (20000, 6 * k - 100):
            lv, lval, flank = _toy(rng, n, k=k)
            a = lval[lv]
            _same(orc.align_overlap(a, flank, p), ctx.align_overlap(a, flank))
    finally:
        ctx.set_align_params(*[float(v) for v in orc.align_params(None)])


def test_flank_beyond_the_strip_limit_takes_the_generic_kernel(ctx, orc):
    rng = np.random.default_rng(4)
    params = orc.align_params(None)
    lv, lval, flank = _toy(rng, 9000, k=8193)
    a = lval[lv]
    _same(orc.align_overlap(a, flank, params), ctx.align_overlap(a, flank))


@pytest.mark.parametrize("s", [1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 22, 25, 26])
@pytest.mark.parametrize("params", [None, [-3, -1, -20, -4, 16, 0]])
def test_other_samples_per_kmer(ctx, orc, s, params):
    """`samples` of the JSON `align` block (STRique.py:513,529) other than 6: the kernels work on the largest
    compiled run length that divides it (6 for multiples of 6; 1 ... 5, 7 ... 11, 13), two classes per lane, strips of
    128 classes."""
    rng = np.random.default_rng(1200 + s)
    p = orc.align_params(None) if params is None else np.array(params, np.float32)
    ctx.set_align_params(*[float(v) for v in p])
    try:
        for k, n in ((20, 3000), (145 if s > 1 else 100, 12000), (7, 5)):
            if s == 17 and k * s > 1024:      # a prime above 13 runs as 17 x 1: 1024 rows at most
                k = 1024 // s
            lv, lval, flank = _toy(rng, n, k=k, s=s)
            a = lval[lv]
            _same(orc.align_overlap(a, flank, p), ctx.align_overlap(a, flank))
    finally:
        ctx.set_align_params(*[float(v) for v in orc.align_params(None)])
