"""The register-resident layout of a profile chain (strq_model_set_positions / VitG2, strique_amd/csrc/viterbi_kernels.h),
checked without a GPU: the tables the library would upload (host-only debug export) drive a plain-Python restatement of
viterbi_g2_kernel's time step, whose log-probability and count must equal the oracle's bit for bit -- on STRique's
flanked models (scripts/STRique.py:384-431) with even and odd repeat profiles, missing observations included."""
import ctypes
import os

import numpy as np
import pytest

ROW_ME, ROW_MO, ROW_IE, ROW_IO, ROW_DE, ROW_DO, ROW_CH, ROWS = 0, 7, 13, 16, 19, 22, 24, 26


def _layout(lib, bk, pos_kind=None, pos_index=None):
    lp = np.zeros(ROWS * 64); own = np.zeros(6 * 64, np.int32); meta = np.zeros(10, np.int32); why = ctypes.create_string_buffer(256)
    c = lambda a, t: np.ascontiguousarray(a, t)
    arrs = [c(bk.in_ptr, np.int32), c(bk.in_src, np.int32), c(bk.in_logp, np.float64), c(bk.emis_kind, np.int32), c(bk.count_inc, np.int32),
            c(bk.pos_kind if pos_kind is None else pos_kind, np.int32), c(bk.pos_index if pos_index is None else pos_index, np.int32)]
    rc = lib.strq_debug_g2_layout(ctypes.c_int32(bk.n_states), ctypes.c_int32(bk.silent_start), ctypes.c_int32(bk.start), ctypes.c_int32(bk.end),
                                  *[a.ctypes.data_as(ctypes.c_void_p) for a in arrs], lp.ctypes.data_as(ctypes.c_void_p),
                                  own.ctypes.data_as(ctypes.c_void_p), meta.ctypes.data_as(ctypes.c_void_p), why, 256)
    return rc, lp.reshape(ROWS, 64), own.reshape(6, 64), meta, why.value.decode()


def _emis(bk, st, x):
    if st < 0:
        return -np.inf
    k = bk.emis_kind[st]; a, b, c = bk.emis_a[st], bk.emis_b[st], bk.emis_c[st]
    if x != x:
        return 0.0
    if k == 1:
        d = x - a
        return c - (d * d) * b
    return c if (a <= x <= b) else -np.inf


def _first_max(cands):
    bv, bc = cands[0]
    for v, c in cands[1:]:
        if v > bv:
            bv, bc = v, c
    return bv, bc


def _insert_even(own_i, own_m, relay, relay_first):
    """The even insert slot: itself, its match, then the delete-type state of its position -- which, in a flagged hub lane whose
    relay took its value from a gather column, stands FIRST (it wins a tie against the other two)."""
    (v, c) = _first_max([own_i, own_m])
    if relay_first and relay[0] == v and v > -np.inf and relay[1] != c:
        FLAG_DECIDED.append(1)          # a finite tie with different payloads that only the flag settles
    if relay[0] > v or (relay_first and relay[0] == v):
        return relay
    return (v, c)


FLAG_DECIDED = []


def _emulate(bk, lp, own, meta, xs):
    """viterbi_g2_kernel, one lane at a time."""
    NEG = -np.inf
    pv = np.full((4, 64), NEG); pc = np.zeros((4, 64), np.int64); dv = np.full((2, 64), NEG); dc = np.zeros((2, 64), np.int64)
    bs0, bl0, bs1, bl1, ss, sl, es, el = [int(v) for v in meta[:8]]
    hub = (int(np.uint32(meta[8])) | (int(np.uint32(meta[9])) << 32))
    front = np.zeros(64, bool)          # hub lanes whose relay kept a gather column's value in the previous time step
    dv[ss, sl] = 0.0

    def sweeps(y, yc):
        while True:
            win_any = False
            y1_old = y[1].copy(); yc1_old = yc[1].copy()
            for l in range(64):
                tin = (y1_old[l - 1] if l > 0 else 0.0) + lp[ROW_CH, l]
                if tin > y[0, l]:
                    y[0, l] = tin; yc[0, l] = yc1_old[l - 1] if l > 0 else 0
                tin = y[0, l] + lp[ROW_CH + 1, l]
                if tin > y[1, l]:
                    y[1, l] = tin; yc[1, l] = yc[0, l]; win_any = True
            if not win_any:
                break

    sweeps(dv, dc)
    inc = np.zeros((4, 64), np.int64)
    for k in range(4):
        for l in range(64):
            if own[k, l] >= 0:
                inc[k, l] = bk.count_inc[own[k, l]]
    sh = lambda a: np.concatenate([[0.0], a[:-1]])
    shc = lambda a: np.concatenate([[0], a[:-1]])
    for x in xs:
        sMe, sMo, sIo, sDo = sh(pv[0]), sh(pv[1]), sh(pv[3]), sh(dv[1])
        cMe, cMo, cIo, cDo = shc(pc[0]), shc(pc[1]), shc(pc[3]), shc(dc[1])
        b0v, b0c = pv[bs0, max(bl0, 0)], pc[bs0, max(bl0, 0)]
        nv = np.full((4, 64), NEG); nc = np.zeros((4, 64), np.int64)
        for l in range(64):
            best = [
                _first_max([(sMe[l] + lp[0, l], cMe[l]), (sIo[l] + lp[1, l], cIo[l]), (sMo[l] + lp[2, l], cMo[l]), (pv[2, l] + lp[3, l], pc[2, l]),
                            (pv[0, l] + lp[4, l], pc[0, l]), (b0v + lp[5, l], b0c), (sDo[l] + lp[6, l], cDo[l])]),
                _first_max([(sMo[l] + lp[7, l], cMo[l]), (pv[2, l] + lp[8, l], pc[2, l]), (pv[0, l] + lp[9, l], pc[0, l]), (pv[3, l] + lp[10, l], pc[3, l]),
                            (pv[1, l] + lp[11, l], pc[1, l]), (dv[0, l] + lp[12, l], dc[0, l])]),
                _insert_even((pv[2, l] + lp[ROW_IE, l], pc[2, l]), (pv[0, l] + lp[ROW_IE + 1, l], pc[0, l]), (dv[0, l] + lp[ROW_IE + 2, l], dc[0, l]), front[l]),
                _first_max([(pv[3, l] + lp[ROW_IO, l], pc[3, l]), (pv[1, l] + lp[ROW_IO + 1, l], pc[1, l]), (dv[1, l] + lp[ROW_IO + 2, l], dc[1, l])])]
            for k, (bv, bc) in enumerate(best):
                nv[k, l] = bv + _emis(bk, own[k, l], x); nc[k, l] = bc + inc[k, l]
        y = np.full((2, 64), NEG); yc = np.zeros((2, 64), np.int64)
        sI, sM = sh(nv[3]), sh(nv[1]); cI, cM = shc(nc[3]), shc(nc[1])
        b1v, b1c = nv[bs1, max(bl1, 0)], nc[bs1, max(bl1, 0)]          # this time step's value of the delete-type states' broadcast source
        g01 = np.full(64, NEG)
        for l in range(64):
            g01[l] = _first_max([(sI[l] + lp[ROW_DE, l], cI[l]), (sM[l] + lp[ROW_DE + 1, l], cM[l])])[0]
            y[0, l], yc[0, l] = _first_max([(sI[l] + lp[ROW_DE, l], cI[l]), (sM[l] + lp[ROW_DE + 1, l], cM[l]), (b1v + lp[ROW_DE + 2, l], b1c)])
            y[1, l], yc[1, l] = _first_max([(nv[2, l] + lp[ROW_DO, l], nc[2, l]), (nv[0, l] + lp[ROW_DO + 1, l], nc[0, l])])
        sweeps(y, yc)
        front = np.array([bool((hub >> l) & 1) and y[0, l] == g01[l] for l in range(64)])
        pv, pc, dv, dc = nv, nc, y, yc
    return dv[es, el], dc[es, el]


@pytest.fixture(scope="module")
def lib():
    from strique_amd import build
    return ctypes.CDLL(build.build_lib())


@pytest.mark.parametrize("name,repeat,flank", [("c9orf72", None, 50), ("fmr1", None, 50), ("c9orf72", "CAGCA", 40), ("c9orf72", "AT", 30),
                                              ("c9orf72", "GGCCTGGCCTGG", 35)])
def test_layout_reproduces_the_oracle(lib, orc, pm, cfg, name, repeat, flank):
    from strique_amd import hmm
    import warnings
    chrom, b, e, rep, prefix, suffix = cfg["repeat"][name]
    rep = repeat or rep
    fm = hmm.FlankedRepeatModel(rep, prefix[-flank:], suffix[:flank], pm, cfg["HMM"])
    bk = fm.baked
    assert bk.pos_kind is not None
    rc, lp, own, meta, why = _layout(lib, bk)
    assert rc == 0, why
    assert (own >= 0).sum() == bk.n_states          # every state has its place
    rng = np.random.default_rng(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for nrep, drop in ((2, 0), (9, 4)):
            seq = prefix[-flank:] + rep * nrep + suffix[:flank]
            seq = seq[:12] + seq[12 + drop:]
            x = np.clip(pm.generate_signal(seq, samples=5, noise=True, rng=rng), pm.model_min + .5, pm.model_max - .5)
            lo, _, co = orc.viterbi(bk, x, want_path=False)
            lg, cg = _emulate(bk, lp, own, meta, x)
            assert np.float64(lo).tobytes() == np.float64(lg).tobytes() and co == cg
        for xn in (np.full(30, np.nan), np.where(np.arange(len(x)) % 3 == 0, np.nan, x)[:300]):
            lo, _, co = orc.viterbi(bk, xn, want_path=False)
            lg, cg = _emulate(bk, lp, own, meta, xn)
            assert np.float64(lo).tobytes() == np.float64(lg).tobytes() and co == cg


def tie_prone(bk, rng):
    """The same topology with every log-probability a small negative integer and every emission Uniform with constant -1:
    all path scores are exact integers, so equal candidates meet at almost every state and time step and the count that
    comes out depends on every tie being broken as the oracle breaks it (first in-edge in ascending source order)."""
    lp = -rng.integers(1, 4, len(bk.in_logp)).astype(np.float64) if rng is not None else np.full(len(bk.in_logp), -1.0)
    if rng is None:
        # Every edge -1: a path's score is its number of hops, every path of emitting states ties with every other.  Then: no
        # shortcuts through silent states, the repeat unit is entered through its first insert state only (whose relayed
        # in-edges -- last prefix insert / match -- stand FIRST in its in-edge order), and inside the unit a path has to move
        # on with every observation.  From the first completed round of the loop on, the first insert state sees, at every time
        # step, "still in the prefix" (count 0, relayed) tie with "round the loop k times" (count k, its own match column).
        import re
        names = list(bk.names)
        lp[bk.in_ptr[bk.silent_start]:] = -50.0
        unit_match = [n for n in names if re.fullmatch(r"repeat\d+m", n)]
        first_match = min(unit_match, key=lambda n: (len(n), n))
        for n in unit_match:
            l = names.index(n)
            for e in range(bk.in_ptr[l], bk.in_ptr[l + 1]):
                src = names[bk.in_src[e]]
                if src == n or (n == first_match and (src.startswith("prefix") or src == first_match[:-1] + "i")):
                    lp[e] = -50.0          # (the first match state is reached from the loop only: it carries the loop's count)
        for n in names:
            if re.fullmatch(r"repeat\d+i", n):
                l = names.index(n)
                for e in range(bk.in_ptr[l], bk.in_ptr[l + 1]):
                    if n != first_match[:-1] + "i" or bk.in_src[e] == l:
                        lp[e] = -50.0          # no other insert state of the unit, no dwelling in the first one
    ne = bk.silent_start
    return bk._replace(in_logp=lp, emis_kind=np.full(ne, 2, np.int32), emis_a=np.full(ne, 0.0), emis_b=np.full(ne, 200.0), emis_c=np.full(ne, -1.0))


@pytest.mark.parametrize("name,repeat", [("c9orf72", None), ("fmr1", None), ("c9orf72", "CAGCA")])
def test_ties_are_broken_like_the_oracle(lib, orc, pm, cfg, name, repeat):
    """Round 4 relays part of an insert-type state's in-edges through a virtual delete state and restores their place in the
    tie order with a flag (VitG2::hub_mask).  Integer log-probabilities make ties the rule: the emulated time step must still
    give the oracle's count, window after window (the emulation counts the finite ties with different payloads that only
    the flag settles, so that the test is known to reach the flagged case)."""
    from strique_amd import hmm
    chrom, b, e, rep, prefix, suffix = cfg["repeat"][name]
    rep = repeat or rep
    bk0 = hmm.FlankedRepeatModel(rep, prefix[-30:], suffix[:30], pm, cfg["HMM"]).baked
    del FLAG_DECIDED[:]
    for seed in range(6):
        rng = np.random.default_rng(100 + seed)
        bk = tie_prone(bk0, rng if seed else None)          # seed 0: every edge -1 (a path's score is its number of hops)
        rc, lp, own, meta, why = _layout(lib, bk)
        assert rc == 0, why
        assert (int(np.uint32(meta[8])) | int(np.uint32(meta[9])) << 32) != 0          # the model has a flagged hub lane
        for T in (40, 75, 130):
            x = rng.uniform(10.0, 150.0, T)
            lo, _, co = orc.viterbi(bk, x, want_path=False)
            lg, cg = _emulate(bk, lp, own, meta, x)
            assert lo == lg and co == cg, (seed, T, lo, lg, co, cg)
    assert FLAG_DECIDED, "no window reached a finite tie with different payloads that the hub flag decides"


def test_models_that_are_no_chain_are_refused(lib, pm, pm_mod, cfg):
    """Wrong positions, a counted silent state, the hub model of the modification pass: no image, a reason, never a wrong one."""
    from strique_amd import hmm
    chrom, b, e, rep, prefix, suffix = cfg["repeat"]["c9orf72"]
    bk = hmm.FlankedRepeatModel(rep, prefix[-50:], suffix[:50], pm, cfg["HMM"]).baked
    pos = bk.pos_index.copy(); pos[7] += 3
    rc, *_, why = _layout(lib, bk, pos_index=pos)
    assert rc == 3 and why
    rc, *_, why = _layout(lib, bk, pos_index=bk.pos_index[::-1].copy())
    assert rc == 3 and why
    inc = bk.count_inc.copy(); inc[bk.silent_start + 5] = 1
    rc, *_, why = _layout(lib, bk._replace(count_inc=inc))
    assert rc == 3 and "counted" in why
    kind = bk.pos_kind.copy(); kind[:] = 1          # a match (Normal emission) declared insert-type
    rc, *_, why = _layout(lib, bk, pos_kind=kind)
    assert rc == 3 and why
    mm = hmm.RepeatModModel("GGCCCC", pm, pm_mod, cfg["HMM"]).baked
    assert mm.pos_kind is None
