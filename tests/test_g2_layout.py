"""The register-resident layout of a profile chain (strq_model_set_positions / VitG2, strique_amd/csrc/viterbi_kernels.h),
checked without a GPU: the tables the library would upload (host-only debug export) drive a plain-Python restatement of
viterbi_g2_kernel's time step, whose log-probability and count must equal the oracle's bit for bit -- on STRique's
flanked models (scripts/STRique.py:384-431) with even and odd repeat profiles, missing observations included."""
import ctypes
import os

import numpy as np
import pytest

ROW_ME, ROW_MO, ROW_IE, ROW_IO, ROW_DE, ROW_DO, ROW_CH, ROWS = 0, 7, 13, 20, 25, 27, 29, 31


def _layout(lib, bk, pos_kind=None, pos_index=None):
    lp = np.zeros(ROWS * 64); own = np.zeros(6 * 64, np.int32); meta = np.zeros(8, np.int32); why = ctypes.create_string_buffer(256)
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


def _emulate(bk, lp, own, meta, xs):
    """viterbi_g2_kernel, one lane at a time."""
    NEG = -np.inf
    pv = np.full((4, 64), NEG); pc = np.zeros((4, 64), np.int64); dv = np.full((2, 64), NEG); dc = np.zeros((2, 64), np.int64)
    bs0, bl0, bs1, bl1, ss, sl, es, el = [int(v) for v in meta]
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
        b1v, b1c = pv[bs1, max(bl1, 0)], pc[bs1, max(bl1, 0)]
        nv = np.full((4, 64), NEG); nc = np.zeros((4, 64), np.int64)
        for l in range(64):
            best = [
                _first_max([(sMe[l] + lp[0, l], cMe[l]), (sIo[l] + lp[1, l], cIo[l]), (sMo[l] + lp[2, l], cMo[l]), (pv[2, l] + lp[3, l], pc[2, l]),
                            (pv[0, l] + lp[4, l], pc[0, l]), (b0v + lp[5, l], b0c), (sDo[l] + lp[6, l], cDo[l])]),
                _first_max([(sMo[l] + lp[7, l], cMo[l]), (pv[2, l] + lp[8, l], pc[2, l]), (pv[0, l] + lp[9, l], pc[0, l]), (pv[3, l] + lp[10, l], pc[3, l]),
                            (pv[1, l] + lp[11, l], pc[1, l]), (dv[0, l] + lp[12, l], dc[0, l])]),
                _first_max([(sIo[l] + lp[13, l], cIo[l]), (sMo[l] + lp[14, l], cMo[l]), (pv[2, l] + lp[15, l], pc[2, l]), (pv[0, l] + lp[16, l], pc[0, l]),
                            (b1v + lp[17, l], b1c), (sDo[l] + lp[18, l], cDo[l]), (dv[0, l] + lp[19, l], dc[0, l])]),
                _first_max([(pv[2, l] + lp[20, l], pc[2, l]), (pv[0, l] + lp[21, l], pc[0, l]), (pv[3, l] + lp[22, l], pc[3, l]), (pv[1, l] + lp[23, l], pc[1, l]),
                            (dv[1, l] + lp[24, l], dc[1, l])])]
            for k, (bv, bc) in enumerate(best):
                nv[k, l] = bv + _emis(bk, own[k, l], x); nc[k, l] = bc + inc[k, l]
        y = np.full((2, 64), NEG); yc = np.zeros((2, 64), np.int64)
        sI, sM = sh(nv[3]), sh(nv[1]); cI, cM = shc(nc[3]), shc(nc[1])
        for l in range(64):
            y[0, l], yc[0, l] = _first_max([(sI[l] + lp[ROW_DE, l], cI[l]), (sM[l] + lp[ROW_DE + 1, l], cM[l])])
            y[1, l], yc[1, l] = _first_max([(nv[2, l] + lp[ROW_DO, l], nc[2, l]), (nv[0, l] + lp[ROW_DO + 1, l], nc[0, l])])
        sweeps(y, yc)
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
