"""The CPU oracle and the host-side mirror against the golden vectors recorded from the reference
(tests/golden/make_golden.py) and against the reference's own test expectations."""
import json
import os

import numpy as np
import pytest
import scipy.ndimage as ndi
import scipy.signal

from conftest import GOLDEN


def test_pore_model_statistics(pm, pm_mod):
    g = json.load(open(os.path.join(GOLDEN, "pore_model.json")))
    for key, p in (("base", pm), ("mod", pm_mod)):
        assert p.kmer == g[key]["kmer"]
        assert float(p.model_median) == g[key]["median"]
        assert float(p.model_MAD) == g[key]["MAD"]
        assert float(p.model_min) == g[key]["min"]
        assert float(p.model_max) == g[key]["max"]
    assert float(pm_mod.scale2stdv(pm)) == g["mod_scale2stdv_base"]
    assert float(pm.scale2stdv(pm_mod)) == g["base_scale2stdv_mod"]


def test_normalize2model_bitwise(pm, pm_mod, orc, opm):
    n = np.load(os.path.join(GOLDEN, "normalize.npz"))
    for key in ("f64", "i16", "u8"):
        x = n[key + "_in"]
        assert np.array_equal(pm.normalize2model(x, mode="minmax"), n[key + "_minmax"])
        assert np.array_equal(pm.normalize2model(x, mode="median"), n[key + "_median"])
        assert np.array_equal(pm_mod.normalize2model(x, mode="minmax"), n[key + "_minmax_mod"])
        assert float(pm.MAD(x)) == float(n[key + "_MAD"])
        # the oracle's independent restatement
        assert np.array_equal(opm.normalize_minmax(x), n[key + "_minmax"])
        assert float(orc.mad(x)) == float(n[key + "_MAD"])
    assert np.array_equal(pm.generate_signal("".join(map(chr, [])) or _seq60(n), samples=8), n["generate_fixed"])


def test_normalize2model_entropy_mode(pm):
    """mode='entropy' (STRique.py:161-171; not on the count path): bit-equal to what the reference's own code computes on a seeded
    signal (tests/golden/make_entropy_golden.py ran it with scikit-image's dilation stood in for by the recalled 1-D window:
    pinned up to that window), with and without the clip; any other mode string is 'median', as in the reference."""
    z = np.load(os.path.join(GOLDEN, "normalize_entropy.npz"))
    assert np.array_equal(pm.normalize2model(z["signal"], mode="entropy"), z["entropy"])
    assert np.array_equal(pm.normalize2model(z["signal"], clip=False, mode="entropy"), z["entropy_noclip"])
    assert not np.array_equal(z["entropy"], pm.normalize2model(z["signal"], mode="median"))
    assert np.array_equal(pm.normalize2model(z["signal"], mode="whatever"), pm.normalize2model(z["signal"], mode="median"))


def _seq60(n):
    # the sequence is not stored; recover it from the fixture generator's seed
    rng = np.random.Generator(np.random.PCG64(20260001))
    return "".join(rng.choice(list("ACGT"), 400))[:60]


def test_flank_templates(pm, cfg):
    from strique_amd.counter import reverse_complement as rc
    z = np.load(os.path.join(GOLDEN, "flank_signals.npz"))
    assert bytes(z["revcomp_out"]).decode() == rc(bytes(z["revcomp_in"]).decode())
    for name, (chrom, b, e, repeat, prefix, suffix) in cfg["repeat"].items():
        pe, p, se, s = prefix.upper(), prefix[-50:].upper(), suffix.upper(), suffix[:50].upper()
        want = {"+": (p, s, pe, se), "-": (rc(s), rc(p), rc(se), rc(pe))}
        for strand, seqs in want.items():
            for field, seq in zip(("prefix", "suffix", "prefix_ext", "suffix_ext"), seqs):
                assert np.array_equal(pm.generate_signal(seq, samples=6), z["%s|%s|%s" % (name, strand, field)])


def test_medfilt_matches_scipy(orc):
    rng = np.random.default_rng(0)
    for dtype in (np.int16, np.float64):
        for n in (1, 2, 3, 10, 1001):
            x = (rng.normal(0, 300, n)).astype(dtype)
            got = orc.medfilt3(x)
            assert got.dtype == x.dtype
            assert np.array_equal(got, scipy.signal.medfilt(x, 3))


def _skimage014_open_close(u8):
    """scikit-image 0.14 opening/closing with rectangle(1, 8), written with scipy.ndimage the way
    skimage.morphology.grey does it (footprint padded to 9, dilation footprint inverted)."""
    img = u8.reshape(1, -1)
    ones = np.ones((1, 8), np.uint8)
    left = np.hstack((np.zeros((1, 1), np.uint8), ones))      # shift False: zero column first
    right = np.hstack((ones, np.zeros((1, 1), np.uint8)))     # shift True: zero column last
    ero = lambda im, fp: ndi.grey_erosion(im, footprint=fp)
    dil = lambda im, fp: ndi.grey_dilation(im, footprint=fp[::-1, ::-1])
    opened = dil(ero(img, left), right)
    return ero(dil(opened, left), right)[0]


def test_morphology_matches_ndimage(orc):
    rng = np.random.default_rng(1)
    for n in (1, 5, 8, 9, 17, 100, 5000):
        u8 = rng.integers(0, 256, n).astype(np.uint8)
        assert np.array_equal(orc.grey_open_close_1x8(u8), _skimage014_open_close(u8))
    step = np.repeat(rng.integers(0, 256, 300), rng.integers(1, 12, 300)).astype(np.uint8)
    assert np.array_equal(orc.grey_open_close_1x8(step), _skimage014_open_close(step))


@pytest.mark.parametrize("repeat,locus,counts", [("GGCCCC", "c9orf72", (100, 200)), ("GCG", "fmr1", (100, 300))])
def test_reference_unit_test_scenarios(pm, cfg, orc, opm, repeat, locus, counts):
    """scripts/STRique_test.py:47-82: noise-free signals, 8 samples per k-mer, n must equal i."""
    chrom, b, e, _, prefix, suffix = cfg["repeat"][locus]
    tc = orc.classifier(repeat, prefix, suffix, "+", opm, None, None)
    rng = np.random.default_rng(3)
    backbone = "".join(rng.choice(list("ACTG"), 2000))
    params = orc.align_params(None)
    for i in counts:
        seq = backbone[:1000] + prefix + repeat * i + suffix + backbone[-1000:]
        res, _ = orc.detect(pm.generate_signal(seq, samples=8), tc, opm, params)
        assert res[0] == i


def test_normalization_scenario_without_backbone(pm, cfg, orc, opm):
    """scripts/STRique_test.py:86-100."""
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    tc = orc.classifier(repeat, prefix, suffix, "+", opm, None, None)
    for i in (10, 50, 90):
        res, _ = orc.detect(pm.generate_signal(prefix + repeat * i + suffix, samples=8), tc, opm, orc.align_params(None))
        assert res[0] == i


# the flanks of the reference's interpolation test differ by one base from configs/repeat_config.tsv (scripts/STRique_test.py:70-71)
_INTERP_PREFIX = 'AGCGGGCCGGGGGTTCGGCCTCAGTCAGGCGCTCAGCTCCGTTTCGGTTTCACTTCCGGTGGAGGGCCGCCTCTGAGCGGGCGGCGGGCCGACGGCGAGCGCGGGCGGCGGCGGTGACGGAGGCGCCGCTGCCAGGGGGCGTGCGGCAGC'
_INTERP_SUFFIX = 'GAGGCGGCGGCGGCGGCGGCGGCGGCGGCGGCTGGGCCTCGAGCGCCCGCAGCCCACCTCTCGGGGGCGGGCTCCCGGCGCTAGCAGGGCTGAAGAGAAGATGGAGGAGCTGGTGGTGGAAGTGCGGGGCTCCAATGGCGCTTTCTACAA'


def test_every_scenario_of_the_reference_unit_tests(pm, pm_mod, cfg, orc, opm, opm_mod):
    """All four tests of scripts/STRique_test.py with their own loops and assertions (`n == i`):
    test_Detection (:43-62), test_Interpolation with its own flanks (:66-82), test_Normalization (:85-100),
    test_Modification (:103-124: noisy signals from the base and the mCpG model, count asserted on the mCpG run).
    The reference draws its backbone and noise unseeded; here they are seeded."""
    import random
    rnd = random.Random(20260102)
    backbone = ''.join(rnd.choice('ACTG') for _ in range(2000))
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    params = orc.align_params(None)                      # repeatCounter(model_file): class defaults, no JSON
    tc = orc.classifier(repeat, prefix, suffix, "+", opm, None, None)
    for i in range(100, 301, 100):                       # test_Detection
        seq = backbone[:1000] + prefix + repeat * i + suffix + backbone[-1000:]
        assert orc.detect(pm.generate_signal(seq, samples=8), tc, opm, params)[0][0] == i
    for i in range(10, 100, 10):                         # test_Normalization
        assert orc.detect(pm.generate_signal(prefix + repeat * i + suffix, samples=8), tc, opm, params)[0][0] == i
    tg = orc.classifier('GCG', _INTERP_PREFIX, _INTERP_SUFFIX, "+", opm, None, None)
    for i in range(100, 301, 100):                       # test_Interpolation
        seq = backbone[:1000] + _INTERP_PREFIX + 'GCG' * i + _INTERP_SUFFIX + backbone[-1000:]
        assert orc.detect(pm.generate_signal(seq, samples=8), tg, opm, params)[0][0] == i
    tm = orc.classifier(repeat, prefix, suffix, "+", opm, opm_mod, None)
    rng = np.random.default_rng(20260103)
    for i in range(100, 301, 100):                       # test_Modification
        seq = backbone[:1000] + prefix + repeat * i + suffix + backbone[-1000:]
        base = orc.detect(pm.generate_signal(seq, samples=8, noise=True, rng=rng), tm, opm, params, pm_mod=opm_mod)[0]
        mod = orc.detect(pm_mod.generate_signal(seq, samples=8, noise=True, rng=rng), tm, opm, params, pm_mod=opm_mod)[0]
        assert mod[0] == i
        assert base[6].count('0') > 0.9 * len(base[6]) and mod[6].count('1') > 0.9 * len(mod[6])
