"""Synthetic nanopore reads for tests and benchmarks (SURVEY.md 8d recipe).

sequence = backbone_L | prefix150 | repeat x n | suffix150 | backbone_R, backbone i.i.d. ACGT;
signal   = per k-mer dwell U{6,7,8,9} samples ~ N(level_mean, level_stdv) -- the distribution of
the reference's generate_signal(noise=True) (scripts/STRique.py:186-194) -- delivered as float64
pA or as int16 DAC counts round(pA * 8192/1400 - 10).  Vectorised: a 50 kb read takes ~10 ms.
"""
import numpy as np

_CODE = np.full(256, -1, np.int64)
for _i, _b in enumerate(b"ACGT"):
    _CODE[_b] = _i
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def read_seed(config_id, read_idx):
    return 20260000 + config_id * 10 ** 6 + read_idx


class KmerTable(object):
    """Dense (4^k) mean / stdv arrays of a pore_model, indexed by the base-4 code of a k-mer."""

    def __init__(self, pm):
        self.k = pm.kmer
        n = 4 ** self.k
        self.mean = np.zeros(n); self.stdv = np.zeros(n)
        for kmer, (m, s) in pm.model_dict.items():
            idx = 0
            for ch in kmer.encode():
                idx = idx * 4 + int(_CODE[ch])
            self.mean[idx] = m; self.stdv[idx] = s

    def indices(self, seq_bytes):
        codes = _CODE[np.frombuffer(seq_bytes, np.uint8)]
        k = self.k
        idx = np.zeros(len(codes) - k + 1, np.int64)
        for j in range(k):
            idx = idx * 4 + codes[j:len(codes) - k + 1 + j]
        return idx


def make_sequence(rng, total_nt, prefix, repeat, n_repeat, suffix, strand="+"):
    core = prefix.upper() + repeat.upper() * n_repeat + suffix.upper()
    n_back = total_nt - len(core)
    if n_back < 2000:
        raise ValueError("read too short for %d repeats" % n_repeat)
    left = int(rng.integers(1000, n_back - 1000 + 1))
    back = rng.integers(0, 4, n_back).astype(np.uint8)
    back = np.frombuffer(b"ACGT", np.uint8)[back].tobytes()
    seq = back[:left] + core.encode() + back[left:]
    if strand == "-":
        seq = seq.translate(_COMP)[::-1]
    return seq


class EmpiricalNoise(object):
    """Dwell times, per-occurrence level offsets and per-sample residuals of the one real read the reference bundles
    (strique_amd/data/empirical_noise.npz, made by tests/golden/make_empirical_noise.py from the oracle's decode of
    data/c9orf72.fast5), resampled independently (bootstrap): a k-mer occurrence gets a dwell from the pool (0 = the k-mer
    is skipped), its level is the table mean plus an offset from the pool, every sample adds a residual from the pool scaled
    by the k-mer's table stdv.
    Calibration (tools/empirical_probe.py, CPU oracle on 28 reads of 50 kb, 200 ... 2000 repeats): with the pools as measured
    (`offset_scale` 1) the median flank score is 0.685 of the maximum (the real read: 0.670 / 0.704) and the oracle recovers the
    planted count within max(2, 1 %) on 76 % of the reads -- a flank at that score level sits at the background's own maximum and
    ~1 search in 8 ends at a wrong place; the default `offset_scale` 0.8 is the value at which the median is 0.70, the top of the
    real read's range, and the count is recovered on 96 % (within +-2 on 68 %: over 1000+ repeat units a decode at this noise
    level loses a few).  `resid_scale` stays 1."""

    def __init__(self, path=None, offset_scale=0.8, resid_scale=1.0):
        import os
        if path is None:
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "empirical_noise.npz")          # package data (written by tests/golden/make_empirical_noise.py)
        z = np.load(path)
        self.dwell = z["dwell"].astype(np.int64)
        self.level_offset = z["level_offset"].astype(np.float64) * offset_scale
        self.resid_z = z["resid_z"].astype(np.float64) * resid_scale
        self.source = str(z["source"])

    def signal(self, rng, table, idx):
        n_k = len(idx)
        dwell = self.dwell[rng.integers(0, len(self.dwell), n_k)]
        level = table.mean[idx] + self.level_offset[rng.integers(0, len(self.level_offset), n_k)]
        n = int(dwell.sum())
        return np.repeat(level, dwell) + np.repeat(table.stdv[idx], dwell) * self.resid_z[rng.integers(0, len(self.resid_z), n)]


def make_signal(rng, table, seq_bytes, as_int16=True, realism=0.0, noise=None):
    """noise = an EmpiricalNoise: dwell, level offsets and sample residuals resampled from the bundled real read (the
    "degraded" workload of bench.py); `realism` is ignored then.
    realism = 0: the reference's generate_signal(noise=True) distribution (dwell U{6..9}, N(mean, stdv) per sample) -- the
    easiest input the flank alignment will ever see.  realism in (0, 2] degrades the read the way real r9.4 signal differs
    from the k-mer table, all of it scaled by `realism`:
      * levels: sample noise wider by (1 + 2.5 x realism); every k-mer occurrence off its table mean by N(0, 1.5 pA x realism)
        (sequence context the 6-mer table does not model); the whole read on a slow baseline drift of up to +-2 pA x realism (a
        random-walk bridge over ~10^4 samples) with a scale error of +-4 % x realism;
      * dwell: 6 % x realism of the k-mers stall (8 + a geometric tail of mean 10 samples), 4 % x realism are hurried through in
        two to four samples, instead of 6..9;
      * spikes: 2 % x realism of the samples are outliers of +-15..45 pA.
    realism = 1 brings the flank scores of the synthetic reads down to those of the read the reference bundles
    (data/c9orf72.fast5: 0.67 / 0.70 of the maximum; synthetic 0.79 at realism 0, 0.73 at 0.5, 0.69 at 1, 0.68 at 1.5:
    tools/realism_probe.py).  At that level most planted counts are still recovered; beyond it the HMM decode degrades faster
    than it does on the real read -- the knob is a stress test of the pipeline's heuristics, not a simulator."""
    idx = table.indices(seq_bytes)
    n_k = len(idx)
    if noise is not None:
        pa = noise.signal(rng, table, idx)
        if as_int16:
            return np.clip(np.round(pa * (8192 / 1400.0) - 10), -32768, 32767).astype(np.int16)
        return pa
    dwell = rng.integers(6, 10, n_k)
    if realism > 0.0:
        r = float(min(realism, 2.0))
        u = rng.random(n_k)
        stall = u < 0.06 * r
        hurry = (u >= 0.06 * r) & (u < 0.10 * r)
        dwell = np.where(stall, 8 + rng.geometric(1.0 / 10.0, n_k), dwell)
        dwell = np.where(hurry, rng.integers(2, 5, n_k), dwell)
        dwell = np.minimum(dwell, 120)
        level = table.mean[idx] + rng.normal(0.0, 1.5 * r, n_k)
        pa = rng.normal(np.repeat(level, dwell), np.repeat(table.stdv[idx] * (1.0 + 2.5 * r), dwell))
        n = len(pa)
        steps = rng.normal(0.0, 1.0, n // 512 + 2).cumsum()
        steps -= np.linspace(steps[0], steps[-1], len(steps))          # a bridge: starts and ends on the baseline
        amp = np.abs(steps).max()
        drift = np.interp(np.arange(n), np.arange(len(steps)) * 512.0, steps * (2.0 * r / amp if amp > 0 else 0.0))
        pa = (pa - 90.0) * (1.0 + rng.uniform(-0.04, 0.04) * r) + 90.0 + drift
        spikes = rng.random(n) < 0.02 * r
        pa = np.where(spikes, pa + rng.choice((-1.0, 1.0), n) * rng.uniform(15.0, 45.0, n), pa)
    else:
        pa = rng.normal(np.repeat(table.mean[idx], dwell), np.repeat(table.stdv[idx], dwell))
    if as_int16:
        return np.clip(np.round(pa * (8192 / 1400.0) - 10), -32768, 32767).astype(np.int16)
    return pa


def make_read(table, config_id, read_idx, total_nt, target, n_repeat, strand=None, as_int16=True, realism=0.0, noise=None):
    """target = (repeat, prefix, suffix).  Returns (signal, strand).  `realism` / `noise`: see make_signal (0 / None = the SURVEY.md 8d recipe)."""
    rng = np.random.Generator(np.random.PCG64(read_seed(config_id, read_idx)))
    if strand is None:
        strand = "+" if rng.random() < 0.5 else "-"
    repeat, prefix, suffix = target
    seq = make_sequence(rng, total_nt, prefix, repeat, n_repeat, suffix, strand)
    return make_signal(rng, table, seq, as_int16, realism, noise), strand
