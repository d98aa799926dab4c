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


def make_signal(rng, table, seq_bytes, as_int16=True):
    idx = table.indices(seq_bytes)
    dwell = rng.integers(6, 10, len(idx))
    pa = rng.normal(np.repeat(table.mean[idx], dwell), np.repeat(table.stdv[idx], dwell))
    if as_int16:
        return np.round(pa * (8192 / 1400.0) - 10).astype(np.int16)
    return pa


def make_read(table, config_id, read_idx, total_nt, target, n_repeat, strand=None, as_int16=True):
    """target = (repeat, prefix, suffix).  Returns (signal, strand)."""
    rng = np.random.Generator(np.random.PCG64(read_seed(config_id, read_idx)))
    if strand is None:
        strand = "+" if rng.random() < 0.5 else "-"
    repeat, prefix, suffix = target
    seq = make_sequence(rng, total_nt, prefix, repeat, n_repeat, suffix, strand)
    return make_signal(rng, table, seq, as_int16), strand
