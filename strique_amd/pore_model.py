"""k-mer pore model: table loading, model statistics, flank template synthesis, normalisation.

Host-side mirror of the reference's `pore_model` (scripts/STRique.py:113-195), same method names
and argument meaning so that calling code reads the same.  Only what the `count` hot path uses is
provided: the unused 'entropy' normalisation mode (STRique.py:161-171) is not.
"""
import numpy as np


class pore_model(object):
    def __init__(self, model_file=None, table=None):
        """`model_file`: tab separated kmer, mean, stdv[, count] (the reference's format);
        `table`: alternatively (kmers, means, stdvs) sequences."""
        kmers, means, stdvs = [], [], []
        if table is not None:
            kmers = [k.decode() if isinstance(k, bytes) else str(k) for k in table[0]]
            means = [float(v) for v in table[1]]; stdvs = [float(v) for v in table[2]]
        else:
            with open(model_file, "r") as fp:
                for line in fp:
                    cols = line.strip().split("\t")           # kmer, mean, stdv[, count]  (STRique.py:115-119)
                    if len(cols) < 3:
                        continue
                    kmers.append(cols[0]); means.append(float(cols[1])); stdvs.append(float(cols[2]))
        if not kmers:
            raise ValueError("pore model %s is empty" % model_file)
        # dict keeps first-seen order like the reference's dict comprehension (later duplicates overwrite)
        self.model_dict = {}
        for k, m, s in zip(kmers, means, stdvs):
            self.model_dict[k] = (m, s)
        self.kmer = len(kmers[0])
        self._means = np.array([v[0] for v in self.model_dict.values()])
        self._stdvs = np.array([v[1] for v in self.model_dict.values()])
        self.model_median = np.median(self._means)                                # STRique.py:121
        self.model_MAD = np.mean(np.absolute(np.subtract(self._means, self.model_median)))   # :122
        lo, hi = int(np.argmin(self._means)), int(np.argmax(self._means))          # first extreme, like min()/max()
        self.model_min = self._means[lo] - 6 * self._stdvs[lo]                    # :123-126
        self.model_max = self._means[hi] + 6 * self._stdvs[hi]
        # the four model-side numbers of the 'minmax' normalisation never change (STRique.py:154,157-158)
        q_lo, q_hi = np.percentile(self._means, [1, 99])
        self.model_tail_lo = np.median(self._means[self._means < q_lo])
        self.model_tail_hi = np.median(self._means[self._means > q_hi])

    def MAD(self, signal):
        """Mean absolute deviation from the median (the reference calls it MAD, STRique.py:142-143)."""
        return np.mean(np.absolute(np.subtract(signal, np.median(signal))))

    def scale2stdv(self, other):
        """Ratio of the median k-mer stdv of `other` to this model's (STRique.py:145-148)."""
        return np.median(other._stdvs) / np.median(self._stdvs)

    def minmax_coefficients(self, signal):
        """(c1, h1, h2, c2) of the 'minmax' map  (x - c1) / h1 * h2 + c2  (STRique.py:152-160)."""
        q_lo, q_hi = np.percentile(signal, [1, 99])
        m_lo = np.median(signal[signal < q_lo])
        m_hi = np.median(signal[signal > q_hi])
        M_lo, M_hi = self.model_tail_lo, self.model_tail_hi
        return (m_lo + (m_hi - m_lo) / 2, (m_hi - m_lo) / 2, (M_hi - M_lo) / 2, M_lo + (M_hi - M_lo) / 2)

    def normalize2model(self, signal, clip=True, mode="median"):
        signal = np.asarray(signal, dtype=np.float64)
        if mode == "minmax":
            c1, h1, h2, c2 = self.minmax_coefficients(signal)
            nrm_signal = (signal - c1) / h1
            nrm_signal = nrm_signal * h2 + c2
        elif mode == "median":
            med = np.median(signal)
            mad = self.MAD(signal)
            nrm_signal = np.divide(np.subtract(signal, med), mad)
            nrm_signal = np.add(np.multiply(nrm_signal, self.model_MAD), self.model_median)
        elif mode == "entropy":
            # STRique.py:161-171 -- not on the count path (detect normalises with 'minmax'); kept for callers of the class.
            # MAD of every 500-sample window (the tail mirrored), the 50 largest jumps of that profile, a mask 750 samples wide around
            # them, then the median / MAD map over the masked samples.  One Python-level MAD per window like the reference's list
            # comprehension: np.mean over a row of a 2-D view sums in another order.
            n = 500
            a = np.append(signal, signal[-1:-1 - (n - 1):-1])
            windows = np.lib.stride_tricks.sliding_window_view(a, n)
            sliding = [self.MAD(w) for w in windows]
            sliding += [sliding[-1]]
            diff_signal = np.abs(np.diff(sliding))
            ind = np.argpartition(diff_signal, -50)[-50:]
            diff_mask = np.zeros(len(diff_signal), dtype=np.uint8)
            diff_mask[ind] = 1
            # skimage.morphology.dilation(mask, rectangle(1, 750)) of scikit-image < 0.15 on a 1 x N image: an even footprint of width W
            # covers in[i - (W/2 - 1) .. i + W/2], borders reflected (recalled, like the 1 x 8 windows of the hot path)
            import scipy.ndimage
            diff_mask = scipy.ndimage.maximum_filter1d(diff_mask, size=750, mode="reflect", origin=-1).astype(bool)
            med = np.median(signal[diff_mask])
            mad = self.MAD(signal[diff_mask])
            nrm_signal = np.divide(np.subtract(signal, med), mad)
            nrm_signal = np.add(np.multiply(nrm_signal, self.model_MAD), self.model_median)
        else:
            # the reference's `else:` branch takes every other value as 'median' (STRique.py:172-176)
            med = np.median(signal)
            mad = self.MAD(signal)
            nrm_signal = np.divide(np.subtract(signal, med), mad)
            nrm_signal = np.add(np.multiply(nrm_signal, self.model_MAD), self.model_median)
        if clip:
            np.clip(nrm_signal, self.model_min + .5, self.model_max - .5, out=nrm_signal)
        return nrm_signal

    def level_means(self, sequence):
        k = self.kmer
        return np.array([self.model_dict[sequence[i:i + k]][0] for i in range(len(sequence) - k + 1)])

    def level_stdvs(self, sequence):
        k = self.kmer
        return np.array([self.model_dict[sequence[i:i + k]][1] for i in range(len(sequence) - k + 1)])

    def generate_signal(self, sequence, samples=10, noise=False, rng=None):
        """Expected (or sampled) signal of a nucleotide sequence (STRique.py:182-195).

        `rng`: optional numpy Generator; the reference draws from the global numpy state, which
        is what happens here too when rng is None."""
        means = self.level_means(sequence)
        draw = rng if rng is not None else np.random
        if samples and not noise:
            return np.repeat(means, samples)
        dwell = draw.uniform(6, 10, len(means)).astype(int)
        if not noise:
            return np.repeat(means, dwell)
        stdvs = self.level_stdvs(sequence)
        return draw.normal(np.repeat(means, dwell), np.repeat(stdvs, dwell))
