#!/usr/bin/env python3
"""normalize_entropy.npz: pore_model.normalize2model(mode='entropy') of the reference itself (scripts/STRique.py:161-171) on a
seeded signal.  Runs ONLY in the build container (/root/reference), like make_golden.py, with the same stub modules -- plus the two
things that mode needs and this image lacks: scikit-image's `dilation` / `rectangle` (stood in for by the 1-D window SURVEY.md A.3
recalls for scikit-image < 0.15: an even footprint of width W covers in[i - (W/2 - 1) .. i + W/2], borders reflected) and
`np.bool` (removed from numpy 1.24 on; the reference predates that).  The vector is therefore the reference's own arithmetic --
sliding MAD, arg-partition, mask, median / MAD over the masked samples -- PINNED UP TO that recalled window.

    python tests/golden/make_entropy_golden.py
"""
import os
import sys

import numpy as np
import scipy.ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg._install_stubs()
    skm = sys.modules["skimage.morphology"]
    skm.rectangle = lambda h, w: np.ones((h, w), np.uint8)

    def dilation(img, selem):
        w = selem.shape[1]
        # window [i - (w/2 - 1), i + w/2] for even w (origin -1 moves scipy's centred window one to the right), reflected borders
        return scipy.ndimage.maximum_filter1d(img, size=w, axis=1, mode="reflect", origin=-1 if w % 2 == 0 else 0)
    skm.dilation = dilation
    if not hasattr(np, "bool"):
        np.bool = bool
    sys.path.insert(0, os.path.join(mg.REF, "scripts"))
    import STRique as ref
    pm = ref.pore_model(os.path.join(mg.REF, "models", "r9_4_450bps.model"))
    rng = np.random.Generator(np.random.PCG64(20260601))
    seq = "".join(rng.choice(list("ACGT"), 700))
    level = np.array([pm.model_dict[seq[i:i + 6]][0] for i in range(len(seq) - 5)])
    sig = np.repeat(level, rng.integers(6, 10, len(level)))
    sig = sig + rng.normal(0, 1.5, len(sig))
    sig[1500:2300] += rng.normal(0, 6.0, 800)          # a noisy stretch: what the mode looks for
    out = pm.normalize2model(sig.copy(), mode="entropy")
    np.savez_compressed(os.path.join(HERE, "normalize_entropy.npz"), signal=sig, entropy=out, entropy_noclip=pm.normalize2model(sig.copy(), clip=False, mode="entropy"))
    print("normalize_entropy.npz:", len(sig), "samples")


if __name__ == "__main__":
    main()
