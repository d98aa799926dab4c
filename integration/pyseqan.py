"""Drop-in replacement for the reference's compiled module `STRique_lib.pyseqan`
(reference src/pyalign.cpp:47-63): class `align_raw` with the same eight float properties and
`align_overlap(a, b) -> (score, a_idx, b_idx)`, computed by libstrique_hip on an MI355X.

A STRique checkout switches over by putting this file at `STRique_lib/pyseqan.py` (and this
repository on PYTHONPATH): `from STRique_lib import fast5Index, pyseqan` (scripts/STRique.py:49)
then resolves to it, `repeatCounter.__init__` sets the properties (STRique.py:517-523) and
`__detect_range__` calls `align_overlap` (STRique.py:539) unchanged.

There is no CPU fallback: constructing `align_raw` without the library or without a GPU raises.
"""
import numpy as np

from strique_amd import ffi

# align_raw_settings defaults (reference src/align_raw.h:51-60)
_DEFAULTS = dict(open_h=-2.0, ext_h=-8.0, open_v=-2.0, ext_v=-8.0, dist_offset=8.0, dist_min=-16.0)
_ORDER = ("open_h", "ext_h", "open_v", "ext_v", "dist_offset", "dist_min")


def _prop(name):
    def get(self):
        return float(np.float32(self._p[name]))

    def set_(self, value):
        self._p[name] = float(np.float32(value))          # the binding stores float (align_raw<float, float>)
        self._push()
    return property(get, set_)


class align_raw(object):
    def __init__(self, device=0, context=None):
        self._ctx = context if context is not None else ffi.Context(device)
        self._p = dict(_DEFAULTS)
        self._push()

    def _push(self):
        self._ctx.set_align_params(*[self._p[k] for k in _ORDER])

    # src/pyalign.cpp:53-58
    gap_open_h = _prop("open_h")
    gap_extension_h = _prop("ext_h")
    gap_open_v = _prop("open_v")
    gap_extension_v = _prop("ext_v")
    dist_offset = _prop("dist_offset")
    dist_min = _prop("dist_min")

    # src/pyalign.cpp:51-52 -> align_raw::set_gap_open / get_gap_open (src/align_raw.h:84-87): the setter
    # writes both directions, the getter returns the horizontal one
    @property
    def gap_open(self):
        return self.gap_open_h

    @gap_open.setter
    def gap_open(self, value):
        self._p["open_h"] = self._p["open_v"] = float(np.float32(value))
        self._push()

    @property
    def gap_extension(self):
        return self.gap_extension_h

    @gap_extension.setter
    def gap_extension(self, value):
        self._p["ext_h"] = self._p["ext_v"] = float(np.float32(value))
        self._push()

    def align_overlap(self, a, b):
        """Semi-global signal alignment (src/pyalign.cpp:59-61): `b` end to end inside `a`.
        Returns (score, a_idx, b_idx): the float score and, as Python lists of ints, the view position
        of every element of `a` and of `b` (src/align_raw.h:141-146)."""
        # pybind11's list caster rounds every element to float (std::vector<float>)
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).astype(np.float32))
        self._push()            # the context may be shared with a repeatCounter that set its own parameters
        score, a_idx, b_idx, _, _, _ = self._ctx.align_overlap(a, b, want_idx=True)
        return float(score), a_idx.tolist(), b_idx.tolist()
