"""repeatCounter: per-read repeat detection on the GPU.

Host-side mirror of the reference's `repeatCounter` (scripts/STRique.py:505-618): same
constructor arguments, `add_target(target_name, repeat, prefix, suffix)` and
`detect(target_name, raw_signal, strand)` with the same return tuple and the same ValueErrors.
`detect_batch` is the throughput entry: many (read, target, strand) triples in one device pipeline.

All numerical work of detect() happens in libstrique_hip (strique_amd/ffi.py); this module only
builds the per-target inputs (flank templates, HMM arrays) and turns results into the reference's
output tuple.
"""
from collections import namedtuple


import numpy as np

from . import ffi
from . import hmm as hmm_mod
from .pore_model import pore_model

ALIGN_DEFAULTS = {'dist_offset': 16.0, 'dist_min': 0.0, 'gap_open_h': -1.0, 'gap_open_v': -16.0,
                  'gap_extension_h': -1.0, 'gap_extension_v': -16.0, 'samples': 6}       # STRique.py:507-513

_COMPLEMENT = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}


def reverse_complement(sequence):
    return "".join(_COMPLEMENT.get(base, base) for base in reversed(sequence))      # STRique.py:534-536


target_classifier = namedtuple('target_classifier',
                               ['prefix', 'suffix', 'prefix_ext', 'suffix_ext', 'repeatHMM', 'modHMM', 'target_id'])


class repeatCounter(object):
    def __init__(self, model_file, mod_model_file=None, align_config=None, HMM_config=None, device=0, context=None):
        cfg = dict(ALIGN_DEFAULTS)
        if align_config and isinstance(align_config, dict):
            cfg.update(align_config)
        self.ctx = context if context is not None else ffi.Context(device)
        self.ctx.set_align_params(cfg['gap_open_h'], cfg['gap_extension_h'], cfg['gap_open_v'], cfg['gap_extension_v'],
                                  cfg['dist_offset'], cfg['dist_min'])
        self.pm = model_file if isinstance(model_file, pore_model) else pore_model(model_file)
        if mod_model_file:
            self.pm_mod = mod_model_file if isinstance(mod_model_file, pore_model) else pore_model(mod_model_file)
        else:
            self.pm_mod = self.pm
        self.ctx.set_pore_stats(self.pm.model_tail_lo, self.pm.model_tail_hi, self.pm.model_min, self.pm.model_max)
        self.samples = cfg['samples']
        self.HMM_config = HMM_config
        self.targets = {}

    # -------------------------------------------------------------------------------------
    def _classifier(self, repeat, prefix, suffix, prefix_ext, suffix_ext):
        flanked = hmm_mod.FlankedRepeatModel(repeat, prefix, suffix, self.pm, self.HMM_config)
        mod = None
        if self.pm is not self.pm_mod:
            mod = hmm_mod.RepeatModModel(repeat, self.pm, self.pm_mod, self.HMM_config)
        sig = lambda s: self.pm.generate_signal(s, samples=self.samples)
        p, s, pe, se = sig(prefix), sig(suffix), sig(prefix_ext), sig(suffix_ext)
        flanked.model_id = self.ctx.model_create(flanked.baked)
        tid = self.ctx.target_add(pe, se, len(pe) - len(p), len(se) - len(s), self.samples, flanked.model_id,
                                  flanked.count_bias)
        if mod is not None:
            mod.model_id = self.ctx.model_create(mod.baked)
            self.ctx.target_set_mod(tid, mod.model_id, mod.model_min, mod.model_max)
        return target_classifier(p, s, pe, se, flanked, mod, tid)

    def add_target(self, target_name, repeat, prefix, suffix):
        if target_name in self.targets:
            raise ValueError("RepeatCounter: Target with name " + str(target_name) + " already defined.")
        prefix_ext = prefix.upper(); prefix = prefix[-50:].upper()            # STRique.py:555-559
        suffix_ext = suffix.upper(); suffix = suffix[:50].upper()
        repeat = repeat.upper()
        rc = reverse_complement
        tc_plus = self._classifier(repeat, prefix, suffix, prefix_ext, suffix_ext)
        # reverse strand: flanks swap roles (STRique.py:569-575)
        tc_minus = self._classifier(rc(repeat), rc(suffix), rc(prefix), rc(suffix_ext), rc(prefix_ext))
        self.targets[target_name] = (tc_plus, tc_minus)

    def _classifier_for(self, target_name, strand):
        if target_name not in self.targets:
            raise ValueError("RepeatCounter: Target with name " + str(target_name) + " not defined.")
        if strand == '+':
            return self.targets[target_name][0]
        if strand == '-':
            return self.targets[target_name][1]
        raise ValueError("RepeatCounter: Strand must be + or -.")

    # -------------------------------------------------------------------------------------
    def detect_batch(self, items):
        """items: iterable of (target_name, raw_signal, strand).  Returns a list of the tuples
        detect() returns, in input order."""
        items = list(items)
        if not items:
            return []
        tcs = [self._classifier_for(t, s) for t, _, s in items]
        sigs = [np.asarray(r) for _, r, _ in items]
        # DAC samples that fit int16 take their order statistics from exact histograms; everything else is float64
        # (radix selection on the GPU, cond_kernels.hip: f64_stats_kernel).  A mixed batch runs as two device batches.
        def fits_int16(s):
            if s.dtype.kind not in 'iu':
                return False
            if s.dtype in (np.int8, np.uint8, np.int16):
                return True
            return s.size == 0 or (int(s.min()) >= -32768 and int(s.max()) <= 32767)
        is_int = [fits_int16(s) for s in sigs]
        out = [None] * len(items)
        for want_int in (True, False):
            idx = [i for i, f in enumerate(is_int) if f == want_int]
            if not idx:
                continue
            arrs = [sigs[i].astype(np.int16 if want_int else np.float64, copy=False) for i in idx]
            res = self.ctx.detect_batch_reads(arrs, [tcs[i].target_id for i in idx])      # one pointer per read: no host-side concatenation
            mods = self.ctx.batch_fetch_mod() if self.pm is not self.pm_mod else ['-'] * len(res)
            for i, r, m in zip(idx, res, mods):
                n = int(r['count']); p = float(r['log_p']) if n or r['log_p'] != 0 else 0
                out[i] = (n, float(r['score_prefix']), float(r['score_suffix']), p, int(r['offset']), int(r['ticks']), m)
        return out

    def detect(self, target_name, raw_signal, strand):
        """(n, score_prefix, score_suffix, log_p, offset, ticks, mod_pattern) -- STRique.py:581-618."""
        return self.detect_batch([(target_name, raw_signal, strand)])[0]
