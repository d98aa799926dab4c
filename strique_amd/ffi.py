"""ctypes binding of libstrique_hip.so (include/strique_hip.h).

This is the only place the package touches the native library.  There is no CPU fallback:
if the library is missing or no GPU is present, creating a `Context` raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STRQ_LIB") or os.path.join(_HERE, "lib", "libstrique_hip.so")      # STRQ_LIB: A/B testing of builds

STRQ_OK, STRQ_ERR_ARG, STRQ_ERR_DEVICE, STRQ_ERR_UNSUPPORTED, STRQ_ERR_NOMEM = 0, 1, 2, 3, 4

_lib = None


class StriqueHipError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libstrique_hip error %d: %s" % (code, message))
        self.code = code


def _torch_hip_runtime():
    """Path of the HIP runtime a PyTorch-ROCm wheel bundles (torch/lib/libamdhip64.so), found WITHOUT importing torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    return cand if os.path.exists(cand) else None


def mapped_hip_runtimes():
    """Files of every HIP runtime mapped into this process (diagnostics, tests): there must never be two."""
    seen = set()
    try:
        for line in open("/proc/self/maps"):
            f = line.split()[-1]
            if "libamdhip64" in os.path.basename(f):
                seen.add(os.path.realpath(f))
    except OSError:
        pass
    return sorted(seen)


def _pin_hip_runtime():
    """One HIP runtime per process, whatever the import order.

    libstrique_hip.so needs `libamdhip64.so.7`; a PyTorch-ROCm wheel brings its own copy with the SAME soname.  If torch is
    imported first, the dynamic linker resolves the library's dependency to torch's already-loaded copy (soname match) and
    the process has one runtime.  The other order used to load /opt/rocm's copy here and torch's copy later -- two
    runtimes, and torch.cuda could no longer initialise (round 3's README note).  So the decision is taken here, before
    the library is opened: when a torch wheel with a bundled runtime is installed, that copy is loaded first (by path,
    RTLD_GLOBAL; torch itself is not imported), and both this library and a later `import torch` -- for RCCL through
    torch.distributed -- bind to it.  STRQ_HIP_RUNTIME=system keeps /opt/rocm's runtime (a process that never imports
    torch), STRQ_HIP_RUNTIME=/path/to/libamdhip64.so names one."""
    choice = os.environ.get("STRQ_HIP_RUNTIME", "auto")
    if choice == "system" or mapped_hip_runtimes():
        return None
    path = _torch_hip_runtime() if choice in ("auto", "torch") else choice
    if path is None:
        if choice == "torch":
            raise ImportError("STRQ_HIP_RUNTIME=torch: no torch wheel with a bundled libamdhip64.so found")
        return None
    ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    return path


def load_library(path=None):
    """Load the shared library (building nothing: see strique_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise ImportError("%s not found: run `python -m strique_amd.build` (hipcc, gfx950) first" % path)
        _pin_hip_runtime()
        lib = ctypes.CDLL(path)
        lib.strq_last_error.restype = ctypes.c_char_p
        lib.strq_last_error.argtypes = [ctypes.c_void_p]
        lib.strq_ctx_destroy.restype = None
        lib.strq_ctx_destroy.argtypes = [ctypes.c_void_p]
        lib.strq_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
        lib.strq_get_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32]
        _lib = lib
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def host_stats(signals, offsets, want_raw=False):
    """strq_host_stats: the six host-side scalars per float64 read (median, MAD, c1, h1 of the median-filtered signal;
    c1, h1 of the raw one), with numpy's arithmetic.  No context and no device involved."""
    lib = load_library()
    signals = _c(signals, np.float64); offsets = _c(offsets, np.int64)
    out = np.zeros((len(offsets) - 1, 6), np.float64)
    rc = lib.strq_host_stats(_ptr(signals), _ptr(offsets), ctypes.c_int64(len(offsets) - 1), ctypes.c_int32(1 if want_raw else 0), _ptr(out))
    if rc != STRQ_OK:
        raise StriqueHipError(rc, "strq_host_stats: bad argument")
    return out


class Context(object):
    """One HIP context / stream / workspace on one GPU (strq_ctx)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        rc = self._lib.strq_ctx_create(ctypes.c_int(device), ctypes.byref(self._h))
        if rc != STRQ_OK:
            self._h = None
            raise StriqueHipError(rc, "cannot create a context on HIP device %d (no GPU?)" % device)
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.strq_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != STRQ_OK:
            raise StriqueHipError(rc, self._lib.strq_last_error(self._h).decode())

    def device_synchronize(self):
        self._check(self._lib.strq_device_synchronize(self._h))

    def set_option(self, key, value):
        """strq_set_option on this context: `value` None removes the entry (the process-wide table / the environment variable
        of the same name decide again), "" means "not set" whatever they say; anything else is stored as its str()."""
        v = None if value is None else str(value).encode()
        self._check(self._lib.strq_set_option(self._h, key.encode(), v))

    def get_option(self, key):
        """Effective value of a switch for this context ("" when unset)."""
        buf = ctypes.create_string_buffer(256)
        self._check(self._lib.strq_get_option(self._h, key.encode(), buf, ctypes.c_int32(256)))
        return buf.value.decode()

    # ---- alignment ------------------------------------------------------------------------
    def set_align_params(self, open_h, ext_h, open_v, ext_v, dist_offset, dist_min):
        p = np.array([open_h, ext_h, open_v, ext_v, dist_offset, dist_min], dtype=np.float32)
        self._check(self._lib.strq_set_align_params(self._h, _ptr(p)))

    def get_align_params(self):
        p = np.zeros(6, np.float32)
        self._check(self._lib.strq_get_align_params(self._h, _ptr(p)))
        return p

    def align_overlap(self, a, b, want_idx=True):
        """(score, a_idx, b_idx, rec, j_end, j0); inputs are rounded to float32 like the reference's
        pybind11 caster does (src/pyalign.cpp:59-61)."""
        a = _c(a, np.float32); b = _c(b, np.float32)
        n, m = len(a), len(b)
        score = ctypes.c_float(); j_end = ctypes.c_int64(); j0 = ctypes.c_int64()
        rec = np.zeros(m, np.int32)
        a_idx = np.zeros(n, np.uint64) if want_idx else None
        b_idx = np.zeros(m, np.uint64) if want_idx else None
        self._check(self._lib.strq_align_overlap(self._h, _ptr(a), ctypes.c_int64(n), _ptr(b), ctypes.c_int64(m),
                                                 ctypes.byref(score), _ptr(a_idx), _ptr(b_idx), _ptr(rec),
                                                 ctypes.byref(j_end), ctypes.byref(j0)))
        return score.value, a_idx, b_idx, rec, j_end.value, j0.value

    def align_batch(self, levels, read_off, level_val, align_read, flank, flank_off, samples=6, want_rec=True):
        levels = _c(levels, np.uint8); read_off = _c(read_off, np.int64); level_val = _c(level_val, np.float32)
        align_read = _c(align_read, np.int32); flank = _c(flank, np.float32); flank_off = _c(flank_off, np.int64)
        na, nr = len(align_read), len(read_off) - 1
        score = np.zeros(na, np.float32); j_end = np.zeros(na, np.int64); j0 = np.zeros(na, np.int64)
        rec = np.zeros(len(flank), np.int32) if want_rec else None
        self._check(self._lib.strq_align_batch(self._h, ctypes.c_int64(na), ctypes.c_int64(nr), _ptr(levels), _ptr(read_off),
                                               _ptr(level_val), _ptr(align_read), _ptr(flank), _ptr(flank_off),
                                               ctypes.c_int32(samples), _ptr(score), _ptr(j_end), _ptr(j0), _ptr(rec)))
        return score, j_end, j0, rec

    def last_timing(self):
        t = np.zeros(8, np.float32)
        self._check(self._lib.strq_last_timing(self._h, _ptr(t)))
        return t

    def last_counters(self):
        t = np.zeros(8, np.float64)
        self._check(self._lib.strq_last_counters(self._h, _ptr(t)))
        return t

    def last_viterbi_launches(self):
        """strq_last_viterbi_launches as a dict (flanked-model decode of the last sub-batch)."""
        g = np.zeros(4, np.int32)
        self._check(self._lib.strq_last_viterbi_launches(self._h, _ptr(g)))
        return dict(zip(("launches", "register_resident", "lane_layout", "general"), (int(v) for v in g)))

    def last_second_round(self):
        """strq_last_second_round: (alignments that ran the second forward round, alignments) of the last batched call."""
        g = np.zeros(2, np.int64)
        self._check(self._lib.strq_last_second_round(self._h, _ptr(g)))
        return int(g[0]), int(g[1])

    def last_screen(self):
        """strq_last_screen as a dict: what the upper-bound screen of the last batched call did."""
        g = np.zeros(8, np.float64)
        self._check(self._lib.strq_last_screen(self._h, _ptr(g)))
        keys = ("ms", "screened", "windowed", "whole_read", "window_columns", "wave_steps", "scale", "candidate_chunks")
        out = dict(zip(keys, (float(v) for v in g)))
        m = np.zeros(8, np.int32)
        self._check(self._lib.strq_last_screen_mode(self._h, _ptr(m)))
        out["mode"] = {0: None, 1: "fine", 2: "coarse"}.get(int(m[0]))
        out["coarse_pause"], out["fine_pause"], out["coarse_margin"], out["merge"] = int(m[1]), int(m[2]), int(m[3]), int(m[4])
        out["second_look"] = int(m[5])          # alignments resolved by the coarse screen's second look (not in strq_last_second_round)
        return out

    def last_overlap(self):
        """strq_last_overlap as a dict: Viterbi ms harvested since the last run call started, and how much of it lay under the
        following sub-batch's screen kernel / whole alignment stage."""
        g = np.zeros(4, np.float64)
        self._check(self._lib.strq_last_overlap(self._h, _ptr(g)))
        return dict(zip(("viterbi_ms", "under_screen_ms", "under_alignment_stage_ms", "sub_batches"), (float(v) for v in g)))

    def last_geometry(self):
        """strq_last_geometry as a dict: which forward-DP kernel instance the last batched call ran."""
        g = np.zeros(8, np.int32)
        self._check(self._lib.strq_last_geometry(self._h, _ptr(g)))
        keys = ("waves_per_alignment", "tables_per_cu", "wpe", "rows_per_lane", "packed", "overlap_first", "overlap_worst", "launch_groups")
        return dict(zip(keys, (int(v) for v in g)))

    # ---- HMM ------------------------------------------------------------------------------
    def model_create(self, baked):
        mid = ctypes.c_int32(-1)
        arrs = [_c(baked.in_ptr, np.int32), _c(baked.in_src, np.int32), _c(baked.in_logp, np.float64),
                _c(baked.emis_kind, np.int32), _c(baked.emis_a, np.float64), _c(baked.emis_b, np.float64),
                _c(baked.emis_c, np.float64), _c(baked.count_inc, np.int32), _c(baked.tag, np.int32)]
        hints = getattr(baked, 'hint_slot', None) is not None and (np.asarray(baked.hint_slot)[:baked.silent_start] >= 0).all()
        arrs += [_c(baked.hint_slot, np.int32) if hints else None, _c(baked.hint_lane, np.int32) if hints else None]
        self._check(self._lib.strq_model_create(self._h, ctypes.c_int32(baked.n_states), ctypes.c_int32(baked.silent_start),
                                                ctypes.c_int32(baked.start), ctypes.c_int32(baked.end),
                                                *[_ptr(a) for a in arrs], ctypes.byref(mid)))
        self.last_positions_rc = None
        if getattr(baked, 'pos_kind', None) is not None:
            # optional register-resident image (profile chains); a model that is no such chain keeps its lane layout
            rc = self._lib.strq_model_set_positions(self._h, mid, _ptr(_c(baked.pos_kind, np.int32)), _ptr(_c(baked.pos_index, np.int32)))
            self.last_positions_rc = rc
            if rc not in (STRQ_OK, STRQ_ERR_UNSUPPORTED):
                self._check(rc)
        return mid.value

    def viterbi_batch(self, model_id, seqs, want_path=False):
        """seqs: list of float64 arrays.  Returns (logp[], counted[], status[], paths or None)."""
        off = np.zeros(len(seqs) + 1, np.int64)
        for i, s in enumerate(seqs):
            off[i + 1] = off[i] + len(s)
        x = np.concatenate([_c(s, np.float64) for s in seqs]) if len(seqs) else np.zeros(0)
        n = len(seqs)
        logp = np.zeros(n); counted = np.zeros(n, np.int64); status = np.zeros(n, np.int32)
        paths = np.zeros(len(x), np.int32) if want_path else None
        self._check(self._lib.strq_viterbi_batch(self._h, ctypes.c_int32(model_id), ctypes.c_int64(n), _ptr(x), _ptr(off),
                                                 _ptr(logp), _ptr(counted), _ptr(status), _ptr(paths)))
        if want_path:
            paths = [paths[off[i]:off[i + 1]] for i in range(n)]
        return logp, counted, status, paths

    def viterbi(self, model_id, x, want_path=True):
        logp, counted, status, paths = self.viterbi_batch(model_id, [x], want_path)
        return logp[0], int(counted[0]), int(status[0]), (paths[0] if want_path else None)

    # ---- detect pipeline ------------------------------------------------------------------
    def set_pore_stats(self, tail_lo, tail_hi, model_min, model_max):
        self._check(self._lib.strq_set_pore_stats(self._h, ctypes.c_double(tail_lo), ctypes.c_double(tail_hi),
                                                  ctypes.c_double(model_min), ctypes.c_double(model_max)))

    def target_add(self, prefix_ext, suffix_ext, trim_prefix, trim_suffix, samples, model_id, count_bias):
        pe = _c(prefix_ext, np.float32); se = _c(suffix_ext, np.float32)
        tid = ctypes.c_int32(-1)
        self._check(self._lib.strq_target_add(self._h, _ptr(pe), ctypes.c_int64(len(pe)), _ptr(se), ctypes.c_int64(len(se)),
                                              ctypes.c_int32(trim_prefix), ctypes.c_int32(trim_suffix), ctypes.c_int32(samples),
                                              ctypes.c_int32(model_id), ctypes.c_int32(count_bias), ctypes.byref(tid)))
        return tid.value

    def target_set_mod(self, target_id, mod_model_id, mod_min, mod_max):
        self._check(self._lib.strq_target_set_mod(self._h, ctypes.c_int32(target_id), ctypes.c_int32(mod_model_id),
                                                  ctypes.c_double(mod_min), ctypes.c_double(mod_max)))

    def batch_fetch_mod(self):
        off = np.zeros(self._n_batch + 1, np.int64)
        self._check(self._lib.strq_batch_fetch_mod(self._h, None, ctypes.c_int64(0), _ptr(off)))
        pool = np.zeros(max(1, int(off[-1])), np.uint8)
        self._check(self._lib.strq_batch_fetch_mod(self._h, _ptr(pool), ctypes.c_int64(len(pool)), _ptr(off)))
        raw = pool.tobytes()
        return [raw[off[i]:off[i + 1]].decode() for i in range(self._n_batch)]

    def batch_upload(self, signals, offsets, target_ids, host_stats=None):
        """signals: one concatenated int16 or float64 array; offsets: n_reads + 1."""
        signals = np.ascontiguousarray(signals)
        if signals.dtype == np.int16:
            dtype = 0
        elif signals.dtype == np.float64:
            dtype = 1
        else:
            raise ValueError("signals must be int16 or float64")
        offsets = _c(offsets, np.int64); target_ids = _c(target_ids, np.int32)
        hs = None if host_stats is None else _c(host_stats, np.float64)
        self._n_batch = len(target_ids)
        self._check(self._lib.strq_batch_upload(self._h, ctypes.c_int64(len(target_ids)), _ptr(signals), ctypes.c_int32(dtype),
                                                _ptr(offsets), _ptr(target_ids), _ptr(hs)))

    def batch_upload_part(self, total_reads, total_samples, first_read, signals, offsets, target_ids):
        """strq_batch_upload_part: reads [first_read, first_read + len(target_ids)) of a resident batch of `total_reads` int16 reads
        (`total_samples` samples in all); parts follow each other, the first one (first_read = 0) sizes the device buffers."""
        signals = np.ascontiguousarray(signals)
        if signals.dtype != np.int16:
            raise ValueError("signals must be int16")
        offsets = _c(offsets, np.int64); target_ids = _c(target_ids, np.int32)
        self._n_batch = int(total_reads)
        self._check(self._lib.strq_batch_upload_part(self._h, ctypes.c_int64(total_reads), ctypes.c_int64(total_samples), ctypes.c_int64(first_read),
                                                     ctypes.c_int64(len(target_ids)), _ptr(signals), ctypes.c_int32(0), _ptr(offsets), _ptr(target_ids)))

    def batch_run(self):
        self._check(self._lib.strq_batch_run(self._h))

    def batch_run_range(self, first, last):
        """Reads [first, last) of the uploaded batch only (strq_batch_run_range)."""
        self._check(self._lib.strq_batch_run_range(self._h, ctypes.c_int64(first), ctypes.c_int64(last)))

    def batch_fetch(self):
        out = np.zeros(self._n_batch, dtype=RESULT_DTYPE)
        self._check(self._lib.strq_batch_fetch(self._h, _ptr(out)))
        return out

    def batch_fetch_range(self, first, last):
        """Rows of reads [first, last) (strq_batch_fetch_range): waits only for the sub-batches in flight that hold them."""
        out = np.zeros(max(0, int(last) - int(first)), dtype=RESULT_DTYPE)
        self._check(self._lib.strq_batch_fetch_range(self._h, ctypes.c_int64(first), ctypes.c_int64(last), _ptr(out)))
        return out

    def detect_batch(self, signals, offsets, target_ids, host_stats=None):
        """strq_detect_batch: signals stay in this (host) buffer and are uploaded one sub-batch ahead of
        the kernels.  Use batch_upload / batch_run / batch_fetch to keep a batch resident in HBM."""
        signals = np.ascontiguousarray(signals)
        if signals.dtype == np.int16:
            dtype = 0
        elif signals.dtype == np.float64:
            dtype = 1
        else:
            raise ValueError("signals must be int16 or float64")
        offsets = _c(offsets, np.int64); target_ids = _c(target_ids, np.int32)
        hs = None if host_stats is None else _c(host_stats, np.float64)
        self._n_batch = len(target_ids)
        out = np.zeros(self._n_batch, dtype=RESULT_DTYPE)
        self._check(self._lib.strq_detect_batch(self._h, ctypes.c_int64(len(target_ids)), _ptr(signals), ctypes.c_int32(dtype),
                                                _ptr(offsets), _ptr(target_ids), _ptr(hs), _ptr(out)))
        return out

    def detect_batch_reads(self, reads, target_ids):
        """strq_detect_batch_reads: `reads` is a list of 1-D arrays, all int16 or all float64 (C-contiguous); nothing
        is concatenated on the host."""
        n = len(reads)
        if n == 0:
            return np.zeros(0, dtype=RESULT_DTYPE)
        dt = reads[0].dtype
        if dt == np.int16:
            dtype = 0
        elif dt == np.float64:
            dtype = 1
        else:
            raise ValueError("signals must be int16 or float64")
        reads = [np.ascontiguousarray(r, dtype=dt) for r in reads]          # views where possible; kept alive through the call
        ptrs = (ctypes.c_void_p * n)(*[r.ctypes.data for r in reads])
        lengths = np.array([len(r) for r in reads], np.int64)
        target_ids = _c(target_ids, np.int32)
        self._n_batch = n
        out = np.zeros(n, dtype=RESULT_DTYPE)
        self._check(self._lib.strq_detect_batch_reads(self._h, ctypes.c_int64(n), ptrs, _ptr(lengths), ctypes.c_int32(dtype),
                                                      _ptr(target_ids), None, _ptr(out)))
        return out

    def debug_conditioning(self, read, n):
        levels = np.zeros(n, np.uint8); lval = np.zeros(256, np.float32); sc = np.zeros(10)
        self._check(self._lib.strq_debug_conditioning(self._h, ctypes.c_int64(read), _ptr(levels), ctypes.c_int64(n), _ptr(lval), _ptr(sc)))
        return levels, lval, sc


RESULT_DTYPE = np.dtype([("count", np.int32), ("status", np.int32), ("score_prefix", np.float64),
                         ("score_suffix", np.float64), ("log_p", np.float64), ("offset", np.int64),
                         ("ticks", np.int64), ("prefix_begin", np.int64), ("prefix_end", np.int64),
                         ("suffix_begin", np.int64), ("suffix_end", np.int64)], align=True)
