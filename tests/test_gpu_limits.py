"""Edges of the library rather than of the algorithm: a read four times longer than BASELINE's, several contexts in
one process, and a context that gives its device memory back."""
import numpy as np
import pytest

from conftest import oracle_tc
from test_gpu_detect import _read

pytestmark = pytest.mark.gpu


def test_ultra_long_read(gpu_counter, orc, opm, targets, cfg, pm):
    """A 200 kb read (N ~ 1.5 M samples, 4x BASELINE configs[2]) with 1500 repeat units: same tuple as the oracle."""
    sig = _read(pm, targets, "c9orf72", "-", 200000, 1500, 77)
    assert len(sig) > 1400000
    params = orc.align_params(cfg["align"])
    w = orc.detect(sig, oracle_tc(orc, opm, targets, "c9orf72", "-", cfg["HMM"]), opm, params)[0]
    g = gpu_counter.detect("c9orf72", sig, "-")
    assert tuple(g[:6]) == tuple(w[:6])
    assert abs(g[0] - 1500) <= 2


def test_two_contexts_in_one_process(pm, cfg, targets):
    """Two repeatCounters (two strq_ctx) on one device, calls interleaved: each gives what it gives alone."""
    from strique_amd.counter import repeatCounter
    a = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    b = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    for c in (a, b):
        for name, t in targets.items():
            c.add_target(name, *t)
    reads = [("c9orf72", _read(pm, targets, "c9orf72", "+", 6000, 20 + k, 500 + k), "+") for k in range(6)]
    alone = a.detect_batch(reads)
    mixed = []
    for k, r in enumerate(reads):
        mixed.append((a if k % 2 else b).detect(*r))
        (b if k % 2 else a).detect(*reads[(k + 1) % len(reads)])
    assert [tuple(x) for x in mixed] == [tuple(x) for x in alone]
    a.ctx.close(); b.ctx.close()


def _free_bytes():
    """hipMemGetInfo of the runtime the library itself is linked against (already loaded with it)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so.7")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_context_returns_its_device_memory(pm, cfg, targets):
    from strique_amd.counter import repeatCounter
    read = ("fmr1", _read(pm, targets, "fmr1", "+", 20000, 40, 900), "+")

    def cycle():
        c = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
        c.add_target("fmr1", *targets["fmr1"])
        out = c.detect(*read)
        c.ctx.close()
        return out
    first = cycle()
    free0 = _free_bytes()
    for _ in range(10):
        assert tuple(cycle()) == tuple(first)
    free1 = _free_bytes()
    assert free0 - free1 < 64 << 20, (free0, free1)


def test_one_buffer_per_read_equals_one_concatenated_buffer(gpu_counter, pm, targets):
    """strq_detect_batch_reads (a pointer per read, what repeatCounter.detect_batch passes) against strq_detect_batch
    on the concatenated copy: int16 and float64, empty reads in between, reads that straddle the staging slots."""
    rng = np.random.default_rng(12)
    pairs = [(_read(pm, targets, "c9orf72", "+-"[k % 2], int(rng.integers(3000, 30000)), int(rng.integers(5, 80)), 4000 + k), "+-"[k % 2]) for k in range(40)]
    empty = (np.zeros(0, np.int16), "+")
    pairs.insert(3, empty); pairs.insert(17, empty); pairs.append(empty)
    reads = [p[0] for p in pairs]; strands = [p[1] for p in pairs]
    tids = [gpu_counter._classifier_for("c9orf72", s).target_id for s in strands]
    ctx = gpu_counter.ctx
    for dtype in (np.int16, np.float64):
        arrs = [r.astype(dtype) for r in reads]
        off = np.zeros(len(arrs) + 1, np.int64); off[1:] = np.cumsum([len(a) for a in arrs])
        flat = ctx.detect_batch(np.concatenate(arrs), off, tids)
        scattered = ctx.detect_batch_reads(arrs, tids)
        assert flat.tobytes() == scattered.tobytes()
        assert (flat["count"] > 0).sum() == 40
