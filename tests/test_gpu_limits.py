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
