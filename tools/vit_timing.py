#!/usr/bin/env python3
"""Where the cycles of a Viterbi time step go (diagnostic build of the library, GPU box):

    hipcc ... -DSTRQ_VIT_TIMING -c strique_amd/csrc/viterbi_kernels.hip  -> tools/bin/lib_vittiming.so
    STRQ_LIB=tools/bin/lib_vittiming.so python tools/vit_timing.py [reads] [read_nt]

Runs `detect` over a batch of BASELINE configs[2] reads and prints the shader cycles per time step spent in the
emitting phase, the silent gather + tournament, the chain sweeps and the rest (stores, loop), from s_memtime stamps
inside viterbi_kernel, for 1 ... 8 waves per CU (STRQ_VIT_WAVES).  The stamps drain the wave's queues: the split is a
guide to the proportions, the total is higher than an uninstrumented step."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    from strique_amd import ffi
    from strique_amd.counter import repeatCounter
    lib = ffi.load_library()
    if not hasattr(lib, "strq_debug_vit_timing"):
        raise SystemExit("this library was not built with -DSTRQ_VIT_TIMING (set STRQ_LIB)")
    pm, cfg = bench.load_inputs()
    sigs, strands, nreps = bench.make_batch(pm, cfg, n, nt, 0)
    for waves in (8, 4, 1):
        os.environ["STRQ_VIT_WAVES"] = str(waves)
        rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
        rc.add_target("c9orf72", *cfg["repeat"]["c9orf72"][3:6])
        items = [("c9orf72", s, st) for s, st in zip(sigs, strands)]
        rc.detect_batch(items[:64])
        buf = (ctypes.c_ulonglong * 8)()
        lib.strq_debug_vit_timing(buf)                       # clear
        rc.detect_batch(items)
        tm = rc.ctx.last_timing()
        lib.strq_debug_vit_timing(buf)
        v = np.array(list(buf), float)
        steps = max(v[4], 1.0)
        print("waves per CU %d: Viterbi %.1f ms | cycles per time step: emitting phase %.0f, silent gather + tournament %.0f, chain sweeps %.0f, "
              "stores + loop %.0f, sum %.0f (%d steps)" % (waves, tm[6], v[0] / steps, v[1] / steps, v[2] / steps, v[3] / steps, v[:4].sum() / steps, int(steps)), flush=True)
        rc.ctx.close()


if __name__ == "__main__":
    main()
