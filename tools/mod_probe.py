"""Throughput of the modification pass (BASELINE configs[4]): reads/s with --mod_model on.
usage (GPU box): python tools/mod_probe.py [n_reads] [read_nt]"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
from strique_amd.counter import repeatCounter
from strique_amd.pore_model import pore_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
pm, cfg = bench.load_inputs()
t = np.load(os.path.join(R, "tests", "golden", "pore_tables.npz"))
pmm = pore_model(table=(t["mod_kmer"], t["mod_mean"], t["mod_stdv"]))
chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
sigs, strands, nreps = bench.make_batch(pm, cfg, n, nt, 0)
for mod in (None, pmm):
    rc = repeatCounter(pm, mod_model_file=mod, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    rc.add_target("c9orf72", repeat, prefix, suffix)
    items = [("c9orf72", s, st) for s, st in zip(sigs, strands)]
    rc.detect_batch(items)                       # warm-up at full size: device buffers are grown once
    t0 = time.time(); out = rc.detect_batch(items); dt = time.time() - t0
    tm = rc.ctx.last_timing()
    print("mod=%s  %d reads  %.2f s  %.0f reads/s  stages(ms) lut %.0f fwd %.0f trace %.0f cond %.0f vit %.0f   counts ok %d  mod len ok %d" % (
        mod is not None, n, dt, n / dt, tm[0], tm[1], tm[2], tm[5], tm[6],
        sum(abs(o[0] - k) <= 2 for o, k in zip(out, nreps)), sum(abs(len(o[6]) - o[0]) <= 3 for o in out)), flush=True)
