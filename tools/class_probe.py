import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from strique_amd.counter import repeatCounter
pm, cfg = bench.load_inputs()
counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
counter.add_target("c9orf72", *cfg["repeat"]["c9orf72"][3:6])
sigs, strands, nreps = bench.make_batch(pm, cfg, 4096, 50000, 0)
items = [("c9orf72", s, st) for s, st in zip(sigs, strands)]
counter.detect_batch(items[:256])
for rep in range(2):
    t0 = time.time(); out = counter.detect_batch(items); dt = time.time() - t0
    print("repeatCounter.detect_batch: %.3f s -> %.0f reads/s" % (dt, len(items) / dt))
t0 = time.time(); flat = np.concatenate(sigs); tc = time.time() - t0
off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
tids = [counter._classifier_for("c9orf72", s).target_id for s in strands]
t0 = time.time(); res = counter.ctx.detect_batch(flat, off, tids); dt = time.time() - t0
print("np.concatenate %.3f s; ctx.detect_batch(flat) %.3f s -> %.0f reads/s" % (tc, dt, len(sigs) / dt))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); counter.detect_batch(items); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
