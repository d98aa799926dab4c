"""Throughput of the other BASELINE.json configs on one GPU (ad hoc; bench.py measures configs[2]).

  configs[1]: 10 kb reads, 30 x GGGGCC (C9orf72)
  configs[3]: 50 kb reads, C9orf72 / FMR1 (CGG) / HTT (CAG) targets 1:1:1, n ~ U{30..1000}
              (HTT: tests/golden/repeat_config_htt.tsv -- the reference's repeat_config.tsv has no HTT row)
  configs[4]: see tools/mod_probe.py

usage (GPU box): python tools/config_probe.py [n_reads]
Signals are resident in HBM for the timed passes (batch_upload once, batch_run timed), like bench.py.
"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
from strique_amd import synth
from strique_amd.counter import repeatCounter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pm, cfg = bench.load_inputs()
from strique_amd.cli import parse_config
HTT = tuple(parse_config(os.path.join(R, "tests", "golden", "repeat_config_htt.tsv"))["repeat"]["htt"][3:6])      # the committed HTT/CAG row
table = synth.KmerTable(pm)


def run(name, config_id, read_nt, picks):
    rc = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    targets = {}
    for t in ("c9orf72", "fmr1"):
        chrom, b, e, repeat, prefix, suffix = cfg["repeat"][t]
        targets[t] = (repeat, prefix, suffix)
    targets["htt"] = HTT
    for t, (repeat, prefix, suffix) in targets.items():
        rc.add_target(t, repeat, prefix, suffix)
    sigs, tids, want = [], [], []
    for i in range(n):
        t, nrep = picks(i)
        s, strand = synth.make_read(table, config_id, i, read_nt, targets[t], nrep)
        sigs.append(s); tids.append(rc._classifier_for(t, strand).target_id); want.append(nrep)
    off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    ctx = rc.ctx
    ctx.batch_upload(np.concatenate(sigs), off, tids)
    ctx.batch_run(); ctx.batch_fetch()
    ctx.device_synchronize()
    t0 = time.time()
    for _ in range(2):
        ctx.batch_run()          # (two sub-batches in flight: the second pass's forward stage runs over the first one's Viterbi launches)
    res = ctx.batch_fetch()      # ... and the last pass's Viterbi launches end inside the clock
    ctx.device_synchronize()
    dt = (time.time() - t0) / 2
    tm = ctx.last_timing()
    ok = int(sum(abs(int(r["count"]) - w) <= 2 for r, w in zip(res, want)))
    print("%s: %d reads, N~%d samples: %.1f ms per pass = %.0f reads/s   stages(ms) cond %.1f tables %.1f fwd %.1f trace %.1f viterbi %.1f   planted count recovered (+-2): %d/%d"
          % (name, n, off[-1] // n, dt * 1e3, n / dt, tm[5], tm[0], tm[1], tm[2], tm[6], ok, n), flush=True)


run("configs[1] 10 kb, 30 x GGGGCC", 2, 10000, lambda i: ("c9orf72", 30))
pick_rng = np.random.Generator(np.random.PCG64(99))
mix = [(("c9orf72", "fmr1", "htt")[i % 3], int(pick_rng.integers(30, 1001))) for i in range(n)]
run("configs[3] 50 kb, C9orf72/FMR1/HTT, n~U{30..1000}", 4, 50000, lambda i: mix[i])
