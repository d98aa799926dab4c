#!/usr/bin/env python3
"""Longer randomised parity run than the test suite holds (GPU box; the CPU oracle takes most of the time):
    python tools/fuzz_detect.py SEED TRIALS [screen]
With `screen`: only configurations the upper-bound screens take (collapsed gaps, samples 6, dist_min 0, flanks of 30 ... 150 nt), reads of any length screened
(STRQ_SCREEN_MIN_N=0), and every trial runs under one of the screen settings (coarse with 2 / 3 / 6 rows per DP row and a random margin / candidate cap, fine on
the two-flank body, fine on the one-flank kernel, default).
Every trial draws an `align` block (collapsed or general affine gaps, dist_offset / dist_min, samples), every HMM
probability and std scale, a repeat unit, flank lengths of 30 ... 400 nt, with or without the modification model,
and three reads (either strand, int16 or float64, sometimes rounded to provoke ties); the whole tuple of
repeatCounter.detect_batch must equal the oracle's."""
import sys, json, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from strique_amd import synth
from strique_amd.counter import repeatCounter
from strique_amd.pore_model import pore_model
from oracle import strique_oracle as orc
orc.lib()
G = os.path.join(ROOT, 'tests', 'golden')
t = np.load(os.path.join(G, 'pore_tables.npz')); cfg = json.load(open(os.path.join(G, 'config.json')))
pm = pore_model(table=(t["base_kmer"], t["base_mean"], t["base_stdv"])); pm_mod = pore_model(table=(t["mod_kmer"], t["mod_mean"], t["mod_stdv"]))
opm = orc.PoreModel(table=(t["base_kmer"], t["base_mean"], t["base_stdv"])); opm_mod = orc.PoreModel(table=(t["mod_kmer"], t["mod_mean"], t["mod_stdv"]))
seed = int(sys.argv[1]); trials = int(sys.argv[2]); screen_mode = len(sys.argv) > 3 and sys.argv[3] == "screen"
rng = np.random.default_rng(seed)
nt = lambda n: "".join(rng.choice(list("ACGT"), n))
bad = 0; t0 = time.time()
for trial in range(trials):
    collapsed = screen_mode or rng.random() < 0.6
    gh = -float(rng.integers(1, 4)); gv = -float(rng.integers(4, 20))
    acfg = dict(gap_open_h=gh, gap_extension_h=gh if collapsed else -float(rng.integers(1, 4)), gap_open_v=gv, gap_extension_v=gv if collapsed else -float(rng.integers(4, 20)),
                dist_offset=float(rng.choice([8.0, 12.0, 16.0, 20.0])), dist_min=float(rng.choice([0.0, 0.0, -1.0, -4.0])), samples=int(rng.choice([6, 6, 6, 6, 5, 8, 12, 7])))
    if screen_mode:
        acfg.update(samples=6, dist_min=0.0)
    hcfg = dict(cfg["HMM"])
    for key in ("match_loop", "match_match", "match_insert", "match_delete", "insert_loop", "insert_match_0", "insert_match_1", "insert_delete", "delete_delete", "delete_insert", "delete_match"):
        hcfg[key] = float(hcfg[key] * rng.uniform(0.5, 1.5))
    hcfg.update(leave_repeat=float(rng.uniform(0.0005, 0.01)), e1_ratio=float(rng.uniform(0.05, 0.5)), seq_std_scale=float(rng.uniform(0.8, 1.5)), rep_std_scale=float(rng.uniform(0.8, 1.6)), rep_std_offset=float(rng.choice([0.0, 0.1])))
    with_mod = rng.random() < 0.25
    unit = "GGCCCC" if with_mod else [nt(int(rng.integers(2, 12))), "CGG", "GGCCCC", "CAG"][trial % 4]
    target = (unit, nt(int(rng.integers(30, 151 if screen_mode else 400))), nt(int(rng.integers(30, 151 if screen_mode else 400))))
    try:
        rc = repeatCounter(pm, mod_model_file=pm_mod if with_mod else None, align_config=acfg, HMM_config=hcfg, device=0)
        rc.add_target("t", *target)
        if screen_mode:
            rc.ctx.set_option("STRQ_SCREEN_MIN_N", "0")
            pick = trial % 6
            if pick < 3:
                rc.ctx.set_option("STRQ_SCREEN_MODE", "coarse")
                rc.ctx.set_option("STRQ_SCREEN2_MARGIN", int(rng.choice([1, 50, 400]))); rc.ctx.set_option("STRQ_SCREEN2_MAX_CAND", int(rng.choice([1, 4, 16])))
            elif pick == 3:
                rc.ctx.set_option("STRQ_SCREEN_MODE", "fine")
            elif pick == 4:
                rc.ctx.set_option("STRQ_NO_SCREEN", "1")
    except Exception as e:
        print("trial", trial, "setup:", str(e)[:80]); continue
    table = synth.KmerTable(pm_mod if with_mod and rng.random() < 0.5 else pm)
    params = orc.align_params(acfg)
    items = []
    for k in range(3):
        strand = "+-"[int(rng.integers(0, 2))]
        nrep = int(rng.integers(3, 80)); need = len(target[1]) + len(target[2]) + nrep * len(unit) + 2200
        sig = synth.make_read(table, 12, 1000 * seed + 10 * trial + k, need + int(rng.integers(0, 9000)), target, nrep, strand=strand, as_int16=(rng.random() < 0.7))[0]
        if rng.random() < 0.2:
            sig = np.round(sig)                      # tie-prone
        items.append(("t", sig, strand))
    got = rc.detect_batch(items)
    for (name, sig, strand), g in zip(items, got):
        tc = orc.classifier(*target, strand, opm, opm_mod if with_mod else None, hcfg, samples=acfg["samples"])
        w = orc.detect(sig, tc, opm, params, pm_mod=opm_mod if with_mod else None)[0]
        if tuple(g) != tuple(w):
            bad += 1; print("MISMATCH trial", trial, acfg, strand, g, w, flush=True)
    rc.ctx.close()
print("seed %d: %d trials, %d mismatches, %.0f s" % (seed, trials, bad, time.time() - t0))
