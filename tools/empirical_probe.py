"""Calibration probe of the empirical noise model (strique_amd.synth.EmpiricalNoise): n synthetic 50 kb reads through the CPU oracle,
planted count recovered / flank score as a fraction of the maximum.  python tools/empirical_probe.py 28 0.8 1.0 [emp | <realism>]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import multiprocessing as mp
G = os.path.join(ROOT, "tests", "golden")
def work(args):
    sig, strand, nrep = args
    from oracle import strique_oracle as orc
    t = np.load(os.path.join(G, "pore_tables.npz"))
    opm = orc.PoreModel(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
    cfg = json.load(open(os.path.join(G, "config.json")))
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    tc = orc.classifier(repeat, prefix, suffix, strand, opm, None, cfg["HMM"])
    params = orc.align_params(cfg["align"])
    res, info = orc.detect(sig, tc, opm, params)
    flt, u8, morph, fltn = orc.condition(sig, opm)
    sp = orc.align_overlap(morph, tc["prefix_ext"], params, want_idx=False)
    ss = orc.align_overlap(morph, tc["suffix_ext"], params, want_idx=False)
    return res[0], sp[0] / (16 * 870), ss[0] / (16 * 870), sp[4], ss[4], len(sig)
if __name__ == "__main__":
    from strique_amd import synth
    from strique_amd.pore_model import pore_model
    t = np.load(os.path.join(G, "pore_tables.npz"))
    pm = pore_model(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
    cfg = json.load(open(os.path.join(G, "config.json")))
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    n = int(sys.argv[1]); osc = float(sys.argv[2]); rsc = float(sys.argv[3]); mode = sys.argv[4] if len(sys.argv) > 4 else "emp"
    noise = synth.EmpiricalNoise(offset_scale=osc, resid_scale=rsc) if mode == "emp" else None
    kt = synth.KmerTable(pm)
    jobs = []
    for i in range(n):
        nrep = [200, 500, 1000, 1500, 2000][i % 5]
        sig, strand = synth.make_read(kt, 7, i, 50000, (repeat, prefix, suffix), nrep, noise=noise, realism=float(mode) if mode != "emp" else 0.0)
        jobs.append((sig, strand, nrep))
    with mp.get_context("spawn").Pool(7) as pool:
        res = pool.map(work, jobs, chunksize=1)
    fr = np.array([[r[1], r[2]] for r in res])
    ok2 = sum(abs(r[0] - j[2]) <= 2 for r, j in zip(res, jobs)); ok1 = sum(abs(r[0] - j[2]) <= max(2, 0.01 * j[2]) for r, j in zip(res, jobs))
    print("mode %s osc %.2f rsc %.2f: within +-2: %d/%d, within max(2, 1%%): %d/%d; flank score fraction median %.3f (p10 %.3f p90 %.3f)" % (mode, osc, rsc, ok2, n, ok1, n, np.median(fr), np.percentile(fr, 10), np.percentile(fr, 90)))
    for r, j in zip(res, jobs): print(j[2], r[0], "%.3f %.3f" % (r[1], r[2]), r[3], r[4], r[5])
