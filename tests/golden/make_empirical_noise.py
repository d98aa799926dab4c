"""Noise model of synthetic reads taken from the ONE real read the reference bundles (data/c9orf72.fast5 -> tests/golden/bundled_read.npz;
documented row: docs/installation/test.md:15-16) -- replaces the normal / uniform generator of scripts/STRique.py:182-195 for the
"degraded" workload of bench.py and tests.  Test infrastructure: run in the build container, the .npz is the committed fixture.

    python tests/golden/make_empirical_noise.py          # writes strique_amd/data/empirical_noise.npz (package data of strique_amd.synth)

The read is decoded with the CPU oracle (conditioning, both flank alignments, flanked-repeat HMM Viterbi with the full path,
oracle/strique_oracle.py).  Along the path every run of samples at one chain position (match or insert state of one k-mer; the two
emitting dummy states of the repeat loop count towards the k-mer before them) is one EVENT = one k-mer occurrence:
    dwell[e]          samples of the event; a k-mer the path skips (delete states) is a dwell of 0
    level_offset[e]   median of the event's samples minus the k-mer's model mean, in the model's pA (events of >= 3 samples),
                      shrunk towards their mean by the share of their variance that is the sampling noise of a median of `dwell`
                      noisy samples (pi/2 x robust sample variance / dwell per event) -- the pool holds the spread of the LEVELS,
                      the samples' own noise is added again from resid_z when a read is synthesised
    resid_z           (sample - model mean - level offset of its event) / model stdv of the k-mer, every sample of every event
                      (samples decoded as insert / dummy states are the outliers of this pool)
strique_amd.synth.EmpiricalNoise resamples the three pools independently (bootstrap)."""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    from oracle import strique_oracle as orc
    t = np.load(os.path.join(HERE, "pore_tables.npz"))
    opm = orc.PoreModel(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
    cfg = json.load(open(os.path.join(HERE, "config.json")))
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    sig = np.load(os.path.join(HERE, "bundled_read.npz"))["signal"]
    tc = orc.classifier(repeat, prefix, suffix, "-", opm, None, cfg["HMM"])
    res, info = orc.detect(sig, tc, opm, orc.align_params(cfg["align"]))
    assert res[4] == 1633 and res[5] == 40758, res          # the documented geometry
    flt, u8, morph, fltn = orc.condition(sig, opm)
    x = fltn[info["prefix_begin"]:info["suffix_end"]]
    model = tc["hmm"]
    logp, path, counted = orc.viterbi(model, x)
    names = [model.names[s] for s in path]
    # match-state parameters per chain position
    mu, sd = {}, {}
    for i, n in enumerate(model.names):
        m = re.match(r"(prefix|repeat|suffix)(\d+)m$", n)
        if m and model.emis_kind[i] == 1:
            key = (m.group(1), int(m.group(2)))
            mu[key] = float(model.emis_a[i])
            sd[key] = float(1.0 / np.sqrt(2.0 * model.emis_b[i])) if hasattr(model, "emis_b") else 1.0
    events = []          # [key, first sample, one past the last sample]
    for tt, n in enumerate(names):
        m = re.match(r"(prefix|repeat|suffix)(\d+)[mi]$", n)
        if m is None:          # dummy state of the repeat loop: noise between two units, kept with the event before it
            if events:
                events[-1][2] = tt + 1
            continue
        key = (m.group(1), int(m.group(2)))
        if events and events[-1][0] == key:
            events[-1][2] = tt + 1
        else:
            events.append([key, tt, tt + 1])
    length = {}
    for key, a, bb in events:
        length[key[0]] = max(length.get(key[0], 0), key[1] + 1)
    dwell, offs, rz, med_var = [], [], [], []
    for k, (key, a, bb) in enumerate(events):
        if k > 0 and events[k - 1][0][0] == key[0]:
            i0, i1 = events[k - 1][0][1], key[1]
            skipped = ((i1 - i0) % length[key[0]] if key[0] == "repeat" else i1 - i0) - 1
            dwell += [0] * max(0, skipped)
        seg = x[a:bb]
        dwell.append(bb - a)
        d = float(np.median(seg) - mu[key])
        if bb - a >= 3:
            offs.append(d)
            robust_sd = 1.4826 * float(np.median(np.abs(seg - np.median(seg))))
            med_var.append(np.pi / 2.0 * robust_sd ** 2 / (bb - a))
        rz.append((seg - mu[key] - d) / sd[key])
    rz = np.concatenate(rz)
    offs = np.array(offs); v_meas = float(offs.var()); v_noise = float(np.mean(med_var))
    shrink = float(np.sqrt(max(0.0, v_meas - v_noise) / v_meas))
    offs_raw_sd = float(offs.std())
    offs = offs.mean() + (offs - offs.mean()) * shrink
    out = dict(dwell=np.array(dwell, np.int32), level_offset=np.array(offs, np.float32), resid_z=rz.astype(np.float32),
               source=np.array("data/c9orf72.fast5 read ce47b364-ed6e-4409-808a-1041c0b5aac2, window [%d, %d), %d events"
                               % (info["prefix_begin"], info["suffix_end"], len(events))))
    np.savez_compressed(os.path.join(HERE, "..", "..", "strique_amd", "data", "empirical_noise.npz"), **out)
    dw = out["dwell"]
    print("events %d (+%d skipped k-mers); dwell mean %.2f median %d p5 %d p95 %d max %d" % (len(events), int((dw == 0).sum()), dw.mean(), np.median(dw), np.percentile(dw, 5), np.percentile(dw, 95), dw.max()))
    print("level offsets as measured: sd %.2f pA, sampling noise of the medians %.2f pA -> shrunk by %.3f" % (offs_raw_sd, np.sqrt(v_noise), shrink))
    print("level offset: mean %.2f sd %.2f pA; resid z: sd %.2f, |z|>3: %.3f, |z|>6: %.4f" % (np.mean(offs), np.std(offs), rz.std(), (np.abs(rz) > 3).mean(), (np.abs(rz) > 6).mean()))


if __name__ == "__main__":
    main()
