#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/collect_profiles.sh into the committed evidence:

    profiles/<tag>_kernel_stats.csv / .md   per-kernel time (rocprofv3 --kernel-trace --stats)
    profiles/<tag>_pmc.md                   FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes)
    profiles/hbm_traffic.json               bytes per alignment of the forward DP (read by bench.py)

usage: python tools/summarize_profiles.py r01 [reads_in_pmc_run=1024]
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[-1]


def counters(path, counter):
    """kernel -> [values] in dispatch order."""
    per = OrderedDict()
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            per.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return per


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    pmc_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)

    stats = one(os.path.join(src, "kt", "**", "*kernel_stats.csv"))
    shutil.copy(stats, os.path.join(out, tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    bench_line = ""
    log = os.path.join(src, "bench_kt.log")
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("{") and '"metric"' in ln:
                bench_line = ln.strip()
    with open(os.path.join(out, tag + "_kernel_stats.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats (%s)\n\n" % tag)
        f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_%s/kt -o %s -- "
                "python3 bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --synth-workers 1 --no-cpu-baseline --no-host-leg --check 0`\n"
                "(3 passes over 4096 reads of 50 kb: 1 warm-up + 2 timed; the warm-up pass cuts the column segments with the initial 8192-column overlap, the timed ones with "
                "the overlap chosen from its scores, so the forward DP's average here sits ~2 ms above bench.py's timed average).  Raw table: `%s_kernel_stats.csv`.\n\n" % (tag, tag, tag))
        f.write("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %.3f | %.1f | %s |\n" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6,
                                                        float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        if bench_line:
            b = json.loads(bench_line)
            f.write("\nbench.py line of the same (profiled) run: value %.1f %s, forward DP avg launch %.2f ms "
                    "(HIP events inside bench.py), stages %s\n" % (b["value"], b["unit"], b["roofline"]["avg_launch_ms"],
                                                                  json.dumps(b.get("stage_ms_per_step"))))

    fetch = counters(one(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv")), "FETCH_SIZE")
    write = counters(one(os.path.join(src, "pmc_write", "**", "*counter_collection.csv")), "WRITE_SIZE")
    with open(os.path.join(out, tag + "_pmc.md"), "w") as f:
        f.write("# HBM traffic per kernel, rocprofv3 PMC (%s)\n\n" % tag)
        f.write("Two separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, no trace domain) of\n"
                "`python3 bench.py --steps 1 --warmup 0 --reads %d --batches 1 --synth-workers 1 --no-cpu-baseline --no-host-leg --check 0` (%d reads of 50 kb = %d alignments).\n"
                "Counter unit: KB.  On gfx950 FETCH_SIZE under-counts wide coalesced streams by 2x (MI355X_MICROARCH.md, HBM\n"
                "section): the x2 column applies that correction as an upper bound; WRITE_SIZE is used as reported.\n"
                "Values are per-dispatch averages.\n\n" % (pmc_reads, pmc_reads, 2 * pmc_reads))
        f.write("| kernel | dispatches | FETCH_SIZE KB | x2 | WRITE_SIZE KB | bytes/launch (x2 + write) |\n|---|---|---|---|---|---|\n")
        for k in fetch:
            fv = sum(fetch[k]) / len(fetch[k])
            wv = sum(write.get(k, [0.0])) / max(1, len(write.get(k, [0.0])))
            f.write("| %s | %d | %.1f | %.1f | %.1f | %.3e |\n" % (k, len(fetch[k]), fv, 2 * fv, wv, (2 * fv + wv) * 1024))
    fk = [k for k in fetch if "align_forward" in k and not second_round(k)]
    if fk:
        k = fk[0]
        n_launch = len(fetch[k])
        fv = sum(fetch[k]) / n_launch
        wv = sum(write[k]) / len(write[k])
        n_align = 2 * pmc_reads / n_launch
        info = {
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 1 --warmup 0 "
                      "--reads %d --no-cpu-baseline --no-host-leg --check 0; see profiles/%s_pmc.md" % (pmc_reads, tag),
            "kernel": k, "alignments_per_launch": n_align,
            "FETCH_SIZE_KB": fv, "WRITE_SIZE_KB": wv,
            "bytes_per_launch_raw": (fv + wv) * 1024, "bytes_per_launch_fetch_x2": (2 * fv + wv) * 1024,
            "align_forward_kernel_bytes_per_alignment": (2 * fv + wv) * 1024 / n_align,
            "note": "FETCH_SIZE counted x2 as MI355X_MICROARCH.md prescribes for gfx950 (upper bound); WRITE_SIZE uncorrected. "
                    "Writes are the wavefront checkpoints (one per 256 steps).",
        }
        json.dump(info, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
        print(json.dumps(info, indent=1))
    consts = {}
    fb = bench_json(os.path.join(src, "bench_fetch.log"))
    if fk and fb:
        cols = fb["roofline"]["columns_computed_over_columns_of_the_reads"] * 2.0 * fb["config"]["reads_per_gpu_per_step"] * \
            float(fb["config"]["workload"].split("N~")[1].split(" ")[0]) / max(1, fb["roofline"]["launches_per_step"])
        consts["hbm_bytes_per_column"] = info["bytes_per_launch_fetch_x2"] / cols
        consts["traffic_source"] = "profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 per MI355X_MICROARCH.md)" % tag
    sq_summary(tag, src, out, pmc_reads, consts)
    if consts:
        json.dump(consts, open(os.path.join(out, "dp_constants.json"), "w"), indent=1)
        print(json.dumps(consts, indent=1))


def second_round(kernel):
    """align_forward_seg_kernel<R, S, PK, SEG, WPE, true>: the (normally empty) second round of the column segments."""
    if "seg_kernel" not in kernel:
        return False
    targs = [x.strip() for x in kernel.split("<", 1)[1].rstrip(">").split(",")]
    return len(targs) >= 6 and targs[5] == "true"


def bench_json(log):
    if not os.path.exists(log):
        return None
    for ln in open(log):
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    return None


def sq_summary(tag, src, out, pmc_reads, consts):
    """Issue / stall split of the waves (SQ counters, quad-cycle units) -> profiles/<tag>_sq.md."""
    hits = sorted(glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True))
    if not hits:
        return
    # VALU instructions per wave-step of the forward DP: SQ_INSTS_VALU of the launch / wave-steps the library counted in the same run
    for sub, log, key in (("pmc_sq", "bench_sq.log", None), ("pmc_sq_packed", "bench_sq_packed.log", "packed")):
        h = sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True))
        b = bench_json(os.path.join(src, log))
        if not h or not b:
            continue
        iv = counters(h[-1], "SQ_INSTS_VALU"); wc = counters(h[-1], "SQ_WAVE_CYCLES"); ai = counters(h[-1], "SQ_ACTIVE_INST_VALU")
        for k in iv:
            if "align_forward" not in k or second_round(k):
                continue
            steps = b["roofline"]["wave_steps_per_launch"]
            targs = [x.strip() for x in k.split("<", 1)[1].rstrip(">").split(",")]
            kind = "packed" if (targs[2] if "seg_kernel" in k else targs[-1]) == "true" else "float32"
            consts.setdefault("valu_insts_per_wave_step", {})[kind] = sum(iv[k]) / len(iv[k]) / steps
            consts.setdefault("valu_insts_per_wave_step_by_kernel", {})[k.replace("strq::", "")] = sum(iv[k]) / len(iv[k]) / steps      # bench.py looks the launched instance up by name
            consts.setdefault("kernels", {})[kind] = k
            consts.setdefault("issue_utilisation_counters", {})[kind] = {
                "SQ_INSTS_VALU": sum(iv[k]) / len(iv[k]), "wave_steps": steps,
                "valu_active_fraction_of_wave_cycles": (sum(ai[k]) / len(ai[k])) / (sum(wc[k]) / len(wc[k])) if k in ai and k in wc else None,
                "avg_launch_ms_in_counter_run": b["roofline"]["avg_launch_ms"]}
        consts["valu_source"] = "profiles/%s_sq.md (rocprofv3 --pmc SQ_INSTS_VALU over bench.py --reads %d; wave-steps from strq_last_counters of the same run)" % (tag, pmc_reads)
    names = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU",
             "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
    per = OrderedDict()
    for n in names:
        for k, v in counters(hits[-1], n).items():
            per.setdefault(k, {})[n] = sum(v) / len(v)
    with open(os.path.join(out, tag + "_sq.md"), "w") as f:
        f.write("# Where the wave cycles go, rocprofv3 SQ counters (%s)\n\n" % tag)
        f.write("One pass of `--pmc %s` over `python3 bench.py --steps 1 --warmup 0 --reads %d --no-cpu-baseline --no-host-leg --check 0`.\n"
                "Cycle counters are in quad-cycles summed over all waves (MI355X_MICROARCH.md): WAVE_CYCLES = resident wave time,\n"
                "ACTIVE_INST_ANY = issuing, WAIT_INST_ANY = issue stalls, WAIT_ANY = parked on s_waitcnt.  Per-dispatch averages.\n\n" % (" ".join(names), pmc_reads))
        f.write("| kernel | issuing | of which VALU | issue stall | waitcnt | VALU instructions | LDS active | LDS bank conflict |\n|---|---|---|---|---|---|---|---|\n")
        for k, v in per.items():
            wc = v.get("SQ_WAVE_CYCLES", 0.0)
            if wc < 1e7:
                continue
            pct = lambda n: "%.0f %%" % (100.0 * v.get(n, 0.0) / wc)
            f.write("| %s | %s | %s | %s | %s | %.3g | %s | %s |\n" % (k, pct("SQ_ACTIVE_INST_ANY"), pct("SQ_ACTIVE_INST_VALU"), pct("SQ_WAIT_INST_ANY"),
                    pct("SQ_WAIT_ANY"), v.get("SQ_INSTS_VALU", 0.0), pct("SQ_LDS_IDX_ACTIVE"), pct("SQ_LDS_BANK_CONFLICT")))
        f.write("\nPercentages are of SQ_WAVE_CYCLES (time a wave is resident).\n")
        if consts.get("valu_insts_per_wave_step"):
            f.write("\nForward DP, VALU instructions per wave-step (SQ_INSTS_VALU / wave-steps counted by the library in the same run): %s.\n"
                    "One wave-step = 2 DP columns x R rows per lane.  Chip-level issue utilisation = SQ_INSTS_VALU / launch time / "
                    "(1024 SIMDs x 2.4 GHz / 2) is computed by bench.py from these constants and its own event timing.\n"
                    % json.dumps(consts["valu_insts_per_wave_step"]))
            for kind, v in consts.get("issue_utilisation_counters", {}).items():
                u = v["SQ_INSTS_VALU"] / (v["avg_launch_ms_in_counter_run"] * 1e-3) / (1024 * 2.4e9 / 2)
                f.write("* %s tables (`%s`): %.4g VALU instructions in a %.2f ms launch (counter run, %d reads) = %.3f of the chip's VALU issue peak.\n"
                        % (kind, consts["kernels"][kind], v["SQ_INSTS_VALU"], v["avg_launch_ms_in_counter_run"], pmc_reads, u))


if __name__ == "__main__":
    main()
