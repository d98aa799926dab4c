#!/usr/bin/env python3
"""profiles/<tag>_*.md from the raw rocprofv3 output of tools/runs_r05/gpu_r5_l.sh (gpurun_out/prof_<tag>/): the default bench line
with the coarse screen (csrc/screen_kernels.hip: align_screen3_kernel, three flank rows per DP row, both flanks of a read per wave).

    python tools/summarize_round_profiles.py r06 [pmc_reads]          # (r05: the same script, round 5)

Writes <tag>_kernel_stats.{csv,md}, <tag>_pmc.md, <tag>_sq.md and the constants bench.py prices its rooflines with into
profiles/dp_constants.json: VALU instructions per wave-step and HBM bytes per column of the screen kernel that ran, VALU instructions
per time step of the flanked-HMM Viterbi."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCREENS = ("align_screen1_kernel", "align_screen2_kernel", "align_screen3_kernel", "align_screen6_kernel", "align_screen_kernel")


def short(name):
    return name.split("(")[0].replace("void ", "")


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    return files[0]


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in calls.items()}


def bench_json(log):
    # round 6: bench.py prints a compact line; the full record of a profiled pass is in <log minus .log>.json (--detail)
    detail = os.path.splitext(log)[0] + ".json"
    if os.path.exists(detail):
        return json.load(open(detail))
    line = None
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("{") and '"metric"' in ln:
                line = ln.strip()
    return json.loads(line) if line else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    pmc_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    out = os.path.join(ROOT, "profiles")
    stats = one(os.path.join(src, "kt", "**", "*kernel_stats.csv"))
    shutil.copy(stats, os.path.join(out, tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    b = bench_json(os.path.join(src, "bench_kt.log"))
    with open(os.path.join(out, tag + "_kernel_stats.md"), "w") as f:
        cmd_file = os.path.join(src, "cmd_kt.txt")
        cmd = open(cmd_file).read().strip() if os.path.exists(cmd_file) else ("bench.py --steps 2 --warmup 1 --reads 4096 --batches 1 --synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0")
        f.write("# rocprofv3 --kernel-trace --stats (%s: the default line, coarse screen on)\n\n" % tag)
        f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_%s/kt -o %s -- python3 %s` (4096 reads of 50 kb per step).  "
                "Raw table: `%s_kernel_stats.csv`.\n\n" % (tag, tag, cmd, tag))
        f.write("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %.3f | %.1f | %s |\n" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        if b:
            f.write("\nbench.py record of the same (profiled) run: value %.1f %s, `roofline.kernel` %s, screen launch %.2f ms on average (HIP events inside bench.py), stages %s\n"
                    % (b["value"], b["unit"], b["roofline"]["kernel"], b["screen"]["ms_per_step"], json.dumps(b.get("stage_ms_per_step"))))
            if b["roofline"].get("co_running"):
                f.write("\nTwo sub-batches in flight: `%s` runs on a second stream under the screen of the following step (%.1f ms of it per step inside the screen's "
                        "launch window: `strq_last_overlap`), so both kernels' durations above are those of launches that SHARE the SIMDs -- except the first timed step's screen "
                        "(nothing to run beside it after the warm-up's rows were taken) and the last step's Viterbi launch (eight waves per CU, alone).  `%s_overlap.txt` lists "
                        "which launches overlapped in this trace.\n" % (b["roofline"]["co_running"]["kernel"], b["roofline"]["co_running"]["ms_under_this_launch_per_step"], tag))
    ov = os.path.join(src, "overlap.txt")
    if os.path.exists(ov):
        shutil.copy(ov, os.path.join(out, tag + "_overlap.txt"))

    consts_path = os.path.join(out, "dp_constants.json")
    consts = json.load(open(consts_path)) if os.path.exists(consts_path) else {}
    fetch, nf = per_kernel(one(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv")))
    write, nw = per_kernel(one(os.path.join(src, "pmc_write", "**", "*counter_collection.csv")))
    fb = bench_json(os.path.join(src, "bench_fetch.log"))
    with open(os.path.join(out, tag + "_pmc.md"), "w") as f:
        f.write("# HBM traffic per kernel, rocprofv3 PMC (%s: coarse screen on)\n\n" % tag)
        f.write("Two separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, no trace domain) of `python3 bench.py --steps 1 --warmup 0 --reads %d --batches 1 "
                "--synth-workers 8 --no-cpu-baseline --no-host-leg --no-legs --check 0` (%d alignments).  Counter unit: KB; on gfx950 FETCH_SIZE under-counts wide coalesced "
                "streams by 2x (MI355X_MICROARCH.md): the x2 column is that correction as an upper bound.  Totals over the run's dispatches.\n\n" % (pmc_reads, 2 * pmc_reads))
        f.write("| kernel | dispatches | FETCH_SIZE KB | x2 | WRITE_SIZE KB | bytes (x2 + write) |\n|---|---|---|---|---|---|\n")
        for k in fetch:
            fv = fetch[k].get("FETCH_SIZE", 0.0); wv = write.get(k, {}).get("WRITE_SIZE", 0.0)
            f.write("| %s | %d | %.1f | %.1f | %.1f | %.3e |\n" % (k, nf[k], fv, 2 * fv, wv, (2 * fv + wv) * 1024))
        sk = [k for k in fetch if any(k.endswith(s) for s in SCREENS)]
        if sk and fb:
            k = sk[0]
            total = (2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024
            n_samples = float(fb["config"]["workload"].split("N~")[1].split(" ")[0]) * fb["config"]["reads_per_gpu_per_step"]
            cols = 2.0 * n_samples          # the two alignments of a read each see all its columns (the pieces' overlaps are in the bytes, not in the divisor)
            consts["screen_hbm_bytes_per_read_column"] = total / cols
            consts["screen_traffic_source"] = "profiles/%s_pmc.md" % tag
            f.write("\n`%s`: %.3e B over %.3e columns (2 alignments x the reads' samples) = %.2f B per column: the uint8 levels ONCE per read and piece (both flanks of a "
                    "read share the wave: half of what the fine screen reads), the two float32 score tables of a read (~25 KB each, staged once per workgroup); "
                    "it writes 2 x 4 B per 128 columns.\n" % (k.split("::")[-1], total, cols, total / cols))

    sq, ns = per_kernel(one(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv")))
    sb = bench_json(os.path.join(src, "bench_sq.log"))
    with open(os.path.join(out, tag + "_sq.md"), "w") as f:
        f.write("# Where the wave cycles go, rocprofv3 SQ counters (%s: coarse screen on)\n\n" % tag)
        f.write("One pass of `--pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT` over "
                "`python3 bench.py --steps 1 --warmup 0 --reads %d ...` as above.  Percentages are of SQ_WAVE_CYCLES (time a wave is resident); totals over the run's dispatches.\n\n" % pmc_reads)
        f.write("| kernel | issuing | of which VALU | issue stall | waitcnt | VALU instructions | LDS active | LDS bank conflict |\n|---|---|---|---|---|---|---|---|\n")
        for k, v in sq.items():
            wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            f.write("| %s | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.3g | %.0f %% | %.0f %% |\n" % (
                k, 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
                100 * v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_INSTS_VALU", 0), 100 * v.get("SQ_LDS_IDX_ACTIVE", 0) / wc, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / wc))
        sk = [k for k in sq if any(k.endswith(s) for s in SCREENS)]
        if sk and sb:
            k = sk[0]; name = k.split("::")[-1]
            steps = sb["screen"]["wave_steps_per_step"] * sb["steps"]
            ip = sq[k]["SQ_INSTS_VALU"] / steps
            consts.setdefault("valu_insts_per_wave_step_by_kernel", {})[name] = ip
            consts["screen_valu_source"] = "profiles/%s_sq.md (rocprofv3 --pmc SQ_INSTS_VALU over bench.py --reads %d; wave-steps from strq_last_screen of the same run)" % (tag, pmc_reads)
            f.write("\n`%s`: %.4g VALU instructions over %.4g wave-steps (strq_last_screen of the same run) = **%.2f VALU instructions per wave-step** "
                    "(one wave-step = 2 DP columns of BOTH flank alignments of a read: 2 x 870 flank rows as 2 x 290 merged rows, 10 per lane).  In the counter run: %.2f ms per launch.\n"
                    % (name, sq[k]["SQ_INSTS_VALU"], steps, ip, sb["screen"]["ms_per_step"]))
        vk = [k for k in sq if "viterbi_g2_kernel" in k]
        if vk and sb:
            k = vk[0]
            ts = sb["roofline_viterbi"]["time_steps_per_step"]
            ipt = sq[k]["SQ_INSTS_VALU"] / ts
            consts["viterbi_valu_insts_per_time_step"] = ipt
            consts["viterbi_kernel"] = k.split("::")[-1]
            consts["viterbi_valu_source"] = "profiles/%s_sq.md (rocprofv3 --pmc SQ_INSTS_VALU of %s over bench.py --reads %d: %.4g) / the Viterbi time steps of the same run (strq_last_counters: %.0f)" % (
                tag, k.split("::")[-1], pmc_reads, sq[k]["SQ_INSTS_VALU"], ts)
            f.write("\n`%s`: %.4g VALU instructions over %.4g time steps = **%.2f VALU instructions per time step** of a window.\n" % (k.split("::")[-1], sq[k]["SQ_INSTS_VALU"], ts, ipt))
    # the fine screen's pass (STRQ_SCREEN_MODE=fine: the fine bound on the two-flank body, align_screen1_kernel)
    fine_dir = os.path.join(src, "pmc_sq_fine")
    if os.path.isdir(fine_dir):
        sqf, _ = per_kernel(one(os.path.join(fine_dir, "**", "*counter_collection.csv")))
        fbj = bench_json(os.path.join(src, "bench_sq_fine.log"))
        with open(os.path.join(out, tag + "_sq.md"), "a") as f:
            for k, v in sqf.items():
                if not any(k.endswith(x) for x in SCREENS):
                    continue
                name = k.split("::")[-1]
                steps = fbj["screen"]["wave_steps_per_step"] * fbj["steps"]
                ip = v["SQ_INSTS_VALU"] / steps
                wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
                consts.setdefault("valu_insts_per_wave_step_by_kernel", {})[name] = ip
                f.write("\n`%s` (the fine screen, `STRQ_SCREEN_MODE=fine`, a pass of its own over the same 1024 reads): %.4g VALU instructions over %.4g wave-steps = **%.2f VALU "
                        "instructions per wave-step** (one wave-step = 2 DP columns of both flank alignments of a read, 30 rows per lane); issuing %.0f %%, VALU %.0f %%, issue stall %.0f %%, "
                        "waitcnt %.0f %%, LDS bank conflicts %.0f %% of the wave cycles; %.2f ms per launch.\n"
                        % (name, v["SQ_INSTS_VALU"], steps, ip, 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                           100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / wc, fbj["screen"]["ms_per_step"]))
    json.dump(consts, open(consts_path, "w"), indent=1)
    print(json.dumps({k: consts[k] for k in consts if "screen" in k or "viterbi" in k or k == "valu_insts_per_wave_step_by_kernel"}, indent=1))


if __name__ == "__main__":
    main()
