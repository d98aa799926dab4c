#!/usr/bin/env python3
"""Turn the output of tools/valu_rates.hip into profiles/<round>_valu_rates.md and profiles/valu_rates.json
(the measured issue ceilings bench.py quotes beside the nominal VALU peak).
    hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/valu_rates.hip && /tmp/valu_rates > gpurun_out/valu_rates.md
    python tools/valu_rates_summary.py gpurun_out/valu_rates.md --round r02
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("raw")
    ap.add_argument("--round", default="r02")
    a = ap.parse_args()
    data = {}
    for l in open(a.raw):
        if not l.startswith("|") or l.startswith("|---") or "instruction" in l:
            continue
        c = [x.strip() for x in l.strip().strip("|").split("|")]
        data.setdefault(c[0], {})[int(c[1])] = dict(ms=float(c[2]), clock_ghz=float(c[4]), g_per_s=float(c[5]))
    cell = "DP cell: 3 v_add_f32 + v_max3_f32, independent cells"
    chain = "DP cell: 3 v_add_f32 + v_max3_f32, chained like a DP column"
    consts = {
        "source": "profiles/%s_valu_rates.md (tools/valu_rates.hip, wave64 instructions per second over the whole chip)" % a.round,
        "nominal_peak": 1228.8,
        "v_add_f32_stream": data["v_add_f32"][4]["g_per_s"],
        "dp_cell_mix_independent": data[cell][4]["g_per_s"],
        "dp_cell_mix_chained": data[chain][4]["g_per_s"],
        "float64_stream_two_waves": min(data[k][2]["g_per_s"] for k in ("v_add_f64", "v_max_f64", "v_cmp_gt_f64")),
        "float64_stream_four_waves": min(data[k][4]["g_per_s"] for k in ("v_add_f64", "v_max_f64", "v_cmp_gt_f64")),
        "waves_per_simd": 4,
    }
    json.dump(consts, open(os.path.join(ROOT, "profiles", "valu_rates.json"), "w"), indent=1)
    out = ["# VALU issue rates on MI355X, measured (%s)" % a.round, "",
           "`tools/valu_rates.hip` (`hipcc --offload-arch=gfx950 -O2`): every kernel runs 32 768 iterations of 32 inline-asm instances of one",
           "instruction (independent registers unless noted) in 1, 2, 4 or 8 waves per SIMD on all 256 CUs; the figure is wave64",
           "instructions per second over the whole chip (G/s), from the HIP-event time of the launch.  The nominal figure used as",
           "`roofline.peak` in `bench.py` is 1024 SIMDs x 2.4 GHz / 2 cycles = **1228.8 G/s**; the shader clock read from",
           "`s_memtime` against `s_memrealtime` during these kernels was %.2f - %.2f GHz." % (
               min(v[4]["clock_ghz"] for v in data.values()), max(v[4]["clock_ghz"] for v in data.values())), "",
           "| instruction | 1 wave / SIMD | 2 | 4 | 8 (more waves than fit: 4 resident) |", "|---|---|---|---|---|"]
    for n in data:
        out.append("| %s | %s |" % (n, " | ".join("%.0f" % data[n][w]["g_per_s"] if w in data[n] else "" for w in (1, 2, 4, 8))))
    out += ["", "## Reading", "",
            "* Only the FMA-class float32 instructions (`v_add_f32`, `v_fma_f32`) issue at the double rate, and even a pure stream",
            "  of them reaches **~%.0f G/s (%.2f of the nominal peak)** at four waves per SIMD; everything else -- `v_max_f32`, `v_max3_f32`," % (
                consts["v_add_f32_stream"], consts["v_add_f32_stream"] / 1228.8),
            "  `v_med3_i32`, compares, `v_cndmask_b32`, DPP moves, and every float64 instruction (`v_add_f64`, `v_max_f64`, `v_mul_f64`,",
            "  `v_fma_f64`, `v_cmp_gt_f64`) -- issues at **~500 - 570 G/s**, half of that.  `v_pk_add_f32` has the rate of a float64",
            "  instruction: two adds for the price of two.  64-bit integer compares cost what float64 compares cost.",
            "* A lone wave issues an independent instruction every ~5.5 cycles and a dependent one every ~8.6, whatever the class;",
            "  two waves per SIMD already reach the full rate on independent streams.",
            "* **The forward DP's own mix** (three `v_add_f32` and one `v_max3_f32` per cell) runs at **%.0f G/s** on independent" % consts["dp_cell_mix_independent"],
            "  cells and **%.0f G/s** with the dependency chain of a DP column (four waves per SIMD): the half-rate `v_max3_f32`" % consts["dp_cell_mix_chained"],
            "  hides between the adds of the other waves, and the mix issues as fast as pure adds.  `align_forward_seg_kernel`",
            "  sustains ~935 G/s over a whole launch (`bench.py`, `roofline.achieved`) -- 0.76 of the nominal peak, **1.00 of",
            "  what this chip issues on the kernel's own instruction mix**.  What is left for the DP is fewer instructions",
            "  per cell (5.25 against the 4 of the bare recurrence), not better issue.",
            "* The Viterbi kernel is float64 throughout (add, compare, max, select per in-edge): its ceiling is the ~%.0f G/s" % consts["float64_stream_two_waves"],
            "  of the half-rate class at two waves per SIMD, and it sustains 325 G/s (2.8e10 instructions per 4096 windows in",
            "  85.4 ms, `%s_sq.md`) -- %.2f of that ceiling, the rest being the LDS round trips and the serial chain sweeps of a" % (a.round, 325.0 / consts["float64_stream_two_waves"]),
            "  time step.",
            "* The lone `v_cndmask_b32 ..., vcc` row reads a VCC nothing writes in that loop; with the mask in an SGPR",
            "  pair (as the DP uses it) or behind a compare the instruction issues at the half rate like the others.", ""]
    open(os.path.join(ROOT, "profiles", "%s_valu_rates.md" % a.round), "w").write("\n".join(out))
    print(json.dumps(consts, indent=1))


if __name__ == "__main__":
    main()
