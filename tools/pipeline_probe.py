#!/usr/bin/env python3
"""Two sub-batches in flight against the serial order, like for like: the same resident batches, the same number of warm-up and timed
steps, per-step wall time and stage times.  (bench.py's legs differ in step counts, and the screens' pause / retry cycle makes that
matter on degraded reads.)
    python tools/pipeline_probe.py [--workload empirical|clean] [--reads 4096] [--batches 2] [--steps 12] [--modes pipelined,serial]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="empirical")
    ap.add_argument("--reads", type=int, default=4096)
    ap.add_argument("--read-nt", type=int, default=50000)
    ap.add_argument("--batches", type=int, default=2)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--modes", default="pipelined,serial")
    ap.add_argument("--opt", action="append", default=[], help="KEY=VALUE switches for every mode")
    args = ap.parse_args()
    from strique_amd.counter import repeatCounter
    from strique_amd import dist as sd
    pm, cfg = bench.load_inputs()
    counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=0)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    counter.add_target("c9orf72", repeat, prefix, suffix)
    ctx = counter.ctx
    for kv in args.opt:
        k, v = kv.split("=", 1); ctx.set_option(k, v)
    workers = max(1, min(32, sd.effective_cpus()))
    lens, strands, nreps, kept, t_gen, t_up = bench.stage_resident(ctx, counter, args, 0, args.batches, workers, args.workload, keep_first=0, keep_all=False)
    print("staged %d x %d reads (%s), synth %.1f s" % (args.batches, args.reads, args.workload, t_gen), flush=True)
    rows = {}
    for mode in args.modes.split(","):
        ctx.set_option("STRQ_SERIAL", "1" if mode == "serial" else None)
        bench.run_steps(ctx, bench.Leg(), args.reads, args.batches, args.warmup, 0)
        ctx.batch_fetch(); ctx.device_synchronize()
        leg = bench.Leg(); per = []
        t0 = time.time(); k = args.warmup
        # step by step, so that each step's wall time is seen (the rows of step k are fetched after step k + 1 is queued)
        prev = None; tlast = t0
        for i in range(args.steps):
            bi = (k + i) % args.batches
            ctx.batch_run_range(bi * args.reads, (bi + 1) * args.reads)
            leg.stats(ctx)
            tm = ctx.last_timing(); scr = ctx.last_screen()
            if prev is not None:
                leg.rows(prev[0], prev[1], ctx.batch_fetch_range(prev[1] * args.reads, (prev[1] + 1) * args.reads))
            prev = (k + i, bi)
            now = time.time()
            per.append((now - tlast, float(tm[5]), float(tm[1]), float(tm[2]), float(tm[6]), scr.get("mode"), float(scr["ms"])))
            tlast = now
        leg.rows(prev[0], prev[1], ctx.batch_fetch_range(prev[1] * args.reads, (prev[1] + 1) * args.reads))
        ctx.device_synchronize()
        el = time.time() - t0
        print("%s: %.1f reads/s, %.1f ms per step over %d steps" % (mode, args.reads * args.steps / el, el / args.steps * 1e3, args.steps))
        for i, p in enumerate(per):
            print("   step %2d: wall %.1f ms | cond %.1f fwd %.1f trace %.1f vit(prev) %.1f | screen %s %.1f" % ((i,) + tuple(1e3 * p[0:1]) + p[1:5] + (p[5], p[6])) if False else
                  "   step %2d: wall %6.1f ms | cond %5.1f fwd %6.1f trace %5.1f vit(harvested) %6.1f | screen %s %6.1f" % (i, p[0] * 1e3, p[1], p[2], p[3], p[4], p[5], p[6]))
        rows[mode] = np.concatenate([leg.last[b] for b in sorted(leg.last)]).tobytes()
    if len(rows) > 1:
        vals = list(rows.values())
        print("rows equal across modes:", all(v == vals[0] for v in vals))


if __name__ == "__main__":
    main()
