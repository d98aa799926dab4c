#!/usr/bin/env python3
"""Benchmark of the STRique `count` hot path on MI355X: reads/s on 50 kb r9.4 signals.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of repeatCounter.detect (conditioning, two flank alignments, flanked-repeat HMM
Viterbi) over one batch of synthetic reads per GPU (BASELINE.json configs[2]: 50 kb reads, C9orf72
GGGGCC repeat counts 200/500/1000/1500/2000 in equal shares, SURVEY.md 8d recipe).  The raw int16
signals are uploaded once and stay resident in HBM; the timed region covers every kernel of the
pipeline and the (tiny) result read-back.  Reads shard over ranks (weak scaling: the batch per GPU is
fixed); with N > 1 the job ends with ONE gather of all steps' result records to rank 0 (RCCL), inside the timed region
(--gather-every-step: after every step instead).

Every step runs a DIFFERENT resident batch (--batches, default 3, all uploaded before the timed region), so the
column-segment overlap a step is cut with was adapted to another batch's scores, never to its own.

Rank 0 prints one JSON line.  `roofline` names what binds the dominant kernel (the forward flank DP):
VALU issue -- wave instructions per second against 1024 SIMDs x 2.4 GHz / 2, with the instruction count
per wave-step taken from the committed SQ_INSTS_VALU profile (profiles/dp_constants.json) and the
wave-steps counted by the library in this run; the SURVEY.md 8d algorithmic-bytes figure is kept under
`roofline.hbm_algorithmic`.  `host_inclusive_reads_per_s` is the same pipeline with the signals starting
in pageable host RAM.  `cpu_baseline` times the CPU oracle (the reference's own arithmetic: full matrix,
one double pow per cell) on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
N_SIMD = 1024                  # 256 CUs x 4 SIMD-32 (MI355X_MICROARCH.md): one wave64 VALU instruction per 2 cycles each
CLOCK_HZ = 2.4e9
FLANK_ROWS = 870               # (150 - 6 + 1) k-mers x 6 samples per flank
REPEAT_SWEEP = (200, 500, 1000, 1500, 2000)


def load_inputs():
    from strique_amd.pore_model import pore_model
    t = np.load(os.path.join(ROOT, "tests", "golden", "pore_tables.npz"))
    pm = pore_model(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "config.json")))
    return pm, cfg


def make_batch(pm, cfg, n_reads, read_nt, first_index, config_id=3):
    from strique_amd import synth
    table = synth.KmerTable(pm)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    sigs, strands, nreps = [], [], []
    for i in range(n_reads):
        nrep = REPEAT_SWEEP[(first_index + i) % len(REPEAT_SWEEP)]
        s, strand = synth.make_read(table, config_id, first_index + i, read_nt, (repeat, prefix, suffix), nrep)
        sigs.append(s); strands.append(strand); nreps.append(nrep)
    return sigs, strands, nreps


def _synth_chunk(args):
    """Worker process: reads [first, first + count) of the synthetic recipe."""
    first, count, read_nt = args
    pm, cfg = load_inputs()
    return make_batch(pm, cfg, count, read_nt, first)


def make_batches_parallel(n_reads, read_nt, first_index, workers):
    """The same reads as make_batch(..., first_index), synthesised by a few worker processes (a 50 kb read takes
    ~7 ms of numpy; 3 x 4096 of them would otherwise dominate the run)."""
    import multiprocessing as mp
    workers = max(1, min(workers, n_reads // 64 or 1))
    if workers == 1:
        pm, cfg = load_inputs()
        return make_batch(pm, cfg, n_reads, read_nt, first_index)
    per = (n_reads + workers - 1) // workers
    # chunks start at multiples of len(REPEAT_SWEEP) offsets of first_index: make_batch derives the repeat count from the index
    jobs = [(first_index + k * per, min(per, n_reads - k * per), read_nt) for k in range(workers) if k * per < n_reads]
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        parts = pool.map(_synth_chunk, jobs)
    sigs, strands, nreps = [], [], []
    for a, b, c in parts:
        sigs += a; strands += b; nreps += c
    return sigs, strands, nreps


def _oracle_side(strand):
    """The CPU oracle's own pore model and classifier (built inside oracle/, nothing from the product)."""
    from oracle import strique_oracle as orc
    t = np.load(os.path.join(ROOT, "tests", "golden", "pore_tables.npz"))
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "config.json")))
    opm = orc.PoreModel(table=(t["base_kmer"], t["base_mean"], t["base_stdv"]))
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    return orc, opm, orc.classifier(repeat, prefix, suffix, strand, opm, None, cfg["HMM"]), orc.align_params(cfg["align"])


def _cpu_one(args):
    """One read through the CPU oracle (worker process): with the reference's per-cell double pow, or
    with memoised scores (same bits, the honest "optimised CPU" variant)."""
    sig, strand, use_lut = args
    orc, opm, tc, params = _oracle_side(strand)
    t0 = time.time()
    res, _ = orc.detect(sig, tc, opm, params, use_lut=use_lut)
    return time.time() - t0, res[0]


def _pin_worker(cpus, counter):
    """Pool initializer: one worker per physical core (first logical CPU of each)."""
    with counter.get_lock():
        k = counter.value; counter.value += 1
    try:
        os.sched_setaffinity(0, {cpus[k % len(cpus)]})
    except (AttributeError, OSError):
        pass


def physical_cores():
    """First logical CPU of every physical core this process may run on (/proc/cpuinfo: physical id, core id)."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    seen = {}
    try:
        cpu = phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("processor"):
                cpu = int(ln.split(":")[1])
            elif ln.startswith("physical id"):
                phys = int(ln.split(":")[1])
            elif ln.startswith("core id"):
                core = int(ln.split(":")[1])
            elif not ln.strip():
                if cpu is not None and cpu in allowed:
                    seen.setdefault((phys, core if core is not None else cpu), cpu)
                cpu = phys = core = None
        if cpu is not None and cpu in allowed:
            seen.setdefault((phys, core if core is not None else cpu), cpu)
    except OSError:
        pass
    return sorted(seen.values()) or sorted(allowed)


def _mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                return int(ln.split()[1]) / 1e6
    except OSError:
        pass
    return 16.0


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sigs, strands, max_workers=0, sweep=(32, 64, 128)):
    """The CPU oracle (reference arithmetic: full matrix, one double pow per cell, float64 Viterbi) on the
    host cores: pinned worker processes, one read per worker -- STRique's --t N.  The full-matrix formulation
    streams ~1.6 GB per alignment, so the host's memory system, not its core count, decides what N delivers most:
    the worker counts of `sweep` (capped by physical cores / memory / --cpu-workers, spread evenly over the
    sockets) are each timed on their own sample and `value` is the BEST of them (measured wall time, nothing
    extrapolated); the whole sweep is reported."""
    import multiprocessing as mp
    cpus = physical_cores()
    host_cores = os.cpu_count() or 1
    # a worker holds the full (N + 1) x 871 float32 matrix and its byte trace: ~2.5 GB at 50 kb
    mem_cap = max(1, int(_mem_available_gb() * 0.8 / 2.6))
    cap = max(1, min(max_workers or len(cpus), len(cpus), len(sigs), mem_cap))
    # a container may show 256 CPUs and be throttled to a quota (cgroup cpu.max: 16 CPUs' worth on the MI355X boxes): that many
    # workers is a sweep point of its own -- more workers only share the same CPU time
    from strique_amd import dist as _sd
    quota = _sd.cpu_quota()
    counts = sorted({min(w, cap) for w in sweep} | ({cap} if cap < min(sweep) else set()) | ({min(cap, max(1, int(quota)))} if quota else set()))
    ctx = mp.get_context("spawn")
    runs = []
    best = None
    for w in counts:
        spread = [cpus[(i * len(cpus)) // w] for i in range(w)]          # evenly over the sockets / CCDs
        sample = [(s, st, False) for s, st in zip(sigs[:w], strands[:w])]
        counter = ctx.Value("i", 0)
        with ctx.Pool(w, initializer=_pin_worker, initargs=(spread, counter)) as pool:
            pool.map(_cpu_one, [(sigs[0][:20000], strands[0], True)] * w, chunksize=1)      # start the workers, load the oracle
            t0 = time.time()
            out = pool.map(_cpu_one, sample, chunksize=1)
            wall = time.time() - t0
            per_core = float(np.mean([o[0] for o in out]))
            run = {"workers": w, "reads_per_s": len(sample) / wall, "wall_s": wall, "seconds_per_read_per_core": per_core}
            runs.append(run)
            if best is None or run["reads_per_s"] > best[0]["reads_per_s"]:
                t0 = time.time()
                out_lut = pool.map(_cpu_one, [(s, st, True) for s, st, _ in sample], chunksize=1)
                wall_lut = time.time() - t0
                best = (run, out, out_lut, wall_lut, len(sample))
    run, out, out_lut, wall_lut, n_sample = best
    cores = run["workers"]; wall = run["wall_s"]; per_core = run["seconds_per_read_per_core"]
    return {"value": run["reads_per_s"], "unit": "reads/s", "cores": cores, "kind": "port",
            "host_cpu_count": host_cores, "physical_cores": len(cpus), "cpu_model": _cpu_model(),
            "cpu_quota_cores": quota,
            "cpu_quota_note": None if not quota else "the job's control group is limited to %.0f CPUs' worth of time (cpu.max): worker counts above it share that time" % quota,
            "cores_note": "best of the sweep: %d pinned worker processes, one per physical core, spread evenly over the %d physical cores" % (cores, len(cpus)),
            "sweep": runs,
            "wall_s": wall, "per_core_reads_per_s": 1.0 / per_core, "seconds_per_read_per_core": per_core,
            "lut_variant": {"value": n_sample / wall_lut, "unit": "reads/s", "wall_s": wall_lut, "workers": cores,
                            "same_counts": [int(o[1]) for o in out_lut] == [int(o[1]) for o in out],
                            "note": "same oracle with scores memoised per (level, class) instead of one pow per cell"},
            "sample": "%d reads of the timed batch (one per worker process, like STRique's --t), full 2x(N+1)x871 "
                      "float32 DP with one double pow per cell + float64 Viterbi; %.1f s per read per core, %.1f s wall"
                      % (n_sample, per_core, wall),
            "counts": [int(o[1]) for o in out]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=4096, help="reads per GPU per step (16 per CU: the forward DP launch ends without a ragged tail)")
    ap.add_argument("--read-nt", type=int, default=50000)
    ap.add_argument("--batches", type=int, default=3, help="distinct resident batches per GPU, one per step in rotation")
    ap.add_argument("--synth-workers", type=int, default=0, help="processes synthesising the reads (0: a share of the host's cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=0, help="cap on the worker processes of the CPU baseline (0: one per physical core)")
    ap.add_argument("--cpu-sweep", default="32,64,128", help="worker counts of the CPU baseline; the best one is reported as cpu_baseline.value")
    ap.add_argument("--gather-every-step", action="store_true", help="N > 1: run the result gather after every step instead of once after the last one")
    ap.add_argument("--dump-rows", default=None, help="testing: rank 0 saves the gathered table of the timed steps (numpy .npy) here")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the PCIe-inclusive strq_detect_batch measurement")
    ap.add_argument("--host-leg-batches", type=int, default=3, help="sub-batches of the PCIe-inclusive leg")
    ap.add_argument("--check", type=int, default=2, help="reads verified against the CPU oracle, all six fields (LUT variant: same bits)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true", help="testing only: all ranks use HIP device 0 (needs --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    n_batches = max(1, args.batches)
    pinned = None
    if world > 1:
        # every rank on its share of the host's CPUs (synthesis workers, upload and statistics threads inherit it)
        from strique_amd import dist as _sd
        pinned = _sd.pin_rank_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    # ---- synthetic reads (before anything touches the GPU): `batches` distinct batches per rank
    t_gen = time.time()
    from strique_amd import dist as _sdq
    synth_workers = args.synth_workers or max(1, min(32, _sdq.effective_cpus() // max(1, world)))          # CPUs the job can really use (affinity, cgroup quota) over the ranks
    first_read = rank * n_batches * args.reads
    sigs, strands, nreps = make_batches_parallel(n_batches * args.reads, args.read_nt, first_read, synth_workers)
    t_gen = time.time() - t_gen

    dist = None
    device = 0 if args.share_device else local
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(device)
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    from strique_amd.counter import repeatCounter
    from strique_amd import ffi, dist as sdist
    pm, cfg = load_inputs()
    counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=device)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    counter.add_target("c9orf72", repeat, prefix, suffix)

    off = np.zeros(len(sigs) + 1, np.int64); off[1:] = np.cumsum([len(s) for s in sigs])
    tids = [counter._classifier_for("c9orf72", s).target_id for s in strands]
    ctx = counter.ctx
    flat = np.concatenate(sigs)
    t_up = time.time()
    ctx.batch_upload(flat, off, tids)           # host -> HBM, not timed: all batches resident before the timed region
    t_up = time.time() - t_up
    del flat

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.device_synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()

    world_seen = [1]
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"

    def gather(parts):
        """The one collective of the job: the result records of `parts` = [(step, records)] of every rank to
        rank 0, positions = ((step x world) + rank) x reads + row.  Returns rank 0's table (None elsewhere)."""
        world_seen[0] = dist.get_world_size()
        k0 = parts[0][0]
        recs = np.concatenate([r for _, r in parts])
        pos = np.concatenate([((k - k0) * world + rank) * args.reads + np.arange(len(r)) for k, r in parts])
        return sdist.gather_records(recs, pos, len(parts) * world * args.reads, device=coll_dev)

    def step(k):
        bi = k % n_batches                      # a different resident batch every step
        lo, hi = bi * args.reads, (bi + 1) * args.reads
        ctx.batch_run_range(lo, hi)
        res = ctx.batch_fetch()[lo:hi]
        if dist is not None and args.gather_every_step:
            gather([(k, res)])
        return bi, res

    k_step = 0
    for _ in range(args.warmup):
        bi, res = step(k_step); k_step += 1
    if dist is not None and args.warmup and not args.gather_every_step:
        gather([(k_step - 1, res)])             # warm the collective up as well (communicator set-up is not part of a step)
    barrier()
    t0 = time.time()
    fwd_ms = 0.0; fwd_launches = 0; stage_ms = np.zeros(8); counters = np.zeros(8)
    last = {}; geoms = []; mine = []
    screen = {"ms": 0.0, "wave_steps": 0.0, "screened": 0.0, "windowed": 0.0, "whole_read": 0.0, "window_columns": 0.0, "scale": 0.0}
    gathered_rows = None
    for _ in range(args.steps):
        bi, res = step(k_step); k_step += 1
        last[bi] = res.copy()
        mine.append((k_step - 1, last[bi]))
        tm = ctx.last_timing(); cn = ctx.last_counters()
        fwd_ms += float(tm[1]); fwd_launches += int(tm[7]); stage_ms += tm
        counters[:3] += cn[:3]; counters[3:7] = cn[3:7]; counters[7] += cn[7]
        geoms.append(ctx.last_geometry())
        scr = ctx.last_screen()
        for key in ("ms", "wave_steps", "screened", "windowed", "whole_read", "window_columns"):
            screen[key] += scr[key]
        screen["scale"] = scr["scale"]
    table = None
    if dist is not None and not args.gather_every_step:
        table = gather(mine)                    # ONE gather of all steps' records, inside the timed region
    barrier()
    elapsed = time.time() - t0
    rank_report = None
    if dist is not None:
        # after the clock: every rank's digest of its own rows, its peak host memory and its CPU share to rank 0 (a second, tiny
        # collective), which checks the gathered table against the digests -- every rank's rows arrived intact, at their positions
        import hashlib
        import resource
        import torch
        digest = hashlib.sha256(np.concatenate([r for _, r in mine]).tobytes()).digest()[:8]
        mine_info = np.zeros(4, np.int64)
        mine_info[0] = int.from_bytes(digest, "little", signed=True)
        mine_info[1] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss          # KiB
        mine_info[2] = len(pinned) if pinned else 0
        mine_info[3] = device
        t = torch.from_numpy(mine_info).to(coll_dev)
        infos = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(infos, t)
        if rank == 0:
            infos = [i.cpu().numpy() for i in infos]
            rows_ok = None
            if table is not None:
                gathered_rows = len(table)
                rows_ok = True
                k0 = mine[0][0]
                for r_ in range(world):
                    part = np.concatenate([table[((k - k0) * world + r_) * args.reads:((k - k0) * world + r_ + 1) * args.reads] for k, _ in mine])
                    d = int.from_bytes(hashlib.sha256(part.tobytes()).digest()[:8], "little", signed=True)
                    rows_ok = rows_ok and d == int(infos[r_][0])
                if not rows_ok:
                    raise SystemExit("bench.py: the gathered table differs from what the ranks computed")
                if args.dump_rows:
                    np.save(args.dump_rows, table)
            rank_report = {"rows_equal_every_ranks_digest": rows_ok, "peak_host_rss_gb_per_rank": [round(float(i[1]) / 1048576.0, 3) for i in infos],
                           "cpus_pinned_per_rank": [int(i[2]) for i in infos], "hip_device_per_rank": [int(i[3]) for i in infos]}
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness spot check against the oracle (same arithmetic, memoised scores): all six fields of a row
    checked = []; check_ok = True
    if rank == 0 and args.check > 0:
        ran = sorted(last)
        for i in range(args.check):
            bi = ran[i % len(ran)]; j = i                  # read j of batch bi (different batches when several ran)
            if j >= args.reads:
                break
            g = bi * args.reads + j
            w = _cpu_check(sigs[g], strands[g])
            r = last[bi][j]
            got = (int(r["count"]), float(r["score_prefix"]), float(r["score_suffix"]), float(r["log_p"]), int(r["offset"]), int(r["ticks"]))
            want = (int(w[0]), float(w[1]), float(w[2]), float(w[3]), int(w[4]), int(w[5]))
            same = got == want
            check_ok = check_ok and same
            checked.append({"batch": bi, "read": j, "expected_repeats": nreps[g], "oracle": list(want), "gpu": list(got), "all_fields_equal": same})

    if rank == 0:
        total_reads = world * args.reads * args.steps
        value = total_reads / elapsed
        n_samples = int(off[-1]) // n_batches                      # samples of one batch (they are equally large)
        launches_per_step = max(1, fwd_launches // max(1, args.steps))
        avg_launch_s = (fwd_ms / 1e3) / max(1, fwd_launches)
        # --- what binds the dominant kernel (the forward flank DP): VALU issue.  SURVEY.md 8d: when the
        # implementation removes the per-cell trace stream, GCUPS / VALU is the binding roofline.
        prof = _profile_constants()
        geo = geoms[-1] if geoms else ctx.last_geometry()
        segs, tables, packed, R, wpe = geo["waves_per_alignment"], geo["tables_per_cu"], geo["packed"], geo["rows_per_lane"], geo["wpe"]
        # the instance the library launched (strq_last_geometry): the last template argument says which of the two kernel bodies --
        # true = round 3's, kept for launches whose flanks are all 870 rows (or fill their last lane) at 14 rows per lane
        known = R == 14 and FLANK_ROWS == 870
        kname = "align_forward_seg_kernel<%d, 6, %s, %d, %d, false, %s>" % (R, "true" if packed else "false", segs, wpe, "true" if known else "false")
        by_kernel = prof.get("valu_insts_per_wave_step_by_kernel", {})
        ipstep = by_kernel.get(kname)
        ipstep_note = None
        if ipstep is None:
            ipstep = prof.get("valu_insts_per_wave_step", {}).get("packed" if packed else "float32")
            known = prof.get("kernels", {}).get("packed" if packed else "float32", "")
            if ipstep is not None and kname not in known:
                ipstep_note = "no committed SQ_INSTS_VALU profile of this kernel instance; constant measured on %s" % known
        wave_steps_per_launch = counters[0] / max(1, fwd_launches)
        valu_peak = N_SIMD * CLOCK_HZ / 2.0 / 1e9                     # wave64 VALU instructions per second, all SIMDs (G/s)
        one = sigs[:args.reads]
        cells = sum(2 * (FLANK_ROWS + 1) * (len(s) + 1) for s in one)
        bytes_per_step = sum(2 * len(s) + 2 * (FLANK_ROWS + 1) * (len(s) + 1) for s in one)
        hbm_alg = bytes_per_step / launches_per_step / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        gcups = cells / (fwd_ms / 1e3 / max(1, args.steps)) / 1e9 if fwd_ms > 0 else None
        roof = {"bound": "valu", "unit": "G wave-instructions/s", "peak": valu_peak,
                "kernel": kname, "avg_launch_ms": avg_launch_s * 1e3, "launches_per_step": launches_per_step,
                "waves_per_alignment": segs, "score_tables_per_cu": tables, "waves_per_cu": segs * tables,
                "overlap_columns_per_step": [g["overlap_first"] for g in geoms], "overlap_worst_case": geo["overlap_worst"],
                "wave_steps_per_launch": wave_steps_per_launch,
                "columns_computed_over_columns_of_the_reads": counters[1] / max(1.0, 2.0 * n_samples * args.steps),
                "lane_utilisation": FLANK_ROWS / float(64 * R) if R else None,
                "gcups": gcups,
                "traffic": (prof.get("hbm_bytes_per_column", 0.0) * counters[1] / max(1, fwd_launches)) or None,
                "traffic_source": prof.get("traffic_source"),
                "hbm_algorithmic": {"achieved": hbm_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_alg / HBM_PEAK_GBS,
                                    "bytes_per_launch": bytes_per_step / launches_per_step,
                                    "note": "SURVEY.md 8d algorithmic bytes (int16 signal once + the reference's 1 B/cell trace) over the measured "
                                            "launch time.  These bytes never move here (checkpoint + recompute instead of a per-cell trace), "
                                            "so this figure is a work rate in the reference's units, not HBM utilisation; it may exceed 1."}}
        if gcups:
            # the bare recurrence is 4 instructions per cell (3 v_add_f32 + 1 v_max3_f32), one wave instruction = 64 cells' worth:
            # the share of the nominal issue peak that is the DP itself, without look-ups, selects, lane shifts or idle lanes
            roof["useful_achieved"] = gcups * 4.0 / 64.0
            roof["useful_frac"] = roof["useful_achieved"] / valu_peak
        if ipstep and avg_launch_s > 0:
            roof["valu_insts_per_wave_step"] = ipstep
            roof["valu_insts_source"] = prof.get("valu_source")
            if ipstep_note:
                roof["valu_insts_note"] = ipstep_note
            roof["achieved"] = ipstep * wave_steps_per_launch / avg_launch_s / 1e9
            roof["frac"] = roof["achieved"] / valu_peak
            roof["instr_per_cell"] = ipstep / (2.0 * R)                    # per lane: R rows x 2 columns per step
            roof["instr_per_cell_floor"] = 4.0                             # 3 v_add_f32 + 1 v_max3_f32
            rates = _measured_rates()
            if rates:
                # what the chip actually issues on this kernel's instruction mix (tools/valu_rates.hip): the nominal
                # peak assumes a 2-cycle issue for every instruction at 2.4 GHz
                roof["measured_issue_ceiling"] = {"unit": "G wave-instructions/s", "v_add_f32_stream": rates.get("v_add_f32_stream"),
                                                  "dp_cell_mix": rates.get("dp_cell_mix_independent"),
                                                  "dp_cell_mix_chained": rates.get("dp_cell_mix_chained"), "source": rates.get("source")}
                if rates.get("dp_cell_mix_independent"):
                    roof["frac_of_measured_ceiling"] = roof["achieved"] / rates["dp_cell_mix_independent"]
        else:
            roof["achieved"] = None; roof["frac"] = None
        if screen["ms"] > 0.5 * fwd_ms:
            # The upper-bound screen (csrc/screen_kernels.hip) ran and is the dominant kernel: an integer DP over the whole read
            # (one add + one max3 per cell) that tells the float32 DP which column windows can hold the optimum.  The line's
            # roofline is its VALU issue rate; the float32 kernel's figures over the windows move to `exact_pass`.
            exact = dict(roof)
            exact["ms_per_step"] = (fwd_ms - screen["ms"]) / args.steps
            exact["note"] = "float32 DP over the screen's windows only (plus the host planning between the two passes, which the forward stage time includes)"
            scr_ip = by_kernel.get("align_screen_kernel")
            scr_note = None
            if scr_ip is None:
                scr_ip = 111.0; scr_note = "no committed SQ_INSTS_VALU profile of this kernel: instruction count of the steady-state loop (ISA)"
            scr_launch_s = screen["ms"] / 1e3 / max(1, fwd_launches)
            scr_steps_per_launch = screen["wave_steps"] / max(1, fwd_launches)
            achieved = scr_ip * scr_steps_per_launch / scr_launch_s / 1e9 if scr_launch_s > 0 else None
            roof = {"bound": "valu", "unit": "G wave-instructions/s", "peak": valu_peak, "kernel": "align_screen_kernel",
                    "avg_launch_ms": scr_launch_s * 1e3, "launches_per_step": launches_per_step,
                    "wave_steps_per_launch": scr_steps_per_launch, "valu_insts_per_wave_step": scr_ip,
                    "valu_insts_source": prof.get("screen_valu_source") if scr_note is None else scr_note,
                    "achieved": achieved, "frac": achieved / valu_peak if achieved else None,
                    "instr_per_cell": scr_ip / 28.0, "instr_per_cell_floor": 2.0,
                    "lane_utilisation": FLANK_ROWS / float(64 * 14),
                    "gcups": gcups,
                    "traffic": (prof.get("screen_hbm_bytes_per_read_column", 0.0) * 2.0 * n_samples / launches_per_step) or None,
                    "traffic_source": prof.get("screen_traffic_source"),
                    "traffic_note": "the uint8 levels once per piece (1 B per column, overlaps included), the float32 score table of every alignment, 4 B written per 128 columns",
                    "scale": screen["scale"],
                    "alignments_screened_per_step": screen["screened"] / args.steps, "with_windows": screen["windowed"] / args.steps,
                    "whole_read": screen["whole_read"] / args.steps,
                    "window_columns_over_columns_of_the_reads": screen["window_columns"] / max(1.0, 2.0 * n_samples * args.steps),
                    "hbm_algorithmic": exact.get("hbm_algorithmic"),
                    "exact_pass": {k: exact.get(k) for k in ("kernel", "ms_per_step", "note", "waves_per_alignment", "score_tables_per_cu", "wave_steps_per_launch",
                                                              "columns_computed_over_columns_of_the_reads", "valu_insts_per_wave_step", "overlap_worst_case")}}
            if achieved:
                roof["useful_achieved"] = achieved * (2.0 * 28.0 / scr_ip) * roof["lane_utilisation"]
                roof["useful_frac"] = roof["useful_achieved"] / valu_peak
        out = {
            "metric": "reads/s for STRique 'count' on 50 kb r9.4 signals", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "value_definition": "whole-job reads/s with the int16 signals resident in HBM when the timed region starts (the benchmark "
                                "contract); `host_inclusive_reads_per_s` is the same pipeline with the signals starting in pageable host "
                                "RAM (SURVEY.md 8d's wording), uploads overlapped with the kernels",
            "vs_baseline_note": "BASELINE.md holds no published number for this metric; `vs_cpu_baseline` = value / cpu_baseline.value of this run",
            "dtype_note": "float32 flank DP over the windows an int32 upper-bound screen leaves (the screen is the dominant kernel when it runs), float64 HMM Viterbi and conditioning statistics",
            "data": "synthetic (SURVEY.md 8d recipe, seeded), int16 signals resident in HBM; %d distinct batches per GPU, a different one every step" % n_batches,
            "config": {"workload": "BASELINE configs[2]: %d reads/GPU/step, %d nt (N~%d samples), C9orf72 GGGGCC x {200,500,1000,1500,2000}"
                                   % (args.reads, args.read_nt, n_samples // max(1, args.reads)),
                       "reads_per_gpu_per_step": args.reads, "read_nt": args.read_nt, "distinct_batches_per_gpu": n_batches,
                       "sharding": "reads over ranks, no data-path collective"},
            "world_size_seen_by_the_collective": world_seen[0],
            "collective": None if world == 1 else {"what": "all_gather of the result records to rank 0 (%s)" % ("RCCL" if args.backend == "nccl" else args.backend),
                                                   "when": "after every step" if args.gather_every_step else "once, after the last timed step, inside the timed region",
                                                   "rows_on_rank_0": gathered_rows, "ranks": rank_report},
            "resident_reads_per_s": value,
            "roofline": roof,
            "stage_ms_per_step": {"conditioning": float(stage_ms[5]) / args.steps, "score_tables": float(stage_ms[0]) / args.steps,
                                  "forward_dp": float(stage_ms[1]) / args.steps, "trace": float(stage_ms[2]) / args.steps,
                                  "viterbi": float(stage_ms[6]) / args.steps},
            # the HBM-bound part of the pipeline (median filter + histogram, quantise + morphology): algorithmic bytes per sample --
            # int16 raw read, int16 filtered written and read again, uint8 levels written = 7 B -- over the stage's kernel time
            "conditioning_hbm": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                 "achieved": 7.0 * n_samples / (float(stage_ms[5]) / args.steps * 1e-3) / 1e9 if stage_ms[5] > 0 else None,
                                 "frac": 7.0 * n_samples / (float(stage_ms[5]) / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms[5] > 0 else None,
                                 "bytes_per_sample": 7, "note": "three streaming kernels + per-read statistics; no per-cell work"},
            "viterbi": {"time_steps_per_step": counters[7] / args.steps,
                        "us_per_time_step_per_wave_slot": (float(stage_ms[6]) / args.steps * 1e3) / max(1.0, counters[7] / args.steps / (8 * N_SIMD / 4))},
            "host": {"synth_s": t_gen, "synth_workers": synth_workers, "upload_s": t_up, "upload_GBs": int(off[-1]) * 2 / t_up / 1e9 if t_up > 0 else None},
            "check": checked, "check_ok": check_ok,
        }
        if not args.no_host_leg and world == 1:
            # SURVEY.md 8d quotes the metric with the signals in host RAM: the boundary's host-buffer entry
            # (strq_detect_batch) over `host_leg_batches` distinct sub-batches, samples in pageable host memory, uploads of
            # sub-batch k + 1 overlapping the kernels of sub-batch k.  Reported beside `value`, never as `value`.
            reps = max(1, args.host_leg_batches)
            pick = [(k % n_batches) for k in range(reps)]
            big = np.concatenate([s for k in pick for s in sigs[k * args.reads:(k + 1) * args.reads]])
            lens = [len(s) for k in pick for s in sigs[k * args.reads:(k + 1) * args.reads]]
            off2 = np.zeros(len(lens) + 1, np.int64); off2[1:] = np.cumsum(lens)
            tids2 = np.array([t for k in pick for t in tids[k * args.reads:(k + 1) * args.reads]], np.int32)
            t1 = time.time()
            res2 = ctx.detect_batch(big, off2, tids2, None)          # first call: sizes the device buffers for this batch (hipMalloc)
            dt_cold = time.time() - t1
            t1 = time.time()
            res2 = ctx.detect_batch(big, off2, tids2, None)          # steady state of a long-running caller
            dt = time.time() - t1
            same = all(bool(np.array_equal(res2[i * args.reads:(i + 1) * args.reads], last[k])) for i, k in enumerate(pick) if k in last)
            out["host_inclusive_reads_per_s"] = len(lens) / dt
            out["host_buffers"] = {"reads": len(lens), "sub_batches": reps, "seconds": dt, "seconds_first_call_with_allocations": dt_cold, "reads_per_s": len(lens) / dt, "GB": big.nbytes / 1e9,
                                   "same_rows_as_resident_run": same,
                                   "note": "PCIe-inclusive: int16 signals start in pageable host memory and go through a pinned staging ring; "
                                           "only the first sub-batch's upload is exposed"}
            del big
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 at N = 1 only
            ctx.close()                                       # the oracle workers want the host's memory, not the GPU's buffers
            sweep = tuple(int(v) for v in args.cpu_sweep.split(",") if v.strip()) or (32, 64, 128)
            out["cpu_baseline"] = cpu_baseline(sigs[:args.reads], strands[:args.reads], args.cpu_workers, sweep)
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] > 0 else None
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not check_ok:
        sys.stderr.write("bench.py: GPU rows differ from the CPU oracle -- see \"check\" in the JSON line\n")
        raise SystemExit(1)


def _cpu_check(sig, strand):
    """The oracle's whole row (n, score_prefix, score_suffix, log_p, offset, ticks, mod) of one read."""
    orc, opm, tc, params = _oracle_side(strand)
    res, _ = orc.detect(sig, tc, opm, params, use_lut=True)
    return res


def _measured_rates():
    p = os.path.join(ROOT, "profiles", "valu_rates.json")
    try:
        return json.load(open(p))
    except (OSError, ValueError):
        return {}


def _profile_constants():
    """Per-unit constants measured with rocprofv3 counters and committed under profiles/ (the counters
    cannot be collected inside the timed run): VALU instructions per wave-step of the forward DP
    (SQ_INSTS_VALU / wave-steps) and HBM bytes per DP column (FETCH_SIZE + WRITE_SIZE)."""
    p = os.path.join(ROOT, "profiles", "dp_constants.json")
    try:
        return json.load(open(p))
    except Exception:
        return {}


if __name__ == "__main__":
    main()
