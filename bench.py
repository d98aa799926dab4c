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


_NOISE = {}


def make_batch(pm, cfg, n_reads, read_nt, first_index, config_id=3, workload="clean"):
    """workload "clean": the SURVEY.md 8d recipe (the reference's own generator, scripts/STRique.py:182-195); "empirical":
    dwell, level offsets and sample residuals resampled from the bundled real read (strique_amd.synth.EmpiricalNoise)."""
    from strique_amd import synth
    table = synth.KmerTable(pm)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    noise = None
    if workload == "empirical":
        if "emp" not in _NOISE:
            _NOISE["emp"] = synth.EmpiricalNoise()
        noise = _NOISE["emp"]; config_id = 7
    sigs, strands, nreps = [], [], []
    for i in range(n_reads):
        nrep = REPEAT_SWEEP[(first_index + i) % len(REPEAT_SWEEP)]
        s, strand = synth.make_read(table, config_id, first_index + i, read_nt, (repeat, prefix, suffix), nrep, noise=noise)
        sigs.append(s); strands.append(strand); nreps.append(nrep)
    return sigs, strands, nreps


def _synth_chunk(args):
    """Worker process: reads [first, first + count) of the synthetic recipe."""
    first, count, read_nt, workload = args
    pm, cfg = load_inputs()
    return make_batch(pm, cfg, count, read_nt, first, workload=workload)


def make_batches_parallel(n_reads, read_nt, first_index, workers, workload="clean"):
    """The same reads as make_batch(..., first_index), synthesised by a few worker processes (a 50 kb read takes
    ~7 ms of numpy; 3 x 4096 of them would otherwise dominate the run)."""
    import multiprocessing as mp
    workers = max(1, min(workers, n_reads // 64 or 1))
    if workers == 1:
        pm, cfg = load_inputs()
        return make_batch(pm, cfg, n_reads, read_nt, first_index, workload=workload)
    per = (n_reads + workers - 1) // workers
    # chunks start at multiples of len(REPEAT_SWEEP) offsets of first_index: make_batch derives the repeat count from the index
    jobs = [(first_index + k * per, min(per, n_reads - k * per), read_nt, workload) for k in range(workers) if k * per < n_reads]
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


def memory_limit_gb():
    """The control group's memory limit in GB (cgroup v2 memory.max, v1 memory.limit_in_bytes), or None when there is none."""
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(path).read().strip()
            if v != "max" and int(v) < (1 << 60):
                return int(v) / 1e9
        except (OSError, ValueError):
            pass
    return None


def cpu_baseline(sigs, strands, max_workers=0, sweep=()):
    """The CPU oracle (reference arithmetic: full matrix, one double pow per cell, float64 Viterbi) on the host cores: pinned
    worker processes, one read per worker -- STRique's --t N (scripts/STRique.py:733-746,912).
    What the job may use is its control group's CPU quota (cpu.max: 16 CPUs' worth of time on the MI355X boxes, whatever
    /proc/cpuinfo shows), so the sweep is {quota, 2 x quota} worker processes (more only time-slice the same CPUs: round 4's sweep
    to 128 workers moved the result by < 4 %), `value` is the better of the two, `cores` is the QUOTA when it binds, and
    `per_core_reads_per_s` comes from the point with one worker per quota CPU.  `extrapolated_physical_cores` says what the
    whole machine would deliver at that per-core rate if nothing else (memory bandwidth) gave way -- an extrapolation, labelled."""
    import multiprocessing as mp
    cpus = physical_cores()
    host_cores = os.cpu_count() or 1
    # a worker holds the full (N + 1) x 871 float32 matrix and its byte trace: ~2.5 GB at 50 kb
    mem_gb = _mem_available_gb()
    lim = memory_limit_gb()
    if lim:
        mem_gb = min(mem_gb, lim)
    mem_cap = max(1, int(mem_gb * 0.8 / 2.6))
    cap = max(1, min(max_workers or len(cpus), len(cpus), len(sigs), mem_cap))
    from strique_amd import dist as _sd
    quota = _sd.cpu_quota()
    base = int(quota) if quota else len(cpus)
    counts = sorted({min(w, cap) for w in (sweep or (base, 2 * base))})
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
            run = {"workers": w, "reads_per_s": len(sample) / wall, "wall_s": wall, "seconds_per_read_per_worker": per_core}
            runs.append(run)
            if best is None or run["reads_per_s"] > best[0]["reads_per_s"]:
                t0 = time.time()
                out_lut = pool.map(_cpu_one, [(s, st, True) for s, st, _ in sample], chunksize=1)
                wall_lut = time.time() - t0
                best = (run, out, out_lut, wall_lut, len(sample))
    run, out, out_lut, wall_lut, n_sample = best
    quota_binds = bool(quota) and quota < len(cpus)
    cores = int(quota) if quota_binds else run["workers"]
    # the rate of one CPU that really runs: the sweep point with (at most) one worker per CPU the job may use
    honest = min(runs, key=lambda r: abs(r["workers"] - (int(quota) if quota_binds else r["workers"])))
    per_core_rate = honest["reads_per_s"] / min(honest["workers"], cores)
    return {"value": run["reads_per_s"], "unit": "reads/s", "cores": cores, "kind": "port",
            "workers_of_value": run["workers"],
            "host_cpu_count": host_cores, "physical_cores": len(cpus), "cpu_model": _cpu_model(),
            "cpu_quota_cores": quota, "memory_limit_gb": lim,
            "cores_note": ("the job's control group is limited to %.0f CPUs' worth of time (cpu.max): `cores` is that quota, the %d worker processes of `value` share it"
                           % (quota, run["workers"])) if quota_binds else "one pinned worker process per physical core used",
            "sweep": runs,
            "wall_s": run["wall_s"], "per_core_reads_per_s": per_core_rate,
            "seconds_per_read_per_core": 1.0 / per_core_rate if per_core_rate > 0 else None,
            "extrapolated_physical_cores": {"cores": len(cpus), "reads_per_s": per_core_rate * len(cpus),
                                            "note": "EXTRAPOLATED, not measured: per_core_reads_per_s x the machine's physical cores, i.e. what the host would "
                                                    "deliver without the quota if the per-core rate held (the full-matrix DP streams ~1.6 GB per alignment)"},
            "lut_variant": {"value": n_sample / wall_lut, "unit": "reads/s", "wall_s": wall_lut, "workers": run["workers"],
                            "same_counts": [int(o[1]) for o in out_lut] == [int(o[1]) for o in out],
                            "note": "same oracle with scores memoised per (level, class) instead of one pow per cell"},
            "sample": "%d reads of the timed batch (one per worker process, like STRique's --t), full 2x(N+1)x871 "
                      "float32 DP with one double pow per cell + float64 Viterbi; %.1f s per read per worker, %.1f s wall"
                      % (n_sample, run["seconds_per_read_per_worker"], run["wall_s"]),
            "counts": [int(o[1]) for o in out]}


def visible_devices():
    """HIP devices this process could open -- counted WITHOUT initialising the GPU (the launcher must stay clean: its rank
    processes are fresh children).  torch.cuda.device_count() does not initialise the device on this image."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: start N fresh rank processes of this script -- before this
    process makes any GPU call -- with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's JSON line, return the worst
    exit code.  Mirrors what `--t N` does in the reference (scripts/STRique.py:733-746,912: N worker processes, one dispatcher).
    Fewer visible devices than N: exit code 2 and a message -- a request for 8 GPUs never ends as an `n_gpus: 1` line."""
    import subprocess
    n = args.gpus
    if not args.share_device:
        have = visible_devices()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d requested, %d HIP device(s) visible -- not running (use --share-device --backend gloo to test N ranks on one GPU)\n" % (n, have))
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), STRQ_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0 prints the one JSON line; everything it writes to stdout is relayed as it comes
    for line in procs[0].stdout:
        sys.stdout.write(line.decode(errors="replace")); sys.stdout.flush()
    worst = 0; deadline = None
    while any(p.poll() is None for p in procs):
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs) and deadline is None:
            deadline = time.time() + 60.0          # a rank failed: the others get a minute to notice, then they are stopped
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    for p in procs:
        rc = p.returncode
        worst = max(worst, rc if rc >= 0 else 128 - rc)
    return worst


def stage_resident(ctx, counter, args, first_read, n_batches, workers, workload, keep_first, keep_all):
    """The resident batches of one rank, one at a time: synthesise -> upload (strq_batch_upload_part, 512 reads per call) -> free,
    so that a rank's peak host footprint is about one batch, not all of them plus a concatenated copy.  Returns the lengths,
    strands and planted counts of all reads and the signals worth keeping on the host: the first `keep_first` of every batch
    (spot check, CPU baseline) or all of them (`keep_all`: the PCIe-inclusive leg at N = 1)."""
    total = n_batches * args.reads
    est = int(total * args.read_nt * (9.6 if workload == "empirical" else 7.6))
    lens, strands_all, nreps_all, kept = [], [], [], {}
    t_gen = t_up = 0.0
    for b in range(n_batches):
        t0 = time.time()
        sigs, strands, nreps = make_batches_parallel(args.reads, args.read_nt, first_read + b * args.reads, workers, workload)
        t_gen += time.time() - t0
        tids = [counter._classifier_for("c9orf72", st).target_id for st in strands]
        t0 = time.time()
        for p0 in range(0, args.reads, 512):
            part = sigs[p0:p0 + 512]
            off = np.zeros(len(part) + 1, np.int64); off[1:] = np.cumsum([len(x) for x in part])
            ctx.batch_upload_part(total, est, b * args.reads + p0, np.concatenate(part), off, tids[p0:p0 + 512])
        t_up += time.time() - t0
        lens += [len(x) for x in sigs]; strands_all += strands; nreps_all += nreps
        for j in range(len(sigs) if keep_all else min(keep_first, len(sigs))):
            kept[b * args.reads + j] = sigs[j]
        del sigs
    return np.array(lens, np.int64), strands_all, nreps_all, kept, t_gen, t_up


def stage_resident_shared(ctx, counter, args, n_batches, workers, workload, rank, world, dist, keep_first):
    """N > 1: the job's reads are synthesised ONCE -- every rank makes 1 / N of a pool of `n_batches` batches (14 ms of numpy per 50 kb
    read: 3 x 4096 reads per rank on 16 / N CPUs each would be minutes of set-up at N = 8) and writes its share to /dev/shm; after a
    barrier every rank uploads the whole pool from there in its own order: rank r starts with pool batch r mod n_batches and rotates
    the reads inside a batch by r x reads / N, so no two ranks run the same launch at the same time.  Returns what stage_resident
    returns (lengths, strands, planted counts in THIS rank's order)."""
    import shutil
    total = n_batches * args.reads
    tag = [None]
    if rank == 0:
        tag[0] = "strq_bench_%d_%d" % (os.getpid(), int(time.time()))
        # where the pool goes: shared memory if it has the room (a container's /dev/shm can be 64 MB), else a temporary directory
        need = int(total * args.read_nt * (9.6 if workload == "empirical" else 7.6) * 2 * 1.15)
        tag.append(None)
        for base in ("/dev/shm", os.environ.get("TMPDIR") or "/tmp", "/tmp"):
            try:
                st = os.statvfs(base)
                if st.f_bavail * st.f_frsize > need:
                    tag[1] = base; break
            except OSError:
                continue
    else:
        tag.append(None)
    dist.broadcast_object_list(tag, src=0)
    if tag[1] is None:          # nowhere to put a shared pool: every rank makes its own reads after all
        return None
    pool_dir = os.path.join(tag[1], tag[0])
    os.makedirs(pool_dir, exist_ok=True)
    t0 = time.time()
    lo, hi = total * rank // world, total * (rank + 1) // world
    for p0 in range(lo, hi, 512):
        cnt = min(512, hi - p0)
        sigs, strands, nreps = make_batches_parallel(cnt, args.read_nt, p0, workers, workload)
        off = np.zeros(cnt + 1, np.int64); off[1:] = np.cumsum([len(x) for x in sigs])
        np.save(os.path.join(pool_dir, "sig_%08d.npy" % p0), np.concatenate(sigs))
        np.savez(os.path.join(pool_dir, "meta_%08d.npz" % p0), off=off, strands=np.array([st == "+" for st in strands]), nreps=np.array(nreps, np.int64))
        del sigs
    t_gen = time.time() - t0
    dist.barrier()
    # the pool's parts, in pool order
    parts = sorted(int(f[4:12]) for f in os.listdir(pool_dir) if f.startswith("sig_"))
    meta = {}
    for p0 in parts:
        m = np.load(os.path.join(pool_dir, "meta_%08d.npz" % p0))
        meta[p0] = (m["off"], m["strands"], m["nreps"])
    pool_len = np.concatenate([np.diff(meta[p0][0]) for p0 in parts]); pool_plus = np.concatenate([meta[p0][1] for p0 in parts]); pool_nrep = np.concatenate([meta[p0][2] for p0 in parts])
    part_of = np.concatenate([np.full(len(meta[p0][2]), p0) for p0 in parts]); idx_in = np.concatenate([np.arange(len(meta[p0][2])) for p0 in parts])
    order = np.concatenate([((b + rank) % n_batches) * args.reads + (np.arange(args.reads) + rank * args.reads // world) % args.reads for b in range(n_batches)])
    est = int(pool_len.sum())
    t0 = time.time()
    lens, strands_all, nreps_all, kept = [], [], [], {}
    cache = {}
    for q0 in range(0, total, 512):
        ids = order[q0:q0 + 512]
        part = []
        for g in ids:
            p0 = int(part_of[g])
            if p0 not in cache:
                if len(cache) > 3:
                    cache.pop(next(iter(cache)))
                cache[p0] = np.load(os.path.join(pool_dir, "sig_%08d.npy" % p0), mmap_mode="r")
            o = meta[p0][0]; j = int(idx_in[g])
            part.append(np.asarray(cache[p0][o[j]:o[j + 1]]))
        st = ["+" if pool_plus[g] else "-" for g in ids]
        tids = [counter._classifier_for("c9orf72", x).target_id for x in st]
        off = np.zeros(len(part) + 1, np.int64); off[1:] = np.cumsum([len(x) for x in part])
        ctx.batch_upload_part(total, est, q0, np.concatenate(part), off, tids)
        lens += [len(x) for x in part]; strands_all += st; nreps_all += [int(pool_nrep[g]) for g in ids]
        for k, x in enumerate(part):
            if (q0 + k) % args.reads < keep_first:
                kept[q0 + k] = x.copy()
    t_up = time.time() - t0
    cache.clear()
    dist.barrier()
    if rank == 0:
        shutil.rmtree(pool_dir, ignore_errors=True)
    return np.array(lens, np.int64), strands_all, nreps_all, kept, t_gen, t_up


class Leg(object):
    """Measurements of a run of steps on one context (what the roofline blocks are priced with)."""

    def __init__(self):
        self.fwd_ms = 0.0; self.fwd_launches = 0; self.stage_ms = np.zeros(8); self.counters = np.zeros(8)
        self.geoms = []; self.last = {}; self.mine = []; self.elapsed = 0.0; self.steps = 0
        self.screen = {"ms": 0.0, "wave_steps": 0.0, "screened": 0.0, "windowed": 0.0, "whole_read": 0.0, "window_columns": 0.0, "scale": 0.0, "candidate_chunks": 0.0}
        self.second_round = [0, 0]; self.screen_mode = None; self.screen_merge = 0
        self.overlap = {"viterbi_ms": 0.0, "under_screen_ms": 0.0, "under_alignment_stage_ms": 0.0, "sub_batches": 0.0}

    def rows(self, k, bi, res):
        self.last[bi] = res.copy()
        self.mine.append((k, self.last[bi]))

    def late_viterbi(self, ms, time_steps):
        """The Viterbi launches of the LAST step, which the final fetch waited for (strq_batch_fetch_range)."""
        self.stage_ms[6] += ms; self.counters[7] += time_steps

    def stats(self, ctx):
        """After a run call: its conditioning / alignment figures, and the Viterbi figures of the sub-batch it took the rows of
        (the one before: the library keeps two sub-batches in flight)."""
        tm = ctx.last_timing(); cn = ctx.last_counters()
        self.fwd_ms += float(tm[1]); self.fwd_launches += int(tm[7]); self.stage_ms += tm
        self.counters[:3] += cn[:3]; self.counters[3:7] = cn[3:7]; self.counters[7] += cn[7]
        self.geoms.append(ctx.last_geometry())
        scr = ctx.last_screen()
        for key in ("ms", "wave_steps", "screened", "windowed", "whole_read", "window_columns", "candidate_chunks"):
            self.screen[key] += scr[key]
        self.screen["scale"] = scr["scale"]
        self.screen_mode = scr.get("mode", self.screen_mode); self.screen_merge = scr.get("merge", 0)
        sr = ctx.last_second_round()
        self.second_round[0] += sr[0]; self.second_round[1] += sr[1]
        ov = ctx.last_overlap()
        for key in self.overlap:
            self.overlap[key] += ov[key]
        self.steps += 1


def run_steps(ctx, leg, reads, n_batches, steps, k0, each=None):
    """`steps` steps over the resident batches in rotation.  A step = strq_batch_run_range over one batch; its rows are fetched
    (strq_batch_fetch_range) after the NEXT step has been queued -- the library runs a step's HMM Viterbi launches under the next
    step's conditioning and flank alignments -- and the last step's rows before this returns: every kernel and every row of the
    `steps` steps lies between the caller's two clock readings.  `each(k, rows)`: called with every step's rows (the gather
    after every step).  Returns k0 + steps."""
    prev = None
    if n_batches < 2 and each is None:
        each = lambda k, rows: None          # one resident batch: a step's rows would be overwritten by the next step's -- fetched right away
    for i in range(steps):
        k = k0 + i; bi = k % n_batches
        ctx.batch_run_range(bi * reads, (bi + 1) * reads)
        leg.stats(ctx)
        if each is not None:          # rows wanted right away: no second step in flight
            v0 = float(ctx.last_timing()[6]); c0 = float(ctx.last_counters()[7])
            rows = ctx.batch_fetch_range(bi * reads, (bi + 1) * reads)
            leg.late_viterbi(float(ctx.last_timing()[6]) - v0, float(ctx.last_counters()[7]) - c0)
            leg.rows(k, bi, rows); each(k, rows)
            continue
        if prev is not None:
            leg.rows(prev[0], prev[1], ctx.batch_fetch_range(prev[1] * reads, (prev[1] + 1) * reads))
        prev = (k, bi)
    if prev is not None:
        v0 = float(ctx.last_timing()[6]); c0 = float(ctx.last_counters()[7])
        leg.rows(prev[0], prev[1], ctx.batch_fetch_range(prev[1] * reads, (prev[1] + 1) * reads))
        leg.late_viterbi(float(ctx.last_timing()[6]) - v0, float(ctx.last_counters()[7]) - c0)
    return k0 + steps


def run_leg(ctx, reads, n_batches, steps, warmup, k0=0):
    """`warmup` untimed + `steps` timed steps over the resident batches in rotation, on one process (the extra legs at N = 1)."""
    k = run_steps(ctx, Leg(), reads, n_batches, warmup, k0)
    ctx.batch_fetch(); ctx.device_synchronize()
    leg = Leg()
    t0 = time.time()
    run_steps(ctx, leg, reads, n_batches, steps, k)
    ctx.device_synchronize()
    leg.elapsed = time.time() - t0
    return leg


# kernel, VALU instructions per wave-step of its steady-state loop (ISA; the committed SQ_INSTS_VALU profile takes precedence), DP cells per lane and step
# (a merged cell of the coarse screens covers 2 / 3 / 6 cells of the reference's matrix), cells of the reference's matrix per lane and step, busy lanes
SCREEN_KERNELS = {("fine", 0): ("align_screen_kernel", 98.0, 28.0, 28.0, FLANK_ROWS / float(64 * 14)),
                  ("fine", 1): ("align_screen1_kernel", 161.0, 60.0, 60.0, 58.0 / 64.0),
                  ("coarse", 2): ("align_screen2_kernel", 101.0, 30.0, 60.0, 58.0 / 64.0),
                  ("coarse", 3): ("align_screen3_kernel", 81.0, 20.0, 60.0, 58.0 / 64.0),
                  ("coarse", 6): ("align_screen6_kernel", 61.0, 10.0, 60.0, 58.0 / 64.0)}


def roofline_blocks(leg, lens_one_batch, prof, ctx_geo=None):
    """`roofline` of a leg: the dominant kernel of its forward stage -- the float32 DP, or the integer screen in front of it when
    that ran (then the float32 kernel's figures over the windows sit under `exact_pass`) -- priced as VALU issue: wave
    instructions per second against 1024 SIMDs x 2.4 GHz / 2, instruction counts per wave-step from the committed SQ_INSTS_VALU
    profiles (profiles/dp_constants.json), wave-steps counted by the library in this run."""
    steps = max(1, leg.steps)
    fwd_ms, fwd_launches, counters, geoms, screen = leg.fwd_ms, leg.fwd_launches, leg.counters, leg.geoms, leg.screen
    n_samples = int(lens_one_batch.sum())
    launches_per_step = max(1, fwd_launches // steps)
    avg_launch_s = (fwd_ms / 1e3) / max(1, fwd_launches)
    geo = geoms[-1] if geoms else ctx_geo
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
        known_k = prof.get("kernels", {}).get("packed" if packed else "float32", "")
        if ipstep is not None and kname not in known_k:
            ipstep_note = "no committed SQ_INSTS_VALU profile of this kernel instance; constant measured on %s" % known_k
    wave_steps_per_launch = counters[0] / max(1, fwd_launches)
    valu_peak = N_SIMD * CLOCK_HZ / 2.0 / 1e9                     # wave64 VALU instructions per second, all SIMDs (G/s)
    cells = float(np.sum(2 * (FLANK_ROWS + 1) * (lens_one_batch + 1)))
    bytes_per_step = float(np.sum(2 * lens_one_batch + 2 * (FLANK_ROWS + 1) * (lens_one_batch + 1)))
    hbm_alg = bytes_per_step / launches_per_step / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    gcups = cells / (fwd_ms / 1e3 / steps) / 1e9 if fwd_ms > 0 else None
    roof = {"bound": "valu", "unit": "G wave-instructions/s", "peak": valu_peak,
            "kernel": kname, "avg_launch_ms": avg_launch_s * 1e3, "launches_per_step": launches_per_step,
            "waves_per_alignment": segs, "score_tables_per_cu": tables, "waves_per_cu": segs * tables,
            "overlap_columns_per_step": [g["overlap_first"] for g in geoms], "overlap_worst_case": geo["overlap_worst"],
            "wave_steps_per_launch": wave_steps_per_launch,
            "columns_computed_over_columns_of_the_reads": counters[1] / max(1.0, 2.0 * n_samples * steps),
            "lane_utilisation": FLANK_ROWS / float(64 * R) if R else None,
            "gcups": gcups, "gcups_note": "cells of the reference's matrices (2 x 871 x (N + 1) per read) per second of the forward stage",
            "traffic": (prof.get("hbm_bytes_per_column", 0.0) * counters[1] / max(1, fwd_launches)) or None,
            "traffic_source": prof.get("traffic_source"),
            "hbm_algorithmic": {"achieved": hbm_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "work_rate_over_hbm_peak": hbm_alg / HBM_PEAK_GBS,
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
        # A screen (csrc/screen_kernels.hip) ran and is the dominant kernel: an integer DP over the whole read that tells the
        # float32 DP which column windows can hold the optimum.  The leg's roofline is its VALU issue rate; the float32 kernel's
        # figures over the windows move to `exact_pass`.
        exact = dict(roof)
        exact["ms_per_step"] = (fwd_ms - screen["ms"]) / steps
        exact["note"] = "float32 DP over the screen's windows only (plus the host planning between the two passes, which the forward stage time includes)"
        mode = leg.screen_mode or "fine"
        sk_name, sk_isa, sk_cells, sk_ref_cells, sk_util = SCREEN_KERNELS.get((mode, leg.screen_merge), SCREEN_KERNELS[("fine", 0)])
        scr_ip = by_kernel.get(sk_name)
        scr_note = None
        if scr_ip is None:
            scr_ip = sk_isa; scr_note = "no committed SQ_INSTS_VALU profile of this kernel: instruction count of the steady-state loop (ISA)"
        # one screen launch per sub-batch of (at most) 4096 reads
        scr_launches = steps * max(1, -(-len(lens_one_batch) // 4096))
        scr_launch_s = screen["ms"] / 1e3 / scr_launches
        scr_steps_per_launch = screen["wave_steps"] / scr_launches
        achieved = scr_ip * scr_steps_per_launch / scr_launch_s / 1e9 if scr_launch_s > 0 else None
        roof = {"bound": "valu", "unit": "G wave-instructions/s", "peak": valu_peak, "kernel": sk_name, "screen_mode": mode,
                "flank_rows_per_dp_row": max(1, leg.screen_merge), "alignments_per_wave": 2 if leg.screen_merge else 1,
                "avg_launch_ms": scr_launch_s * 1e3, "launches_per_step": scr_launches // steps,
                "wave_steps_per_launch": scr_steps_per_launch, "valu_insts_per_wave_step": scr_ip,
                "valu_insts_source": prof.get("screen_valu_source") if scr_note is None else scr_note,
                "achieved": achieved, "frac": achieved / valu_peak if achieved else None,
                "instr_per_cell": scr_ip / sk_cells, "instr_per_cell_floor": 2.0, "instr_per_cell_of_the_reference_matrix": scr_ip / sk_ref_cells,
                "lane_utilisation": sk_util,
                "gcups": gcups, "gcups_note": "cells of the reference's matrices (2 x 871 x (N + 1) per read) per second of the forward stage: the screen "
                                              "covers every one of them with an integer bound, the float32 DP recomputes the windows",
                "traffic": (prof.get("screen_hbm_bytes_per_read_column", 0.0) * 2.0 * n_samples / max(1, scr_launches // steps)) or None,
                "traffic_source": prof.get("screen_traffic_source"),
                "traffic_note": "the uint8 levels once per piece (1 B per column, overlaps included; the coarse screen reads them once for both alignments of a read), the float32 score "
                                "table of every alignment, 4 B written per 128 columns and alignment; measured with the kernel named in traffic_source",
                "scale": screen["scale"],
                "alignments_screened_per_step": screen["screened"] / steps, "with_windows": screen["windowed"] / steps,
                "whole_read": screen["whole_read"] / steps,
                "window_columns_over_columns_of_the_reads": screen["window_columns"] / max(1.0, 2.0 * n_samples * steps),
                "hbm_algorithmic": exact.get("hbm_algorithmic"),
                "exact_pass": {k: exact.get(k) for k in ("kernel", "ms_per_step", "note", "waves_per_alignment", "score_tables_per_cu", "wave_steps_per_launch",
                                                          "columns_computed_over_columns_of_the_reads", "valu_insts_per_wave_step", "overlap_worst_case", "frac")}}
        if achieved:
            roof["useful_achieved"] = achieved * (2.0 * sk_cells / scr_ip) * roof["lane_utilisation"]
            roof["useful_frac"] = roof["useful_achieved"] / valu_peak
        ov = leg.overlap
        vit_ip = prof.get("viterbi_valu_insts_per_time_step")
        if achieved and vit_ip and ov["under_screen_ms"] > 0 and ov["viterbi_ms"] > 0 and scr_launch_s > 0:
            # Two sub-batches in flight: the previous step's Viterbi launch (one wave per SIMD) runs on the same SIMDs while this launch
            # lasts.  The issue rate of the launch window is the sum of both kernels' instructions over it; the screen's own rate is kept
            # beside it, and the rate of the screen ALONE on the GPU comes from the serial leg (`kernel_alone`, filled in by main()).
            vit_instr_per_step = vit_ip * leg.counters[7] / steps
            in_window = vit_instr_per_step * ov["under_screen_ms"] / ov["viterbi_ms"]
            co = in_window / max(1, scr_launches // steps) / scr_launch_s / 1e9
            roof["achieved_own"] = achieved; roof["frac_own"] = achieved / valu_peak
            roof["co_running"] = {"kernel": (prof.get("viterbi_kernel", "viterbi_g2_kernel<false, 8, 2>")).replace("<false, 8,", "<false, 4,") + " of the previous step, second stream",
                                  "ms_under_this_launch_per_step": ov["under_screen_ms"] / steps, "viterbi_ms_per_step": ov["viterbi_ms"] / steps,
                                  "achieved": co, "valu_insts_per_time_step": vit_ip}
            roof["achieved"] = achieved + co; roof["frac"] = (achieved + co) / valu_peak
            # float64 instructions hold the issue port for two slots (profiles/r02_valu_rates.md: every f64 VALU instruction at half rate): the share of
            # the window's issue CYCLES the two kernels fill -- what is left is what neither of them could use
            roof["issue_cycles_frac"] = min(1.0, (achieved + 2.0 * co) / valu_peak)
            roof["kernel"] = sk_name + " + co-running " + roof["co_running"]["kernel"].split(" of ")[0]
            roof["achieved_definition"] = "VALU instructions of BOTH kernels issued inside this launch's window (HIP events) / its duration"
    return roof


def viterbi_roofline(leg, prof):
    """`roofline_viterbi`: the flanked-repeat HMM decode (viterbi_g2_kernel), float64 throughout -- VALU instructions per time step
    from the committed SQ_INSTS_VALU profile x the time steps the library counted in this run / the Viterbi stage time, against the
    nominal issue peak and against the half rate at which gfx950 issues every float64 instruction (profiles/r02_valu_rates.md)."""
    steps = max(1, leg.steps)
    vit_ms = float(leg.stage_ms[6]) / steps
    tsteps = leg.counters[7] / steps
    valu_peak = N_SIMD * CLOCK_HZ / 2.0 / 1e9
    ip = prof.get("viterbi_valu_insts_per_time_step")
    out = {"bound": "valu", "unit": "G wave-instructions/s", "peak": valu_peak, "kernel": prof.get("viterbi_kernel", "viterbi_g2_kernel<false, 8, 2>"),
           "ms_per_step": vit_ms, "time_steps_per_step": tsteps,
           "us_per_time_step_per_wave_slot": (vit_ms * 1e3) / max(1.0, tsteps / (8 * N_SIMD / 4)),
           "valu_insts_per_time_step": ip, "valu_insts_source": prof.get("viterbi_valu_source")}
    if leg.overlap["under_alignment_stage_ms"] > 0:
        out["kernel"] = out["kernel"].replace("<false, 8,", "<false, 4,")
        out["co_running_share"] = leg.overlap["under_alignment_stage_ms"] / max(1e-9, leg.overlap["viterbi_ms"])
        out["note"] = "the launch runs on a second stream under the next step's flank-alignment kernels (one wave per SIMD, sharing them): ms_per_step is its duration there, not GPU time it has to itself"
    if ip and vit_ms > 0:
        out["achieved"] = ip * tsteps / (vit_ms * 1e-3) / 1e9
        out["frac"] = out["achieved"] / valu_peak
        out["float64_issue_peak"] = valu_peak / 2.0
        out["frac_of_float64_issue"] = out["achieved"] / (valu_peak / 2.0)
    else:
        out["achieved"] = None; out["frac"] = None
    return out


def leg_summary(leg, reads, lens_one_batch, prof, nreps=None, n_batches=1):
    steps = max(1, leg.steps)
    out = {"value": reads * steps / leg.elapsed if leg.elapsed > 0 else None, "unit": "reads/s", "steps": leg.steps, "ms_per_step": leg.elapsed / steps * 1e3,
           "overlap_ms_per_step": {k: v / steps for k, v in leg.overlap.items() if k != "sub_batches"},
           "stage_ms_per_step": {"conditioning": float(leg.stage_ms[5]) / steps, "score_tables": float(leg.stage_ms[0]) / steps,
                                 "forward_dp": float(leg.stage_ms[1]) / steps, "trace": float(leg.stage_ms[2]) / steps, "viterbi": float(leg.stage_ms[6]) / steps},
           "roofline": roofline_blocks(leg, lens_one_batch, prof),
           "roofline_viterbi": viterbi_roofline(leg, prof),
           "screen": {"mode": leg.screen_mode, "ms_per_step": leg.screen["ms"] / steps,
                      "share_of_alignments_with_windows": leg.screen["windowed"] / max(1.0, leg.screen["screened"]) if leg.screen["screened"] else 0.0,
                      "alignments_screened_per_step": leg.screen["screened"] / steps, "wave_steps_per_step": leg.screen["wave_steps"] / steps,
                      "window_columns_over_columns_of_the_reads": leg.screen["window_columns"] / max(1.0, 2.0 * float(lens_one_batch.sum()) * steps),
                      "candidate_chunks_per_alignment": leg.screen["candidate_chunks"] / max(1.0, leg.screen["screened"])},
           "second_round_share": leg.second_round[0] / max(1, leg.second_round[1])}
    if nreps is not None and leg.last:
        ok = tot = 0
        for bi, res in leg.last.items():
            want = np.array(nreps[bi * reads:(bi + 1) * reads])
            ok += int(np.sum(np.abs(res["count"] - want) <= np.maximum(2, 0.01 * want))); tot += len(want)
        out["planted_count_recovered"] = {"within_max_2_or_1_percent": ok, "of": tot}
    return out


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
    ap.add_argument("--cpu-sweep", default="", help="worker counts of the CPU baseline (default: the job's CPU quota and twice that); the best one is cpu_baseline.value")
    ap.add_argument("--gather-every-step", action="store_true", help="N > 1: run the result gather after every step instead of once after the last one")
    ap.add_argument("--dump-rows", default=None, help="testing: rank 0 saves the gathered table of the timed steps (numpy .npy) here")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the PCIe-inclusive strq_detect_batch measurement")
    ap.add_argument("--host-leg-batches", type=int, default=8, help="sub-batches of the PCIe-inclusive leg (the resident batches in rotation: only the first one's upload is exposed)")
    ap.add_argument("--no-legs", action="store_true", help="skip the A/B legs (no screen, fine screen, degraded reads) that follow the headline measurement at N = 1")
    ap.add_argument("--leg-steps", type=int, default=3, help="timed steps of every A/B leg (after one untimed step)")
    ap.add_argument("--degraded-reads", type=int, default=0, help="reads of the degraded-read leg (0: --reads)")
    ap.add_argument("--workload", default="clean", choices=["clean", "empirical"], help="reads of the headline measurement (clean = BASELINE configs[2])")
    ap.add_argument("--check", type=int, default=2, help="reads verified against the CPU oracle, all six fields (LUT variant: same bits)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the result gather for N > 1 (nccl = RCCL)")
    ap.add_argument("--private-reads", action="store_true", help="N > 1: every rank synthesises its own batches (default: one pool for the job, made once, 1 / N per rank, read by all in different orders)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: --reads per GPU per step whatever N (the default); strong: --total-reads per step for the whole job, 1 / N of them per GPU")
    ap.add_argument("--total-reads", type=int, default=0, help="--scaling strong: reads of one step over all GPUs (BASELINE configs[3]: 100000)")
    ap.add_argument("--detail", default=None, help="where rank 0 writes the full record (default: gpurun_out/bench_detail.json); the line on stdout is the compact one")
    ap.add_argument("--share-device", action="store_true", help="testing only: all ranks use HIP device 0 (needs --backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus, or plainly (bench.py starts its own rank processes)" % (args.gpus, world))
    n_batches = max(1, args.batches)
    if args.scaling == "strong":
        # a fixed job: --total-reads per step, rank r takes its 1 / N (rounded up: the last ranks' rows of padding are real reads too, so the
        # job never does less than it says); two resident batches per GPU so that the footprint stays that of the weak mode
        if args.total_reads <= 0:
            raise SystemExit("bench.py: --scaling strong needs --total-reads")
        args.reads = -(-args.total_reads // world)
        n_batches = max(1, min(n_batches, 2))
    device = 0 if args.share_device else local
    if world > 1 and not args.share_device:
        have = visible_devices()
        if device >= have:
            sys.stderr.write("bench.py: rank %d needs HIP device %d, %d visible\n" % (rank, device, have))
            raise SystemExit(2)
    pinned = None
    if world > 1:
        # every rank on its share of the host's CPUs (synthesis workers, upload and statistics threads inherit it)
        from strique_amd import dist as _sd
        pinned = _sd.pin_rank_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    from strique_amd import dist as _sdq
    synth_workers = args.synth_workers or max(1, min(32, _sdq.effective_cpus() // max(1, world)))          # CPUs the job can really use (affinity, cgroup quota) over the ranks

    # ---- the process group: gloo is the control plane (barriers, the max over ranks); the result gather runs over RCCL
    # (--backend nccl) when that initialises on every rank, over gloo otherwise -- said in the line, never fatal
    dist = None; coll_group = None; coll_dev = "cpu"; coll_backend = None; nccl_error = None
    if world > 1:
        import datetime
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
        coll_backend = "gloo"
        if args.backend == "nccl":
            ok = 1
            try:
                torch.cuda.set_device(device)
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(minutes=5))
                t = torch.ones(1, device="cuda")
                dist.all_reduce(t, group=g)
                torch.cuda.synchronize()
                if int(t.item()) != world:
                    raise RuntimeError("all_reduce over RCCL returned %r for %d ranks" % (t.item(), world))
            except Exception as e:          # noqa: BLE001 -- whatever RCCL raises: the job finishes over gloo
                ok = 0; nccl_error = "%s: %s" % (type(e).__name__, str(e)[:300])
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                coll_group = g; coll_dev = "cuda"; coll_backend = "nccl"
            elif nccl_error is None:
                nccl_error = "RCCL did not initialise on another rank"

    from strique_amd.counter import repeatCounter
    from strique_amd import ffi, dist as sdist
    pm, cfg = load_inputs()
    counter = repeatCounter(pm, align_config=cfg["align"], HMM_config=cfg["HMM"], device=device)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    counter.add_target("c9orf72", repeat, prefix, suffix)
    ctx = counter.ctx

    # ---- synthetic reads, one resident batch at a time (synthesise -> upload -> free): `batches` distinct batches per rank
    first_read = rank * n_batches * args.reads
    want_host_leg = (not args.no_host_leg) and world == 1
    keep_first = max(args.check, 0)
    staged = None
    if world > 1 and not args.private_reads:
        staged = stage_resident_shared(ctx, counter, args, n_batches, synth_workers, args.workload, rank, world, dist, keep_first)
        if staged is None:
            args.private_reads = True
    if staged is not None:
        lens, strands, nreps, kept, t_gen, t_up = staged
    else:
        lens, strands, nreps, kept, t_gen, t_up = stage_resident(ctx, counter, args, first_read, n_batches, synth_workers, args.workload,
                                                                  keep_first=max(keep_first, 256 if (world == 1 and not args.no_cpu_baseline) else 0), keep_all=want_host_leg)
    tids = [counter._classifier_for("c9orf72", st).target_id for st in strands]

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.device_synchronize()
        if dist is not None and coll_dev == "cuda":
            import torch
            torch.cuda.synchronize()

    world_seen = [1]

    def gather(parts):
        """The one collective of the job: the result records of `parts` = [(step, records)] of every rank to
        rank 0, positions = ((step x world) + rank) x reads + row.  Returns rank 0's table (None elsewhere)."""
        world_seen[0] = dist.get_world_size(coll_group)
        k0 = parts[0][0]
        recs = np.concatenate([r for _, r in parts])
        pos = np.concatenate([((k - k0) * world + rank) * args.reads + np.arange(len(r)) for k, r in parts])
        return sdist.gather_records(recs, pos, len(parts) * world * args.reads, device=coll_dev, group=coll_group)

    each = (lambda k, rows: gather([(k, rows)])) if (dist is not None and args.gather_every_step) else None
    warm_leg = Leg()
    k_step = run_steps(ctx, warm_leg, args.reads, n_batches, args.warmup, 0, each)          # a different resident batch every step
    ctx.batch_fetch()
    if dist is not None and args.warmup and not args.gather_every_step:
        gather([warm_leg.mine[-1]])             # warm the collective up as well (communicator set-up is not part of a step)
    barrier()
    t0 = time.time()
    main_leg = Leg()
    gathered_rows = None
    k_step = run_steps(ctx, main_leg, args.reads, n_batches, args.steps, k_step, each)
    mine = main_leg.mine; last = main_leg.last
    table = None
    if dist is not None and not args.gather_every_step:
        table = gather(mine)                    # ONE gather of all steps' records, inside the timed region
    barrier()
    elapsed = time.time() - t0
    rank_report = None
    import resource
    if dist is not None:
        # after the clock: every rank's digest of its own rows, its peak host memory and its CPU share to rank 0 (a second, tiny
        # collective), which checks the gathered table against the digests -- every rank's rows arrived intact, at their positions
        import hashlib
        import torch
        digest = hashlib.sha256(np.concatenate([r for _, r in mine]).tobytes()).digest()[:8]
        mine_info = np.zeros(5, np.int64)
        mine_info[0] = int.from_bytes(digest, "little", signed=True)
        mine_info[1] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss          # KiB
        mine_info[2] = len(pinned) if pinned else 0
        mine_info[3] = device
        mine_info[4] = int(elapsed * 1e6)          # this rank's own time over the steps (the line's ms_per_step is the maximum)
        t = torch.from_numpy(mine_info)
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
                           "cpus_pinned_per_rank": [int(i[2]) for i in infos], "hip_device_per_rank": [int(i[3]) for i in infos],
                           "ms_per_step_per_rank": [float(i[4]) / 1e3 / max(1, args.steps) for i in infos]}
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    main_leg.elapsed = elapsed
    peak_rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0

    # ---- correctness spot check against the oracle (same arithmetic, memoised scores): all six fields of a row
    checked = []; check_ok = True
    if rank == 0 and args.check > 0:
        ran = sorted(last)
        for i in range(args.check):
            bi = ran[i % len(ran)]; j = i                  # read j of batch bi (different batches when several ran)
            if j >= args.reads:
                break
            g = bi * args.reads + j
            w = _cpu_check(kept[g], strands[g])
            r = last[bi][j]
            got = (int(r["count"]), float(r["score_prefix"]), float(r["score_suffix"]), float(r["log_p"]), int(r["offset"]), int(r["ticks"]))
            want = (int(w[0]), float(w[1]), float(w[2]), float(w[3]), int(w[4]), int(w[5]))
            same = got == want
            check_ok = check_ok and same
            checked.append({"batch": bi, "read": j, "expected_repeats": nreps[g], "oracle": list(want), "gpu": list(got), "all_fields_equal": same})

    if rank == 0:
        total_reads = world * args.reads * args.steps
        value = total_reads / elapsed
        prof = _profile_constants()
        one = lens[:args.reads]
        n_samples = int(lens.sum()) // n_batches                      # samples of one batch (they are equally large)
        stage_ms = main_leg.stage_ms
        roof = roofline_blocks(main_leg, one, prof, ctx.last_geometry())
        mem_lim = memory_limit_gb()
        out = {
            "metric": "reads/s for STRique 'count' on 50 kb r9.4 signals", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "value_definition": "whole-job reads/s with the int16 signals resident in HBM when the timed region starts (the benchmark "
                                "contract); `host_inclusive_reads_per_s` is the same pipeline with the signals starting in pageable host "
                                "RAM (SURVEY.md 8d's wording), uploads overlapped with the kernels",
            "vs_baseline_note": "BASELINE.md holds no published number for this metric; `vs_cpu_baseline` = value / cpu_baseline.value of this run",
            "dtype_note": "float32 flank DP over the windows an int32 upper-bound screen leaves (the screen is the dominant kernel when it runs), float64 HMM Viterbi and conditioning statistics",
            "data": "synthetic (%s, seeded), int16 signals resident in HBM; %d distinct batches per GPU, a different one every step"
                    % ("SURVEY.md 8d recipe" if args.workload == "clean" else "noise resampled from the bundled real read: strique_amd.synth.EmpiricalNoise", n_batches),
            "config": {"workload": "BASELINE configs[2]: %d reads/GPU/step, %d nt (N~%d samples), C9orf72 GGGGCC x {200,500,1000,1500,2000}%s"
                                   % (args.reads, args.read_nt, n_samples // max(1, args.reads), "" if args.workload == "clean" else " -- EMPIRICAL noise, not the BASELINE recipe"),
                       "reads_per_gpu_per_step": args.reads, "total_reads": (args.total_reads if args.scaling == "strong" else None), "read_nt": args.read_nt, "distinct_batches_per_gpu": n_batches,
                       "reads_pool": None if world == 1 else ("one pool per rank" if args.private_reads else "one pool for the job (every rank makes 1/N of it, all ranks run all of it in rotated orders)"),
                       "boundary": "int16 signals resident in HBM when the clock starts (benchmark contract); host_inclusive_reads_per_s = the same from pageable host RAM (SURVEY.md 8d)",
                       "sharding": "reads over ranks, no data-path collective"},
            "world_size_seen_by_the_collective": world_seen[0],
            "collective": None if world == 1 else {"what": "all_gather of the result records to rank 0 (%s)" % ("RCCL" if coll_backend == "nccl" else coll_backend),
                                                   "backend": coll_backend, "backend_requested": args.backend, "nccl_error": nccl_error,
                                                   "control_plane": "gloo (barriers, max over ranks)",
                                                   "launched_by": "bench.py itself (fresh rank processes)" if os.environ.get("STRQ_BENCH_SELF_LAUNCHED") else "an external launcher (RANK / WORLD_SIZE in the environment)",
                                                   "when": "after every step" if args.gather_every_step else "once, after the last timed step, inside the timed region",
                                                   "rows_on_rank_0": gathered_rows, "ranks": rank_report},
            "resident_reads_per_s": value,
            "roofline": roof,
            "roofline_viterbi": viterbi_roofline(main_leg, prof),
            "stage_ms_per_step": {"conditioning": float(stage_ms[5]) / args.steps, "score_tables": float(stage_ms[0]) / args.steps,
                                  "forward_dp": float(stage_ms[1]) / args.steps, "trace": float(stage_ms[2]) / args.steps,
                                  "viterbi": float(stage_ms[6]) / args.steps},
            "screen": leg_summary(main_leg, args.reads * world, one, prof)["screen"],
            # the HBM-bound part of the pipeline (median filter + histogram, quantise + morphology): algorithmic bytes per sample --
            # int16 raw read, int16 filtered written and read again, uint8 levels written = 7 B -- over the stage's kernel time
            "conditioning_hbm": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                 "achieved": 7.0 * n_samples / (float(stage_ms[5]) / args.steps * 1e-3) / 1e9 if stage_ms[5] > 0 else None,
                                 "frac": 7.0 * n_samples / (float(stage_ms[5]) / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms[5] > 0 else None,
                                 "bytes_per_sample": 7, "note": "three streaming kernels + per-read statistics; no per-cell work"},
            "viterbi": {"time_steps_per_step": main_leg.counters[7] / args.steps,
                        "us_per_time_step_per_wave_slot": (float(stage_ms[6]) / args.steps * 1e3) / max(1.0, main_leg.counters[7] / args.steps / (8 * N_SIMD / 4))},
            "host": {"synth_s": t_gen, "synth_workers": synth_workers, "upload_s": t_up, "upload_GBs": int(lens.sum()) * 2 / t_up / 1e9 if t_up > 0 else None,
                     "peak_host_rss_gb_per_rank": round(peak_rss_gb, 3) if world == 1 else (rank_report or {}).get("peak_host_rss_gb_per_rank"),
                     "peak_host_rss_note": "resident batches are staged one at a time (synthesise -> upload in 512-read parts -> free); at N = 1 the signals of the PCIe-inclusive leg stay on the host as well",
                     "memory_limit_gb": mem_lim, "cpu_quota_cores": _sdq.cpu_quota()},
            "check": checked, "check_ok": check_ok,
        }
        if want_host_leg:
            # SURVEY.md 8d quotes the metric with the signals in host RAM: the boundary's host-buffer entry
            # (strq_detect_batch_reads: one buffer per read, nothing concatenated) over `host_leg_batches` distinct sub-batches,
            # samples in pageable host memory, uploads of sub-batch k + 1 overlapping the kernels of sub-batch k.
            # Reported beside `value`, never as `value`.
            reps = max(1, args.host_leg_batches)
            pick = [(k % n_batches) for k in range(reps)]
            reads2 = [kept[k * args.reads + j] for k in pick for j in range(args.reads)]
            tids2 = np.array([t for k in pick for t in tids[k * args.reads:(k + 1) * args.reads]], np.int32)
            nbytes = sum(r.nbytes for r in reads2)
            t1 = time.time()
            res2 = ctx.detect_batch_reads(reads2, tids2)          # first call: sizes the device buffers for this batch (hipMalloc)
            dt_cold = time.time() - t1
            t1 = time.time()
            res2 = ctx.detect_batch_reads(reads2, tids2)          # steady state of a long-running caller
            dt = time.time() - t1
            same = all(bool(np.array_equal(res2[i * args.reads:(i + 1) * args.reads], last[k])) for i, k in enumerate(pick) if k in last)
            out["host_inclusive_reads_per_s"] = len(reads2) / dt
            out["host_buffers"] = {"reads": len(reads2), "sub_batches": reps, "seconds": dt, "seconds_first_call_with_allocations": dt_cold, "reads_per_s": len(reads2) / dt, "GB": nbytes / 1e9,
                                   "same_rows_as_resident_run": same,
                                   "note": "PCIe-inclusive: int16 signals start in pageable host memory (one buffer per read) and go through a pinned staging ring; "
                                           "only the first sub-batch's upload is exposed"}
            del reads2
            # the resident batches of the headline measurement were replaced by that call: stage them again for the legs below
            if not args.no_legs:
                total = n_batches * args.reads
                est = int(lens.sum())
                for p0 in range(0, total, 512):
                    part = [kept[g] for g in range(p0, min(total, p0 + 512))]
                    off = np.zeros(len(part) + 1, np.int64); off[1:] = np.cumsum([len(x) for x in part])
                    ctx.batch_upload_part(total, est, p0, np.concatenate(part), off, tids[p0:p0 + 512])
        cpu_sigs = [kept[j] for j in range(min(args.reads, 256)) if j in kept]
        if world == 1 and not args.no_legs:
            # ---- A/B legs on the SAME resident batches, in this process (strq_set_option), each with its own roofline block:
            # the forward stage without any screen (float32 DP over whole reads), with the fine screen only, and the whole
            # pipeline on degraded reads (noise resampled from the bundled real read)
            legs = {}
            ctx.set_option("STRQ_NO_SCREEN", "1")
            legs["no_screen"] = leg_summary(run_leg(ctx, args.reads, n_batches, args.leg_steps, 1, k_step), args.reads, one, prof, nreps, n_batches)
            ctx.set_option("STRQ_NO_SCREEN", None)
            ctx.set_option("STRQ_SCREEN_MODE", "fine")
            legs["fine_screen"] = leg_summary(run_leg(ctx, args.reads, n_batches, args.leg_steps, 1, k_step), args.reads, one, prof, nreps, n_batches)
            ctx.set_option("STRQ_SCREEN_MODE", None)
            ctx.set_option("STRQ_SERIAL", "1")          # one sub-batch at a time, every kernel alone on the GPU (the order of rounds 1-5)
            legs["serial"] = leg_summary(run_leg(ctx, args.reads, n_batches, args.leg_steps, 1, k_step), args.reads, one, prof, nreps, n_batches)
            ctx.set_option("STRQ_SERIAL", None)
            out["value_serial"] = legs["serial"]["value"]
            sr_, sv_ = legs["serial"]["roofline"], legs["serial"]["roofline_viterbi"]
            out["roofline"]["kernel_alone"] = {"kernel": sr_.get("kernel"), "avg_launch_ms": sr_.get("avg_launch_ms"), "achieved": sr_.get("achieved"), "frac": sr_.get("frac"),
                                               "source": "legs.serial of this run (STRQ_SERIAL=1: the launch has the GPU to itself)"}
            out["roofline_viterbi"]["kernel_alone"] = {"kernel": sv_.get("kernel"), "ms_per_step": sv_.get("ms_per_step"), "frac": sv_.get("frac"), "frac_of_float64_issue": sv_.get("frac_of_float64_issue")}
            out["value_no_screen"] = legs["no_screen"]["value"]
            out["value_fine_screen"] = legs["fine_screen"]["value"]
            out["roofline_no_screen"] = legs["no_screen"]["roofline"]
            # degraded reads: their own resident batch (the clean ones are dropped)
            kept_clean = kept; kept = None
            dargs = argparse.Namespace(**vars(args)); dargs.reads = args.degraded_reads or args.reads
            dnb = 2                                                             # two batches in rotation: two steps in flight, as in the headline
            dl, dstr, dn, dkept, dgen, dup = stage_resident(ctx, counter, dargs, 0, dnb, synth_workers, "empirical", keep_first=1, keep_all=False)
            dleg = run_leg(ctx, dargs.reads, dnb, 2 * args.leg_steps, 4, 0)      # four untimed steps: the screens' own adaptation (overlap plan, pauses, margin) settles
            legs["degraded"] = leg_summary(dleg, dargs.reads, dl[:dargs.reads], prof, dn, dnb)
            legs["degraded"]["workload"] = "%d reads, %d nt, noise resampled from the bundled real read (strique_amd.synth.EmpiricalNoise: N~%d samples)" % (dargs.reads, args.read_nt, int(dl.mean()))
            ctx.set_option("STRQ_NO_SCREEN", "1")
            dleg2 = run_leg(ctx, dargs.reads, dnb, args.leg_steps, 1, 0)
            ctx.set_option("STRQ_NO_SCREEN", None)
            ctx.set_option("STRQ_SERIAL", "1")
            dleg3 = run_leg(ctx, dargs.reads, dnb, args.leg_steps, 2, 0)
            ctx.set_option("STRQ_SERIAL", None)
            legs["degraded"]["value_serial"] = dargs.reads * dleg3.steps / dleg3.elapsed
            legs["degraded"]["viterbi_ms_serial"] = float(dleg3.stage_ms[6]) / max(1, dleg3.steps)
            out["value_degraded_serial"] = legs["degraded"]["value_serial"]
            legs["degraded"]["value_no_screen"] = dargs.reads * dleg2.steps / dleg2.elapsed
            legs["degraded"]["forward_dp_ms_no_screen"] = float(dleg2.stage_ms[1]) / max(1, dleg2.steps)
            if args.check > 0:
                w = _cpu_check(dkept[0], dstr[0]); r = dleg.last[0][0]
                got = (int(r["count"]), float(r["score_prefix"]), float(r["score_suffix"]), float(r["log_p"]), int(r["offset"]), int(r["ticks"]))
                want = (int(w[0]), float(w[1]), float(w[2]), float(w[3]), int(w[4]), int(w[5]))
                legs["degraded"]["check"] = {"oracle": list(want), "gpu": list(got), "all_fields_equal": got == want}
                check_ok = check_ok and got == want
                out["check_ok"] = check_ok
            out["value_degraded"] = legs["degraded"]["value"]
            out["roofline_degraded"] = legs["degraded"]["roofline"]
            out["legs"] = legs
            kept = kept_clean
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 at N = 1 only
            ctx.close()                                       # the oracle workers want the host's memory, not the GPU's buffers
            sweep = tuple(int(v) for v in args.cpu_sweep.split(",") if v.strip())
            out["cpu_baseline"] = cpu_baseline(cpu_sigs, strands[:len(cpu_sigs)], args.cpu_workers, sweep)
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] > 0 else None
            out["vs_cpu_baseline_note"] = "GPU / the CPU port on the %s CPUs the job may use; against the extrapolated whole machine: %.0f x" % (
                out["cpu_baseline"]["cores"], value / out["cpu_baseline"]["extrapolated_physical_cores"]["reads_per_s"])
        detail_path = write_detail(out, args.detail)
        print(json.dumps(compact_line(out, detail_path), separators=(",", ":")))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not check_ok:
        sys.stderr.write("bench.py: GPU rows differ from the CPU oracle -- see \"check\" in the JSON line\n")
        raise SystemExit(1)


def _r(v, sig=6):
    """Floats of the compact line: six significant digits."""
    if isinstance(v, float):
        return float("%.*g" % (sig, v))
    return v


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


COMPACT_LIMIT = 4096


def write_detail(out, path):
    """The full record (every leg, every roofline block, the CPU sweep, notes) goes to a FILE: the driver keeps an 8 KB tail of
    stdout, and round 5's 25 KB line could not be parsed from it.  Returns the path written (None when nothing could be)."""
    cands = [path] if path else [os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                                 os.path.join(os.environ.get("TMPDIR", "/tmp"), "strique_bench_detail.json")]
    for p in cands:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, "w") as f:
                json.dump(out, f)
            return p
        except OSError:
            continue
    return None


def compact_line(out, detail_path=None):
    """The ONE line rank 0 prints: the benchmark contract's keys, `roofline` of the dominant kernel, `roofline_viterbi`,
    `cpu_baseline`, the A/B values, the collective's state -- under COMPACT_LIMIT bytes whatever the run was (a CPU test holds
    it to that with round 5's 25 KB record).  Only rates that are fractions of a peak (<= 1 by construction) are called `frac`."""
    roof = out.get("roofline") or {}
    line = {k: _r(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "boundary", "reads_per_gpu_per_step", "total_reads", "read_nt", "distinct_batches_per_gpu", "sharding"))
    line["roofline"] = _pick(roof, ("bound", "kernel", "unit", "achieved", "peak", "frac", "traffic", "avg_launch_ms", "launches_per_step",
                                    "wave_steps_per_launch", "valu_insts_per_wave_step", "valu_insts_source", "lane_utilisation", "gcups"))
    line["roofline"].setdefault("traffic", None)
    for k in ("achieved_own", "frac_own", "issue_cycles_frac"):
        if roof.get(k) is not None:
            line["roofline"][k] = _r(roof[k])
    if isinstance(roof.get("co_running"), dict):
        line["roofline"]["co_running"] = _pick(roof["co_running"], ("kernel", "ms_under_this_launch_per_step", "achieved"))
    if isinstance(roof.get("kernel_alone"), dict):
        line["roofline"]["kernel_alone"] = _pick(roof["kernel_alone"], ("avg_launch_ms", "frac"))
    ex = roof.get("exact_pass")
    if isinstance(ex, dict):
        line["roofline"]["exact_pass"] = _pick(ex, ("kernel", "ms_per_step", "frac", "columns_computed_over_columns_of_the_reads"))
    if out.get("roofline_viterbi"):
        line["roofline_viterbi"] = _pick(out["roofline_viterbi"], ("kernel", "ms_per_step", "achieved", "peak", "frac", "frac_of_float64_issue", "valu_insts_per_time_step", "co_running_share"))
        if isinstance(out["roofline_viterbi"].get("kernel_alone"), dict):
            line["roofline_viterbi"]["kernel_alone"] = _pick(out["roofline_viterbi"]["kernel_alone"], ("ms_per_step", "frac", "frac_of_float64_issue"))
    if out.get("stage_ms_per_step"):
        line["stage_ms_per_step"] = {k: _r(v, 4) for k, v in out["stage_ms_per_step"].items()}
    cb = out.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "per_core_reads_per_s", "workers_of_value", "physical_cores"))
        c["sample"] = str(cb.get("sample", ""))[:220]
        if cb.get("lut_variant"):
            c["lut_variant_value"] = _r(cb["lut_variant"].get("value"))
        if cb.get("extrapolated_physical_cores"):
            c["extrapolated"] = {"cores": cb["extrapolated_physical_cores"].get("cores"), "reads_per_s": _r(cb["extrapolated_physical_cores"].get("reads_per_s")), "measured": False}
        line["cpu_baseline"] = c
        line["vs_cpu_baseline"] = _r(out.get("vs_cpu_baseline"))
    for k in ("value_serial", "value_no_screen", "value_fine_screen", "value_degraded", "value_degraded_serial", "host_inclusive_reads_per_s", "resident_reads_per_s"):
        if out.get(k) is not None:
            line[k] = _r(out[k])
    hb = out.get("host_buffers")
    if hb:
        line["host_buffers"] = _pick(hb, ("reads", "sub_batches", "seconds", "same_rows_as_resident_run"))
    coll = out.get("collective")
    if coll:
        line["collective"] = _pick(coll, ("backend", "backend_requested", "rows_on_rank_0"))
        line["collective"]["nccl_error"] = (coll.get("nccl_error") or None) and str(coll["nccl_error"])[:160]
        ranks = coll.get("ranks") or {}
        if ranks:
            line["collective"]["rows_equal_every_ranks_digest"] = ranks.get("rows_equal_every_ranks_digest")
            if ranks.get("ms_per_step_per_rank"):
                line["collective"]["ms_per_step_per_rank"] = [_r(v, 5) for v in ranks["ms_per_step_per_rank"]]
    line["world_size_seen_by_the_collective"] = out.get("world_size_seen_by_the_collective")
    if out.get("host"):
        line["host"] = _pick(out["host"], ("synth_s", "upload_s", "cpu_quota_cores"))
    line["check_ok"] = out.get("check_ok"); line["checked_reads"] = len(out.get("check") or [])
    line["detail"] = detail_path
    s = json.dumps(line, separators=(",", ":"))
    # whatever a future field adds: drop the optional blocks, least important first, until the line fits
    for k in ("host", "stage_ms_per_step", "host_buffers", "resident_reads_per_s", "data"):
        if len(s) < COMPACT_LIMIT:
            break
        line.pop(k, None); s = json.dumps(line, separators=(",", ":"))
    assert len(s) < COMPACT_LIMIT, len(s)
    return line


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
