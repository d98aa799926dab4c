#!/usr/bin/env python3
"""What the fast5 reader alone delivers (no GPU involved): reads/s of Fast5Index.get_raw on gzip-compressed bulk files.

    python tools/reader_probe.py [n_reads] [read_nt]

Writes n_reads synthetic reads into bulk fast5 files with deflate-compressed chunks (what h5py / MinKNOW write), indexes
them, and reads every signal back with 1 ... 48 threads in tasks of 32 reads (the `count` command's reader pattern): one native
inflate call per read, one per task (strq_inflate_many), the latter into huge-page slabs (fast5.SlabAllocator); with libdeflate and with zlib.  Tells whether the
readers or something behind them bound `count` on compressed files."""
import io
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from contextlib import redirect_stdout

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
PER_FILE = 512


def write_file(args):
    import bench
    from strique_amd import h5write, synth
    f0, n, nt, data = args
    pm, cfg = bench.load_inputs()
    table = synth.KmerTable(pm)
    chrom, b, e, repeat, prefix, suffix = cfg["repeat"]["c9orf72"]
    reads = []
    for i in range(f0, min(n, f0 + PER_FILE)):
        sig, strand = synth.make_read(table, 12, i, nt, (repeat, prefix, suffix), 10 + i % 60)
        reads.append(("%08x-3333-4000-8000-%012d" % (i, i), sig))
    open(os.path.join(data, "batch_%d.fast5" % (f0 // PER_FILE)), "wb").write(h5write.multi_read_fast5(reads, compression="gzip"))
    return [r for r, _ in reads]


def measure(data):
    from strique_amd import cli, fast5, ffi
    cli._tune_allocator()
    idx = cli.Fast5Index(os.path.join(data, "reads.fofn"))
    ids = sorted(idx.index)
    backend = "libdeflate" if ffi.load_library().strq_inflate_backend() else "zlib"
    tasks = [ids[i:i + 32] for i in range(0, len(ids), 32)]

    def task(names, mode):
        alloc = None          # (fast5.SlabAllocator, the "slabs" mode, was removed in round 6)
        if mode == "one call per read":
            return sum(len(idx.get_raw(q, alloc)) for q in names)
        plans = [idx.get_raw(q, alloc, True) for q in names]          # located under the interpreter lock, inflated in one native call
        assert all(e is None for e in fast5.inflate_plans(plans))
        return sum(len(p.out) for p in plans)

    for q in ids[:64]:
        idx.get_raw(q)
    for mode in ("one call per read", "one call per task", "slabs"):
        for threads in (1, 8, 16, 32, 48):
            sub = tasks if threads > 1 else tasks[:16]
            import ctypes
            st = (ctypes.c_int64 * 3)()
            lib = ffi.load_library(); lib.strq_inflate_stats.restype = None
            lib.strq_inflate_stats(st, 1)
            t0 = time.time()
            with ThreadPoolExecutor(threads) as ex:
                total = sum(ex.map(lambda nm: task(nm, mode), sub))
            dt = time.time() - t0
            lib.strq_inflate_stats(st, 1)
            n_reads = sum(len(s) for s in sub)
            # inflate ms per read = time inside libdeflate / zlib; busy = that time over (threads x wall): what share of the pool's time is the inflate itself
            print("%-10s %-17s %2d threads: %6.0f reads/s  (%.2f GB/s of samples; inflate %.2f ms per read, %.0f %% of the threads' wall time)" % (
                backend, mode, threads, n_reads / dt, total * 2 / dt / 1e9, st[0] / 1e6 / max(1, n_reads), 100.0 * st[0] / 1e9 / (threads * dt)), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--measure":
        return measure(sys.argv[2])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    tmp = tempfile.mkdtemp(prefix="strq_rd_")
    import atexit, shutil
    atexit.register(shutil.rmtree, tmp, True)
    data = os.path.join(tmp, "data"); os.makedirs(data)
    t0 = time.time()
    with mp.get_context("fork").Pool(min(16, max(1, (n + PER_FILE - 1) // PER_FILE))) as pool:
        pool.map(write_file, [(f0, n, nt, data) for f0 in range(0, n, PER_FILE)])
    from strique_amd import cli
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main(["index", data])
    open(os.path.join(data, "reads.fofn"), "w").write(buf.getvalue())
    print("wrote and indexed %d reads in %.1f s" % (n, time.time() - t0), flush=True)
    for env in ({}, {"STRQ_NO_LIBDEFLATE": "1"}):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--measure", data], env=dict(os.environ, **env), check=False)


if __name__ == "__main__":
    main()
