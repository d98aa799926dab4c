#!/usr/bin/env python3
"""End-to-end throughput of the `count` command on files: synthetic 10 kb reads written into bulk fast5 files
(strique_amd/h5write.py, contiguous int16), indexed, routed through a SAM file, counted and written as TSV.
usage (GPU box): python tools/cli_probe.py [n_reads] [read_nt] [--t N] [--batch A,B,...]   (reads per GPU batch of `count`; default 4096,8192: one pass pair each)"""
import io
import os
import sys
import tempfile
import time
from contextlib import redirect_stdout

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
from strique_amd import cli, h5write, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
threads = int(sys.argv[sys.argv.index("--t") + 1]) if "--t" in sys.argv else 8
compression = sys.argv[sys.argv.index("--compression") + 1] if "--compression" in sys.argv else None      # gzip | vbz: chunked datasets like MinKNOW's bulk files
pm, cfg = bench.load_inputs()
table = synth.KmerTable(pm)
tmp = tempfile.mkdtemp(prefix="strq_cli_")
import atexit, shutil
atexit.register(shutil.rmtree, tmp, True)          # 16 -- 50 GB of fast5 per run: never leave them behind
t = np.load(os.path.join(R, "tests", "golden", "pore_tables.npz"))
with open(os.path.join(tmp, "r9.model"), "w") as fp:
    for k, m, s in zip(t["base_kmer"], t["base_mean"], t["base_stdv"]):
        fp.write("%s\t%r\t%r\t1\n" % (k.decode(), float(m), float(s)))
with open(os.path.join(tmp, "repeat.tsv"), "w") as fp:
    fp.write("chr\tbegin\tend\tname\trepeat\tprefix\tsuffix\n")
    for name, (chrom, b, e, repeat, prefix, suffix) in cfg["repeat"].items():
        fp.write("\t".join([chrom, str(b), str(e), name, repeat, prefix, suffix]) + "\n")
import json
json.dump({"align": cfg["align"], "HMM": cfg["HMM"]}, open(os.path.join(tmp, "cfg.json"), "w"))
t0 = time.time()
sam = ["@HD\tVN:1.0"]; planted = {}
data = os.path.join(tmp, "data"); os.makedirs(data)
per_file = 512


def write_file(args):
    """Worker process: one bulk fast5 of `per_file` synthetic reads; returns its SAM lines and planted counts."""
    f0, n_, nt_, data_, compression_ = args
    pm_, cfg_ = bench.load_inputs()
    table_ = synth.KmerTable(pm_)
    reads, lines, planted_ = [], [], {}
    for i in range(f0, min(n_, f0 + per_file)):
        name = ["c9orf72", "fmr1"][i % 2]
        chrom, b, e, repeat, prefix, suffix = cfg_["repeat"][name]
        nrep = 10 + i % 60
        sig, strand = synth.make_read(table_, 11, i, nt_, (repeat, prefix, suffix), nrep)
        rid = "%08x-2222-4000-8000-%012d" % (i, i)
        planted_[rid] = nrep
        reads.append((rid, sig))
        lines.append("\t".join([rid, "16" if strand == "-" else "0", chrom, str(b - 3000), "60", "10S%dM5S" % nt_, "*", "0", "0", "*", "*"]))
    open(os.path.join(data_, "batch_%d.fast5" % (f0 // per_file)), "wb").write(h5write.multi_read_fast5(reads, compression=compression_))
    return lines, planted_


import multiprocessing as mp
with mp.get_context("fork").Pool(min(16, max(1, (n + per_file - 1) // per_file))) as pool:          # before anything touches the GPU
    for lines, pl in pool.map(write_file, [(f0, n, nt, data, compression) for f0 in range(0, n, per_file)]):
        sam += lines; planted.update(pl)
open(os.path.join(tmp, "aln.sam"), "w").write("\n".join(sam) + "\n")
print("wrote %d reads in %.1f s" % (n, time.time() - t0), flush=True)
buf = io.StringIO()
t0 = time.time()
with redirect_stdout(buf):
    cli.main(["index", data])
open(os.path.join(data, "reads.fofn"), "w").write(buf.getvalue())
print("index: %.2f s" % (time.time() - t0), flush=True)
argv = ["count", os.path.join(data, "reads.fofn"), os.path.join(tmp, "r9.model"), os.path.join(tmp, "repeat.tsv"), "--config", os.path.join(tmp, "cfg.json"),
        "--algn", os.path.join(tmp, "aln.sam"), "--out", os.path.join(tmp, "out.tsv"), "--t", str(threads), "--batch", "4096"]
batch_sizes = [int(v) for v in sys.argv[sys.argv.index("--batch") + 1].split(",")] if "--batch" in sys.argv else [4096, 8192]
if "--profile" in sys.argv:
    import cProfile, pstats
    cProfile.run("cli.main(argv)", os.path.join(tmp, "prof"))
    pstats.Stats(os.path.join(tmp, "prof")).sort_stats("cumtime").print_stats(22)
import gc
for bs in batch_sizes:
  argv[-1] = str(bs)
  for rep in range(2):
    gc.collect()                                         # the previous pass's context (tens of GB of device buffers) goes first
    t0 = time.time(); cli.main(argv); dt = time.time() - t0
    rows = [l.split("\t") for l in open(os.path.join(tmp, "out.tsv")).read().splitlines()[1:]]
    ok = sum(abs(int(r[3]) - planted[r[0]]) <= 2 for r in rows)
    print("count --batch %d pass %d: %d rows in %.2f s = %.0f reads/s end to end (files -> TSV), planted count recovered %d/%d" % (bs, rep, len(rows), dt, len(rows) / dt, ok, len(rows)), flush=True)
