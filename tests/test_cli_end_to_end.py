"""The documented `count` run of the reference on its bundled files (docs/installation/test.md:8-16):

    python3 scripts/STRique.py index data/ > data/reads.fofn
    cat data/c9orf72.sam | python3 scripts/STRique.py count data/reads.fofn models/r9_4_450bps.model
        configs/repeat_config.tsv --config configs/STRique.json

`tests/golden/c9orf72.fast5` / `.sam` are byte-for-byte copies of the reference's data files
(tests/golden/make_golden.py); the model, repeat table and JSON config are written back into the
reference's file formats from the golden values recorded there.
"""
import io
import json
import os
import shutil
from contextlib import redirect_stdout

import numpy as np
import pytest

from conftest import GOLDEN

READ_ID = "ce47b364-ed6e-4409-808a-1041c0b5aac2"
DOCS_ROW = dict(count=735, score_prefix=6.3155927807600545, score_suffix=6.031860427335506,
                log_p=-119860.52066647023, offset=1633, ticks=40758)


@pytest.fixture()
def workdir(tmp_path, tables, cfg):
    data = tmp_path / "data"
    data.mkdir()
    shutil.copyfile(os.path.join(GOLDEN, "c9orf72.fast5"), data / "c9orf72.fast5")
    shutil.copyfile(os.path.join(GOLDEN, "c9orf72.sam"), data / "c9orf72.sam")
    for key, name in (("base", "r9_4_450bps.model"), ("mod", "r9_4_450bps_mCpG.model")):
        with open(tmp_path / name, "w") as fp:
            for k, m, s in zip(tables[key + "_kmer"], tables[key + "_mean"], tables[key + "_stdv"]):
                k = k.decode() if isinstance(k, bytes) else str(k)
                fp.write("%s\t%s\t%s\t1\n" % (k, repr(float(m)), repr(float(s))))
    with open(tmp_path / "repeat_config.tsv", "w") as fp:
        fp.write("chr\tbegin\tend\tname\trepeat\tprefix\tsuffix\n")
        for name, (chrom, b, e, repeat, prefix, suffix) in cfg["repeat"].items():
            fp.write("\t".join([chrom, str(b), str(e), name, repeat, prefix, suffix]) + "\n")
    with open(tmp_path / "STRique.json", "w") as fp:
        json.dump({"align": cfg["align"], "HMM": cfg["HMM"]}, fp)
    return tmp_path


def _index(workdir):
    from strique_amd import cli
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main(["index", str(workdir / "data")])
    return buf.getvalue()


def test_index_and_raw_reader_on_the_bundled_fast5(workdir):
    """`index` output format (fast5Index.py:53,163-179) and the HDF5 subset reader on the real file
    (superblock v0, deflate, two chunks) against the committed raw-signal fixture."""
    from strique_amd import cli, fast5
    text = _index(workdir)
    assert text == "c9orf72.fast5\t%s\n" % READ_ID
    fofn = workdir / "data" / "reads.fofn"
    fofn.write_text(text)
    raw = cli.Fast5Index(str(fofn)).get_raw(READ_ID)
    want = np.load(os.path.join(GOLDEN, "bundled_read.npz"))["signal"]
    assert raw.dtype == np.int16 and np.array_equal(raw, want)
    assert cli.Fast5Index(str(fofn)).get_raw("no-such-read") is None
    (rid, sig), = fast5.read_raw(str(workdir / "data" / "c9orf72.fast5"))
    assert rid == READ_ID and len(sig) == 284184 and int(sig.min()) == -4096 and int(sig.max()) == 3008


def test_sam_router_on_the_bundled_alignment(workdir, cfg):
    """FLAG / POS / CIGAR-derived span and clips of the real record, and the locus it selects."""
    from strique_amd import cli
    want = json.load(open(os.path.join(GOLDEN, "sam.json")))["records"][0]
    loci = {}
    for name, (chrom, b, e, *_r) in cfg["repeat"].items():
        loci.setdefault(chrom, []).append((name, b, e))
    recs = [cli.decode_sam(l) for l in open(workdir / "data" / "c9orf72.sam") if not l.startswith("@")]
    assert len(recs) == 1
    sr = recs[0]
    got = dict(QNAME=sr.QNAME, FLAG=sr.FLAG, RNAME=sr.RNAME, POS=sr.POS, TLEN=sr.TLEN, CLIP_BEGIN=sr.CLIP_BEGIN,
               CLIP_END=sr.CLIP_END, targets=cli.intersect_targets(sr, loci))
    assert got == {k: want[k] for k in got}


@pytest.mark.gpu
@pytest.mark.parametrize("with_mod", [False, True])
def test_count_command_end_to_end(workdir, with_mod):
    from strique_amd import cli
    fofn = workdir / "data" / "reads.fofn"
    fofn.write_text(_index(workdir))
    out = workdir / "out.tsv"
    argv = ["count", str(fofn), str(workdir / "r9_4_450bps.model"), str(workdir / "repeat_config.tsv"),
            "--config", str(workdir / "STRique.json"), "--algn", str(workdir / "data" / "c9orf72.sam"), "--out", str(out)]
    if with_mod:
        argv += ["--mod_model", str(workdir / "r9_4_450bps_mCpG.model")]
    cli.main(argv)
    lines = out.read_text().splitlines()
    assert lines[0].split("\t") == ["ID", "target", "strand", "count", "score_prefix", "score_suffix", "log_p", "offset", "ticks", "mod"]
    assert len(lines) == 2
    rid, target, strand, count, sp, ss, lp, offset, ticks, mod = lines[1].split("\t")
    assert (rid, target, strand) == (READ_ID, "c9orf72", "-")
    assert int(offset) == DOCS_ROW["offset"] and int(ticks) == DOCS_ROW["ticks"]     # geometry: exact
    assert abs(int(count) - DOCS_ROW["count"]) <= 2
    assert abs(float(sp) / DOCS_ROW["score_prefix"] - 1) < 0.01 and abs(float(ss) / DOCS_ROW["score_suffix"] - 1) < 0.01
    assert abs(float(lp) / DOCS_ROW["log_p"] - 1) < 0.02
    if with_mod:
        assert set(mod) <= set("01") and abs(len(mod) - int(count)) <= 3
    else:
        assert mod == "-"


@pytest.mark.gpu
def test_count_on_a_multi_read_fast5_with_mixed_targets(workdir, pm, cfg):
    """`index` + `count` on a bulk (multi-read) fast5 written by tests/h5write.py and a SAM with one
    record per read over both bundled loci and both strands: one row per record, in input order, target
    and strand from the SAM record, planted repeat count recovered."""
    import h5write
    from strique_amd import cli, synth
    table = synth.KmerTable(pm)
    plan = [("c9orf72", "+", 14), ("fmr1", "-", 33), ("c9orf72", "-", 57), ("fmr1", "+", 8), ("c9orf72", "+", 90)]
    reads, sam = [], ["@HD\tVN:1.0"]
    for i, (name, strand, nrep) in enumerate(plan):
        chrom, b, e, repeat, prefix, suffix = cfg["repeat"][name]
        sig, st = synth.make_read(table, 7, i, 6000 + 500 * i, (repeat, prefix, suffix), nrep, strand=strand)
        rid = "%08d-0000-4000-8000-%012d" % (i, i)
        reads.append((rid, sig))
        sam.append("\t".join([rid, "16" if strand == "-" else "0", chrom, str(b - 3000), "60", "12S6000M5S", "*", "0", "0", "ACGT", "*"]))
    sam.append("\t".join(["unmapped-elsewhere", "0", "chr1", "1000", "60", "100M", "*", "0", "0", "ACGT", "*"]))
    bulk = workdir / "bulk"; bulk.mkdir()
    (bulk / "batch_0.fast5").write_bytes(h5write.multi_read_fast5(reads))
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main(["index", str(bulk)])
    (bulk / "reads.fofn").write_text(buf.getvalue())
    (workdir / "aln.sam").write_text("\n".join(sam) + "\n")
    out = workdir / "multi.tsv"
    cli.main(["count", str(bulk / "reads.fofn"), str(workdir / "r9_4_450bps.model"), str(workdir / "repeat_config.tsv"),
              "--config", str(workdir / "STRique.json"), "--algn", str(workdir / "aln.sam"), "--out", str(out), "--batch", "3"])
    rows = [l.split("\t") for l in out.read_text().splitlines()[1:]]
    assert [(r[0], r[1], r[2]) for r in rows] == [(rid, name, strand) for (rid, _), (name, strand, _) in zip(reads, plan)]
    assert [int(r[3]) for r in rows] == [n for _, _, n in plan]


@pytest.mark.gpu
def test_two_rank_count_equals_single_rank(workdir, pm, pm_mod, cfg):
    """SURVEY.md 8e on real kernels: two fresh rank processes (gloo, both on HIP device 0) run `count`
    with the modification model on 12 mixed reads -- the (read, target) pairs dealt by read length, one
    gather of records + modification strings -- and rank 0's merged TSV equals the single-process TSV."""
    import subprocess
    import sys
    import h5write
    from conftest import ROOT
    from strique_amd import cli, synth
    table, table_mod = synth.KmerTable(pm), synth.KmerTable(pm_mod)
    rng = np.random.default_rng(17)
    reads, sam = [], ["@HD\tVN:1.0"]
    for i in range(12):
        name = ["c9orf72", "fmr1"][i % 2]; strand = "+-"[(i // 2) % 2]
        chrom, b, e, repeat, prefix, suffix = cfg["repeat"][name]
        nt = int(rng.integers(4000, 9000))
        sig, _ = synth.make_read(table_mod if i % 3 == 0 else table, 8, i, nt, (repeat, prefix, suffix), int(rng.integers(5, 80)), strand=strand)
        rid = "%08d-1111-4000-8000-%012d" % (i, i)
        reads.append((rid, sig))
        sam.append("\t".join([rid, "16" if strand == "-" else "0", chrom, str(b - 3000), "60", "12S%dM5S" % nt, "*", "0", "0", "*", "*"]))
    bulk = workdir / "bulk2"; bulk.mkdir()
    (bulk / "batch_0.fast5").write_bytes(h5write.multi_read_fast5(reads[:7]))
    (bulk / "batch_1.fast5").write_bytes(h5write.multi_read_fast5(reads[7:]))
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main(["index", str(bulk)])
    (bulk / "reads.fofn").write_text(buf.getvalue())
    (workdir / "aln2.sam").write_text("\n".join(sam) + "\n")
    base = [str(bulk / "reads.fofn"), str(workdir / "r9_4_450bps.model"), str(workdir / "repeat_config.tsv"),
            "--config", str(workdir / "STRique.json"), "--algn", str(workdir / "aln2.sam"),
            "--mod_model", str(workdir / "r9_4_450bps_mCpG.model"), "--batch", "4"]
    one = workdir / "one.tsv"
    cli.main(["count"] + base + ["--out", str(one)])
    two = workdir / "two.tsv"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "STRique.py"), "count"] + base + ["--out", str(two), "--backend", "gloo", "--share-device"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert len(one.read_text().splitlines()) == 13
    assert two.read_text() == one.read_text()
