"""Host-side logic of the `count` command against the golden vectors recorded from the reference."""
import json
import os

import pytest

from conftest import GOLDEN, ROOT


def test_sam_decoding_matches_reference():
    from strique_amd import cli
    g = json.load(open(os.path.join(GOLDEN, "sam.json")))
    ops = cli.decode_cigar(g["cigar"])
    assert [list(o) for o in ops] == g["ops"]
    assert cli.ops_length(ops) == g["len_MIS=X"] and cli.ops_length(ops, "MDN=X") == g["len_MDN=X"]
    rec = g["records"][0]
    # rebuild a SAM line with a CIGAR that has the documented span / clips
    tlen, cb, ce = rec["TLEN"], rec["CLIP_BEGIN"], rec["CLIP_END"]
    cigar = "%dS%dM%dS" % (cb, tlen, ce)
    line = "\t".join([rec["QNAME"], str(rec["FLAG"]), rec["RNAME"], str(rec["POS"]), "60", cigar, "*", "0", "0", "ACGT", "*"])
    sr = cli.decode_sam(line)
    assert (sr.QNAME, sr.FLAG, sr.RNAME, sr.POS, sr.TLEN, sr.CLIP_BEGIN, sr.CLIP_END) == \
        (rec["QNAME"], rec["FLAG"], rec["RNAME"], rec["POS"], tlen, cb, ce)
    cfg = json.load(open(os.path.join(GOLDEN, "config.json")))
    loci = {}
    for name, (chrom, b, e, *_rest) in cfg["repeat"].items():
        loci.setdefault(chrom, []).append((name, b, e))
    assert cli.intersect_targets(sr, loci) == rec["targets"]
    assert cli.decode_sam("too\tshort").QNAME == ""


def test_parse_config_roundtrip(tmp_path, cfg):
    from strique_amd import cli
    tsv = tmp_path / "repeat_config.tsv"
    rows = ["chr\tbegin\tend\tname\trepeat\tprefix\tsuffix"]
    for name, (chrom, b, e, repeat, prefix, suffix) in cfg["repeat"].items():
        rows.append("\t".join([chrom, str(b), str(e), name, repeat, prefix, suffix]))
    rows.append("broken line")
    tsv.write_text("\n".join(rows) + "\n")
    js = tmp_path / "STRique.json"
    js.write_text(json.dumps({"align": cfg["align"], "HMM": cfg["HMM"], "extra": 1}))
    got = cli.parse_config(str(tsv), str(js))
    assert {k: list(v) for k, v in got["repeat"].items()} == {k: list(v) for k, v in cfg["repeat"].items()}
    assert got["align"] == cfg["align"] and got["HMM"] == cfg["HMM"]
    assert cli.parse_config(str(tsv))["align"] is None
    js.write_text(json.dumps({"align": {}}))
    with pytest.raises(SystemExit):
        cli.parse_config(str(tsv), str(js))


def test_fast5_layouts_written_by_hand(tmp_path):
    """Multi-read fast5 (/read_<id>/Raw/Signal), single-read files inside a tar archive, and the index
    lines / raw lookups for both (STRique_lib/fast5Index.py:53,163-179,220-233): files written byte by
    byte by tests/h5write.py."""
    import io
    import tarfile
    import numpy as np
    import h5write
    from strique_amd import cli, fast5
    rng = np.random.default_rng(3)
    reads = [("%08x-aaaa-bbbb-cccc-%012x" % (i, i * 7919), rng.integers(-3000, 3000, 50 + 13 * i).astype(np.int16)) for i in range(9)]
    (tmp_path / "batch").mkdir()
    bulk = tmp_path / "batch" / "bulk_0.fast5"
    bulk.write_bytes(h5write.multi_read_fast5(reads[:6]))
    got = dict(fast5.read_raw(str(bulk)))
    assert set(got) == {r for r, _ in reads[:6]} and all(np.array_equal(got[r], s) for r, s in reads[:6])
    with tarfile.open(tmp_path / "batch" / "singles.tar", "w") as tar:
        for k, (rid, sig) in enumerate(reads[6:]):
            blob = h5write.single_read_fast5(rid, sig, read_number=100 + k)
            info = tarfile.TarInfo("sub/read_%d.fast5" % k); info.size = len(blob)
            tar.addfile(info, io.BytesIO(blob))
    lines = list(cli.Fast5Index.index_records(str(tmp_path / "batch")))
    assert sorted(l.split("\t")[1] for l in lines) == sorted(r for r, _ in reads)
    assert any(l.startswith("bulk_0.fast5/read_") for l in lines) and any(l.startswith("singles.tar/sub/read_") for l in lines)
    fofn = tmp_path / "batch" / "reads.fofn"
    fofn.write_text("\n".join(lines) + "\n")
    idx = cli.Fast5Index(str(fofn))
    for rid, sig in reads:
        raw = idx.get_raw(rid)
        assert raw is not None and raw.dtype == np.int16 and np.array_equal(raw, sig), rid
    assert len(idx._open) <= 4


def test_reader_threads_keep_the_row_order(cfg):
    """`count` fetches raw signals with a thread pool ahead of the GPU batches; rows must come out exactly
    as without it (input order, failed reads skipped)."""
    import io
    import time
    import numpy as np
    from strique_amd import cli
    loci = {}
    for name, (chrom, b, e, *_r) in cfg["repeat"].items():
        loci.setdefault(chrom, []).append((name, b, e))
    lines = ["\t".join(["read%d" % i, "16" if i % 2 else "0", "chr9" if i % 3 else "chrX", str(27570000 if i % 3 else 146990000),
                        "60", "8000M", "*", "0", "0", "ACGT", "*"]) for i in range(40)]

    class FakeCounter(object):
        def detect_batch(self, items):
            return [(len(raw) % 89, 1.0, 2.0, -1.0 * len(t), int(raw[0]), 3, "-") for t, raw, s in items]

    def get_raw(qname):
        i = int(qname[4:])
        time.sleep(0.002 * ((i * 7) % 5))                 # finish out of order
        if i == 13:
            raise IOError("broken file")
        return None if i == 21 else np.arange(50 + i, 400 + 3 * i)

    log = cli.Log("error")
    outs = []
    for readers in (0, 4):
        buf = io.StringIO()
        cli.run_count(iter(lines), loci, get_raw, FakeCounter(), log, 7, 0, 1, buf, readers=readers)
        outs.append(buf.getvalue())
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == 1 + 38


def test_device_fault_stops_the_engine_without_running_the_queued_batch(cfg):
    """A device fault in batch k: `run_count` raises DeviceFault at once -- the batch already queued behind k is cancelled,
    not handed to the faulted device, and nobody waits for it (the caller has to tell the other ranks, dist.any_rank)."""
    import io
    import threading
    import time
    import numpy as np
    import pytest
    from strique_amd import cli, ffi
    loci = {}
    for name, (chrom, b, e, *_r) in cfg["repeat"].items():
        loci.setdefault(chrom, []).append((name, b, e))
    lines = ["\t".join(["read%d" % i, "0", "chr9", "27570000", "60", "8000M", "*", "0", "0", "ACGT", "*"]) for i in range(60)]
    calls = []

    class FaultyCounter(object):
        def detect_batch(self, items):
            calls.append(len(items))
            if len(calls) == 2:
                time.sleep(0.3)                      # batch 2 is running while batch 3 is prepared and queued behind it
                raise ffi.StriqueHipError(ffi.STRQ_ERR_DEVICE, "hipErrorLaunchFailure")
            return [(1, 1.0, 2.0, -1.0, 0, 3, "-") for _ in items]

    def get_raw(qname):
        return np.arange(100)

    t0 = time.time()
    with pytest.raises(cli.DeviceFault):
        cli.run_count(iter(lines), loci, get_raw, FaultyCounter(), cli.Log("error"), 7, 0, 1, io.StringIO(), readers=0)
    assert time.time() - t0 < 5.0
    time.sleep(0.2)
    assert calls == [7, 7], calls                    # nothing ran on the device after the fault


def test_reader_tasks_inflate_their_reads_in_one_call(tmp_path, cfg, monkeypatch):
    """`count` locates the compressed datasets of a reader task in Python and inflates them with ONE native call
    (strq_inflate_many): the same rows as read by read (STRQ_READ_ONE_BY_ONE=1) and as with the huge-page slabs, a damaged read
    is reported and skipped without taking its task's other reads with it."""
    import io
    import zlib
    import numpy as np
    from strique_amd import cli, h5write
    rng = np.random.default_rng(8)
    data = tmp_path / "data"; data.mkdir()
    reads = [("%08x-4444-4000-8000-%012d" % (i, i), rng.integers(300, 900, 9000 + 37 * i).astype(np.int16)) for i in range(70)]
    blob = bytearray(h5write.multi_read_fast5(reads[:40], compression="gzip"))
    pos = bytes(blob).find(zlib.compress(reads[11][1][:8192].tobytes(), 4)[:16])
    assert pos > 0
    blob[pos + 30] ^= 0xFF; blob[pos + 31] ^= 0xFF; blob[pos + 55] ^= 0xFF          # read 11: a damaged chunk
    (data / "a.fast5").write_bytes(bytes(blob))
    (data / "b.fast5").write_bytes(h5write.multi_read_fast5(reads[40:]))              # contiguous datasets in the same run
    from contextlib import redirect_stdout
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main(["index", str(data)])
    (data / "reads.fofn").write_text(buf.getvalue())
    loci = {}
    for name, (chrom, b, e, *_r) in cfg["repeat"].items():
        loci.setdefault(chrom, []).append((name, b, e))
    lines = ["\t".join([rid, "0", "chr9", "27570000", "60", "8000M", "*", "0", "0", "ACGT", "*"]) for rid, _ in reads]

    class FakeCounter(object):
        def detect_batch(self, items):
            return [(int(raw[:50].sum()) % 997, 1.0, 2.0, -1.0, len(raw), int(raw[-1]), "-") for t, raw, s in items]

    outs = {}
    for mode, env in (("batched", {}), ("one_by_one", {"STRQ_READ_ONE_BY_ONE": "1"})):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            idx = cli.Fast5Index(str(data / "reads.fofn"))
            out = io.StringIO(); err = []
            log = lambda msg, level='info': err.append((level, msg))
            cli.run_count(iter(lines), loci, idx.get_raw, FakeCounter(), log, 16, 0, 1, out, readers=3)
            outs[mode] = out.getvalue()
            assert any(reads[11][0] in m for _, m in err), (mode, err)
    assert outs["batched"] == outs["one_by_one"]
    rows = outs["batched"].splitlines()[1:]
    assert len(rows) == 69 and all(reads[11][0] not in r for r in rows)
    by_id = {r.split("\t")[0]: r.split("\t") for r in rows}
    for rid, sig in reads[:11] + reads[12:]:
        assert int(by_id[rid][7]) == len(sig) and int(by_id[rid][8]) == int(sig[-1])          # offset / ticks columns carry the fake values


def test_native_locator_equals_the_python_reader_and_survives_damage(tmp_path, monkeypatch):
    """strq_h5_locate resolves /read_<id>/Raw/Signal on the mapped file in one call (old-style groups, version-1 headers, contiguous
    or deflate-chunked 1-D datasets); the Python reader stays the authority for everything else.  Same arrays from both on bulk
    files of every kind the writer knows, a group of 3000 reads (several symbol-table nodes: the B-tree descent by key), and --
    every address in a file being untrusted -- a few hundred randomly damaged copies: an exception or an array, never a crash,
    and whenever both paths return something, the same thing."""
    import numpy as np
    from strique_amd import fast5, h5write
    rng = np.random.default_rng(12)
    reads = [("%08x-5555-4000-8000-%012d" % (i * 7919 % 100003, i), rng.integers(-500, 900, int(rng.integers(1, 20000))).astype(np.int16)) for i in range(3000)]
    for compression in (None, "gzip"):
        blob = h5write.multi_read_fast5(reads[:3000 if compression is None else 300], compression=compression)
        f = fast5.H5File(blob)
        assert fast5._locate_fn() is not None
        for rid, sig in reads[:3000 if compression is None else 300:7]:
            a = f.dataset("/read_%s/Raw/Signal" % rid)
            assert f._fast_dataset("/read_%s/Raw/Signal" % rid, None, False) is not None          # the native path took it
            monkeypatch.setenv("STRQ_H5_PYTHON", "1")
            b = f.dataset("/read_%s/Raw/Signal" % rid)
            monkeypatch.delenv("STRQ_H5_PYTHON")
            assert a.dtype == b.dtype == np.int16 and np.array_equal(a, sig) and np.array_equal(b, sig)
        with pytest.raises(KeyError):
            f.dataset("/read_nope/Raw/Signal")
    # VBZ-compressed files (what MinKNOW writes; both layouts of the variable-byte layer): chunks decoded by the library in one call
    # (strq_vbz_chunks: zstd through dlopen + strq_svb_decode), the same samples as the per-chunk Python decoder
    for version in (0, 1):
        f = fast5.H5File(h5write.multi_read_fast5(reads[:40], compression="vbz", vbz_version=version))
        for rid, sig in reads[:40:3]:
            a = f._fast_dataset("/read_%s/Raw/Signal" % rid, None, False)
            assert a is not None and a.dtype == np.int16 and np.array_equal(a, sig)
            monkeypatch.setenv("STRQ_H5_PYTHON", "1")
            assert np.array_equal(f.dataset("/read_%s/Raw/Signal" % rid), sig)
            monkeypatch.delenv("STRQ_H5_PYTHON")
    vclean = bytearray(h5write.multi_read_fast5([(rid, sig[:3000]) for rid, sig in reads[:12]], compression="vbz"))
    for trial in range(200):
        blob = bytearray(vclean)
        for _ in range(int(rng.integers(1, 4))):
            blob[int(rng.integers(8, len(blob)))] = int(rng.integers(0, 256))
        got = {}
        for mode in ("native", "python"):
            if mode == "python":
                monkeypatch.setenv("STRQ_H5_PYTHON", "1")
            try:
                got[mode] = np.array(fast5.H5File(bytes(blob)).dataset("/read_%s/Raw/Signal" % reads[trial % 12][0]))
            except Exception as e:
                got[mode] = type(e).__name__
            if mode == "python":
                monkeypatch.delenv("STRQ_H5_PYTHON")
        if isinstance(got["native"], np.ndarray) and isinstance(got["python"], np.ndarray):
            assert np.array_equal(got["native"], got["python"])
        else:
            assert not isinstance(got["native"], np.ndarray) or isinstance(got["python"], np.ndarray), got          # never samples where the Python decoder refuses
    # damage
    small = [(rid, sig[:3000]) for rid, sig in reads[:12]]
    clean = bytearray(h5write.multi_read_fast5(small, compression="gzip"))
    agree = 0
    for trial in range(400):
        blob = bytearray(clean)
        for _ in range(int(rng.integers(1, 5))):
            pos = int(rng.integers(8, len(blob)))
            blob[pos] = int(rng.integers(0, 256))
        rid = small[trial % len(small)][0]
        got = {}
        for mode in ("native", "python"):
            if mode == "python":
                monkeypatch.setenv("STRQ_H5_PYTHON", "1")
            try:
                got[mode] = np.array(fast5.H5File(bytes(blob)).dataset("/read_%s/Raw/Signal" % rid))
            except Exception as e:          # whatever it is, it is an exception
                got[mode] = type(e).__name__
            if mode == "python":
                monkeypatch.delenv("STRQ_H5_PYTHON")
        if isinstance(got["native"], np.ndarray) and isinstance(got["python"], np.ndarray):
            assert np.array_equal(got["native"], got["python"]); agree += 1
    assert agree > 100          # most damage hits signal bytes or unused space: both readers return the same samples


def test_vlen_read_id_and_user_block(tmp_path):
    """read_id stored as a variable-length string (global heap), and a file with a user block in front of
    the superblock (non-zero base address): both forms real fast5 writers produce."""
    import io
    from contextlib import redirect_stdout
    import numpy as np
    import h5write
    from strique_amd import cli, fast5
    sig = (np.arange(3000) % 701 - 300).astype(np.int16)
    blob = h5write.single_read_fast5("0f0f0f0f-vlen-4000-8000-000000000001", sig, vlen_id=True)
    (rid, got), = fast5.read_raw(blob)
    assert rid == "0f0f0f0f-vlen-4000-8000-000000000001" and np.array_equal(got, sig)
    # a user block of 1024 bytes: the superblock is found at 1024 and every address is relative to it
    blob = h5write.single_read_fast5("base-0001", sig, user_block=1024)
    (rid, got), = fast5.read_raw(blob)
    assert rid == "base-0001" and np.array_equal(got, sig)
    d = tmp_path / "d"; d.mkdir()
    (d / "a.fast5").write_bytes(h5write.single_read_fast5("id-a", sig, vlen_id=True))
    (d / "broken.fast5").write_bytes(b"this is not an hdf5 file")
    buf = io.StringIO(); err = io.StringIO()
    import contextlib
    with redirect_stdout(buf), contextlib.redirect_stderr(err):
        cli.main(["index", str(d)])
    assert buf.getvalue() == "a.fast5\tid-a\n"                 # the broken file is reported and skipped
    assert "broken.fast5" in err.getvalue()


def test_fast5_masker_and_plot(tmp_path):
    """scripts/fast5Masker.py:45-92 and the `plot` command (scripts/STRique.py:948-1024): the repeat
    region [offset, offset + ticks) of every counted read is cut out of its signal, the masked reads
    are written (single files or one bulk file) with their metadata, indexed, and can be read back."""
    import numpy as np
    import h5write
    from strique_amd import cli, fast5, masker
    rng = np.random.default_rng(3)
    src = tmp_path / "src"; src.mkdir()
    sigs = {}
    tree = {"attrs": {"file_version": "2.0"}, "groups": {}}
    for i in range(3):
        rid = "aaaaaaaa-%04d-4000-8000-000000000000" % i
        sigs[rid] = rng.integers(200, 900, 5000 + 100 * i).astype(np.int16)
        tree["groups"]["read_" + rid] = {"groups": {
            "Raw": {"attrs": {"read_id": rid, "duration": len(sigs[rid]), "read_number": 10 + i, "start_time": 12345, "median_before": 201.5},
                    "datasets": {"Signal": (sigs[rid], {})}},
            "channel_id": {"attrs": {"channel_number": "77", "digitisation": 8192.0, "offset": 10.0, "range": 1400.5, "sampling_rate": 4000.0}},
            "tracking_id": {"attrs": {"run_id": "run-x", "flow_cell_id": "FAK00000"}}}}
    (src / "batch.fast5").write_bytes(h5write.write_tree(tree))
    import io
    from contextlib import redirect_stdout
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main(["index", str(src)])
    (src / "reads.fofn").write_text(buf.getvalue())
    ids = sorted(sigs)
    counts = tmp_path / "counts.tsv"
    rows = ["\t".join(cli.HEADER)]
    plan = {ids[0]: (1000, 700), ids[1]: (2500, 1200), ids[2]: (0, 300)}
    for rid, (off, ticks) in plan.items():
        rows.append("\t".join([rid, "c9orf72", "+", "17", "6.5", "6.1", "-1234.5", str(off), str(ticks), "-"]))
    rows.append("\t".join(["not-in-the-index", "c9orf72", "+", "3", "1.0", "1.0", "-1.0", "5", "5", "-"]))
    counts.write_text("\n".join(rows) + "\n")
    for fmt in ("bulk", "single"):
        out = tmp_path / ("masked_" + fmt)
        n = masker.run(str(src / "reads.fofn"), str(counts), str(out), fmt, log=io.StringIO())
        assert n == 3
        assert (out / "reads.txt").read_text().split() == ids + ["not-in-the-index"]
        idx = cli.Fast5Index(str(out / "reads.fofn"))
        assert sorted(idx.index) == ids
        for rid, (off, ticks) in plan.items():
            got = idx.get_raw(rid)
            want = np.concatenate([sigs[rid][:off], sigs[rid][off + ticks:]])
            assert got.dtype == np.int16 and np.array_equal(got, want)
        files = sorted(p.name for p in out.glob("*.fast5"))
        assert files == (["reads.fast5"] if fmt == "bulk" else [r + ".fast5" for r in ids])
        f = fast5.H5File(str(out / files[0]))
        grp = ("/read_%s" % ids[0]) if fmt == "bulk" else "/UniqueGlobalKey"
        ch = f.attrs(grp + "/channel_id")
        assert ch["channel_number"] == "77" and ch["range"] == 1400.5 and ch["sampling_rate"] == 4000.0
        raw = f.attrs(("/read_%s/Raw" % ids[0]) if fmt == "bulk" else "/Raw/Reads/Read_10")
        assert raw["duration"] == len(sigs[ids[0]]) - 700 and raw["read_number"] == 10 and raw["median_before"] == 201.5
    plots = tmp_path / "plots"
    cli.main(["plot", str(src / "reads.fofn"), "--counts", str(counts), "--output", str(plots), "--zoom", "200", "--width", "6", "--height", "4"])
    made = sorted(p.name for p in plots.glob("*.png"))
    assert made == sorted("c9orf72_17_%s.png" % r for r in ids) and all((plots / m).stat().st_size > 2000 for m in made)
    # what the figures shade is the TSV's [offset, offset + ticks), panel by panel (strique_amd/plotting.py)
    from matplotlib.figure import Figure
    from strique_amd import plotting
    with open(counts) as fh:
        rows_ = {r.read_id: r for r in plotting.parse_counts(fh)}
    assert sorted(rows_) == sorted(ids + ["not-in-the-index"])
    for rid, (off, ticks) in plan.items():
        n = len(sigs[rid])
        axes = plotting.draw(Figure(figsize=(6, 4)), sigs[rid], rows_[rid], extension=0.1, zoom=200)
        assert axes["overview"].strique_span == (off, off + ticks)
        assert axes["left"].strique_span == (off, min(off + ticks, off + 200, n))
        assert axes["right"].strique_span == (max(off, off + ticks - 200), off + ticks)
        for name, ax in axes.items():
            spans = [p for p in ax.patches]
            assert len(spans) == 1
            x0 = float(spans[0].get_x())          # axvspan draws a Rectangle in data x-coordinates
            assert (x0, x0 + float(spans[0].get_width())) == tuple(float(v) for v in ax.strique_span)
        assert [l.get_xdata()[0] for l in axes["left"].lines[1:]] == [off] and [l.get_xdata()[0] for l in axes["right"].lines[1:]] == [off + ticks]
    w = plotting.Windows(1000, 950, 400, 0.1, 100)          # a repeat that runs past the end of the signal is clipped
    assert w.repeat == (950, 1000) and w.overview == (910, 1000) and w.right == (900, 1000) and w.shaded("left") == (950, 1000)


def test_sam_edge_cases_equal_the_reference():
    """tests/golden/sam_cases.json: twenty hand-written SAM lines and what the reference's __decode_sam__,
    __intersect_target__ and strand rule make of them (recorded by make_golden.py --only sam_cases): clips counted
    from the first / last two CIGAR operations (twice when there are only two), '*' fields, short lines, signs."""
    import io
    from strique_amd import cli
    cases = json.load(open(os.path.join(GOLDEN, "sam_cases.json")))
    cfg = json.load(open(os.path.join(GOLDEN, "config.json")))
    loci = {}
    for name, (chrom, begin, end, repeat, prefix, suffix) in cfg["repeat"].items():
        loci.setdefault(chrom, []).append((name, begin, end))
    assert len(cases) >= 20
    for c in cases:
        sr = cli.decode_sam(c["line"])
        got = {k: getattr(sr, k) for k in c["record"]}
        assert got == c["record"], c["what"]
        assert bool(sr.QNAME) == c["parsed"], c["what"]
        assert cli.intersect_targets(sr, loci) == c["targets"], c["what"]
        routed = list(cli.route(io.StringIO(c["line"] + "\n"), loci, lambda *a: None))
        if c["parsed"] and c["targets"]:
            assert [(r[0], r[1], r[2]) for r in routed] == [(c["record"]["QNAME"], c["strand"], c["targets"])], c["what"]
        else:
            assert routed == [], c["what"]


def test_vbz_chunks(tmp_path):
    """VBZ (HDF5 filter 32020).  No VBZ file or plugin exists in this image, so the decoder is held to hand-worked
    vectors of the published layout, to its own encoder on signals with every code length, and to its length checks."""
    import numpy as np
    from strique_amd import fast5, h5write, vbz
    # [1, 2, 3, -1] as int16, version 0, no zstd: 8 bytes; deltas 1 1 1 -4 -> zig-zag 2 2 2 7; one key byte (all one-byte
    # codes), four data bytes
    assert vbz.decode(bytes.fromhex("08000000" "00" "02020207"), (0, 2, 1, 0)) == np.array([1, 2, 3, -1], "<i2").tobytes()
    # version 1 (one key bit per 16-bit integer): [1, 2, 3, -1, 300]: zig-zag 2 2 2 7 602; key bits 0 0 0 0 1 -> 0x10;
    # 602 = 0x025a little-endian
    assert vbz.decode(bytes.fromhex("0a000000" "10" "02020207" "5a02"), (1, 2, 1, 0)) == np.array([1, 2, 3, -1, 300], "<i2").tobytes()
    # version 0 without delta: -2 widens to 0xfffffffe (four data bytes, code 3), 258 = 0x0102 (code 1)
    assert vbz.decode(bytes.fromhex("04000000" "07" "feffffff" "0201"), (0, 2, 0, 0)) == np.array([-2, 258], "<i2").tobytes()
    rng = np.random.default_rng(8)
    sig = np.clip(np.cumsum(rng.normal(0, 40, 20000)) + 400, -32768, 32767).astype(np.int16)
    sig[100:104] = [-32768, 32767, -32768, 32767]                     # three-byte codes in version 0
    for version in (0, 1):
        for level in (0, 1, 3):
            chunk, cd = vbz.encode(sig, version=version, level=level)
            assert np.array_equal(np.frombuffer(vbz.decode(chunk, cd), "<i2"), sig)
            saved = vbz._native
            try:                                   # the numpy path (library not built) and the library's host helper agree
                vbz._native = None
                assert np.array_equal(np.frombuffer(vbz.decode(chunk, cd), "<i2"), sig)
            finally:
                vbz._native = saved
            with pytest.raises(ValueError):
                vbz.decode(chunk[:-1] if level == 0 else chunk[:len(chunk) // 2], cd)      # truncated
            if level == 0:
                with pytest.raises(ValueError):
                    vbz.decode(chunk + b"\x00", cd)                                          # trailing bytes
    # through the file reader: a multi-read file with chunked VBZ signals, like the bulk files MinKNOW writes
    reads = [("a%d" % i, np.roll(sig, 7 * i)[: 5000 + 4000 * i]) for i in range(4)]
    for version in (0, 1):
        p = tmp_path / ("bulk_v%d.fast5" % version)
        p.write_bytes(h5write.multi_read_fast5(reads, compression="vbz", vbz_version=version))
        got = dict(fast5.read_raw(str(p)))
        assert all(np.array_equal(got[r], s) for r, s in reads)
    p = tmp_path / "bulk_gzip.fast5"
    p.write_bytes(h5write.multi_read_fast5(reads, compression="gzip"))
    got = dict(fast5.read_raw(str(p)))
    assert all(np.array_equal(got[r], s) for r, s in reads)


def test_deflate_chunks_native_and_python_paths(tmp_path, monkeypatch):
    """Chunked deflate datasets (what h5py / MinKNOW write; the bundled c9orf72.fast5 is one) are inflated by the
    library's host helper strq_inflate_chunks in one call per dataset; the per-chunk Python loop stays as the fallback.
    Both must return the samples exactly: plain deflate, shuffle + deflate, ragged last chunk, a dataset shorter than a
    chunk, a damaged chunk."""
    import zlib
    import numpy as np
    from strique_amd import fast5, h5write
    rng = np.random.default_rng(3)

    def shuffled(part):
        return np.frombuffer(part.astype("<i2").tobytes(), np.uint8).reshape(-1, 2).T.tobytes()

    def bulk(signals, shuffle):
        f = h5write._File()
        top = {}
        for i, sig in enumerate(signals):
            if shuffle:
                ds = h5write._chunked_dataset(f, sig, 4096, [(2, "shuffle", (2,)), (1, "deflate", (4,))],
                                              lambda part: zlib.compress(shuffled(part), 4))
            else:
                ds = h5write._deflate_dataset(f, sig)
            raw = h5write._group(f, {"Signal": ds}, attrs=[("read_id", "r%d" % i)])
            top["read_r%d" % i] = h5write._group(f, {"Raw": raw})
        return h5write._finish(f, h5write._group(f, top))

    signals = [rng.integers(-3000, 3000, n).astype(np.int16) for n in (1, 17, 4096, 8192, 8193, 50001, 375013)]
    for shuffle in (False, True):
        path = tmp_path / ("bulk%d.fast5" % shuffle)
        path.write_bytes(bulk(signals, shuffle))
        f = fast5.H5File(str(path))
        located = [f.dataset("/read_r%d/Raw/Signal" % i) for i in range(len(signals))]          # strq_h5_locate + strq_inflate_chunks
        monkeypatch.setenv("STRQ_H5_PYTHON", "1")                                               # the Python parse, native inflate
        native = [f.dataset("/read_r%d/Raw/Signal" % i) for i in range(len(signals))]
        calls = []
        orig = fast5.H5File._native_inflate
        monkeypatch.setattr(fast5.H5File, "_native_inflate", lambda self, *a: calls.append(1) and False)
        python = [f.dataset("/read_r%d/Raw/Signal" % i) for i in range(len(signals))]           # the per-chunk Python loop
        monkeypatch.setattr(fast5.H5File, "_native_inflate", orig)
        monkeypatch.delenv("STRQ_H5_PYTHON")
        assert len(calls) == len(signals)
        for s, a, b, c in zip(signals, native, python, located):
            assert np.array_equal(s, a) and np.array_equal(s, b) and np.array_equal(s, c) and c.dtype == np.int16
    # the reference's own data file goes through the same path
    rid, sig = fast5.read_raw(os.path.join(GOLDEN, "c9orf72.fast5"))[0]
    assert rid == "ce47b364-ed6e-4409-808a-1041c0b5aac2" and len(sig) == 284184 and int(sig.min()) == -4096 and int(sig.max()) == 3008
    # a damaged chunk is reported, not returned as zeros
    blob = bytearray(bulk([signals[-1]], False))
    pos = bytes(blob).find(zlib.compress(signals[-1][:8192].tobytes(), 4)[:16])
    assert pos > 0
    blob[pos + 20] ^= 0xFF; blob[pos + 21] ^= 0xFF; blob[pos + 40] ^= 0xFF
    with pytest.raises(Exception):
        fast5.H5File(bytes(blob)).dataset("/read_r0/Raw/Signal")
    # the helper itself, with libdeflate (when the image has libdeflate.so.0) and with zlib: same bytes; what the file's B-tree
    # says is never trusted -- an address near INT64_MAX, a chunk that inflates to less than a whole chunk, a bad stream
    import ctypes
    import subprocess
    import sys
    from strique_amd import ffi
    lib = ffi.load_library()
    lib.strq_inflate_chunks.restype = ctypes.c_int64
    sig = signals[-1]
    chunks = [zlib.compress(np.pad(sig[i:i + 8192], (0, max(0, 8192 - len(sig[i:i + 8192])))).tobytes(), 4) for i in range(0, len(sig), 8192)]
    base = np.frombuffer(b"".join(chunks), np.uint8)
    csz = np.array([len(c) for c in chunks], np.int32); addr = np.concatenate([[0], np.cumsum(csz)[:-1]]).astype(np.int64)
    eoff = np.arange(len(chunks), dtype=np.int64) * 8192

    def call(addr_, csz_, n_chunks=len(chunks)):
        out = np.zeros(len(sig), np.int16)
        rc = lib.strq_inflate_chunks(ctypes.c_void_p(base.ctypes.data), ctypes.c_int64(base.size), ctypes.c_int64(n_chunks), ctypes.c_void_p(addr_.ctypes.data),
                                     ctypes.c_void_p(csz_.ctypes.data), ctypes.c_void_p(eoff.ctypes.data), ctypes.c_int32(2), ctypes.c_int32(0), ctypes.c_int64(8192),
                                     ctypes.c_int64(len(sig)), ctypes.c_void_p(out.ctypes.data))
        return rc, out

    rc, out = call(addr, csz)
    assert rc == 0 and np.array_equal(out, sig)
    bad = addr.copy(); bad[3] = np.iinfo(np.int64).max - 5
    assert call(bad, csz)[0] == -(3 + 2)
    short = zlib.compress(sig[:100].tobytes(), 4)                       # a valid stream of less than a chunk
    base = np.frombuffer(b"".join(chunks) + short, np.uint8)
    bad = addr.copy(); badsz = csz.copy(); bad[2] = int(np.sum(csz)); badsz[2] = len(short)
    assert call(bad, badsz)[0] == -(2 + 2)
    base = np.frombuffer(b"".join(chunks), np.uint8)
    prog = ("import sys, ctypes; sys.path.insert(0, %r)\nfrom strique_amd import ffi\nprint(ffi.load_library().strq_inflate_backend())" % ROOT)
    backends = {}
    for name, env in (("default", {}), ("zlib", {"STRQ_NO_LIBDEFLATE": "1"})):
        backends[name] = int(subprocess.run([sys.executable, "-c", prog], env=dict(os.environ, **env), stdout=subprocess.PIPE, check=True).stdout.split()[-1])
    assert backends["zlib"] == 0 and backends["default"] == (1 if os.path.exists("/usr/lib/x86_64-linux-gnu/libdeflate.so.0") else backends["default"])
