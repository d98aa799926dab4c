"""Mask the repeat region out of raw nanopore reads: the reference's scripts/fast5Masker.py.

    fast5Masker.py <index> <counts.tsv> <output dir> [--format single|bulk]

Same steps as the reference (scripts/fast5Masker.py:45-92): read the `count` TSV, write the evaluated
read IDs to `<output>/reads.txt`, extract those reads from the indexed archive into `<output>` (one
`<read_id>.fast5` per read, or one bulk `reads.fast5`), index the output (`<output>/reads.fofn`) and
remove the samples `[offset, offset + ticks)` of every record from its signal.

The reference does the last step in place through h5py (`fast5Index.set_raw`, STRique_lib/fast5Index.py:
86-95); h5py is not available here, so the extracted files are *written* with the masked signal by
strique_amd/h5write.py: the read's groups and their string / integer / float attributes (`Raw`,
`channel_id`, `context_tags`, `tracking_id`) are carried over, `duration` is set to the new length, the
signal is stored uncompressed.  Basecall tables under `Analyses` are not copied.  Host code only.
"""
import argparse
import os
import re
import sys
from collections import namedtuple

import numpy as np

from . import fast5, h5write
from .cli import Fast5Index

STRique_record = namedtuple('STRique_record', ['ID', 'target', 'strand', 'count', 'score_prefix', 'score_suffix',
                                               'log_p', 'offset', 'ticks', 'mod'])
_META = ("channel_id", "context_tags", "tracking_id")


def read_counts(path):
    with open(path, 'r') as fp:
        rows = (row.strip().split('\t') for row in fp if row.strip() and not row.startswith('ID'))
        return [STRique_record(*row[:3], int(row[3]), *[float(x) for x in row[4:7]], int(row[7]), int(row[8]), row[9])
                for row in rows]


def _source(index, read_id):
    """(H5File, path of the read's Raw group, {meta group name: path}) of an indexed read, or None."""
    if read_id not in index.index:
        return None
    parts = re.split(r'(\.fast5|\.tar)/', index.index[read_id])
    if len(parts) == 1:
        f = index._file(os.path.join(index.dir, parts[0]))
    elif parts[1] == '.fast5':
        f = index._file(os.path.join(index.dir, parts[0] + '.fast5'))
    else:
        import tarfile
        with tarfile.open(os.path.join(index.dir, parts[0] + '.tar')) as tar:
            f = fast5.H5File(tar.extractfile(tar.getmember(parts[2])).read())
    if len(parts) > 1 and parts[1] == '.fast5':
        top = "/" + parts[2].strip('/')
        raw = top + "/Raw"
        meta = {m: top + "/" + m for m in _META if m in f.listdir(top)}
    else:
        raw = "/Raw/Reads/" + f.listdir("/Raw/Reads")[0]
        keys = f.listdir("/UniqueGlobalKey") if "UniqueGlobalKey" in f.listdir("/") else []
        meta = {m: "/UniqueGlobalKey/" + m for m in _META if m in keys}
    return f, raw, meta


def mask_signal(signal, records):
    """Samples outside every record's [offset, offset + ticks) (scripts/fast5Masker.py:88-90)."""
    keep = np.ones(signal.shape, dtype=bool)
    for r in records:
        keep[r.offset:r.offset + r.ticks] = False
    return signal[keep]


def run(index_file, counts_file, output, fmt='bulk', log=sys.stderr):
    records = read_counts(counts_file)
    os.makedirs(output, exist_ok=True)
    with open(os.path.join(output, 'reads.txt'), 'w') as fp:
        fp.write('\n'.join(r.ID for r in records) + '\n')
    index = Fast5Index(index_file)
    by_id = {}
    for r in records:
        by_id.setdefault(r.ID, []).append(r)
    bulk = {"attrs": {"file_version": "2.0"}, "groups": {}, "datasets": {}}
    n_done = 0
    for rid, recs in by_id.items():
        src = None
        try:
            src = _source(index, rid)
        except Exception as e:
            print("[WARNING] Could not get raw signal for read %s (%s)" % (rid, e), file=log)
        if src is None:
            print("[WARNING] Could not get raw signal for read %s" % rid, file=log)
            continue
        f, raw_path, meta = src
        try:
            signal = f.dataset(raw_path + "/Signal")
        except Exception as e:
            print("[WARNING] Could not get raw signal for read %s (%s)" % (rid, e), file=log)
            continue
        masked = mask_signal(signal, recs).astype(np.int16)
        raw_attrs = dict(f.attrs(raw_path)); raw_attrs["duration"] = len(masked); raw_attrs.setdefault("read_id", rid)
        raw_node = {"attrs": raw_attrs, "datasets": {"Signal": (masked, {})}}
        meta_nodes = {m: {"attrs": dict(f.attrs(p))} for m, p in meta.items()}
        if fmt == 'single':
            number = raw_attrs.get("read_number", 0)
            tree = {"attrs": {"file_version": "1.1"},
                    "groups": {"Raw": {"groups": {"Reads": {"groups": {"Read_%s" % number: raw_node}}}},
                               "UniqueGlobalKey": {"groups": meta_nodes}}}
            with open(os.path.join(output, "%s.fast5" % rid), 'wb') as fp:
                fp.write(h5write.write_tree(tree))
        else:
            bulk["groups"]["read_" + rid] = {"attrs": {"run_id": meta_nodes.get("tracking_id", {}).get("attrs", {}).get("run_id", "")},
                                             "groups": dict(meta_nodes, Raw=raw_node)}
        n_done += 1
    if fmt != 'single' and bulk["groups"]:
        with open(os.path.join(output, "reads.fast5"), 'wb') as fp:
            fp.write(h5write.write_tree(bulk))
    with open(os.path.join(output, 'reads.fofn'), 'w') as fp:
        for record in Fast5Index.index_records(output):
            fp.write(record + '\n')
    return n_done


def main(argv=None):
    parser = argparse.ArgumentParser(description="Mask region in raw nanopore fast5 file.")
    parser.add_argument("index", help="Path to input fast5 index")
    parser.add_argument("counts", help="Path to STRique count output file")
    parser.add_argument("output", help="Path to output .fast5 directory with masked reads")
    parser.add_argument("--format", default='bulk', choices=['single', 'bulk'], help="Output fast5 format")
    args = parser.parse_args(argv)
    print("[INFO] Reading repeat counts, extracting and masking evaluated reads into %s" % args.output, file=sys.stderr)
    n = run(args.index, args.counts, args.output, args.format)
    print("[INFO] %d read(s) written." % n, file=sys.stderr)


if __name__ == '__main__':
    main()
