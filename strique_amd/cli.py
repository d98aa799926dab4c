"""`STRique.py`-compatible command line: `index` and `count`.

Mirrors the reference's CLI surface (scripts/STRique.py:874-945): same positional arguments and
flags, the same `repeat_config.tsv` / JSON config format (parse_config, :836-868), SAM record
decoding and locus intersection (repeatDetector, :648-705) and TSV output (outputWriter, :711-727).
What changes is the engine: instead of `--t` worker processes calling the CPU aligner one read at
a time, the records of the SAM stream are collected and handed to the GPU in batches
(`repeatCounter.detect_batch`); rows are written in input order.
"""
import argparse
import glob
import json
import os
import re
import sys
import tarfile
import tempfile
import threading
from collections import defaultdict, deque

import numpy as np

HEADER = ['ID', 'target', 'strand', 'count', 'score_prefix', 'score_suffix', 'log_p', 'offset', 'ticks', 'mod']
LEVELS = ['error', 'warning', 'info', 'debug']


class Log(object):
    """Messages to stderr.  The engine thread of `run_count` logs while the main thread does: one write per message,
    under a lock, so that lines never interleave."""

    def __init__(self, level='warning'):
        import threading
        self.level = LEVELS.index(level)
        self._lock = threading.Lock()

    def __call__(self, message, level='info'):
        if LEVELS.index(level) <= self.level:
            with self._lock:
                sys.stderr.write("[%s] %s\n" % (level.upper(), message))
                sys.stderr.flush()


def parse_config(repeat_config_file, param_config_file=None, log=None):
    """{'repeat': {name: (chr, begin, end, repeat, prefix, suffix)}, 'align': dict|None, 'HMM': dict|None}"""
    repeats = {}
    with open(repeat_config_file, 'r') as fp:
        next(fp)                                         # header line
        for line in fp:
            cols = line.rstrip().split()
            if len(cols) == 7:
                repeats[cols[3]] = (cols[0], int(cols[1]), int(cols[2]), cols[4], cols[5], cols[6])
            elif log:
                log("Config: Repeat config column mismatch while parsing \n%s" % line, 'error')
    config = {'repeat': repeats, 'align': None, 'HMM': None}
    if param_config_file:
        with open(param_config_file) as fp:
            ld_conf = json.load(fp)
        if not isinstance(ld_conf, dict):
            raise SystemExit('Config: file format broken')
        for key in ('align', 'HMM'):
            if key not in ld_conf:
                raise SystemExit('Config: Error loading HMM config file, missing %s' % key)
            if not isinstance(ld_conf[key], dict):
                raise SystemExit('Config: file format broken')
        config['align'] = ld_conf['align']
        config['HMM'] = ld_conf['HMM']
    return config


class SamRecord(object):
    __slots__ = ('QNAME', 'FLAG', 'RNAME', 'POS', 'TLEN', 'CLIP_BEGIN', 'CLIP_END', 'QLEN')

    def __init__(self):
        self.QNAME = ''; self.FLAG = 0; self.RNAME = ''; self.POS = 0; self.TLEN = 0; self.CLIP_BEGIN = 0; self.CLIP_END = 0; self.QLEN = 0


def decode_cigar(cigar):
    return [(int(op[:-1]), op[-1]) for op in re.findall(r'(\d*\D)', cigar)]


def ops_length(ops, recOps='MIS=X'):
    return sum(n for n, op in ops if op in recOps)


def decode_sam(sam_line):
    """QNAME, FLAG, RNAME, POS, reference span from the CIGAR, soft/hard clips (STRique.py:656-671)."""
    cols = sam_line.rstrip().split('\t')
    sr = SamRecord()
    if len(cols) >= 11:
        try:
            sr.QNAME = cols[0]; sr.FLAG = int(cols[1]); sr.RNAME = cols[2]; sr.POS = int(cols[3])
            ops = decode_cigar(cols[5])
            sr.TLEN = ops_length(ops, recOps='MDN=X')
            sr.CLIP_BEGIN = sum(n for n, op in ops[:2] if op in 'SH')
            sr.CLIP_END = sum(n for n, op in ops[-2:] if op in 'SH')
            sr.QLEN = max(ops_length(ops, recOps='MIS=XH'), len(cols[9]) if cols[9] != '*' else 0)      # read length: the cost of its alignments
        except Exception:
            return SamRecord()
    return sr


def intersect_targets(sr, loci):
    """Targets whose locus lies inside the (clip-extended) alignment (STRique.py:673-679)."""
    return [name for name, begin, end in loci.get(sr.RNAME, [])
            if begin > sr.POS - sr.CLIP_BEGIN and end < sr.POS + sr.TLEN + sr.CLIP_END]


class Fast5Index(object):
    """`path[.fast5/group | .tar/member]<TAB>read_id` index (STRique_lib/fast5Index.py:45-60,220-233)."""

    def __init__(self, index_file):
        if not os.path.exists(index_file):
            raise RuntimeError("[Error] Raw fast5 index file %s not found." % index_file)
        with open(index_file) as fp:
            self.index = {rid: path for path, rid in (line.split('\t') for line in fp.read().split('\n') if line)}
        self.dir = os.path.dirname(index_file)
        self._joined = {}
        self._open = {}                  # path -> H5File, the few most recently used (bulk files hold thousands of reads)
        self._lock = threading.Lock()

    def _join(self, rel):
        """os.path.join(self.dir, rel), memoised per file (thousands of reads share a bulk file)."""
        full = self._joined.get(rel)
        if full is None:
            full = self._joined[rel] = os.path.join(self.dir, rel)
        return full

    def _file(self, path):
        from . import fast5
        with self._lock:                 # get_raw may be called from the reader threads of `count`
            f = self._open.pop(path, None)
            if f is None:
                f = fast5.H5File(path)
                # dozens of reader threads work on tasks from several bulk files at once: with only a few files kept, every
                # task re-opened (mmap) and dropped (munmap: a TLB shoot-down on every CPU the process runs on) its file
                while len(self._open) >= 32:
                    self._open.pop(next(iter(self._open)))
            self._open[path] = f         # most recently used last
            return f

    def get_raw(self, read_id, alloc=None, defer=False):
        """The raw signal of a read, or None.  `alloc`, `defer`: see fast5.H5File.dataset (compressed datasets: where the output
        array comes from; an InflatePlan instead of the samples, for fast5.inflate_plans to fill a whole task's reads at once)."""
        from . import fast5
        where = self.index.get(read_id)
        if where is None:
            return None
        cut = where.find('.fast5/')          # the common case, a read of a bulk file, without the regular expression
        if cut >= 0 and '.tar/' not in where:
            f = self._file(self._join(where[:cut + 6]))
            return f.dataset("/%s/Raw/Signal" % where[cut + 7:].strip('/'), alloc, defer)
        parts = re.split(r'(\.fast5|\.tar)/', where)
        if len(parts) == 1:
            f = self._file(os.path.join(self.dir, parts[0]))
            grp = "/Raw/Reads/" + f.listdir("/Raw/Reads")[0]
            return f.dataset(grp + "/Signal", alloc, defer)
        if parts[1] == '.fast5':
            f = self._file(os.path.join(self.dir, parts[0] + '.fast5'))
            return f.dataset("/%s/Raw/Signal" % parts[2].strip('/'), alloc, defer)
        with tarfile.open(os.path.join(self.dir, parts[0] + '.tar')) as tar:
            data = tar.extractfile(tar.getmember(parts[2])).read()
        f = fast5.H5File(data)
        return f.dataset("/Raw/Reads/" + f.listdir("/Raw/Reads")[0] + "/Signal")

    @staticmethod
    def index_records(path, recursive=False, out_prefix=""):
        from . import fast5
        if os.path.isfile(path):
            files = [path]
        elif recursive:
            files = [os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs if f.endswith(('.fast5', '.tar'))]
        else:
            files = glob.glob(os.path.join(path, '*.fast5')) + glob.glob(os.path.join(path, '*.tar'))
        for fpath in sorted(files):
            rel = os.path.normpath(os.path.join(out_prefix, os.path.dirname(os.path.relpath(fpath, start=path)), os.path.basename(fpath)))
            if fpath.endswith('.tar'):
                with tarfile.open(fpath) as tar:
                    for m in tar.getmembers():
                        if m.name.endswith('.fast5'):
                            try:
                                rid = fast5.read_raw(tar.extractfile(m).read())[0][0]
                                yield "\t".join([os.path.normpath(os.path.join(rel, m.name)), rid])
                            except Exception:
                                print("[ERROR] Failed to open %s, skip file for indexing" % m.name, file=sys.stderr)
                continue
            try:                                      # like the reference: a file that cannot be opened is reported and skipped
                f = fast5.H5File(fpath)
                top = f.listdir("/")
                recs = []
                if "Raw" in top:
                    rd = f.listdir("/Raw/Reads")[0]
                    grp = "/Raw/Reads/" + rd
                    rid = f.attrs(grp).get("read_id")
                    if rid is None:                       # attribute in a form the subset reader does not decode
                        rid = os.path.splitext(os.path.basename(fpath))[0]
                    recs.append("\t".join([rel, rid]))
                else:
                    for g in top:
                        if g.startswith("read_"):
                            recs.append("\t".join([os.path.join(rel, g), f.attrs("/%s/Raw" % g).get("read_id", g[5:])]))
            except Exception as e:
                print("[ERROR] Failed to open %s (%s), skip file for indexing" % (fpath, e), file=sys.stderr)
                continue
            for r in recs:
                yield r


def count(argv):
    parser = argparse.ArgumentParser(description="STR Detection in raw nanopore data")
    parser.add_argument("f5Index", help="Fast5 index")
    parser.add_argument("model", help="Pore model")
    parser.add_argument("repeat", help="Repeat region config file")
    parser.add_argument("--out", default=None, help="Output file name, if not given print to stdout")
    parser.add_argument("--algn", default=None, help="Alignment in sam format, if not given read from stdin")
    parser.add_argument("--mod_model", default=None, help="Base modification pore model")
    parser.add_argument("--config", help="Config file with HMM transition probabilities")
    parser.add_argument("--t", type=int, default=0, help="Reader threads that fetch and inflate raw signals ahead of the GPU batches (the reference's worker-process count); "
                                                          "0 (default): this rank's share of the CPUs the job may use (affinity mask, cgroup quota), at most 24")
    parser.add_argument("--log_level", default='warning', choices=LEVELS, help="Log level")
    parser.add_argument("--batch", type=int, default=2048, help="Reads per GPU batch.  (The library works in sub-batches of 16 reads per CU, two of them in flight, so a batch of "
                                                                "8192 and more keeps the HMM decode of one sub-batch under the alignments of the next -- but `count` is bound by reading "
                                                                "the files, and smaller batches keep readers and GPU busy at the same time: profiles/r06_cli_probe.txt)")
    parser.add_argument("--device", type=int, default=0, help="HIP device")
    parser.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="torch.distributed backend when launched with torchrun (default: nccl = RCCL)")
    parser.add_argument("--share-device", action="store_true", help="testing: every rank uses --device instead of its LOCAL_RANK")
    parser.add_argument("--strict", action="store_true", help="Exit with status 2 when any read could not be processed (the reference only logs such reads and exits 0)")
    args = parser.parse_args(argv)
    log = Log(args.log_level)
    config = parse_config(args.repeat, args.config, log)
    for path, what in ((args.f5Index, "Fast5 index file"), (args.model, "Pore model file")):
        if not os.path.isfile(path):
            log("Main: %s does not exist." % what, 'error'); raise SystemExit(1)
    if args.mod_model and not os.path.isfile(args.mod_model):
        log("Main: Modification pore model file does not exist.", 'error'); raise SystemExit(1)
    from . import dist as sdist
    rank, world, local = sdist.env_rank_world()
    if world > 1:
        # one process per GPU (torchrun): the accepted (read, target) pairs are dealt to the ranks by read
        # length (strique_amd.dist.shard_indices), rank 0 gathers fixed-size result records plus the
        # modification strings once at the end (strique_amd.dist.gather_results) and writes the rows in input order
        if not args.algn:
            log("Main: --algn FILE is required when running on several GPUs (stdin cannot be shared).", 'error'); raise SystemExit(1)
        # this rank's share of the host's CPUs FIRST: sched_setaffinity pins the calling thread and what it creates afterwards, so the
        # threads torch.distributed / RCCL / gloo start in init_process_group -- and the reader, upload and statistics threads -- follow
        sdist.pin_rank_cpus()
        sdist.init_process_group(backend=args.backend)
    _tune_allocator()
    from .counter import repeatCounter
    device = args.device if (world == 1 or args.share_device) else local
    counter = repeatCounter(args.model, mod_model_file=args.mod_model, align_config=config['align'],
                            HMM_config=config['HMM'], device=device)
    loci = defaultdict(list)
    for name, (chrom, begin, end, repeat, prefix, suffix) in config['repeat'].items():
        try:
            counter.add_target(name, repeat, prefix, suffix)
        except ValueError:
            raise
        except Exception as e:                # e.g. a flank longer than the compiled kernel shapes cover
            log("Main: target %s is not supported by the GPU engine (%s); its reads are skipped." % (name, e), 'error')
            continue
        loci[chrom].append((name, begin, end))
    f5 = Fast5Index(args.f5Index)
    stream = open(args.algn) if args.algn else sys.stdin
    out = (open(args.out, 'w') if args.out else sys.stdout) if rank == 0 else None
    readers = args.t
    if readers <= 0:
        # one process per GPU: every rank takes its share of the cores (LOCAL_WORLD_SIZE is set by torchrun) for its reader threads, at most
        # 24; the staging threads of the library and the engine thread run beside them
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        if share == (os.cpu_count() or 1):          # not pinned: an equal share by count
            share //= local_world
        quota = sdist.cpu_quota()                   # a container's CPU quota counts, not the CPUs it shows (16 of 256 on the MI355X boxes)
        if quota:
            share = min(share, max(1, int(quota) // local_world))
        readers = max(1, min(24, share))            # inflating is what the readers do: one per core they can get, 16 ... 24 measure the same end to end
    stats = {}
    fault = 0
    try:
        rows = run_count(stream, loci, f5.get_raw, counter, log, args.batch, rank, world, out if world == 1 else None, readers=readers, stats=stats)
    except DeviceFault:
        if world == 1:
            raise SystemExit(3)
        fault = 1; rows = []
    if world > 1:
        import torch.distributed as dist
        # a rank that lost its device must not leave the others waiting in the gather: every rank learns about it
        # here (the faulty rank arrives at once, the others when their share is done) and all of them exit 3
        if sdist.any_rank(fault):
            if rank == 0:
                log("Main: a rank reported a device error; no output written.", 'error')
            dist.destroy_process_group()
            raise SystemExit(3)
        merged = gather_rows(rows, stats["items"], sdist)
        if rank == 0:
            write_rows(out, merged)
        dist.barrier()
        dist.destroy_process_group()
    if args.out and out is not None:
        out.close()
    if stats.get("failed"):
        # like the reference (STRique.py:704-713): reads that fail are logged, the run itself succeeds
        log("Main: %d read(s) could not be processed (see warnings above)." % stats["failed"], 'error')
        if args.strict:
            raise SystemExit(2)


def _tune_allocator():
    """The reader threads allocate one array per read (0.2 ... 8 MB).  glibc serves such sizes with a fresh mmap each time:
    every read then costs a map, a page fault per 4 KB while it is filled and an unmap, all under the process-wide mm lock,
    and the readers stop scaling beyond a few threads.  Raise the mmap threshold to its maximum (32 MB) and keep freed memory:
    the arrays come out of the per-thread arenas and their pages are reused."""
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
        libc.mallopt(M_MMAP_THRESHOLD, 32 << 20)
        libc.mallopt(M_TRIM_THRESHOLD, (1 << 31) - 1)
    except (OSError, AttributeError):
        pass


class DeviceFault(Exception):
    """The GPU engine reported a device error (fault, out of memory): it will not go away read by read."""


ROW_DTYPE = np.dtype([("count", np.int32), ("valid", np.int32), ("score_prefix", np.float64), ("score_suffix", np.float64),
                      ("log_p", np.float64), ("offset", np.int64), ("ticks", np.int64)])


def format_row(qname, target, strand, res):
    return '\t'.join(str(x) for x in (qname, target, strand) + tuple(res))


def gather_rows(rows, items, sdist):
    """Rows of this rank -> fixed-size records + modification strings -> one gather -> on rank 0 the
    merged [(sequence number, TSV row)] in input order (None elsewhere).  `items`: every accepted
    (qname, strand, target) of the input, which each rank derives from the same SAM file."""
    rec = np.zeros(len(rows), ROW_DTYPE); mods = []; idx = np.zeros(len(rows), np.int64)
    for k, (seq, res) in enumerate(rows):
        idx[k] = seq
        if res is None:
            mods.append("")
            continue
        n, sp, ss, p, offset, ticks, mod = res
        rec[k] = (n, 1, sp, ss, float(p), offset, ticks)
        mods.append(mod)
    full, full_mods = sdist.gather_results(rec, idx, len(items), mods)
    if full is None:
        return None
    merged = []
    for seq, (qname, strand, target) in enumerate(items):
        r = full[seq]
        if not r["valid"]:
            continue
        n = int(r["count"]); lp = float(r["log_p"])
        p = lp if (n or lp != 0) else 0              # the reference prints the integer 0 for a failed gate (STRique.py:602,616)
        merged.append((seq, format_row(qname, target, strand, (n, float(r["score_prefix"]), float(r["score_suffix"]), p,
                                                                 int(r["offset"]), int(r["ticks"]), full_mods[seq]))))
    return merged


def write_rows(out, rows, header=True):
    if header:
        print('\t'.join(HEADER), file=out)
    for _, row in rows:
        print(row, file=out)
    out.flush()


def route(stream, loci, log):
    """Accepted SAM records of `stream`: (qname, strand, [targets], read length)."""
    for line in stream:
        if line.startswith('@'):
            continue
        sr = decode_sam(line)
        if not sr.QNAME:
            log("Detector: Error parsing alignment \n%s" % line, 'error'); continue
        targets = intersect_targets(sr, loci)
        if not targets:
            log("Detector: No target for %s" % sr.QNAME, 'debug'); continue
        yield sr.QNAME, ('+' if sr.FLAG & 0x10 == 0 else '-'), targets, sr.QLEN


def run_count(stream, loci, get_raw, counter, log, batch_size, rank=0, world=1, out=None, readers=0, stats=None):
    """Route the SAM records of `stream` to their targets, run this rank's share through
    `counter.detect_batch` and return [(sequence number, result tuple or TSV row)].

    Single process: rows [(seq, TSV row)] are also written to `out` as soon as their batch is done.
    Several ranks: the records are read first (the SAM carries the read lengths), the accepted
    (read, target) pairs are dealt to the ranks by descending read length
    (strique_amd.dist.shard_indices: the DP cost of a read is proportional to its length), and the
    return value is [(seq, result tuple)] for `gather_rows`.
    `readers` > 0: raw signals are fetched by that many threads ahead of the GPU batches (inflating
    the deflate chunks of a fast5 releases the GIL and is what bounds a `count` run on real files);
    the order of the rows does not change."""
    from .ffi import StriqueHipError, STRQ_ERR_ARG, STRQ_ERR_UNSUPPORTED
    if stats is None:
        stats = {}
    stats.setdefault("failed", 0)
    if out is not None:
        print('\t'.join(HEADER), file=out)
    rows = []
    records = route(stream, loci, log)
    mine_set = None
    if world > 1:
        from . import dist as sdist
        records = list(records)
        items, cost = [], []
        for qname, strand, targets, qlen in records:
            for t in targets:
                items.append((qname, strand, t)); cost.append(qlen)
        mine_set = set(int(i) for i in sdist.shard_indices(len(items), rank, world, cost))
        stats["items"] = items

    import threading
    faulted = threading.Event()

    def run_batch(batch):
        """Engine thread: one batch through the GPU pipeline (the library releases the GIL for the whole call), its rows
        formatted; returns (rows, number of failed reads)."""
        failed = 0
        results = None
        if faulted.is_set():                                      # queued behind the batch that faulted: the device is not touched again
            raise DeviceFault("not run: the device failed in an earlier batch")
        try:
            results = counter.detect_batch([(t, raw, s) for _, _, t, s, raw in batch])
        except StriqueHipError as e:
            if e.code not in (STRQ_ERR_ARG, STRQ_ERR_UNSUPPORTED):
                # a device fault or an out-of-memory condition will not go away read by read
                faulted.set()
                log("Detector: device error, giving up: %s" % e, 'error')
                raise DeviceFault(str(e))
            log("Detector: batch rejected (%s), retrying read by read" % e, 'warning')
        except Exception as e:                                    # a bad batch never kills the run
            log("Detector: batch failed (%s), retrying read by read" % e, 'warning')
        if results is None:
            results = []
            for _, _, t, s, raw in batch:
                try:
                    results.append(counter.detect(t, raw, s))
                except StriqueHipError as e1:
                    if e1.code not in (STRQ_ERR_ARG, STRQ_ERR_UNSUPPORTED):
                        faulted.set()
                        log("Detector: device error, giving up: %s" % e1, 'error')
                        raise DeviceFault(str(e1))
                    log("Detector: read failed: %s" % e1, 'warning'); results.append(None); failed += 1
                except Exception as e1:
                    log("Detector: read failed: %s" % e1, 'warning'); results.append(None); failed += 1
        done = []
        for (seq, qname, target, strand, _), res in zip(batch, results):
            if world > 1:
                done.append((seq, res))
            elif res is not None:
                done.append((seq, format_row(qname, target, strand, res)))
        return done, failed

    # The batches run on an engine thread, one at a time and in order, while this thread routes the next SAM records and
    # collects their signals: the GPU call of batch k overlaps the host-side preparation of batch k + 1 (at 50 kb per read
    # that preparation -- SAM decode, index look-ups, waiting for the reader threads -- costs about as much as the call).
    from concurrent.futures import ThreadPoolExecutor
    engine = ThreadPoolExecutor(max_workers=1)
    in_flight = deque()

    def collect(keep):
        while len(in_flight) > keep:
            done, failed = in_flight.popleft().result()          # re-raises DeviceFault from the engine thread
            stats["failed"] += failed
            rows.extend(done)
            if out is not None:
                write_rows(out, done, header=False)

    def flush(batch):
        if not batch:
            return
        collect(1)                                               # at most one batch running and one waiting
        in_flight.append(engine.submit(run_batch, batch))

    import inspect
    try:
        takes_alloc = "alloc" in inspect.signature(get_raw).parameters
    except (TypeError, ValueError):
        takes_alloc = False

    def fetch(qname, alloc=None, defer=False):
        try:
            return get_raw(qname, alloc, defer) if takes_alloc else get_raw(qname)
        except NotImplementedError as e:          # a storage layout / filter the HDF5 subset reader does not cover
            log("Detector: cannot read %s: %s" % (qname, e), 'error'); stats["failed"] += 1
            return None
        except Exception as e:
            log("Detector: cannot read %s: %s" % (qname, e), 'warning')
            return None

    pool = None
    if readers > 1:
        pool = ThreadPoolExecutor(max_workers=readers)
    CHUNK = 32                             # reads per reader task: one future per read costs more Python time than a contiguous read does
    pending = deque()                      # (future or list of raw signals, [(qname, strand, [(seq, target)])]), in input order
    group = []
    batch = []

    def fetch_many(qnames):
        alloc = None          # (huge-page slabs shared by the reads of a task measured no gain over one inflate call per task: profiles/r04_reader.md -- removed)
        if not takes_alloc or os.environ.get("STRQ_READ_ONE_BY_ONE"):
            return [fetch(q, alloc) for q in qnames]
        # compressed datasets: located first (Python, under the interpreter lock), then inflated together in one native call
        from .fast5 import InflatePlan, inflate_plans
        got = [fetch(q, alloc, True) for q in qnames]
        plans = [(i, g) for i, g in enumerate(got) if isinstance(g, InflatePlan)]
        if plans:
            try:
                errors = inflate_plans([p for _, p in plans])
            except Exception as e:
                errors = [str(e)] * len(plans)
            for (i, p), err in zip(plans, errors):
                if err is None:
                    got[i] = p.out
                else:
                    log("Detector: cannot read %s: %s" % (qnames[i], err), 'warning'); got[i] = None
        return got

    def push_group():
        nonlocal group
        if group:
            names = [g[0] for g in group]
            pending.append((pool.submit(fetch_many, names) if pool is not None else fetch_many(names), group))
            group = []

    def drain(keep):
        nonlocal batch
        while len(pending) > keep:
            raws, grp = pending.popleft()
            if pool is not None:
                raws = raws.result()
            for (qname, strand, mine), raw in zip(grp, raws):
                if raw is None:
                    log("Detector: No fast5 for ID %s" % qname, 'warning'); continue
                for sq, t in mine:
                    batch.append((sq, qname, t, strand, raw))
                if len(batch) >= batch_size:
                    flush(batch); batch = []

    seq = 0
    lookahead = max(1, 2 * batch_size // CHUNK) if pool is not None else 0
    finished = False
    try:
        for qname, strand, targets, _qlen in records:
            mine = [(seq + i, t) for i, t in enumerate(targets) if mine_set is None or (seq + i) in mine_set]
            seq += len(targets)
            if not mine:
                continue
            group.append((qname, strand, mine))
            if len(group) >= CHUNK:
                push_group()
                drain(lookahead)
        push_group()
        drain(0)
        flush(batch)
        collect(0)
        finished = True
    finally:
        if finished:
            engine.shutdown(wait=True)
            if pool is not None:
                pool.shutdown()
        else:
            # an exception is on its way out (a DeviceFault from the engine thread, a broken input): the batch queued behind
            # the failed one must not be handed to a device that may be hung, and nobody waits for it -- the caller exits,
            # or tells the other ranks (dist.any_rank), right away
            for fut in in_flight:
                fut.cancel()
            engine.shutdown(wait=False, cancel_futures=True)
            if pool is not None:
                pool.shutdown(wait=False, cancel_futures=True)
    return rows


def index(argv):
    parser = argparse.ArgumentParser(description="Fast5 raw data archive indexing")
    parser.add_argument("input", help="Input batch or directory of batches")
    parser.add_argument("--recursive", action='store_true', help="Recursively scan input")
    parser.add_argument("--out_prefix", default="", help="Prefix for file paths in output")
    parser.add_argument("--tmp_prefix", default=None, help="Prefix for temporary data")
    args = parser.parse_args(argv)
    for record in Fast5Index.index_records(args.input, recursive=args.recursive, out_prefix=args.out_prefix):
        print(record)


def plot(argv):
    """The `plot` command (argument contract of scripts/STRique.py:948-960).  The figures themselves are this
    package's own: strique_amd/plotting.py draws them from the TSV columns `offset` / `ticks` / scores."""
    parser = argparse.ArgumentParser(description="Signal plots over STR expansions")
    parser.add_argument("f5Index", help="Fast5 index")
    parser.add_argument("--counts", default=None, help="Repeat count output from STRique, if not given read from stdin")
    parser.add_argument("--output", default=None, help="Output directory for plots, use instead of interactive GUI")
    parser.add_argument("--format", default='png', choices=["png", "pdf", "svg"], help="Output format when writing to files")
    parser.add_argument("--width", default=16, type=int, help="Plot width")
    parser.add_argument("--height", default=9, type=int, help="Plot height")
    parser.add_argument("--dpi", default=80, type=int, help="Resolution of plot")
    parser.add_argument("--extension", type=float, default=0.1, help="Extension as fraction of repeat signal around STR region to plot")
    parser.add_argument("--zoom", type=int, default=500, help="Region around prefix and suffix to plot")
    parser.add_argument("--log_level", default='warning', choices=LEVELS, help="Log level")
    args = parser.parse_args(argv)
    log = Log(args.log_level)
    if not os.path.isfile(args.f5Index):
        log("Main: Fast5 index file does not exist.", 'error'); raise SystemExit(1)
    from . import plotting
    import matplotlib
    if args.output:
        matplotlib.use("Agg")
        os.makedirs(args.output, exist_ok=True)
    from matplotlib.figure import Figure
    reads = Fast5Index(args.f5Index)
    made = []
    with (open(args.counts) if args.counts else sys.stdin) as stream:
        for row in plotting.parse_counts(stream):
            raw = reads.get_raw(row.read_id)
            if raw is None:
                log("Plot: No fast5 for ID %s" % row.read_id, 'warning'); continue
            if args.output:
                fig = Figure(figsize=(args.width, args.height), dpi=args.dpi, layout="constrained")
            else:
                import matplotlib.pyplot as plt
                fig = plt.figure(figsize=(args.width, args.height), dpi=args.dpi, layout="constrained")
            plotting.draw(fig, raw, row, extension=args.extension, zoom=args.zoom)
            if args.output:
                path = os.path.join(args.output, plotting.figure_name(row, args.format))
                fig.savefig(path); made.append(path)
            else:
                plt.show()
    return made


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    parser = argparse.ArgumentParser(description='STRique: a nanopore raw signal repeat detection pipeline (MI355X engine)',
                                     usage='''STRique.py <command> [<args>]
Available commands are:
   index      Index batch(es) of bulk-fast5 or tar archived single fast5
   count      Count single read repeat expansions
   plot       Plot repeat signal after counting
''')
    parser.add_argument('command', help='Subcommand to run')
    args = parser.parse_args(argv[:1])
    if args.command == 'count':
        count(argv[1:])
    elif args.command == 'index':
        index(argv[1:])
    elif args.command == 'plot':
        plot(argv[1:])
    else:
        print('Unrecognized command', file=sys.stderr)
        parser.print_help(file=sys.stderr)
        raise SystemExit(1)


if __name__ == '__main__':
    main()
