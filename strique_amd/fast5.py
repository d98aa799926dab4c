"""Minimal HDF5 reader for nanopore fast5 raw signals (no h5py in the target image).

Covers what STRique's fast5Index.get_raw needs for single- and multi-read fast5 files
(reference STRique_lib/fast5Index.py:76-84,220-233): old-style groups (symbol tables, B-tree v1,
local heaps), version-1 object headers with continuation blocks, chunked int16 datasets with the
deflate (and optional shuffle) or VBZ (strique_amd/vbz.py) filter, contiguous datasets, and the string / integer attributes that
carry `read_id` (fixed-length and variable-length strings, the latter through the global heap), files with a
user block (non-zero base address).  HDF5 features outside that subset raise NotImplementedError.
"""
import mmap
import os
import threading
import struct
import zlib

import numpy as np

from . import vbz

_SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File(object):
    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            self.buf = bytes(path_or_bytes)
        else:
            # bulk fast5 files hold thousands of reads in hundreds of MB: map, do not read
            with open(path_or_bytes, "rb") as fp:
                try:
                    self.buf = mmap.mmap(fp.fileno(), 0, access=mmap.ACCESS_READ)
                except (ValueError, OSError):          # empty file, or a file system without mmap
                    self.buf = fp.read()
        b = self.buf
        sb = 0                                    # the superblock sits at 0 or, behind a user block, at 512, 1024, 2048, ...
        while bytes(b[sb:sb + 8]) != _SIG:
            sb = 512 if sb == 0 else sb * 2
            if sb + 8 > len(b):
                raise ValueError("not an HDF5 file")
        if sb:
            self.buf = b = memoryview(b)[sb:]     # addresses are relative to the base address = start of the superblock here
        ver = b[8]
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d" % ver)
        self.so, self.sl = b[13], b[14]          # size of offsets / lengths
        if self.so != 8 or self.sl != 8:
            raise NotImplementedError("only 8-byte offsets/lengths")
        self.leaf_k, self.int_k = struct.unpack_from("<HH", b, 16)
        p = 24 if ver == 0 else 28
        self.base, = struct.unpack_from("<Q", b, p)
        p += 32                                   # base, free-space, eof, driver-info addresses
        if self.base not in (0, UNDEF) and self.base != sb:
            raise NotImplementedError("HDF5 base address %d differs from the superblock offset %d" % (self.base, sb))
        # root group symbol table entry
        self.root = self._symbol_entry(p)

    # ---- low level -----------------------------------------------------------------------
    def _symbol_entry(self, p):
        name_off, ohdr, cache_type = struct.unpack_from("<QQI", self.buf, p)
        scratch = bytes(self.buf[p + 24:p + 40])
        return {"name_off": name_off, "ohdr": ohdr, "cache": cache_type, "scratch": scratch}

    def _messages(self, addr):
        """Yield (type, flags, payload bytes) of a version-1 object header incl. continuations."""
        b = self.buf
        ver = b[addr]
        if ver != 1:
            raise NotImplementedError("object header version %d" % ver)
        nmsg, = struct.unpack_from("<H", b, addr + 2)
        hsize, = struct.unpack_from("<I", b, addr + 8)
        blocks = [(addr + 16, hsize)]
        seen = 0
        while blocks and seen < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and seen < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, p)
                payload = bytes(b[p + 8:p + 8 + msize])
                p += 8 + msize
                seen += 1
                if mtype == 0x10:                    # continuation
                    caddr, clen = struct.unpack_from("<QQ", payload, 0)
                    blocks.append((caddr, clen))
                else:
                    yield mtype, mflags, payload

    def _heap_string(self, heap_addr, off):
        b = self.buf
        if bytes(b[heap_addr:heap_addr + 4]) != b"HEAP":
            raise ValueError("bad local heap")
        data_addr, = struct.unpack_from("<Q", b, heap_addr + 24)
        p = data_addr + off
        if hasattr(b, "find"):
            e = b.find(b"\x00", p, p + 1024)           # mmap and bytes search in place: no 1 KB copy per name
        else:                                          # a memoryview (file with a user block)
            e = bytes(b[p:p + 1024]).find(b"\x00")
            e = e + p if e >= 0 else e
        if e < 0:
            raise ValueError("unterminated heap string")
        return bytes(b[p:e]).decode()

    def _group_entries(self, ohdr):
        """name -> object header address of a group (memoised: a bulk file's root group holds thousands of reads)."""
        cache = self.__dict__.setdefault("_gcache", {})
        if ohdr not in cache:
            cache[ohdr] = self._group_entries_uncached(ohdr)
        return cache[ohdr]

    def _group_entries_uncached(self, ohdr):
        btree = heap = None
        out = {}
        dense = False
        for mtype, _, pl in self._messages(ohdr):
            if mtype == 0x11:
                btree, heap = struct.unpack_from("<QQ", pl, 0)
            elif mtype == 0x06:                         # new-style group, link stored in the header
                flags = pl[1]
                p = 2
                ltype = 0
                if flags & 0x08:
                    ltype = pl[p]; p += 1
                if flags & 0x04:
                    p += 8
                if flags & 0x10:
                    p += 1
                lsz = 1 << (flags & 0x03)
                nlen = int.from_bytes(pl[p:p + lsz], "little"); p += lsz
                name = pl[p:p + nlen].decode(); p += nlen
                if ltype == 0:
                    out[name], = struct.unpack_from("<Q", pl, p)
            elif mtype == 0x02:                         # link info: dense storage lives in a fractal heap
                flags = pl[1]
                p = 2 + (8 if flags & 0x01 else 0)
                fheap, = struct.unpack_from("<Q", pl, p)
                dense = fheap != UNDEF
        if btree is not None:
            self._walk_group_btree(btree, heap, out)
        elif dense and not out:
            raise NotImplementedError("new-style group with dense link storage")
        return out

    def _walk_group_btree(self, addr, heap, out):
        b = self.buf
        if bytes(b[addr:addr + 4]) == b"SNOD":
            n, = struct.unpack_from("<H", b, addr + 6)
            p = addr + 8
            for _ in range(n):
                name_off, ohdr = struct.unpack_from("<QQ", b, p)          # symbol table entry: link name offset, object header address
                out[self._heap_string(heap, name_off)] = ohdr
                p += 40
            return
        if bytes(b[addr:addr + 4]) != b"TREE":
            raise ValueError("bad group B-tree node")
        level = b[addr + 5]
        n, = struct.unpack_from("<H", b, addr + 6)
        p = addr + 24                                   # skip siblings
        for i in range(n):
            p += 8                                      # key i
            child, = struct.unpack_from("<Q", b, p); p += 8
            self._walk_group_btree(child, heap, out)

    # ---- public ---------------------------------------------------------------------------
    def listdir(self, path="/"):
        return sorted(self._group_entries(self._resolve(path)).keys())

    def _resolve(self, path):
        addr = self.root["ohdr"]
        for part in [x for x in path.split("/") if x]:
            ents = self._group_entries(addr)
            if part not in ents:
                raise KeyError(path)
            addr = ents[part]
        return addr

    def attrs(self, path):
        out = {}
        for mtype, _, pl in self._messages(self._resolve(path)):
            if mtype != 0x0C:
                continue
            ver = pl[0]
            if ver not in (1, 2, 3):
                continue
            nsz, tsz, ssz = struct.unpack_from("<HHH", pl, 2)
            p = 8 if ver == 1 else (8 if ver == 2 else 9)
            pad = (lambda x: (x + 7) & ~7) if ver == 1 else (lambda x: x)
            name = pl[p:p + nsz].split(b"\x00")[0].decode(); p += pad(nsz)
            dt = pl[p:p + tsz]; p += pad(tsz)
            p += pad(ssz)
            cls = dt[0] & 0x0F
            size, = struct.unpack_from("<I", dt, 4)
            raw = pl[p:p + size]
            raw = bytes(raw)
            if cls == 3:
                out[name] = raw.split(b"\x00")[0].decode(errors="replace")
            elif cls == 9 and (dt[1] & 0x0F) == 1 and len(raw) >= 16:
                # variable-length string (what h5py / ont_fast5_api write for read_id): length, global heap
                # collection address, object index
                _ln, gaddr, gidx = struct.unpack_from("<IQI", raw, 0)
                val = self._global_heap_object(gaddr, gidx)
                if val is not None:
                    out[name] = val.split(b"\x00")[0].decode(errors="replace")
            elif cls == 0 and size in (1, 2, 4, 8):
                signed = bool(dt[1] & 0x08)
                out[name] = int.from_bytes(raw, "little", signed=signed)
            elif cls == 1 and size in (4, 8):
                out[name] = struct.unpack("<f" if size == 4 else "<d", raw)[0]
        return out

    def _global_heap_object(self, addr, index):
        """Object `index` of the global heap collection at `addr` (HDF5 spec III.E), or None."""
        b = self.buf
        if addr in (0, UNDEF) or bytes(b[addr:addr + 4]) != b"GCOL":
            return None
        size, = struct.unpack_from("<Q", b, addr + 8)
        p, end = addr + 16, addr + size
        while p + 16 <= end:
            idx, _refs, _res, osize = struct.unpack_from("<HHIQ", b, p)
            if idx == 0:
                break                                   # free space object: end of the used part
            if idx == index:
                return bytes(b[p + 16:p + 16 + osize])
            p += 16 + ((osize + 7) & ~7)
        return None

    def dataset(self, path, alloc=None, defer=False):
        """The values of a dataset.  `alloc(n, dtype)` (optional) supplies the ZERO-FILLED output array of a chunked dataset
        instead of numpy.  `defer`: a deflate-compressed dataset comes back as an InflatePlan -- its output
        array allocated, its chunks located, nothing inflated yet -- for inflate_plans() to fill together with others."""
        fast = self._fast_dataset(path, alloc, defer)          # the common layout, resolved by the library in one call
        if fast is not None:
            return fast
        b = self.buf
        shape = None; dtype = None; layout = None; filters = []
        for mtype, _, pl in self._messages(self._resolve(path)):
            if mtype == 0x01:
                ver, rank, flags = pl[0], pl[1], pl[2]
                p = 8 if ver == 1 else 4
                shape = struct.unpack_from("<%dQ" % rank, pl, p)
            elif mtype == 0x03:
                cls = pl[0] & 0x0F
                size, = struct.unpack_from("<I", pl, 4)
                if cls == 0:
                    dtype = np.dtype("<%s%d" % ("i" if pl[1] & 0x08 else "u", size))
                elif cls == 1:
                    dtype = np.dtype("<f%d" % size)
                else:
                    raise NotImplementedError("datatype class %d" % cls)
            elif mtype == 0x08:
                ver = pl[0]
                if ver != 3:
                    raise NotImplementedError("data layout version %d" % ver)
                lclass = pl[1]
                if lclass == 1:
                    addr, size = struct.unpack_from("<QQ", pl, 2)
                    layout = ("contiguous", addr, size)
                elif lclass == 2:
                    rank = pl[2]
                    addr, = struct.unpack_from("<Q", pl, 3)
                    dims = struct.unpack_from("<%dI" % rank, pl, 11)
                    layout = ("chunked", addr, dims)
                else:
                    raise NotImplementedError("compact layout")
            elif mtype == 0x0B:
                ver, nf = pl[0], pl[1]
                p = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid, nlen, fflags, ncd = struct.unpack_from("<HHHH", pl, p); p += 8
                    if ver == 1 or fid >= 256:
                        p += (nlen + 7) & ~7 if ver == 1 else nlen
                    cd = struct.unpack_from("<%dI" % ncd, pl, p)
                    p += 4 * ncd
                    if ver == 1 and ncd % 2:
                        p += 4
                    filters.append((fid, cd))
        if shape is None or dtype is None or layout is None:
            raise ValueError("not a dataset: %s" % path)
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == "contiguous":
            return np.frombuffer(b, dtype, n, layout[1]).reshape(shape)      # a read-only view of the mapped file (it keeps the mapping alive)
        if len(shape) != 1:
            raise NotImplementedError("only 1-D chunked datasets")
        csize = layout[2][0]
        fids = [fid for fid, _ in filters]
        out = alloc(n, dtype) if alloc is not None else np.zeros(n, dtype)          # chunks that were never written read as the fill value
        if fids and set(fids) <= {1, 2} and fids.count(1) == 1 and (2 not in fids or fids.index(2) < fids.index(1)):
            # deflate (after an optional shuffle): all chunks of the dataset in one native call, outside the interpreter lock
            if defer and _inflate_many_fn() is not None:
                return InflatePlan(self, self._chunk_arrays(layout[1], len(shape)), out, csize, 2 in fids)
            if self._native_inflate(layout[1], len(shape), out, csize, 2 in fids):
                return out
        for off, data in self._chunks(layout[1], len(shape)):
            for fid, cd in reversed(filters):
                if fid == 1:
                    data = zlib.decompress(data)
                elif fid == 2:
                    a = np.frombuffer(data, np.uint8).reshape(dtype.itemsize, -1)
                    data = a.T.tobytes()
                elif fid == vbz.FILTER_ID:
                    data = vbz.decode(data, cd)
                else:
                    raise NotImplementedError("HDF5 filter %d" % fid)
            vals = np.frombuffer(data, dtype)
            k = min(len(vals), csize, n - off)
            out[off:off + k] = vals[:k]
        return out

    def _chunk_table(self, addr, rank, rows):
        """Leaf entries of the chunk B-tree as (element offset, file address, stored size) rows, without touching the chunks."""
        b = self.buf
        if addr == UNDEF:
            return
        if bytes(b[addr:addr + 4]) != b"TREE":
            raise ValueError("bad chunk B-tree node")
        level = b[addr + 5]
        n, = struct.unpack_from("<H", b, addr + 6)
        ent = _CHUNK_ENTRY.get(rank)
        if ent is None:
            keysz = 8 + 8 * (rank + 1)
            ent = _CHUNK_ENTRY[rank] = np.dtype({"names": ["csize", "fmask", "off0", "child"], "formats": ["<u4", "<u4", "<u8", "<u8"],
                                                 "offsets": [0, 4, 8, keysz], "itemsize": keysz + 8})
        if addr + 24 + n * ent.itemsize > len(b):
            raise ValueError("chunk B-tree node runs past the end of the file")
        tab = np.frombuffer(b, ent, n, addr + 24)
        if level == 0:
            if tab["fmask"].any():
                raise NotImplementedError("chunk with skipped filters")
            rows.append(tab)
        else:
            for child in tab["child"]:
                self._chunk_table(int(child), rank, rows)

    def _fast_dataset(self, path, alloc, defer):
        """strq_h5_locate (csrc/h5_locate.hip): version-1 headers, old-style groups, a 1-D dataset that is contiguous or chunked
        behind deflate -- what bulk fast5 files look like.  None for everything else (and without the library): the Python code
        below then does the work, and reports what is wrong with a file."""
        fn = _locate_fn()
        if fn is None or os.environ.get("STRQ_H5_PYTHON"):
            return None
        comps = [x for x in path.split("/") if x]
        if not comps:
            return None
        start = self.root["ohdr"]
        if len(comps) > 1:
            try:
                start = self._group_entries(start).get(comps[0])          # memoised: a bulk file's root group holds thousands of reads
            except (NotImplementedError, ValueError, struct.error):
                return None
            if start is None:
                return None
            comps = comps[1:]
        self._base()
        tl = _TLS
        if getattr(tl, "cap", 0) == 0:
            tl.cap = 256
            tl.meta = np.zeros(16, np.int64); tl.ca = np.zeros(tl.cap, np.int64); tl.cs = np.zeros(tl.cap, np.int32); tl.co = np.zeros(tl.cap, np.int64)
        rel = "/".join(comps).encode()
        while True:
            rows = fn(self._base_ptr, self._base_u8.size, start, rel, tl.meta.ctypes.data, tl.ca.ctypes.data, tl.cs.ctypes.data, tl.co.ctypes.data, tl.cap)
            if rows != -101:
                break
            tl.cap *= 4
            tl.ca = np.zeros(tl.cap, np.int64); tl.cs = np.zeros(tl.cap, np.int32); tl.co = np.zeros(tl.cap, np.int64)
        if rows < 0:
            return None
        n, esize, kind, layout, addr, chunk_elems, filters = (int(v) for v in tl.meta[:7])
        dtype = _DTYPES.get((kind, esize))
        if dtype is None:
            return None
        if layout == 1:
            if n == 0:
                return np.zeros(0, dtype)                                # an empty dataset has no storage (undefined address)
            return np.frombuffer(self.buf, dtype, n, addr)               # a read-only view of the mapped file
        # chunks that were never written read as the fill value: zero-fill unless the chunks tile the dataset (then the inflate writes
        # every element, and zeroing 750 KB per read is half of what is left of this call's time under the interpreter lock)
        out = alloc(n, dtype) if alloc is not None else (np.empty(n, dtype) if tl.meta[7] else np.zeros(n, dtype))
        chunks = (tl.ca[:rows].copy(), tl.cs[:rows].copy(), tl.co[:rows].copy()) if rows else None
        if filters & 4:
            # VBZ (what MinKNOW writes): zstd + variable-byte decode of all chunks in one native call (strq_vbz_chunks); a layout it
            # does not decode (-1) goes to the Python decoder below, which also says what is wrong with a chunk
            vfn = _vbz_fn()
            if vfn is None:
                return None
            if chunks is not None:
                version, isize, zigzag, level = (int(v) for v in tl.meta[8:12])
                rc = vfn(self._base_ptr, self._base_u8.size, rows, chunks[0].ctypes.data, chunks[1].ctypes.data, chunks[2].ctypes.data,
                         out.dtype.itemsize, version, isize, zigzag, level, chunk_elems, out.size, out.ctypes.data)
                if rc != 0:
                    return None
            return out
        if defer and _inflate_many_fn() is not None:
            return InflatePlan(self, chunks, out, chunk_elems, bool(filters & 2))
        inflate = _inflate_fn()
        if inflate is None:
            return None
        if chunks is not None:
            rc = inflate(self._base_ptr, self._base_u8.size, rows, chunks[0].ctypes.data, chunks[1].ctypes.data, chunks[2].ctypes.data,
                         out.dtype.itemsize, 1 if filters & 2 else 0, chunk_elems, out.size, out.ctypes.data)
            if rc != 0:
                raise ValueError("chunk %d of a deflate-compressed dataset is damaged" % (-rc - 2) if rc < -1 else "bad chunk table")
        return out

    def _chunk_arrays(self, addr, rank):
        """(file addresses, stored sizes, first elements) of the chunks of a dataset, or None when it has none."""
        rows = []
        self._chunk_table(addr, rank, rows)
        if not rows:
            return None
        tab = np.concatenate(rows) if len(rows) > 1 else rows[0]
        return (np.ascontiguousarray(tab["child"], np.int64), np.ascontiguousarray(tab["csize"], np.int32), np.ascontiguousarray(tab["off0"], np.int64))

    def _base(self):
        base = self.__dict__.get("_base_u8")
        if base is None:
            base = self._base_u8 = np.frombuffer(self.buf, np.uint8)
            self._base_ptr = base.ctypes.data
        return base

    def _native_inflate(self, addr, rank, out, chunk_elems, shuffle):
        """strq_inflate_chunks (csrc/h5_chunks.hip).  False when the library is not built: the Python loop takes over."""
        fn = _inflate_fn()
        if fn is None:
            return False
        arrs = self._chunk_arrays(addr, rank)
        if arrs is None:
            return True
        caddr, csz, eoff = arrs
        base = self._base()
        rc = fn(self._base_ptr, base.size, len(caddr), caddr.ctypes.data, csz.ctypes.data, eoff.ctypes.data,
                out.dtype.itemsize, 1 if shuffle else 0, chunk_elems, out.size, out.ctypes.data)
        if rc != 0:
            raise ValueError("chunk %d of a deflate-compressed dataset is damaged" % (-rc - 2) if rc < -1 else "bad chunk table")
        return True

    def _chunks(self, addr, rank):
        b = self.buf
        if addr == UNDEF:
            return
        if bytes(b[addr:addr + 4]) != b"TREE":
            raise ValueError("bad chunk B-tree node")
        level = b[addr + 5]
        n, = struct.unpack_from("<H", b, addr + 6)
        p = addr + 24
        keysz = 8 + 8 * (rank + 1)
        for i in range(n):
            csize, fmask = struct.unpack_from("<II", b, p)
            offs = struct.unpack_from("<%dQ" % (rank + 1), b, p + 8)
            child, = struct.unpack_from("<Q", b, p + keysz)
            p += keysz + 8
            if level == 0:
                if fmask:
                    raise NotImplementedError("chunk with skipped filters")
                yield offs[0], bytes(b[child:child + csize])
            else:
                for x in self._chunks(child, rank):
                    yield x


class InflatePlan(object):
    """A deflate-compressed dataset whose chunks are located and whose output array exists, but which is not inflated yet."""
    __slots__ = ("file", "chunks", "out", "chunk_elems", "shuffle")

    def __init__(self, file, chunks, out, chunk_elems, shuffle):
        self.file = file; self.chunks = chunks; self.out = out; self.chunk_elems = chunk_elems; self.shuffle = shuffle


def inflate_plans(plans):
    """Fill the output arrays of `plans` with ONE native call (strq_inflate_many): the interpreter lock is released once for
    the whole list.  Returns a list with None for every plan that inflated and the error text for every one that did not."""
    import ctypes
    fn = _inflate_many_fn()
    n = len(plans)
    if n == 0:
        return []
    first = np.zeros(n + 1, np.int64)
    for i, p in enumerate(plans):
        first[i + 1] = first[i] + (len(p.chunks[0]) if p.chunks is not None else 0)
    cat = lambda j, dt: (np.concatenate([p.chunks[j] for p in plans if p.chunks is not None]) if first[-1] else np.zeros(0, dt))
    addr = cat(0, np.int64); csz = cat(1, np.int32); eoff = cat(2, np.int64)
    bases = (ctypes.c_void_p * n)(); outs = (ctypes.c_void_p * n)()
    blen = np.zeros(n, np.int64); esz = np.zeros(n, np.int32); shuf = np.zeros(n, np.int32); cel = np.zeros(n, np.int64); ntot = np.zeros(n, np.int64)
    for i, p in enumerate(plans):
        b = p.file._base()
        bases[i] = p.file._base_ptr; blen[i] = b.size; outs[i] = p.out.ctypes.data
        esz[i] = p.out.dtype.itemsize; shuf[i] = 1 if p.shuffle else 0; cel[i] = p.chunk_elems; ntot[i] = p.out.size
    status = np.zeros(n, np.int64)
    rc = fn(n, bases, blen.ctypes.data, first.ctypes.data, addr.ctypes.data, csz.ctypes.data, eoff.ctypes.data, esz.ctypes.data,
            shuf.ctypes.data, cel.ctypes.data, ntot.ctypes.data, outs, status.ctypes.data)
    if rc < 0:
        raise ValueError("strq_inflate_many: bad argument")
    return [None if s == 0 else ("chunk %d of a deflate-compressed dataset is damaged" % (-s - 2) if s < -1 else "bad chunk table") for s in status]


_CHUNK_ENTRY = {}          # rank -> numpy dtype of a chunk B-tree leaf entry
_INFLATE = []


def _inflate_fn():
    """strq_inflate_chunks with its argument types set once (plain ints go in, no ctypes objects per call), or None."""
    if not _INFLATE:
        import ctypes
        try:
            from . import ffi
            fn = ffi.load_library().strq_inflate_chunks
            fn.restype = ctypes.c_int64
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
            _INFLATE.append(fn)
        except (ImportError, OSError, AttributeError):
            _INFLATE.append(None)
    return _INFLATE[0]


_TLS = threading.local()          # per reader thread: the scratch arrays strq_h5_locate fills
_DTYPES = {(0, 1): np.dtype("<u1"), (0, 2): np.dtype("<u2"), (0, 4): np.dtype("<u4"), (0, 8): np.dtype("<u8"),
           (1, 1): np.dtype("<i1"), (1, 2): np.dtype("<i2"), (1, 4): np.dtype("<i4"), (1, 8): np.dtype("<i8"),
           (2, 4): np.dtype("<f4"), (2, 8): np.dtype("<f8")}
_LOCATE = []


def _locate_fn():
    if not _LOCATE:
        import ctypes
        try:
            from . import ffi
            fn = ffi.load_library().strq_h5_locate
            fn.restype = ctypes.c_int64
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
            _LOCATE.append(fn)
        except (ImportError, OSError, AttributeError):
            _LOCATE.append(None)
    return _LOCATE[0]


_VBZ = []


def _vbz_fn():
    if not _VBZ:
        import ctypes
        try:
            from . import ffi
            fn = ffi.load_library().strq_vbz_chunks
            fn.restype = ctypes.c_int64
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                           ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
            _VBZ.append(fn)
        except (ImportError, OSError, AttributeError):
            _VBZ.append(None)
    return _VBZ[0]


_INFLATE_MANY = []


def _inflate_many_fn():
    if not _INFLATE_MANY:
        import ctypes
        try:
            from . import ffi
            fn = ffi.load_library().strq_inflate_many
            fn.restype = ctypes.c_int64
            fn.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            _INFLATE_MANY.append(fn)
        except (ImportError, OSError, AttributeError):
            _INFLATE_MANY.append(None)
    return _INFLATE_MANY[0]


def read_raw(path):
    """[(read_id, int16 signal)] of a single-read (`/Raw/Reads/Read_*`) or multi-read
    (`/read_*/Raw`) fast5 file."""
    f = H5File(path)
    out = []
    top = f.listdir("/")
    if "Raw" in top:
        for rd in f.listdir("/Raw/Reads"):
            g = "/Raw/Reads/" + rd
            out.append((f.attrs(g).get("read_id", rd), f.dataset(g + "/Signal")))
    else:
        for rd in top:
            if rd.startswith("read_"):
                g = "/%s/Raw" % rd
                out.append((f.attrs(g).get("read_id", rd[5:]), f.dataset(g + "/Signal")))
    return out
