"""Minimal HDF5 writer for fast5 files (no h5py in the target image): superblock v0, old-style groups
(local heap + symbol-table nodes under a B-tree v1), contiguous int16 datasets, string / integer /
float attributes -- byte by byte from the HDF5 file-format specification.  Used by the read masker
(strique_amd/masker.py: the reference's scripts/fast5Masker.py rewrites signals through h5py) and by
the tests, which exercise the reader in strique_amd/fast5.py on layouts the one bundled file does not
have (multi-read fast5, several reads per group, tar archives).
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class _File(object):
    def __init__(self):
        self.buf = bytearray(96)          # superblock (v0: 24 + 32 + 40 bytes) written last

    def alloc(self, data, align=8):
        while len(self.buf) % align:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr


def _msg(mtype, payload, flags=0):
    payload = bytes(payload)
    payload += b"\x00" * (-len(payload) % 8)
    return struct.pack("<HHB3x", mtype, len(payload), flags) + payload


def _object_header(f, msgs):
    body = b"".join(msgs)
    hdr = struct.pack("<BxHII4x", 1, len(msgs), 1, len(body))      # version, #messages, refcount, header size
    return f.alloc(hdr + body)


def _string_attr(name, value):
    name_b = name.encode() + b"\x00"
    val = value.encode()
    dt = struct.pack("<BBBBI", 0x13, 0, 0, 0, len(val))             # version 1, class 3 (string), null-terminated ascii
    ds = struct.pack("<BBBx4x", 1, 0, 0)                              # dataspace version 1, scalar
    pad = lambda x: x + b"\x00" * (-len(x) % 8)
    return _msg(0x0C, struct.pack("<BxHHH", 1, len(name_b), len(dt), len(ds)) + pad(name_b) + pad(dt) + pad(ds) + val)


def _attr(name, value):
    """String, integer (int64 / uint64) or float (float64) scalar attribute."""
    if isinstance(value, str):
        return _string_attr(name, value if value else "\x00")
    name_b = name.encode() + b"\x00"
    ds = struct.pack("<BBBx4x", 1, 0, 0)
    pad = lambda x: x + b"\x00" * (-len(x) % 8)
    if isinstance(value, float):
        # IEEE double: class 1, little endian, sign bit 63, exponent 52/11, mantissa 0/52, bias 1023
        dt = struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        data = struct.pack("<d", value)
    else:
        value = int(value)
        signed = value < 0
        dt = struct.pack("<BBBBIHH", 0x10, 0x08 if signed else 0x00, 0, 0, 8, 0, 64)
        data = struct.pack("<q" if signed else "<Q", value)
    return _msg(0x0C, struct.pack("<BxHHH", 1, len(name_b), len(dt), len(ds)) + pad(name_b) + pad(dt) + pad(ds) + data)


def _vlen_string_attr(f, name, value):
    """The form h5py / ont_fast5_api give `read_id`: a variable-length string whose bytes live in a
    global heap collection (HDF5 spec III.E); the attribute holds (length, collection address, index)."""
    val = value.encode()
    obj = struct.pack("<HHIQ", 1, 1, 0, len(val)) + val + b"\x00" * (-len(val) % 8)
    free = struct.pack("<HHIQ", 0, 0, 0, 0)
    size = 16 + len(obj) + len(free)
    gcol = f.alloc(b"GCOL" + struct.pack("<B3xQ", 1, size) + obj + free)
    name_b = name.encode() + b"\x00"
    # datatype: version 1, class 9 (variable length), type = string (bits 0-3 of the class bit field), base type char
    base = struct.pack("<BBBBI", 0x13, 0, 0, 0, 1)
    dt = struct.pack("<BBBBI", 0x19, 0x01, 0, 0, 16) + base
    ds = struct.pack("<BBBx4x", 1, 0, 0)
    pad = lambda x: x + b"\x00" * (-len(x) % 8)
    data = struct.pack("<IQI", len(val), gcol, 1)
    return _msg(0x0C, struct.pack("<BxHHH", 1, len(name_b), len(dt), len(ds)) + pad(name_b) + pad(dt) + pad(ds) + data)


def _chunked_dataset(f, array, chunk, filters, encode, attrs=()):
    """1-D int16 dataset in chunks of `chunk` samples behind a filter pipeline: filters = [(id, name, cd_values)],
    encode(int16 chunk) -> stored bytes.  One leaf node of a version-1 B-tree indexes the chunks."""
    a = array.astype("<i2")
    n = len(a)
    nchunks = max(1, (n + chunk - 1) // chunk)
    stored = []
    for c in range(nchunks):
        part = a[c * chunk:(c + 1) * chunk]
        if len(part) < chunk:
            part = np.concatenate([part, np.zeros(chunk - len(part), "<i2")])       # edge chunks are stored whole
        data = encode(part)
        stored.append((f.alloc(data), len(data)))
    node = b"TREE" + struct.pack("<BBHQQ", 1, 0, nchunks, UNDEF, UNDEF)
    for c, (addr, size) in enumerate(stored):
        node += struct.pack("<IIQQ", size, 0, c * chunk, 0) + struct.pack("<Q", addr)
    node += struct.pack("<IIQQ", 0, 0, nchunks * chunk, 0)
    tree = f.alloc(node)
    space = struct.pack("<BBBx4xQ", 1, 1, 0, n)
    dtype = struct.pack("<BBBBIHH", 0x10, 0x08, 0, 0, 2, 0, 16)
    layout = struct.pack("<BBBQII", 3, 2, 2, tree, chunk, 2)           # version 3, chunked, rank + 1, B-tree, chunk dims, element size
    pipe = struct.pack("<BB6x", 1, len(filters))
    for fid, name, cd in filters:
        nm = name.encode() + b"\x00"
        nm += b"\x00" * (-len(nm) % 8)
        pipe += struct.pack("<HHHH", fid, len(nm), 1, len(cd)) + nm + struct.pack("<%dI" % len(cd), *cd)
        if len(cd) % 2:
            pipe += b"\x00" * 4
    return _object_header(f, [_msg(0x01, space), _msg(0x03, dtype), _msg(0x08, layout), _msg(0x0B, pipe)] + [_attr(k, v) for k, v in attrs])


def _vbz_dataset(f, array, attrs=(), version=0, level=1, chunk=8192):
    from . import vbz
    cd = vbz.encode(np.zeros(1, "<i2"), version=version, level=level)[1]
    return _chunked_dataset(f, array, chunk, [(vbz.FILTER_ID, "vbz", cd)], lambda part: vbz.encode(part, version=version, level=level)[0], attrs)


def _deflate_dataset(f, array, attrs=(), chunk=8192):
    import zlib
    return _chunked_dataset(f, array, chunk, [(1, "deflate", (4,))], lambda part: zlib.compress(part.tobytes(), 4), attrs)


def _dataset(f, array, attrs=()):
    data = array.astype("<i2").tobytes()
    addr = f.alloc(data)
    space = struct.pack("<BBBx4xQ", 1, 1, 0, len(array))            # version 1, rank 1, no max dims
    dtype = struct.pack("<BBBBIHH", 0x10, 0x08, 0, 0, 2, 0, 16)      # version 1 class 0 (fixed point), signed, 2 bytes
    layout = struct.pack("<BBQQ", 3, 1, addr, len(data))             # version 3, contiguous
    return _object_header(f, [_msg(0x01, space), _msg(0x03, dtype), _msg(0x08, layout)] + [_attr(k, v) for k, v in attrs])


def _group(f, entries, attrs=()):
    """entries: {name: object header address}.  Symbol-table nodes of up to 2 * leaf_k = 128 names under one
    B-tree v1 node (up to 2 * internal_k = 64 nodes: 8192 names per group)."""
    names = sorted(entries)
    heap_data = bytearray(8)                                          # offset 0: the empty string
    offs = []
    for n in names:
        offs.append(len(heap_data))
        heap_data += n.encode() + b"\x00"
        heap_data += b"\x00" * (-len(heap_data) % 8)
    data_addr = f.alloc(bytes(heap_data))
    heap = f.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), UNDEF, data_addr))
    per = 128
    if len(names) > per * 64:
        raise ValueError("more than 8192 entries in one group")
    children = []                                                     # (node address, heap offset of its last name)
    for c0 in range(0, max(len(names), 1), per):
        part = list(zip(names[c0:c0 + per], offs[c0:c0 + per]))
        snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(part)))
        for n, o in part:
            snod += struct.pack("<QQI4x16x", o, entries[n], 0)
        snod += b"\x00" * (40 * (per - len(part)))                    # nodes are allocated at full size
        children.append((f.alloc(bytes(snod)), part[-1][1] if part else 0))
    # B-tree v1, node type 0 (group), level 0: key0 | child0 | key1 | child1 | ... | keyN
    tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(children), UNDEF, UNDEF) + struct.pack("<Q", 0))
    for addr, last in children:
        tree += struct.pack("<QQ", addr, last)
    tree_addr = f.alloc(bytes(tree))
    return _object_header(f, [_msg(0x11, struct.pack("<QQ", tree_addr, heap))] + [_attr(k, v) for k, v in attrs])


def _finish(f, root, base=0):
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, 64, 32, 0)
    sb += struct.pack("<QQQQ", base, UNDEF, len(f.buf), UNDEF)
    sb += struct.pack("<QQI4x16x", 0, root, 0)                        # root group symbol table entry
    f.buf[:len(sb)] = sb
    return bytes(f.buf)


def single_read_fast5(read_id, signal, read_number=7, vlen_id=False, user_block=0):
    """/Raw/Reads/Read_<n>/Signal with the read_id attribute on the Read group.
    vlen_id: read_id as a variable-length string (global heap); user_block: bytes in front of the
    superblock (every address in the file is then relative to that base address)."""
    f = _File()
    sig = _dataset(f, signal)
    if vlen_id:
        rd = _group(f, {"Signal": sig})
        # append the attribute message to the group's object header: rebuild it with the extra message
        f2 = _File(); sig = _dataset(f2, signal)
        attr = _vlen_string_attr(f2, "read_id", read_id)
        rd = _group_with_messages(f2, {"Signal": sig}, [attr])
        f = f2
    else:
        rd = _group(f, {"Signal": sig}, attrs=[("read_id", read_id)])
    reads = _group(f, {"Read_%d" % read_number: rd})
    raw = _group(f, {"Reads": reads})
    blob = _finish(f, _group(f, {"Raw": raw}), base=user_block)
    return b"\x00" * user_block + blob


def _group_with_messages(f, entries, extra_msgs):
    """_group() with ready-made extra header messages."""
    addr = _group(f, entries)
    # the object header just written is the last allocation: re-emit it with the extra messages
    nmsg, = struct.unpack_from("<H", f.buf, addr + 2)
    hsize, = struct.unpack_from("<I", f.buf, addr + 8)
    body = bytes(f.buf[addr + 16:addr + 16 + hsize]) + b"".join(extra_msgs)
    hdr = struct.pack("<BxHII4x", 1, nmsg + len(extra_msgs), 1, len(body))
    return f.alloc(hdr + body)


def multi_read_fast5(reads, compression=None, vbz_version=0):
    """reads: [(read_id, signal)] -> /read_<id>/Raw/Signal, read_id attribute on every Raw group.
    compression: None (contiguous), "gzip" or "vbz" (chunked, as MinKNOW writes bulk files)."""
    f = _File()
    top = {}
    for rid, signal in reads:
        if compression == "vbz":
            sig = _vbz_dataset(f, signal, version=vbz_version)
        elif compression == "gzip":
            sig = _deflate_dataset(f, signal)
        else:
            sig = _dataset(f, signal)
        raw = _group(f, {"Signal": sig}, attrs=[("read_id", rid)])
        top["read_" + rid] = _group(f, {"Raw": raw})
    return _finish(f, _group(f, top))


def write_tree(tree):
    """tree: nested dict {"attrs": {name: str | int | float}, "groups": {name: tree}, "datasets": {name: (int16 array, attrs)}}
    -> bytes of an HDF5 file whose root group is `tree`."""
    f = _File()

    def emit(node):
        entries = {}
        for name, (arr, attrs) in node.get("datasets", {}).items():
            entries[name] = _dataset(f, arr, sorted(attrs.items()))
        for name, sub in node.get("groups", {}).items():
            entries[name] = emit(sub)
        return _group(f, entries, sorted(node.get("attrs", {}).items()))
    return _finish(f, emit(tree))
