// Host-side helper of the fast5 reader (strique_amd/fast5.py): resolve a dataset below a group of an HDF5 file of the common
// kind -- version-1 object headers, old-style groups (symbol-table message, version-1 B-tree, local heap), a 1-D dataset that is
// contiguous or chunked behind deflate (optionally after shuffle) -- and list its chunks, in one call on the mapped file.
// What the reference gets from h5py when it opens /read_<id>/Raw/Signal (STRique_lib/fast5Index.py:76-84,220-233).
//
// Why native: the Python reader needs ~50 us of interpreter time per read for this (three object headers, two group B-trees, two
// heap look-ups, the chunk B-tree), and with 16 and more reader threads that serial share -- not the inflate, which keeps its
// 1.5 ms per read and thread -- capped the readers at ~9.5 k reads/s (profiles/r04_reader.md).  The Python code stays the
// authority: anything this function does not recognise (new-style groups, other filters, compact layout, higher ranks, skipped
// filters ...) returns STRQ_H5_UNHANDLED and the caller takes the Python path, which also produces the error messages.
// Every address and size comes from the file and is checked against the mapping before it is used.
#include <stdint.h>
#include <string.h>
#include "../../include/strique_hip.h"

namespace {

struct View {
    const uint8_t* p; int64_t n;
    bool ok(uint64_t off, uint64_t len) const { return off <= (uint64_t)n && len <= (uint64_t)n - off; }
    uint16_t u16(uint64_t o) const { uint16_t v; memcpy(&v, p + o, 2); return v; }
    uint32_t u32(uint64_t o) const { uint32_t v; memcpy(&v, p + o, 4); return v; }
    uint64_t u64(uint64_t o) const { uint64_t v; memcpy(&v, p + o, 8); return v; }
};

const uint64_t UNDEF = 0xFFFFFFFFFFFFFFFFull;

// messages of a version-1 object header, continuation blocks included: f(type, payload offset, payload size) -> false stops
template <class F>
bool messages(const View& b, uint64_t addr, F f)
{
    if (!b.ok(addr, 16) || b.p[addr] != 1) return false;
    const int nmsg = b.u16(addr + 2);
    struct Block { uint64_t p, size; } blocks[16];
    int nb = 0, seen = 0;
    blocks[nb++] = {addr + 16, b.u32(addr + 8)};
    for (int bi = 0; bi < nb && seen < nmsg; ++bi) {
        uint64_t p = blocks[bi].p; const uint64_t size = blocks[bi].size;
        if (!b.ok(p, size)) return false;
        const uint64_t end = p + size;
        while (p + 8 <= end && seen < nmsg) {
            const int mtype = b.u16(p), msize = b.u16(p + 2);
            if (!b.ok(p + 8, (uint64_t)msize) || p + 8 + (uint64_t)msize > end) return false;
            ++seen;
            if (mtype == 0x10) {                      // continuation
                if (msize < 16 || nb >= 16) return false;
                blocks[nb++] = {b.u64(p + 8), b.u64(p + 16)};
            } else if (!f(mtype, p + 8, (uint64_t)msize)) return true;
            p += 8 + (uint64_t)msize;
        }
    }
    return true;
}

// name of a link in a local heap, compared with `want` (length wl): <0, 0, >0 like strcmp(heap name, want); -2 on a bad heap
int heap_cmp(const View& b, uint64_t heap, uint64_t off, const char* want, size_t wl)
{
    if (!b.ok(heap, 32) || memcmp(b.p + heap, "HEAP", 4) != 0) return -2;
    const uint64_t data = b.u64(heap + 24);
    if (!b.ok(data, off) || !b.ok(data + off, 1)) return -2;
    const uint8_t* s = b.p + data + off;
    const uint64_t room = (uint64_t)b.n - (data + off);
    size_t i = 0;
    for (;; ++i) {
        if (i >= room) return -2;
        const uint8_t c = s[i];
        const uint8_t w = i < wl ? (uint8_t)want[i] : 0;
        if (c != w) return c < w ? -1 : 1;
        if (c == 0) return 0;
    }
}

// object header address of link `name` in the old-style group whose object header is at `ohdr`; UNDEF = not found / not handled
uint64_t group_lookup(const View& b, uint64_t ohdr, const char* name, size_t nl)
{
    uint64_t btree = UNDEF, heap = UNDEF; bool other = false;
    if (!messages(b, ohdr, [&](int t, uint64_t p, uint64_t sz) {
            if (t == 0x11 && sz >= 16) { btree = b.u64(p); heap = b.u64(p + 8); }
            else if (t == 0x06 || t == 0x02) other = true;          // new-style links: the Python reader handles them
            return true; })) return UNDEF;
    if (btree == UNDEF || other) return UNDEF;
    uint64_t node = btree;
    for (int depth = 0; depth < 16; ++depth) {
        if (!b.ok(node, 8)) return UNDEF;
        if (memcmp(b.p + node, "SNOD", 4) == 0) {
            const int n = b.u16(node + 6);
            if (!b.ok(node + 8, (uint64_t)n * 40)) return UNDEF;
            for (int i = 0; i < n; ++i) {
                const uint64_t e = node + 8 + (uint64_t)i * 40;
                const int c = heap_cmp(b, heap, b.u64(e), name, nl);
                if (c == -2) return UNDEF;
                if (c == 0) return b.u64(e + 8);
            }
            return UNDEF;
        }
        if (memcmp(b.p + node, "TREE", 4) != 0 || b.p[node + 4] != 0) return UNDEF;
        const int n = b.u16(node + 6);
        // node: signature, type, level, entries used, left / right sibling (2 x 8), then key 0, child 0, key 1, ..., key n: child i holds
        // the names greater than key i and not greater than key i + 1 (keys are heap offsets of names)
        if (n < 1 || !b.ok(node + 24, (uint64_t)(2 * n + 1) * 8)) return UNDEF;
        uint64_t next = UNDEF;
        for (int i = 0; i < n; ++i) {
            const uint64_t key_hi = b.u64(node + 24 + (uint64_t)(2 * i + 2) * 8);
            const int c = heap_cmp(b, heap, key_hi, name, nl);
            if (c == -2) return UNDEF;
            if (c >= 0) { next = b.u64(node + 24 + (uint64_t)(2 * i + 1) * 8); break; }      // name <= key i + 1
        }
        if (next == UNDEF) return UNDEF;
        node = next;
    }
    return UNDEF;
}

// leaf entries of a version-1 chunk B-tree (type 1), in tree order
int64_t chunk_rows(const View& b, uint64_t addr, int rank, int64_t* caddr, int32_t* csize, int64_t* coff, int64_t cap, int64_t have, int depth)
{
    if (addr == UNDEF) return have;
    if (depth > 16 || !b.ok(addr, 24) || memcmp(b.p + addr, "TREE", 4) != 0 || b.p[addr + 4] != 1) return -1;
    const int level = b.p[addr + 5], n = b.u16(addr + 6);
    const uint64_t keysz = 8 + 8 * (uint64_t)(rank + 1), ent = keysz + 8;
    if (!b.ok(addr + 24, (uint64_t)n * ent)) return -1;
    for (int i = 0; i < n; ++i) {
        const uint64_t e = addr + 24 + (uint64_t)i * ent;
        const uint64_t child = b.u64(e + keysz);
        if (level == 0) {
            if (b.u32(e + 4) != 0) return -1;                     // a chunk with skipped filters
            if (have >= cap) return -2;
            caddr[have] = (int64_t)child; csize[have] = (int32_t)b.u32(e); coff[have] = (int64_t)b.u64(e + 8); ++have;
        } else {
            have = chunk_rows(b, child, rank, caddr, csize, coff, cap, have, depth + 1);
            if (have < 0) return have;
        }
    }
    return have;
}

}  // namespace

extern "C" int64_t strq_h5_locate(const uint8_t* base, int64_t base_len, int64_t group_ohdr, const char* path, int64_t meta[16],
                                  int64_t* chunk_addr, int32_t* chunk_size, int64_t* chunk_off, int64_t max_chunks)
{
    if (!base || base_len < 0 || group_ohdr < 0 || !path || !meta) return STRQ_H5_UNHANDLED;
    const View b{base, base_len};
    uint64_t ohdr = (uint64_t)group_ohdr;
    for (const char* s = path; *s;) {
        while (*s == '/') ++s;
        const char* e = s;
        while (*e && *e != '/') ++e;
        if (e == s) break;
        ohdr = group_lookup(b, ohdr, s, (size_t)(e - s));
        if (ohdr == UNDEF) return STRQ_H5_UNHANDLED;
        s = e;
    }
    // the dataset's own header: dataspace, datatype, layout, filter pipeline
    int64_t n = -1, esize = 0, tclass = -1, is_signed = 0, layout = 0, addr = -1, chunk_elems = 0, filters = 0; bool bad = false;
    int64_t vbz_cd[4] = {0, 0, 0, 0};
    int rank = 0;
    if (!messages(b, ohdr, [&](int t, uint64_t p, uint64_t sz) {
            if (t == 0x01) {
                if (sz < 8) { bad = true; return false; }
                const int ver = b.p[p]; rank = b.p[p + 1];
                const uint64_t q = p + (ver == 1 ? 8 : 4);
                if ((ver != 1 && ver != 2) || rank != 1 || q + 8 > p + sz) { bad = true; return false; }
                n = (int64_t)b.u64(q);
            } else if (t == 0x03) {
                if (sz < 8) { bad = true; return false; }
                tclass = b.p[p] & 0x0F; is_signed = (b.p[p + 1] & 0x08) ? 1 : 0; esize = b.u32(p + 4);
                if (tclass != 0 && tclass != 1) { bad = true; return false; }
            } else if (t == 0x08) {
                if (sz < 3 || b.p[p] != 3) { bad = true; return false; }
                layout = b.p[p + 1];
                if (layout == 1) { if (sz < 18) { bad = true; return false; } addr = (int64_t)b.u64(p + 2); }
                else if (layout == 2) {
                    const int lr = b.p[p + 2];
                    if (lr != 2 || sz < 11 + 4 * (uint64_t)lr) { bad = true; return false; }
                    addr = (int64_t)b.u64(p + 3); chunk_elems = b.u32(p + 11);
                } else { bad = true; return false; }
            } else if (t == 0x0B) {
                if (sz < 2) { bad = true; return false; }
                const int ver = b.p[p], nf = b.p[p + 1];
                uint64_t q = p + (ver == 1 ? 8 : 2);
                if ((ver != 1 && ver != 2) || nf > 2) { bad = true; return false; }
                int seen_shuffle = 0;
                for (int i = 0; i < nf; ++i) {
                    // filter description: id, [name length], flags, number of client values.  A version-2 message leaves the name
                    // length field (and the name) out for the library's own filters (id < 256: deflate, shuffle): a 6-byte header
                    if (q + 6 > p + sz) { bad = true; return false; }
                    const int fid = b.u16(q);
                    int nlen = 0, ncd = 0;
                    if (ver == 2 && fid < 256) { ncd = b.u16(q + 4); q += 6; }
                    else {
                        if (q + 8 > p + sz) { bad = true; return false; }
                        nlen = b.u16(q + 2); ncd = b.u16(q + 6); q += 8;
                        q += ver == 1 ? (uint64_t)((nlen + 7) & ~7) : (uint64_t)nlen;
                    }
                    const uint64_t cd_at = q;
                    q += 4 * (uint64_t)ncd;
                    if (ver == 1 && (ncd % 2)) q += 4;
                    if (q > p + sz) { bad = true; return false; }
                    if (fid == 2 && !(filters & 1)) { filters |= 2; seen_shuffle = 1; }          // shuffle, before the deflate
                    else if (fid == 1 && !(filters & 1)) filters |= 1;
                    else if (fid == 32020 && nf == 1) {                                          // VBZ, alone in the pipeline
                        filters |= 4;
                        for (int j = 0; j < 4 && j < ncd; ++j) vbz_cd[j] = b.u32(cd_at + 4 * (uint64_t)j);
                    }
                    else { bad = true; return false; }
                    (void)seen_shuffle;
                }
            }
            return true; }) || bad) return STRQ_H5_UNHANDLED;
    if (n < 0 || tclass < 0 || layout == 0 || !(esize == 1 || esize == 2 || esize == 4 || esize == 8) || (tclass == 1 && esize < 4)) return STRQ_H5_UNHANDLED;
    meta[0] = n; meta[1] = esize; meta[2] = tclass == 1 ? 2 : (is_signed ? 1 : 0); meta[3] = layout; meta[4] = addr; meta[5] = chunk_elems; meta[6] = filters; meta[7] = 0;
    for (int j = 0; j < 4; ++j) { meta[8 + j] = vbz_cd[j]; meta[12 + j] = 0; }
    if (layout == 1) {
        if (filters) return STRQ_H5_UNHANDLED;
        if (n > 0 && (addr < 0 || (uint64_t)addr == UNDEF || !b.ok((uint64_t)addr, (uint64_t)n * (uint64_t)esize))) return STRQ_H5_UNHANDLED;
        return 0;
    }
    if (!(filters & 5) || chunk_elems < 1 || !chunk_addr || !chunk_size || !chunk_off) return STRQ_H5_UNHANDLED;      // chunked without deflate / VBZ: the Python loop
    const int64_t rows = chunk_rows(b, (uint64_t)addr, 1, chunk_addr, chunk_size, chunk_off, max_chunks, 0, 0);
    if (rows == -2) return STRQ_H5_MORE_CHUNKS;
    if (rows < 0) return STRQ_H5_UNHANDLED;
    // do the chunks tile the dataset (chunk i starts at element i * chunk_elems, the last one reaches the end)?  Then every element
    // of the output is written by the inflate and the caller need not zero-fill it first
    bool tiled = rows * chunk_elems >= n;
    for (int64_t i = 0; i < rows && tiled; ++i) tiled = chunk_off[i] == i * chunk_elems;
    meta[7] = tiled ? 1 : 0;
    return rows;
}
