// Host-side helper of the fast5 reader (strique_amd/fast5.py): the chunks of one 1-D chunked HDF5 dataset behind the
// deflate (and optional shuffle) filter, inflated straight from the mapped file into the caller's array.
// Replaces what the reference gets from h5py / libhdf5 when it reads /Raw/.../Signal (STRique_lib/fast5Index.py:76-84,
// 220-233).  Pure C++ on the host (zlib); it lives in this library so that the `count` command's reader threads spend
// their time here, outside the interpreter lock, instead of in a per-chunk Python loop (46 chunks per 50 kb read).
#include <stdint.h>
#include <string.h>
#include <vector>
#include <zlib.h>
#include "../../include/strique_hip.h"

extern "C" int64_t strq_inflate_chunks(const uint8_t* base, int64_t base_len, int64_t n_chunks, const int64_t* addr,
                                       const int32_t* csize, const int64_t* elem_off, int32_t elem_size, int32_t shuffle,
                                       int64_t chunk_elems, int64_t n_total, void* out)
{
    if (!base || n_chunks < 0 || (n_chunks > 0 && (!addr || !csize || !elem_off)) || elem_size < 1 || elem_size > 8 ||
        chunk_elems < 1 || n_total < 0 || !out) return -1;
    const size_t raw = (size_t)chunk_elems * (size_t)elem_size;          // edge chunks are stored whole
    std::vector<uint8_t> tmp(raw), tmp2(shuffle ? raw : 0);
    for (int64_t k = 0; k < n_chunks; ++k) {
        if (addr[k] < 0 || csize[k] < 0 || addr[k] + csize[k] > base_len || elem_off[k] < 0) return -(k + 2);
        uLongf got = (uLongf)raw;
        if (uncompress(tmp.data(), &got, base + addr[k], (uLong)csize[k]) != Z_OK) return -(k + 2);
        const uint8_t* src = tmp.data();
        if (shuffle) {
            // HDF5 shuffle: byte b of element i sits at b * n + i
            const size_t ne = (size_t)got / (size_t)elem_size;
            for (size_t i = 0; i < ne; ++i)
                for (int b = 0; b < elem_size; ++b) tmp2[i * elem_size + b] = tmp[(size_t)b * ne + i];
            src = tmp2.data();
        }
        if (elem_off[k] >= n_total) continue;
        int64_t take = (int64_t)got / elem_size;
        if (take > chunk_elems) take = chunk_elems;
        if (take > n_total - elem_off[k]) take = n_total - elem_off[k];
        memcpy(static_cast<uint8_t*>(out) + (size_t)elem_off[k] * elem_size, src, (size_t)take * elem_size);
    }
    return 0;
}
