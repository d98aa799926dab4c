// Host-side helper of the fast5 reader (strique_amd/fast5.py): the chunks of one 1-D chunked HDF5 dataset behind the
// deflate (and optional shuffle) filter, inflated straight from the mapped file into the caller's array.
// Replaces what the reference gets from h5py / libhdf5 when it reads /Raw/.../Signal (STRique_lib/fast5Index.py:76-84,
// 220-233).  Pure C++ on the host; it lives in this library so that the `count` command's reader threads spend
// their time here, outside the interpreter lock, instead of in a per-chunk Python loop (46 chunks per 50 kb read).
//
// Round 4: the inflate itself was what bounded `count` on gzip-compressed files (4.8 of the 5.5 ms a 375 k-sample read took
// its reader thread: zlib at ~160 MB/s of int16 signal).  libdeflate -- in the image as libdeflate.so.0, without headers:
// its three entry points are declared here and resolved with dlopen -- decodes the same zlib streams about three times as
// fast; zlib stays as the fallback when the library is absent (or STRQ_NO_LIBDEFLATE=1: A/B runs).  Same bytes either way
// (tests/test_cli_host.py::test_deflate_chunks_native_and_python_paths).
#include "strq_opt.h"
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>
#include <zlib.h>
#include "../../include/strique_hip.h"

namespace {

// libdeflate's public C API (libdeflate.h, stable since 1.0): a decompressor object per thread, one call per zlib stream
typedef struct libdeflate_decompressor* (*ld_alloc_fn)(void);
typedef int (*ld_zlib_fn)(struct libdeflate_decompressor*, const void* in, size_t in_nbytes, void* out, size_t out_nbytes_avail,
                          size_t* actual_out_nbytes_ret);          // 0 = LIBDEFLATE_SUCCESS
typedef void (*ld_free_fn)(struct libdeflate_decompressor*);

struct LibDeflate {
    ld_alloc_fn alloc = nullptr; ld_zlib_fn zlib_decompress = nullptr; ld_free_fn release = nullptr;
    bool ok = false;
};

const LibDeflate& libdeflate()
{
    static LibDeflate L;
    static std::once_flag once;
    std::call_once(once, [] {
        if (strq::opt("STRQ_NO_LIBDEFLATE")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        L.alloc = reinterpret_cast<ld_alloc_fn>(dlsym(h, "libdeflate_alloc_decompressor"));
        L.zlib_decompress = reinterpret_cast<ld_zlib_fn>(dlsym(h, "libdeflate_zlib_decompress"));
        L.release = reinterpret_cast<ld_free_fn>(dlsym(h, "libdeflate_free_decompressor"));
        L.ok = L.alloc && L.zlib_decompress && L.release;
    });
    return L;
}

// one decompressor per reader thread, freed with the thread
struct ThreadDecompressor {
    struct libdeflate_decompressor* d = nullptr;
    ~ThreadDecompressor() { if (d) libdeflate().release(d); }
};

// inflate one zlib stream of at most `cap` bytes; returns the number of bytes produced or -1
int64_t inflate_one(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    const LibDeflate& L = libdeflate();
    if (L.ok) {
        static thread_local ThreadDecompressor td;
        if (!td.d) td.d = L.alloc();
        if (td.d) {
            size_t got = 0;
            if (L.zlib_decompress(td.d, src, n, dst, cap, &got) == 0) return (int64_t)got;
            return -1;
        }
    }
    uLongf got = (uLongf)cap;
    if (uncompress(dst, &got, src, (uLong)n) != Z_OK) return -1;
    return (int64_t)got;
}

// time spent inside the inflate itself, all threads together (diagnostics: strq_inflate_stats)
std::atomic<int64_t> g_inflate_ns{0}, g_inflate_streams{0}, g_inflate_bytes{0};

}  // namespace

// Diagnostics (tools/reader_probe.py): out[0] = nanoseconds spent inside the inflate calls since the last reset, summed over all
// threads; out[1] = zlib streams inflated; out[2] = bytes produced.  reset != 0 clears the counters after reading them.
extern "C" void strq_inflate_stats(int64_t out[3], int32_t reset)
{
    if (out) { out[0] = g_inflate_ns.load(); out[1] = g_inflate_streams.load(); out[2] = g_inflate_bytes.load(); }
    if (reset) { g_inflate_ns = 0; g_inflate_streams = 0; g_inflate_bytes = 0; }
}

// 1 when libdeflate serves strq_inflate_chunks in this process, 0 when zlib does
extern "C" int strq_inflate_backend(void) { return libdeflate().ok ? 1 : 0; }

extern "C" int64_t strq_inflate_chunks(const uint8_t* base, int64_t base_len, int64_t n_chunks, const int64_t* addr,
                                       const int32_t* csize, const int64_t* elem_off, int32_t elem_size, int32_t shuffle,
                                       int64_t chunk_elems, int64_t n_total, void* out)
{
    if (!base || base_len < 0 || n_chunks < 0 || (n_chunks > 0 && (!addr || !csize || !elem_off)) || elem_size < 1 || elem_size > 8 ||
        chunk_elems < 1 || n_total < 0 || !out) return -1;
    const size_t raw = (size_t)chunk_elems * (size_t)elem_size;          // edge chunks are stored whole
    std::vector<uint8_t> tmp(raw), tmp2(shuffle ? raw : 0);
    for (int64_t k = 0; k < n_chunks; ++k) {
        // addresses and sizes come from the file's B-tree: no arithmetic on them that could wrap
        if (addr[k] < 0 || csize[k] < 0 || (int64_t)csize[k] > base_len || addr[k] > base_len - (int64_t)csize[k] || elem_off[k] < 0) return -(k + 2);
        const bool direct = !shuffle && elem_off[k] < n_total && n_total - elem_off[k] >= chunk_elems;      // a whole chunk inside the array: no staging copy
        uint8_t* dst = direct ? static_cast<uint8_t*>(out) + (size_t)elem_off[k] * elem_size : tmp.data();
        const auto t0 = std::chrono::steady_clock::now();
        const int64_t got = inflate_one(base + addr[k], (size_t)csize[k], dst, raw);
        g_inflate_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        ++g_inflate_streams; g_inflate_bytes += got > 0 ? got : 0;
        // libhdf5 stores every chunk whole, the last one of a dataset padded to the chunk size; other writers (the file the
        // reference bundles, data/c9orf72.fast5) end the last chunk with the data.  Either way a chunk must deliver every
        // element of the dataset that falls into it -- anything shorter is a damaged file, not zeros
        int64_t need = elem_off[k] < n_total ? n_total - elem_off[k] : 0;
        if (need > chunk_elems) need = chunk_elems;
        if (got < 0 || got % elem_size != 0 || got < need * elem_size) return -(k + 2);
        if (direct) continue;
        const uint8_t* src = tmp.data();
        if (shuffle) {
            // HDF5 shuffle: byte b of element i sits at b * n + i, n = the elements the chunk holds
            const size_t ne = (size_t)got / (size_t)elem_size;
            for (size_t i = 0; i < ne; ++i)
                for (int b = 0; b < elem_size; ++b) tmp2[i * elem_size + b] = tmp[(size_t)b * ne + i];
            src = tmp2.data();
        }
        if (elem_off[k] >= n_total) continue;
        int64_t take = chunk_elems;
        if (take > n_total - elem_off[k]) take = n_total - elem_off[k];
        memcpy(static_cast<uint8_t*>(out) + (size_t)elem_off[k] * elem_size, src, (size_t)take * elem_size);
    }
    return 0;
}

// The datasets of one reader task in one call (the `count` command reads 32 reads per task): dataset i has the chunks
// [chunk_first[i], chunk_first[i + 1]) of the concatenated addr / csize / elem_off arrays and its own mapped file.  One call
// instead of 32 means the reader thread takes the interpreter lock once per task, not twice per read -- with 16 threads and
// more the handovers, not the inflate, were what the readers waited for (tools/reader_probe.py).  status[i] = the return
// value strq_inflate_chunks would give for dataset i; returns the number of datasets that failed.
extern "C" int64_t strq_inflate_many(int64_t n_ds, const uint8_t* const* base, const int64_t* base_len, const int64_t* chunk_first,
                                     const int64_t* addr, const int32_t* csize, const int64_t* elem_off, const int32_t* elem_size,
                                     const int32_t* shuffle, const int64_t* chunk_elems, const int64_t* n_total, void* const* out,
                                     int64_t* status)
{
    if (n_ds < 0 || (n_ds > 0 && (!base || !base_len || !chunk_first || !elem_size || !shuffle || !chunk_elems || !n_total || !out || !status))) return -1;
    int64_t failed = 0;
    for (int64_t i = 0; i < n_ds; ++i) {
        const int64_t c0 = chunk_first[i], c1 = chunk_first[i + 1];
        if (c0 < 0 || c1 < c0) { status[i] = -1; ++failed; continue; }
        status[i] = strq_inflate_chunks(base[i], base_len[i], c1 - c0, addr ? addr + c0 : nullptr, csize ? csize + c0 : nullptr,
                                        elem_off ? elem_off + c0 : nullptr, elem_size[i], shuffle[i], chunk_elems[i], n_total[i], out[i]);
        if (status[i] != 0) ++failed;
    }
    return failed;
}

// ------------------------------------------------------------------------------------------------------------------
// VBZ chunks (HDF5 filter 32020, strique_amd/vbz.py for the format and its provenance -- UNPINNED like that module: this is the
// same decode, moved out of a per-chunk Python loop): u32 size of the decoded chunk, then a zstd frame (zstd level != 0) around the
// variable-byte stream that strq_svb_decode reads.  zstd comes from the system's libzstd.so.1 through dlopen (no headers in the
// image: the three entry points are declared here).  Every layer is length-checked as in vbz.decode: the frame must decode, the
// stream must be consumed exactly, the decoded size must be the size the chunk states.
namespace {

typedef size_t (*zstd_decompress_fn)(void* dst, size_t cap, const void* src, size_t n);
typedef unsigned (*zstd_iserror_fn)(size_t code);
typedef unsigned long long (*zstd_content_size_fn)(const void* src, size_t n);
struct LibZstd { zstd_decompress_fn decompress = nullptr; zstd_iserror_fn is_error = nullptr; zstd_content_size_fn content_size = nullptr; bool ok = false; };

const LibZstd& libzstd()
{
    static LibZstd L;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        L.decompress = reinterpret_cast<zstd_decompress_fn>(dlsym(h, "ZSTD_decompress"));
        L.is_error = reinterpret_cast<zstd_iserror_fn>(dlsym(h, "ZSTD_isError"));
        L.content_size = reinterpret_cast<zstd_content_size_fn>(dlsym(h, "ZSTD_getFrameContentSize"));
        L.ok = L.decompress && L.is_error && L.content_size;
    });
    return L;
}

}  // namespace

// Same contract as strq_inflate_chunks for chunks behind the VBZ filter with client data {version, integer size, zig-zag flag,
// zstd level}; elem_size must equal the integer size (2 or 4).  Returns 0, -1 on a bad argument or a combination this helper does
// not decode (the caller's Python path then does, or says why not), -(k + 2) when chunk k is out of bounds or damaged.
extern "C" int64_t strq_vbz_chunks(const uint8_t* base, int64_t base_len, int64_t n_chunks, const int64_t* addr, const int32_t* csize,
                                   const int64_t* elem_off, int32_t elem_size, int32_t version, int32_t isize, int32_t zigzag, int32_t level,
                                   int64_t chunk_elems, int64_t n_total, void* out)
{
    if (!base || base_len < 0 || n_chunks < 0 || (n_chunks > 0 && (!addr || !csize || !elem_off)) || chunk_elems < 1 || n_total < 0 || !out) return -1;
    if ((version != 0 && version != 1) || (isize != 2 && isize != 4) || isize != elem_size) return -1;
    const LibZstd& Z = libzstd();
    if (level && !Z.ok) return -1;
    const int key_bits = (version == 1 && isize == 2) ? 1 : 2;
    std::vector<uint8_t> stream, vals((size_t)chunk_elems * (size_t)isize);
    for (int64_t k = 0; k < n_chunks; ++k) {
        if (addr[k] < 0 || csize[k] < 4 || (int64_t)csize[k] > base_len || addr[k] > base_len - (int64_t)csize[k] || elem_off[k] < 0) return -(k + 2);
        const uint8_t* chunk = base + addr[k];
        uint32_t size; memcpy(&size, chunk, 4);
        if (size % (uint32_t)isize != 0 || (int64_t)(size / (uint32_t)isize) > chunk_elems) return -(k + 2);
        const int64_t n = size / (uint32_t)isize;
        const uint8_t* payload = chunk + 4; size_t plen = (size_t)csize[k] - 4;
        if (level) {
            const size_t bound = (size_t)((n + 3) / 4 + 4 * n) + size + 64;
            const unsigned long long stated = Z.content_size(payload, plen);
            const size_t cap = stated < (1ull << 40) ? (size_t)stated : bound;          // unknown / error codes are huge values
            if (cap > bound) return -(k + 2);
            stream.resize(cap > 0 ? cap : 1);
            const size_t got = Z.decompress(stream.data(), cap, payload, plen);
            if (Z.is_error(got)) return -(k + 2);
            payload = stream.data(); plen = got;
        }
        const bool direct = elem_off[k] < n_total && n_total - elem_off[k] >= n;      // the whole chunk lies inside the array
        uint8_t* dst = direct ? static_cast<uint8_t*>(out) + (size_t)elem_off[k] * (size_t)isize : vals.data();
        const int64_t used = strq_svb_decode(payload, (int64_t)plen, n, key_bits, zigzag ? 1 : 0, isize, dst);
        if (used != (int64_t)plen) return -(k + 2);                                    // the stream must be consumed exactly
        int64_t need = elem_off[k] < n_total ? n_total - elem_off[k] : 0;
        if (need > chunk_elems) need = chunk_elems;
        if (n < need) return -(k + 2);                                                 // a chunk delivers every element that falls into it
        if (!direct && elem_off[k] < n_total) memcpy(static_cast<uint8_t*>(out) + (size_t)elem_off[k] * (size_t)isize, vals.data(), (size_t)need * (size_t)isize);
    }
    return 0;
}
