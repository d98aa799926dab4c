// Host-side decoding of the variable-byte layer of VBZ-compressed fast5 signals (no device code in this file; see
// strique_amd/vbz.py for the format and its provenance).  The reader threads of `count` call it through ctypes,
// which releases the GIL: a 375 k-sample read decodes in well under a millisecond instead of ~15 ms in numpy.
#include <cstdint>
#include <cstring>
#include "../../include/strique_hip.h"

// stream: key bytes, then data bytes.  key_bits 2: classic StreamVByte (1-4 data bytes per integer); 1: the 16-bit
// variant (1-2 data bytes).  zigzag: the integers are zig-zag coded differences (first to 0).  out: n integers of
// `isize` bytes (2 or 4).  Returns the number of stream bytes consumed, or -1 if the stream is shorter than its keys say.
extern "C" int64_t strq_svb_decode(const uint8_t* stream, int64_t stream_len, int64_t n, int32_t key_bits, int32_t zigzag,
                                   int32_t isize, void* out)
{
    if (!stream || n < 0 || (key_bits != 1 && key_bits != 2) || (isize != 2 && isize != 4) || (n > 0 && !out)) return -1;
    const int64_t nk = key_bits == 2 ? (n + 3) / 4 : (n + 7) / 8;
    if (stream_len < nk) return -1;
    const uint8_t* keys = stream;
    const uint8_t* p = stream + nk;
    const uint8_t* end = stream + stream_len;
    uint32_t acc32 = 0; uint16_t acc16 = 0;
    for (int64_t i = 0; i < n; ++i) {
        int len;
        if (key_bits == 2) len = ((keys[i >> 2] >> ((i & 3) * 2)) & 3) + 1;
        else len = ((keys[i >> 3] >> (i & 7)) & 1) + 1;
        if (end - p < len) return -1;
        uint32_t u = 0;
        for (int b = 0; b < len; ++b) u |= (uint32_t)p[b] << (8 * b);
        p += len;
        if (key_bits == 1) {               // 16-bit arithmetic
            uint16_t v = (uint16_t)u;
            if (zigzag) { v = (uint16_t)((v >> 1) ^ (uint16_t)(0 - (v & 1))); acc16 = (uint16_t)(acc16 + v); v = acc16; }
            if (isize == 2) static_cast<uint16_t*>(out)[i] = v; else static_cast<uint32_t*>(out)[i] = v;
        } else {
            uint32_t v = u;
            if (zigzag) { v = (v >> 1) ^ (0u - (v & 1u)); acc32 += v; v = acc32; }
            if (isize == 2) static_cast<uint16_t*>(out)[i] = (uint16_t)v; else static_cast<uint32_t*>(out)[i] = v;
        }
    }
    return (int64_t)(p - stream);
}
