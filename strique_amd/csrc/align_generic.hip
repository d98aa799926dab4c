// Generic semi-global alignment for align_overlap(a, b) with arbitrary float inputs (API parity with
// the reference's pyseqan.align_raw.align_overlap, src/pyalign.cpp:59-61, src/align_raw.h:106-158):
// any number of distinct values in `a`, any flank `b` (no run structure, any length).
//
// This is not the throughput path (detect always passes an 8-bit signal and a 6-run template and goes
// through align_kernels.hip); it exists so that the boundary accepts everything the reference accepts.
//
//   * both sequences are dictionary-encoded on the host; the score of every (distinct a, distinct b)
//     pair is evaluated once on the device with the reference's arithmetic,
//     max(off - (float)pow((double)|h - v|, 1.2), dmin), and the entries whose double pow lands within a
//     few ulp of a float rounding boundary are re-evaluated with the host libm (same scheme as
//     lut_kernels.hip), so the table is bit-equal to what the reference computes per cell;
//   * one workgroup of 1024 threads marches an anti-diagonal wavefront over a strip of 1024 flank rows
//     (thread = row), strips one after the other with the strip's last row {S, V} streamed through HBM;
//     full affine recurrence with the oracle's tie rules, one trace byte per cell like the reference;
//   * the host walks the trace back (strq_align_api.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "align_generic.h"

namespace strq {

#define STRQ_NINF (-3.4028234663852886e38f / 2)
#define STRQ_HARD_ULPS 64
// tie rules -- keep identical to oracle/align_oracle.c
#define STRQ_TIE_EXT(ext, opn)  ((ext) >= (opn))
#define STRQ_TIE_H_OVER_V(h, v) ((h) >= (v))
#define STRQ_TIE_D_OVER_G(d, g) ((d) >= (g))

__global__ void __launch_bounds__(256)
generic_table_kernel(const float* __restrict__ va, int na, const float* __restrict__ vb, int nb,
                     float* __restrict__ table, GenericHard* __restrict__ hard, int* __restrict__ hard_count,
                     int hard_cap, AlignParams p)
{
    const size_t total = (size_t)na * nb;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ib = (int)(e / na), ia = (int)(e - (size_t)ib * na);
        const float h = va[ia], v = vb[ib];
        const float d = h > v ? h - v : v - h;
        const double y = pow((double)d, 1.2);
        const uint64_t bits = __builtin_bit_cast(uint64_t, y);
        const int64_t low = (int64_t)(bits & 0x1FFFFFFFull) - 0x10000000ll;      // distance to the float midpoint
        const float x = (float)y;
        const float s = p.dist_offset - x;
        bool hd = (low >= -STRQ_HARD_ULPS && low <= STRQ_HARD_ULPS) || !(y == y) || !(d == d);
        if (hd && y == y) {
            const uint32_t xb = __builtin_bit_cast(uint32_t, x);
            const float x2 = __builtin_bit_cast(float, (double)x < y ? xb + 1u : (xb ? xb - 1u : 0u));
            const float s2 = p.dist_offset - x2;
            if (!(s > p.dist_min) && !(s2 > p.dist_min)) hd = false;         // clipped either way
        }
        table[e] = s > p.dist_min ? s : p.dist_min;
        if (hd) {
            const int slot = atomicAdd(hard_count, 1);
            if (slot < hard_cap) { hard[slot].ia = ia; hard[slot].ib = ib; }
        }
    }
}

__global__ void generic_patch_kernel(float* __restrict__ table, int na, const GenericHard* __restrict__ hard,
                                     const float* __restrict__ vals, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) table[(size_t)hard[i].ib * na + hard[i].ia] = vals[i];
}

// thread t of the (single) workgroup owns flank row row0 + t + 1 of the current strip
__global__ void __launch_bounds__(1024)
generic_align_kernel(GenericAlignArgs a)
{
    __shared__ float shS[2][1025];
    __shared__ float shV[2][1025];
    const int t = threadIdx.x;
    const AlignParams p = a.p;
    const int n = a.n, m = a.m;
    float best = a.col0[m]; int best_j = 0;
    for (int row0 = 0, strip = 0; row0 < m; row0 += 1024, ++strip) {
        const int rows = m - row0 < 1024 ? m - row0 : 1024;
        const bool mine = t < rows;
        const int i = row0 + t + 1;                              // DP row
        const bool last_strip = row0 + rows == m;
        const float* bin_S = strip ? a.bnd_S[(strip - 1) & 1] : nullptr;
        const float* bin_V = strip ? a.bnd_V[(strip - 1) & 1] : nullptr;
        float* bout_S = a.bnd_S[strip & 1]; float* bout_V = a.bnd_V[strip & 1];
        const float* trow = mine ? a.table + (size_t)a.code_b[i - 1] * a.na : nullptr;
        uint8_t* tr = mine ? a.trace + (size_t)i * ((size_t)n + 1) : nullptr;
        // column 0 (not free): S = V = col0, H = -inf
        float S_left = mine ? a.col0[i] : 0.0f, H_left = STRQ_NINF;
        float diag = mine ? a.col0[i - 1] : 0.0f;               // S[i-1][0]
        shS[0][t + 1] = S_left; shV[0][t + 1] = S_left;          // V[i][0] == S[i][0]
        shS[1][t + 1] = 0.0f; shV[1][t + 1] = STRQ_NINF;
        if (t == 0) { shS[0][0] = shS[1][0] = 0.0f; shV[0][0] = shV[1][0] = STRQ_NINF; }
        __syncthreads();
        // step k: thread t works on column j = k - t (1 <= j <= n)
        for (int k = 1; k <= n + rows - 1; ++k) {
            const int j = k - t;
            const int cur = k & 1, prev = cur ^ 1;
            float Sn = 0.0f, Vn = STRQ_NINF;
            if (mine && j >= 1 && j <= n) {
                float upS, upV;
                if (t == 0) {
                    if (strip) {          // written by this workgroup one strip ago: read past the L1
                        upS = __hip_atomic_load(bin_S + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        upV = __hip_atomic_load(bin_V + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else { upS = 0.0f; upV = STRQ_NINF; }      // free top row
                } else { upS = shS[prev][t]; upV = shV[prev][t]; }
                const float sc = trow[a.code_a[j - 1]];
                const float D = diag + sc;
                const float hext = H_left + p.ext_h, hopn = S_left + p.open_h;
                const bool he = STRQ_TIE_EXT(hext, hopn);
                const float Hn = he ? hext : hopn;
                const float vext = upV + p.ext_v, vopn = upS + p.open_v;
                const bool ve = STRQ_TIE_EXT(vext, vopn);
                Vn = ve ? vext : vopn;
                const bool gh = STRQ_TIE_H_OVER_V(Hn, Vn);
                const float G = gh ? Hn : Vn;
                const bool dd = STRQ_TIE_D_OVER_G(D, G);
                Sn = dd ? D : G;
                tr[j] = (uint8_t)((dd ? 0u : (gh ? 1u : 2u)) | (he ? 4u : 0u) | (ve ? 8u : 0u));
                S_left = Sn; H_left = Hn; diag = upS;
                if (t == rows - 1) {
                    if (last_strip) { if (Sn > best) { best = Sn; best_j = j; } }      // strict: leftmost maximum
                    else { bout_S[j] = Sn; bout_V[j] = Vn; }
                }
            }
            shS[cur][t + 1] = Sn; shV[cur][t + 1] = Vn;
            __syncthreads();
        }
        __threadfence();                 // boundary row visible to the next strip's reads (same workgroup)
        __syncthreads();
    }
    if (t == (m - 1) % 1024) { a.result->best = best; a.result->j_end = best_j; a.result->j0 = 0; a.result->status = 0; }
}

int launch_generic_table(hipStream_t st, const float* va, int na, const float* vb, int nb, float* table,
                         GenericHard* hard, int* hard_count, int hard_cap, const AlignParams& p)
{
    const size_t total = (size_t)na * nb;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(generic_table_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, va, na, vb, nb, table, hard, hard_count, hard_cap, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_generic_patch(hipStream_t st, float* table, int na, const GenericHard* hard, const float* vals, int n)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(generic_patch_kernel, dim3((n + 255) / 256), dim3(256), 0, st, table, na, hard, vals, n);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_generic_align(hipStream_t st, const GenericAlignArgs& a)
{
    hipLaunchKernelGGL(generic_align_kernel, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq
