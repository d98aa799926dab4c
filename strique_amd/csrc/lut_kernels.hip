// Per-alignment score tables for the flank DP.
//
// Replaces the per-cell evaluation of Score<float,Distance>::score (reference
// src/score_distance.h:115-122):  s = max(dist_offset - (float)pow((double)|h - v|, 1.2), dist_min)
// by one evaluation per (k-mer class of the flank, 8-bit level of the read) pair.
//
// The table is banded and ragged: for class k only a contiguous level range [e_l, e_r] is stored
// (rows back to back; band_lo[k] packs e_l | (e_r - e_l) << 8 | row offset << 16) and the DP
// kernel clamps the level into it with one v_med3_i32.  A typical 145-class flank needs ~6400
// floats, which lets six DP waves share one CU's LDS.  The range is chosen so that clamping
// cannot change a score: left of e_l (right of e_r) every level scores like e_l (e_r) -- either
// because they all clip to dist_min, or because they all sit on the plateau that
// normalize2model's np.clip (scripts/STRique.py:178-179) makes of the lowest / highest levels.
//
// Bit-exactness versus the host libm: the reference casts a double pow to float.  Two pow
// implementations that are both accurate to a few ulp(double) give the same float unless the
// double result falls within that error of a float rounding boundary.  Entries whose device
// result is within STRQ_HARD_ULPS ulp(double) of such a boundary are listed for the host, which
// re-evaluates exactly those with its own pow (expected: ~1 entry per 100 tables).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lut_kernels.h"

namespace strq {

#define STRQ_HARD_ULPS 64

static __device__ __forceinline__ float cell_score_dev(const AlignParams& p, float h, float v, bool* hard)
{
    const float d = h > v ? h - v : v - h;
    const double y = pow((double)d, 1.2);
    const uint64_t bits = __builtin_bit_cast(uint64_t, y);
    const int64_t low = (int64_t)(bits & 0x1FFFFFFFull) - 0x10000000ll;   // distance to the float midpoint
    bool hd = (low >= -STRQ_HARD_ULPS && low <= STRQ_HARD_ULPS);
    const float x = (float)y;
    const float s = p.dist_offset - x;
    if (hd) {
        // the other float the host could have rounded to; if both give a clipped score the entry is
        // dist_min on the host as well and needs no second opinion
        // x = (float)y >= 0 here (a power of a non-negative number): neighbours by bit pattern
        const uint32_t xb = __builtin_bit_cast(uint32_t, x);
        const float x2 = __builtin_bit_cast(float, (double)x < y ? xb + 1u : (xb ? xb - 1u : 0u));
        const float s2 = p.dist_offset - x2;
        if (!(s > p.dist_min) && !(s2 > p.dist_min)) hd = false;
    }
    *hard = hd;
    return s > p.dist_min ? s : p.dist_min;
}

// One workgroup per table.  The level values of a read are non-decreasing in the level (they are an
// affine map of the level, clipped), so for a class the levels that score above dist_min form one
// contiguous range around the level closest to the class value: the range is found by bisection
// (~16 pow evaluations per class) and only its entries are evaluated (~44 per class instead of 256).
// A table whose level values are not monotone is handed to the host (n_hard = -1).
__global__ void __launch_bounds__(256)
lut_build_kernel(const LutJob* __restrict__ jobs, LutInfo* __restrict__ info, HardEntry* __restrict__ hard,
                 int* __restrict__ hard_count, int hard_cap, AlignParams p)
{
    __shared__ float v[256];
    __shared__ float cls[STRQ_LUT_MAX_K];
    __shared__ int lo[STRQ_LUT_MAX_K], hi[STRQ_LUT_MAX_K + 1];      // later: e_l | e_r << 8 and the row offsets
    __shared__ int width, n_local, plat_lo, plat_hi, rebuild, unpackable;
    __shared__ unsigned char dup[STRQ_LUT_MAX_K];
    __shared__ unsigned short local_hard[STRQ_LUT_LOCAL_HARD][2];
    const LutJob jb = jobs[blockIdx.x];
    const int q = threadIdx.x, lane = q & 63, wave = q >> 6;
    v[q] = jb.level_val[q];
    for (int k = q; k < jb.k; k += 256) cls[k] = jb.cls_val[k];
    if (q == 0) { width = 0; n_local = 0; plat_lo = 255; plat_hi = 0; rebuild = 0; unpackable = 0; }
    __syncthreads();
    {
        // plateaus: levels 0..plat_lo share the value of level 0, levels plat_hi..255 that of level 255
        const float v0 = v[0], v255 = v[255], vq = v[q];
        if (__builtin_bit_cast(uint32_t, vq) != __builtin_bit_cast(uint32_t, v0)) atomicMin(&plat_lo, q - 1);
        if (__builtin_bit_cast(uint32_t, vq) != __builtin_bit_cast(uint32_t, v255)) atomicMax(&plat_hi, q + 1);
        if (q > 0 && !(v[q - 1] <= vq)) rebuild = 1;          // not monotone (or NaN): no contiguous bands
    }
    __syncthreads();
    if (rebuild) { if (q == 0) { info[blockIdx.x].total = 0; info[blockIdx.x].need = 0; info[blockIdx.x].packed = 0; info[blockIdx.x].n_hard = -1; } return; }
    const int plat_lo_r = plat_lo, plat_hi_r = plat_hi;
    auto in_band = [&](int lv, float c) { bool hd; return cell_score_dev(p, v[lv], c, &hd) > p.dist_min; };
    // ---- band of every class: first / last level scoring above dist_min
    for (int k = q; k < jb.k; k += 256) {
        const float c = cls[k];
        int a = 0, b = 256;                                   // lower bound of c among the level values
        while (a < b) { const int m = (a + b) >> 1; if (v[m] < c) a = m + 1; else b = m; }
        int best = a < 256 ? a : 255;
        if (a > 0 && a < 256) { const float dl = c - v[a - 1], dr = v[a] - c; if (dl <= dr) best = a - 1; }
        int l = 256, h = -1;
        bool hd_best;
        const bool best_in = cell_score_dev(p, v[best], c, &hd_best) > p.dist_min;
        if (hd_best) rebuild = 1;                             // borderline at the centre of the band: let the host decide
        if (best_in) {
            int x = 0, y = best;                              // first level in [0, best] inside the band
            while (x < y) { const int m = (x + y) >> 1; if (in_band(m, c)) y = m; else x = m + 1; }
            l = x;
            x = best; y = 255;                                // last level in [best, 255] inside the band
            while (x < y) { const int m = (x + y + 1) >> 1; if (in_band(m, c)) x = m; else y = m - 1; }
            h = x;
        }
        const bool none = h < l;
        const int el = none ? 0 : (l <= plat_lo_r ? plat_lo_r : l - 1);
        const int er = none ? 0 : (h >= plat_hi_r ? plat_hi_r : h + 1);
        lo[k] = el | (er << 8) | ((!none && l <= plat_lo_r) ? 0 : 1 << 16) | ((!none && h >= plat_hi_r) ? 0 : 1 << 17);   // bits 16/17: the edge entry is a clipped one
        hi[k] = er - el + 1;
        atomicMax(&width, er - el + 1);
    }
    __syncthreads();
    // classes with the same value (a k-mer that occurs more than once in the flank: 27 of the 145 of
    // the C9orf72 prefix) share one row: dup[k] = first class with the same bits
    for (int k = q; k < jb.k; k += 256) {
        int first = k;
        const uint32_t bits = __builtin_bit_cast(uint32_t, cls[k]);
        for (int j = 0; j < k; ++j) if (__builtin_bit_cast(uint32_t, cls[j]) == bits) { first = j; break; }
        dup[k] = (unsigned char)first;
    }
    __syncthreads();
    if (q == 0) {
        int off = 0;
        for (int k = 0; k < jb.k; ++k) {
            if (dup[k] == k) { const int w = hi[k]; hi[k] = off; off += w; }
            else hi[k] = hi[dup[k]];
        }
        hi[jb.k] = off;
    }
    __syncthreads();
    const int* roff = hi;
    // ---- the stored entries, one wave per class at a time
    for (int k = wave; k < jb.k; k += 4) {
        const int d = lo[k], el = d & 255, er = (d >> 8) & 255;
        const float c = cls[k];
        if (dup[k] != k) {           // row already written for the first occurrence
            if (lane == 0) jb.band_lo[k] = (int32_t)((uint32_t)el | ((uint32_t)(er - el) << 8) | ((uint32_t)roff[k] << 16));
            continue;
        }
        for (int lv = el + lane; lv <= er; lv += 64) {
            bool hd;
            const float s = cell_score_dev(p, v[lv], c, &hd);
            jb.table[roff[k] + lv - el] = s;
            {   // 24-bit fixed point in units of 2^-20, when that is the same number
                const float r = s * 1048576.0f;
                const uint32_t u = r >= 0.0f && r < 16777216.0f ? (uint32_t)r : 0u;
                if (!((float)u == r) || !(r < 16777216.0f) || __builtin_bit_cast(uint32_t, r) == 0x80000000u) unpackable = 1;
                // two planes: the high 16 bits of every entry, then (dword aligned) the low 8 bits
                const int e = roff[k] + lv - el, total = roff[jb.k];
                reinterpret_cast<uint16_t*>(jb.table3)[e] = (uint16_t)(u >> 8);
                jb.table3[((2 * total + 3) & ~3) + e] = (uint8_t)u;
            }
            if (hd) {
                // a borderline value in a clipped edge entry could come out above dist_min on the host: the
                // band itself would be wrong, so let the host rebuild this table
                if ((lv == el && (d & (1 << 16))) || (lv == er && (d & (1 << 17)))) rebuild = 1;
                else {
                    const int slot = atomicAdd(&n_local, 1);
                    if (slot < STRQ_LUT_LOCAL_HARD) { local_hard[slot][0] = (unsigned short)k; local_hard[slot][1] = (unsigned short)lv; }
                }
            }
        }
        if (lane == 0) jb.band_lo[k] = (int32_t)((uint32_t)el | ((uint32_t)(er - el) << 8) | ((uint32_t)roff[k] << 16));
    }
    __syncthreads();
    if (q == 0) {
        int nh = n_local;
        if (nh > STRQ_LUT_LOCAL_HARD || rebuild) nh = -1;    // host rebuilds the whole table
        info[blockIdx.x].total = roff[jb.k];
        info[blockIdx.x].need = width;
        info[blockIdx.x].packed = (nh == 0 && !unpackable) ? 1 : 0;      // patched or rebuilt tables stay float32
        info[blockIdx.x].n_hard = nh;
        for (int i = 0; i < nh; ++i) {
            const int k = local_hard[i][0], lv = local_hard[i][1];
            const int slot = atomicAdd(hard_count, 1);
            if (slot < hard_cap) {
                HardEntry e;
                e.job = blockIdx.x; e.k = k; e.level = lv;
                e.index = roff[k] + (lv - (lo[k] & 255));
                hard[slot] = e;
            }
        }
    }
}

__global__ void lut_patch_kernel(const LutJob* __restrict__ jobs, const HardEntry* __restrict__ hard,
                                 const float* __restrict__ vals, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) jobs[hard[i].job].table[hard[i].index] = vals[i];
}

int launch_lut_build(hipStream_t stream, const LutJob* jobs, LutInfo* info, int n_jobs, int max_k,
                     HardEntry* hard, int* hard_count, int hard_cap, const AlignParams& p)
{
    if (max_k > STRQ_LUT_MAX_K) return 2;
    hipLaunchKernelGGL(lut_build_kernel, dim3(n_jobs), dim3(256), 0, stream, jobs, info, hard, hard_count, hard_cap, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_lut_patch(hipStream_t stream, const LutJob* jobs, const HardEntry* hard, const float* vals, int n)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lut_patch_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, jobs, hard, vals, n);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq
