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

__global__ void __launch_bounds__(256)
lut_build_kernel(const LutJob* __restrict__ jobs, LutInfo* __restrict__ info, HardEntry* __restrict__ hard,
                 int* __restrict__ hard_count, int hard_cap, AlignParams p)
{
    extern __shared__ float sc_all[];          // k x 256 scores
    __shared__ int lo[STRQ_LUT_MAX_K], hi[STRQ_LUT_MAX_K + 1];
    __shared__ int width, n_local, plat_lo, plat_hi;
    __shared__ unsigned short local_hard[STRQ_LUT_LOCAL_HARD][2];
    const LutJob jb = jobs[blockIdx.x];
    const int q = threadIdx.x;
    for (int k = q; k < jb.k; k += 256) { lo[k] = 256; hi[k] = -1; }
    if (q == 0) { width = 0; n_local = 0; plat_lo = 255; plat_hi = 0; }
    __syncthreads();
    const float v = jb.level_val[q];
    // plateaus: levels 0..plat_lo share the value of level 0, levels plat_hi..255 that of level 255
    {
        const float v0 = jb.level_val[0], v255 = jb.level_val[255];
        if (__builtin_bit_cast(uint32_t, v) != __builtin_bit_cast(uint32_t, v0)) atomicMin(&plat_lo, q - 1);
        if (__builtin_bit_cast(uint32_t, v) != __builtin_bit_cast(uint32_t, v255)) atomicMax(&plat_hi, q + 1);
    }
    __syncthreads();
    const int plat_lo_r = plat_lo, plat_hi_r = plat_hi;
    for (int k = 0; k < jb.k; ++k) {
        bool hd;
        const float s = cell_score_dev(p, v, jb.cls_val[k], &hd);
        sc_all[k * 256 + q] = s;
        if (s > p.dist_min) { atomicMin(&lo[k], q); atomicMax(&hi[k], q); }
        if (hd && q >= plat_lo_r && q <= plat_hi_r) {     // levels inside a plateau duplicate its end level
            const int slot = atomicAdd(&n_local, 1);
            if (slot < STRQ_LUT_LOCAL_HARD) { local_hard[slot][0] = (unsigned short)k; local_hard[slot][1] = (unsigned short)q; }
        }
    }
    __syncthreads();
    // stored range of class k: [e_l, e_r]; from here on lo[] holds e_l | e_r << 8 and hi[] the row
    // offset (ragged rows: class k stores exactly its range, rows back to back)
    {
        int el[(STRQ_LUT_MAX_K + 255) / 256], er[(STRQ_LUT_MAX_K + 255) / 256];
        int x = 0;
        for (int k = q; k < jb.k; k += 256, ++x) {
            const bool none = hi[k] < lo[k];
            el[x] = none ? 0 : (lo[k] <= plat_lo ? plat_lo : lo[k] - 1);
            er[x] = none ? 0 : (hi[k] >= plat_hi ? plat_hi : hi[k] + 1);
            atomicMax(&width, er[x] - el[x] + 1);
        }
        __syncthreads();
        x = 0;
        for (int k = q; k < jb.k; k += 256, ++x) { lo[k] = el[x] | (er[x] << 8); hi[k] = er[x] - el[x] + 1; }
    }
    __syncthreads();
    if (q == 0) {
        int off = 0;
        for (int k = 0; k < jb.k; ++k) { const int w = hi[k]; hi[k] = off; off += w; }
        hi[jb.k] = off;
    }
    __syncthreads();
    auto edge_l = [&](int k) { return lo[k] & 255; };
    auto edge_r = [&](int k) { return lo[k] >> 8; };
    const int* roff = hi;
    for (int k = 0; k < jb.k; ++k) {
        const int el = edge_l(k), er = edge_r(k);
        if (q >= el && q <= er) jb.table[roff[k] + q - el] = sc_all[k * 256 + q];
        if (q == 0) jb.band_lo[k] = (int32_t)((uint32_t)el | ((uint32_t)(er - el) << 8) | ((uint32_t)roff[k] << 16));
    }
    if (q == 0) {
        int nh = n_local;
        if (nh > STRQ_LUT_LOCAL_HARD) nh = -1;    // host rebuilds the whole table
        // a borderline entry outside the stored band could turn out above dist_min on the host:
        // the band itself would be wrong, so let the host rebuild this table
        for (int i = 0; i < nh; ++i) {
            const int k = local_hard[i][0], lv = local_hard[i][1];
            if (lv < edge_l(k) || lv > edge_r(k)) { nh = -1; break; }
        }
        info[blockIdx.x].total = roff[jb.k];
        info[blockIdx.x].need = width;
        info[blockIdx.x].pad_ = 0;
        info[blockIdx.x].n_hard = nh;
        for (int i = 0; i < nh; ++i) {
            const int k = local_hard[i][0], lv = local_hard[i][1];
            const int slot = atomicAdd(hard_count, 1);
            if (slot < hard_cap) {
                HardEntry e;
                e.job = blockIdx.x; e.k = k; e.level = lv;
                e.index = roff[k] + (lv - edge_l(k));
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
    const size_t lds = (size_t)max_k * 256 * 4;
    (void)hipFuncSetAttribute((const void*)lut_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(lut_build_kernel, dim3(n_jobs), dim3(256), lds, stream, jobs, info, hard, hard_count, hard_cap, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_lut_patch(hipStream_t stream, const LutJob* jobs, const HardEntry* hard, const float* vals, int n)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lut_patch_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, jobs, hard, vals, n);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq
