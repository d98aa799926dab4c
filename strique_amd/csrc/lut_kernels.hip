// Per-alignment score tables for the flank DP.
//
// Replaces the per-cell evaluation of Score<float,Distance>::score (reference
// src/score_distance.h:115-122):  s = max(dist_offset - (float)pow((double)|h - v|, 1.2), dist_min)
// by one evaluation per (k-mer class of the flank, 8-bit level of the read) pair.
//
// The table is banded: for class k only the contiguous level range [lo_k, hi_k] that contains
// every level scoring above dist_min is stored, framed by two guard entries equal to dist_min,
// so the DP kernel clamps the level into the band with one v_med3_i32.
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
    *hard = (low >= -STRQ_HARD_ULPS && low <= STRQ_HARD_ULPS);
    const float s = p.dist_offset - (float)y;
    return s > p.dist_min ? s : p.dist_min;
}

__global__ void __launch_bounds__(256)
lut_build_kernel(const LutJob* __restrict__ jobs, LutInfo* __restrict__ info, HardEntry* __restrict__ hard,
                 int* __restrict__ hard_count, int hard_cap, AlignParams p)
{
    extern __shared__ float sc_all[];          // k x 256 scores
    __shared__ int lo[STRQ_LUT_MAX_K], hi[STRQ_LUT_MAX_K];
    __shared__ int width, n_local;
    __shared__ unsigned short local_hard[STRQ_LUT_LOCAL_HARD][2];
    const LutJob jb = jobs[blockIdx.x];
    const int q = threadIdx.x;
    for (int k = q; k < jb.k; k += 256) { lo[k] = 256; hi[k] = -1; }
    if (q == 0) { width = 0; n_local = 0; }
    __syncthreads();
    const float v = jb.level_val[q];
    for (int k = 0; k < jb.k; ++k) {
        bool hd;
        const float s = cell_score_dev(p, v, jb.cls_val[k], &hd);
        sc_all[k * 256 + q] = s;
        if (s > p.dist_min) { atomicMin(&lo[k], q); atomicMax(&hi[k], q); }
        if (hd) {
            const int slot = atomicAdd(&n_local, 1);
            if (slot < STRQ_LUT_LOCAL_HARD) { local_hard[slot][0] = (unsigned short)k; local_hard[slot][1] = (unsigned short)q; }
        }
    }
    __syncthreads();
    for (int k = q; k < jb.k; k += 256) if (hi[k] >= lo[k]) atomicMax(&width, hi[k] - lo[k] + 1);
    __syncthreads();
    const int need = width + 2;
    const int tw = need <= 64 ? 64 : (need <= 128 ? 128 : 258);
    const int stride = tw + 1;
    for (int idx = q; idx < jb.k * stride; idx += 256) {
        const int k = idx / stride, w = idx - k * stride;
        const int blo = (hi[k] >= lo[k] ? lo[k] : 1) - 1;
        const int lv = blo + w;
        float s = p.dist_min;
        if (w < tw && lv >= lo[k] && lv <= hi[k]) s = sc_all[k * 256 + lv];
        jb.table[idx] = s;
        if (w == 0) jb.band_lo[k] = blo;
    }
    if (q == 0) {
        int nh = n_local;
        if (nh > STRQ_LUT_LOCAL_HARD) nh = -1;    // host rebuilds the whole table
        // a borderline entry outside the stored band could turn out above dist_min on the host:
        // the band itself would be wrong, so let the host rebuild this table
        for (int i = 0; i < nh; ++i) {
            const int k = local_hard[i][0], lv = local_hard[i][1];
            if (lv < lo[k] || lv > hi[k]) { nh = -1; break; }
        }
        info[blockIdx.x].tw = tw;
        info[blockIdx.x].n_hard = nh;
        for (int i = 0; i < nh; ++i) {
            const int k = local_hard[i][0], lv = local_hard[i][1];
            const int slot = atomicAdd(hard_count, 1);
            if (slot < hard_cap) {
                HardEntry e;
                e.job = blockIdx.x; e.k = k; e.level = lv;
                e.index = k * stride + (lv - (lo[k] - 1));
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
