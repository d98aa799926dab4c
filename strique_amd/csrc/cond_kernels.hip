// Signal conditioning of repeatCounter.detect on gfx950 -- reference scripts/STRique.py:590-597:
//   flt   = scipy.signal.medfilt(raw, 3)                       zero padded, dtype preserved
//   z     = (flt - median(flt)) / MAD(flt)                     MAD = mean |x - median|      (:142-143)
//   u8    = clip(z*24 + 127, 0, 255).astype(uint8)
//   u8    = closing(opening(u8, rectangle(1,8)), rectangle(1,8))            scikit-image 0.14
//   morph = normalize2model(u8, 'minmax');  fltn = normalize2model(flt, 'minmax')            (:150-180)
//
// Everything numpy computes with order statistics (median, percentile, medians of the tails) is
// taken from exact histograms: 65536 bins for int16 samples, 256 bins for the 8-bit signal.  The
// sums involved are sums of multiples of 0.5 and therefore exact in float64 in any order, and the
// remaining scalar arithmetic repeats numpy's operations one rounding at a time
// (-ffp-contract=off), so the results are bit-identical to the CPU oracle (oracle/strique_oracle.py).
//
// These kernels are byte/short streaming: coalesced 2-byte loads, LDS tiles for the four sliding
// min/max passes, histograms accumulated in an LDS window per tile and flushed bin by bin.
#include "strq_opt.h"
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>
#include "cond_kernels.h"

namespace strq {

#define COND_TILE 2048      // samples per workgroup (256 threads x 8)
#define COND_HALO 16        // >= 14 = 3+4+4+3 samples of context on each side of a tile

template <class T> static __device__ __forceinline__ T med3(T a, T b, T c)
{
    const T lo = a < b ? a : b, hi = a < b ? b : a;
    const T m = hi < c ? hi : c;
    return lo > m ? lo : m;
}

// scipy.ndimage 'reflect' (d c b a | a b c d | d c b a), valid for any offset
static __device__ __forceinline__ int reflect_idx(int i, int n)
{
    const int period = 2 * n;
    int r = i % period; if (r < 0) r += period;
    return r >= n ? period - 1 - r : r;
}

template <class T>
__global__ void __launch_bounds__(256)
medfilt_kernel(const T* __restrict__ raw_all, T* __restrict__ flt_all, const ReadCond* __restrict__ rc_all,
               uint32_t* __restrict__ hist_flt, uint32_t* __restrict__ hist_raw)
{
    const ReadCond rc = rc_all[blockIdx.y];
    const int n = rc.n;
    const int base = blockIdx.x * COND_TILE;
    if (base >= n) return;
    const T* raw = raw_all + rc.off;
    T* flt = flt_all + rc.off;
    for (int k = 0; k < COND_TILE / 256; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        if (i >= n) break;
        const T c = raw[i];
        const T a = i > 0 ? raw[i - 1] : (T)0;           // zero padding (scipy medfilt)
        const T b = i + 1 < n ? raw[i + 1] : (T)0;
        const T m = med3<T>(a, c, b);
        flt[i] = m;
    }
    (void)hist_flt; (void)hist_raw;
}

// 65536-bin histogram of an int16 signal.  One workgroup takes HIST_TILE consecutive samples of
// one read, histograms them in LDS in a window of HIST_WIN values anchored at the tile minimum
// (nanopore DAC values of a tile span one or two thousand counts; 16 KB of LDS leave room for eight workgroups per CU,
// which is what hides the latency of this streaming kernel), and flushes only the non-empty bins
// with global atomics: ~20x fewer global atomics than one per sample.  Samples outside the window
// (never seen on real signals) take the global atomic directly, so the result is exact either way.
#define HSTAT_LDS_BINS 14336      // bins hist_stats_kernel can stage in LDS (56 KB)
#define HIST_TILE 16384
#define HIST_WIN 4096
__global__ void __launch_bounds__(256)
hist16_kernel(const int16_t* __restrict__ sig_all, const ReadCond* __restrict__ rc_all, uint32_t* __restrict__ hist_all,
              uint32_t* __restrict__ range_all, int range_stride)
{
    __shared__ uint32_t bins[HIST_WIN];
    __shared__ int tmin, tmax;
    const ReadCond rc = rc_all[blockIdx.y];
    const int n = rc.n;
    const int base = blockIdx.x * HIST_TILE;
    if (base >= n) return;
    const int16_t* sig = sig_all + rc.off;
    uint32_t* hist = hist_all + (size_t)blockIdx.y * 65536;
    const int end = base + HIST_TILE < n ? base + HIST_TILE : n;
    if (threadIdx.x == 0) { tmin = 32767; tmax = -32768; }
    for (int b = threadIdx.x; b < HIST_WIN; b += 256) bins[b] = 0;
    __syncthreads();
    int mn = 32767, mx = -32768;
    for (int i = base + threadIdx.x; i < end; i += 256) { const int v = sig[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(mn, o, 64); mn = x < mn ? x : mn; const int y = __shfl_xor(mx, o, 64); mx = y > mx ? y : mx; }
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin, mn); atomicMax(&tmax, mx); }
    __syncthreads();
    const int lo = tmin;
    if (threadIdx.x == 0 && range_all) {
        // occupied bin range of the read, both ends as maxima over zero-initialised words: [0] = highest bin,
        // [1] = 65535 - lowest bin
        atomicMax(&range_all[(size_t)blockIdx.y * range_stride], (uint32_t)(tmax + 32768));
        atomicMax(&range_all[(size_t)blockIdx.y * range_stride + 1], (uint32_t)(65535 - (tmin + 32768)));
    }
    for (int i = base + threadIdx.x; i < end; i += 256) {
        const int v = sig[i], w = v - lo;
        if (w < HIST_WIN) atomicAdd(&bins[w], 1u);
        else atomicAdd(&hist[v + 32768], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < HIST_WIN; b += 256) { const uint32_t c = bins[b]; if (c) atomicAdd(&hist[lo + b + 32768], c); }
}

// medfilt3 + histogram of its output in one pass over an int16 signal (the filtered samples of a tile
// stay in registers between the filter, the tile-minimum reduction and the LDS histogram).
__global__ void __launch_bounds__(256)
medfilt_hist16_kernel(const int16_t* __restrict__ raw_all, int16_t* __restrict__ flt_all, const ReadCond* __restrict__ rc_all,
                      uint32_t* __restrict__ hist_all, uint32_t* __restrict__ range_all, int range_stride)
{
    __shared__ uint32_t bins[HIST_WIN];
    __shared__ int tmin, tmax;
    const ReadCond rc = rc_all[blockIdx.y];
    const int n = rc.n;
    const int base = blockIdx.x * HIST_TILE;
    if (base >= n) return;
    const int16_t* raw = raw_all + rc.off;
    int16_t* flt = flt_all + rc.off;
    uint32_t* hist = hist_all + (size_t)blockIdx.y * 65536;
    if (threadIdx.x == 0) { tmin = 32767; tmax = -32768; }
    for (int b = threadIdx.x; b < HIST_WIN; b += 256) bins[b] = 0;
    constexpr int PER = HIST_TILE / 256;
    int16_t m[PER];
    int mn = 32767, mx = -32768;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        int16_t v = 0;
        if (i < n) {
            const int16_t c = raw[i];
            const int16_t a = i > 0 ? raw[i - 1] : (int16_t)0;           // zero padding (scipy medfilt)
            const int16_t b = i + 1 < n ? raw[i + 1] : (int16_t)0;
            v = med3<int16_t>(a, c, b);
            flt[i] = v;
            mn = v < mn ? v : mn; mx = v > mx ? v : mx;
        }
        m[k] = v;
    }
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(mn, o, 64); mn = x < mn ? x : mn; const int y = __shfl_xor(mx, o, 64); mx = y > mx ? y : mx; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin, mn); atomicMax(&tmax, mx); }
    __syncthreads();
    const int lo = tmin;
    if (threadIdx.x == 0 && range_all) {
        atomicMax(&range_all[(size_t)blockIdx.y * range_stride], (uint32_t)(tmax + 32768));
        atomicMax(&range_all[(size_t)blockIdx.y * range_stride + 1], (uint32_t)(65535 - (tmin + 32768)));
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        if (i < n) {
            const int v = m[k], w = v - lo;
            if (w < HIST_WIN) atomicAdd(&bins[w], 1u);
            else atomicAdd(&hist[v + 32768], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < HIST_WIN; b += 256) { const uint32_t c = bins[b]; if (c) atomicAdd(&hist[lo + b + 32768], c); }
}

// The same, 16 bytes per lane: every thread takes eight consecutive samples with one aligned 16-byte load and stores the
// eight filtered samples with one 16-byte store (2-byte accesses -- 128 bytes per wave instruction -- leave the kernel
// waiting on memory latency at a quarter of the HBM rate).  Reads sit back to back in the batch, so a read starts anywhere
// inside a 16-byte line: tiles are laid out in aligned vector space, `a0` samples before the read's first sample, and the
// vectors that straddle the read's ends are masked (loads stay inside the batch buffer, which has slack at both ends of a
// sub-batch; stores of partial vectors go sample by sample -- the bytes next to them belong to the neighbouring read).
// `raw` and `flt` must have the same alignment phase (strq_detect_api.hip shifts the filtered buffer accordingly).
__global__ void __launch_bounds__(256)
medfilt_hist16_vec_kernel(const int16_t* __restrict__ raw_all, int16_t* __restrict__ flt_all, const ReadCond* __restrict__ rc_all,
                          uint32_t* __restrict__ hist_all, uint32_t* __restrict__ range_all, int range_stride)
{
    __shared__ uint32_t bins[HIST_WIN];
    __shared__ int tmin, tmax;
    const ReadCond rc = rc_all[blockIdx.y];
    const int n = rc.n;
    const int16_t* raw = raw_all + rc.off;
    int16_t* flt = flt_all + rc.off;
    const int a0 = (int)((reinterpret_cast<uintptr_t>(raw) >> 1) & 7);      // samples between the 16-byte boundary in front of the read and its first sample
    const int v_base = blockIdx.x * (HIST_TILE / 8);                        // first vector of the tile (vector v holds samples 8v - a0 ... 8v - a0 + 7)
    if (8 * v_base - a0 >= n) return;
    const int4* raw_al = reinterpret_cast<const int4*>(raw - a0);
    int4* flt_al = reinterpret_cast<int4*>(flt - a0);
    uint32_t* hist = hist_all + (size_t)blockIdx.y * 65536;
    if (threadIdx.x == 0) { tmin = 32767; tmax = -32768; }
    for (int b = threadIdx.x; b < HIST_WIN; b += 256) bins[b] = 0;
    constexpr int PER = HIST_TILE / 8 / 256;      // vectors per thread
    int16_t m[PER][8];
    int mn = 32767, mx = -32768;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int v = v_base + k * 256 + threadIdx.x;
        const int i0 = 8 * v - a0;
        if (i0 + 7 >= 0 && i0 < n) {
            const int4 w = raw_al[v];
            int sx[10];      // sx[e + 1] = sample i0 + e; sx[0], sx[9]: the neighbours
            const int ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) sx[e + 1] = (int)(int16_t)((uint32_t)ww[e >> 1] >> (16 * (e & 1)));
            sx[0] = i0 - 1 >= 0 ? (int)raw[i0 - 1] : 0;
            sx[9] = i0 + 8 < n ? (int)raw[i0 + 8] : 0;
            const bool whole = i0 >= 0 && i0 + 7 < n;
            int o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = i0 + e;
                // zero padding at the ends of the read (scipy medfilt): what lies beyond belongs to another read
                const int a = idx - 1 >= 0 ? sx[e] : 0, c = sx[e + 1], b = idx + 1 < n ? sx[e + 2] : 0;
                o[e] = med3<int>(a, c, b);
            }
            if (whole) {
                int4 q;
                q.x = (o[0] & 0xffff) | (o[1] << 16); q.y = (o[2] & 0xffff) | (o[3] << 16);
                q.z = (o[4] & 0xffff) | (o[5] << 16); q.w = (o[6] & 0xffff) | (o[7] << 16);
                flt_al[v] = q;
#pragma unroll
                for (int e = 0; e < 8; ++e) { m[k][e] = (int16_t)o[e]; mn = o[e] < mn ? o[e] : mn; mx = o[e] > mx ? o[e] : mx; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int idx = i0 + e;
                    m[k][e] = (int16_t)o[e];
                    if (idx >= 0 && idx < n) { flt[idx] = (int16_t)o[e]; mn = o[e] < mn ? o[e] : mn; mx = o[e] > mx ? o[e] : mx; }
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) m[k][e] = 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(mn, o, 64); mn = x < mn ? x : mn; const int y = __shfl_xor(mx, o, 64); mx = y > mx ? y : mx; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin, mn); atomicMax(&tmax, mx); }
    __syncthreads();
    const int lo = tmin;
    if (threadIdx.x == 0 && range_all && tmax >= tmin) {
        atomicMax(&range_all[(size_t)blockIdx.y * range_stride], (uint32_t)(tmax + 32768));
        atomicMax(&range_all[(size_t)blockIdx.y * range_stride + 1], (uint32_t)(65535 - (tmin + 32768)));
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i0 = 8 * (v_base + k * 256 + threadIdx.x) - a0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = i0 + e;
            if (idx >= 0 && idx < n) {
                const int v = m[k][e], w = v - lo;
                if (w < HIST_WIN) atomicAdd(&bins[w], 1u);
                else atomicAdd(&hist[v + 32768], 1u);
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < HIST_WIN; b += 256) { const uint32_t c = bins[b]; if (c) atomicAdd(&hist[lo + b + 32768], c); }
}

// numpy's _lerp (np.percentile, method 'linear')
static __device__ __forceinline__ double np_lerp(double a, double b, double t)
{
    const double d = b - a;
    return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}

// One workgroup per read: order statistics of a histogram, then the constants of the 'minmax'
// normalisation (STRique.py:152-160) and, for the filtered signal, median and MAD.
__global__ void __launch_bounds__(256)
hist_stats_kernel(const uint32_t* __restrict__ hist_all, int nbins, int bias, ReadCond* __restrict__ rc_all,
                  PoreStats ps, int which, float* __restrict__ level_val_all,
                  const uint32_t* __restrict__ range_all, int range_stride)
{
    __shared__ uint32_t staged[HSTAT_LDS_BINS];
    __shared__ uint32_t excl[257];
    __shared__ long long ranks[8];
    __shared__ int vals[8];
    __shared__ double red[256];
    __shared__ unsigned long long cnt_lo, cnt_hi;
    const int t = threadIdx.x;
    ReadCond& rc = rc_all[blockIdx.x];
    const uint32_t* hist = hist_all + (size_t)blockIdx.x * nbins;
    const int n = rc.n;
    if (n <= 0) { if (t == 0) rc.status = COND_DEGENERATE; return; }
    // The passes below walk the histogram in bin order, one contiguous chunk per thread.  When the
    // occupied bin range is known (hist16_kernel) and small -- a nanopore read spans a few thousand DAC
    // values -- it is staged in LDS with one coalesced sweep and everything else runs on the copy.
    if (range_all) {
        const int hi_bin = (int)range_all[(size_t)blockIdx.x * range_stride];
        const int lo_bin = 65535 - (int)range_all[(size_t)blockIdx.x * range_stride + 1];
        const int nb = hi_bin - lo_bin + 1;
        if (nb >= 1 && nb <= HSTAT_LDS_BINS) {
            for (int b = t; b < nb; b += 256) staged[b] = hist[lo_bin + b];
            __syncthreads();
            hist = staged; nbins = nb; bias += lo_bin;
        }
    }
    const int bpt = (nbins + 255) / 256, b0 = t * bpt;
    auto H = [&](int b) -> uint32_t { return b < nbins ? hist[b] : 0u; };
    uint32_t mysum = 0;
    for (int b = 0; b < bpt; ++b) mysum += H(b0 + b);
    excl[t + 1] = mysum;
    if (t == 0) excl[0] = 0;
    __syncthreads();
    if (t == 0) for (int i = 1; i <= 256; ++i) excl[i] += excl[i - 1];
    __syncthreads();
    const uint32_t mybase = excl[t];

    auto select_ranks = [&](int k) {     // ranks[0..k) -> vals[0..k)  (value = bin + bias)
        __syncthreads();
        if (mysum) {
            uint32_t c = mybase;
            for (int b = 0; b < bpt; ++b) {
                const uint32_t h = H(b0 + b);
                if (h) for (int i = 0; i < k; ++i) if (ranks[i] >= (long long)c && ranks[i] < (long long)c + h) vals[i] = b0 + b + bias;
                c += h;
            }
        }
        __syncthreads();
    };
    // --- median ranks and the two percentiles' neighbours
    double g_lo = 0, g_hi = 0;
    {
        const double vi_lo = (double)(n - 1) * 0.01, vi_hi = (double)(n - 1) * 0.99;
        long long p_lo = (long long)floor(vi_lo), p_hi = (long long)floor(vi_hi);
        long long n_lo = p_lo + 1, n_hi = p_hi + 1;
        if (vi_lo >= (double)(n - 1)) { p_lo = n_lo = -1; }
        if (vi_hi >= (double)(n - 1)) { p_hi = n_hi = -1; }
        g_lo = vi_lo - (double)p_lo; g_hi = vi_hi - (double)p_hi;
        if (t == 0) {
            ranks[0] = (n - 1) / 2; ranks[1] = n / 2;
            ranks[2] = p_lo < 0 ? n - 1 : p_lo; ranks[3] = n_lo < 0 ? n - 1 : n_lo;
            ranks[4] = p_hi < 0 ? n - 1 : p_hi; ranks[5] = n_hi < 0 ? n - 1 : n_hi;
        }
    }
    select_ranks(6);
    const double med = ((double)vals[0] + (double)vals[1]) / 2;
    const double q_lo = np_lerp((double)vals[2], (double)vals[3], g_lo);
    const double q_hi = np_lerp((double)vals[4], (double)vals[5], g_hi);
    // --- sizes of the two tails (strict comparisons, STRique.py:155-156)
    if (t == 0) { cnt_lo = 0; cnt_hi = 0; }
    __syncthreads();
    {
        unsigned long long cl = 0, ch = 0;
        for (int b = 0; b < bpt; ++b) {
            const double v = (double)(b0 + b + bias);
            const uint32_t h = H(b0 + b);
            if (v < q_lo) cl += h;
            if (v > q_hi) ch += h;
        }
        if (cl) atomicAdd(&cnt_lo, cl);
        if (ch) atomicAdd(&cnt_hi, ch);
    }
    __syncthreads();
    const long long c_lo = (long long)cnt_lo, c_hi = (long long)cnt_hi;
    bool degenerate = (c_lo == 0 || c_hi == 0);
    // an empty tail: numpy's median of nothing is NaN, and so are the constants of the map and every sample it is applied to
    double c1 = __builtin_nan(""), h1 = __builtin_nan("");
    if (!degenerate) {
        if (t == 0) {
            ranks[0] = (c_lo - 1) / 2; ranks[1] = c_lo / 2;
            ranks[2] = n - c_hi + (c_hi - 1) / 2; ranks[3] = n - c_hi + c_hi / 2;
        }
        select_ranks(4);
        const double m_lo = ((double)vals[0] + (double)vals[1]) / 2;
        const double m_hi = ((double)vals[2] + (double)vals[3]) / 2;
        c1 = m_lo + (m_hi - m_lo) / 2;
        h1 = (m_hi - m_lo) / 2;
        if (!(h1 > 0.0)) degenerate = true;
    }
    const double h2 = (ps.M_hi - ps.M_lo) / 2, c2 = ps.M_lo + (ps.M_hi - ps.M_lo) / 2;
    if (which == 0) {
        // MAD = mean |x - median|: every term is a multiple of 0.5, the sum is exact in any order
        double s = 0;
        for (int b = 0; b < bpt; ++b) {
            const uint32_t h = H(b0 + b);
            if (h) s += (double)h * fabs((double)(b0 + b + bias) - med);
        }
        red[t] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (t < w) red[t] += red[t + w]; __syncthreads(); }
        const double mad = red[0] / (double)n;
        if (!(mad > 0.0)) degenerate = true;
        if (t == 0) { rc.med = med; rc.mad = mad; rc.f_c1 = c1; rc.f_h1 = h1; rc.h2 = h2; rc.c2 = c2; rc.status = degenerate ? COND_DEGENERATE : COND_OK; }
    } else if (which == 1) {
        if (t == 0) { rc.m_c1 = c1; rc.m_h1 = h1; if (degenerate) rc.status = COND_DEGENERATE; }
        // value of every 8-bit level after normalize2model + clip, rounded to float32 like the
        // pybind11 list caster does on the way into align_overlap (src/pyalign.cpp:59-61)
        double v = ((double)t - c1) / h1;
        v = v * h2 + c2;
        v = v < ps.clip_lo ? ps.clip_lo : v;
        v = v > ps.clip_hi ? ps.clip_hi : v;
        // empty tails: the reference's medians are NaN and so is every value it hands to align_overlap, which
        // then scores every cell dist_min (src/align_raw.h:100 -- the comparison with NaN is false)
        level_val_all[(size_t)blockIdx.x * 256 + t] = degenerate ? __builtin_nanf("") : (float)v;
    } else {
        if (t == 0) { rc.r_c1 = c1; rc.r_h1 = h1; if (degenerate) rc.status = COND_DEGENERATE; }
    }
}

// z-score -> 8 bit -> grey opening and closing with a 1x8 footprint, one LDS tile per workgroup.
// scikit-image 0.14 pads the even footprint to 9 with a zero on the left in the first stage of
// opening/closing and on the right in the second, so the 1-D windows are
//   opening = max_{-4..+3}( min_{-3..+4} ),  closing = min_{-4..+3}( max_{-3..+4} ),
// each stage reading its input with scipy.ndimage 'reflect' borders.
template <class T>
__global__ void __launch_bounds__(256)
quant_morph_kernel(const T* __restrict__ flt_all, uint8_t* __restrict__ levels_all, const ReadCond* __restrict__ rc_all,
                   uint32_t* __restrict__ hist8)
{
    __shared__ __attribute__((aligned(16))) uint8_t bufA[COND_TILE + 2 * COND_HALO + 16], bufB[COND_TILE + 2 * COND_HALO + 16];
    __shared__ uint32_t h8[256];
    const ReadCond rc = rc_all[blockIdx.y];
    const int n = rc.n;
    const T* flt = flt_all + rc.off;
    // int16: tiles start `a0` samples in front of the read, on a 16-byte boundary of the filtered signal, so that an interior
    // tile reads it with aligned 16-byte loads (a read starts anywhere inside a line: the reads of a batch sit back to back)
    const int a0 = sizeof(T) == 2 ? (int)((reinterpret_cast<uintptr_t>(flt) >> 1) & 7) : 0;
    const int t0 = blockIdx.x * COND_TILE - a0;
    if (t0 >= n || !(rc.mad > 0.0)) return;      // constant signal: z-score undefined, no levels (hist_stats reports NaN level values)
    const int lo = t0 - COND_HALO, span = COND_TILE + 2 * COND_HALO;
    h8[threadIdx.x] = 0;
    uint8_t* levels = levels_all + rc.off;
    if (lo >= 0 && lo + span <= n) {
        // ---- interior tile (all but the first and last of a read): no border logic, eight consecutive
        // samples per thread, the sliding 8-wide min / max of a stage from 16 input bytes in registers
        // (suffix extrema of bytes 0..7, prefix extrema of bytes 8..15, one combine per output)
        constexpr int NG = (COND_TILE + 2 * COND_HALO) / 8;      // groups of 8 samples
        auto quantv = [&](double f) -> uint32_t {
            double z = (f - rc.med) / rc.mad;
            z = z * 24.0 + 127.0;
            z = z < 0.0 ? 0.0 : z;
            z = z > 255.0 ? 255.0 : z;
            return (uint32_t)(uint8_t)z;
        };
        for (int g = threadIdx.x; g < NG; g += 256) {
            const int i = lo + 8 * g;
            double f[8];
            if constexpr (sizeof(T) == 2) {
                const int4 w = *reinterpret_cast<const int4*>(flt + i);          // aligned: lo + a0 is a multiple of 8
                const int ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (double)(int16_t)((uint32_t)ww[e >> 1] >> (16 * (e & 1)));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (double)flt[i + e];
            }
            const uint32_t w0 = quantv(f[0]) | (quantv(f[1]) << 8) | (quantv(f[2]) << 16) | (quantv(f[3]) << 24);
            const uint32_t w1 = quantv(f[4]) | (quantv(f[5]) << 8) | (quantv(f[6]) << 16) | (quantv(f[7]) << 24);
            reinterpret_cast<uint32_t*>(bufA)[2 * g] = w0; reinterpret_cast<uint32_t*>(bufA)[2 * g + 1] = w1;
        }
        __syncthreads();
        // window of output j of the group: input bytes [j + SH, j + SH + 7] of the 16 bytes starting at 8g - 4
        auto stage8 = [&](const uint8_t* src, uint8_t* dst, auto sh_c, auto min_c) {
            constexpr int SH = decltype(sh_c)::value; constexpr bool MIN = decltype(min_c)::value;
            auto op = [](int a, int b) { return MIN ? (a < b ? a : b) : (a > b ? a : b); };
            for (int g = threadIdx.x; g < NG; g += 256) {
                const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
                const int d0 = 2 * g - 1;                         // dword holding byte 8g - 4 (g = 0: never consumed)
                uint32_t w[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) w[x] = s32[d0 + x < 0 ? 0 : d0 + x];
                int b[16];
#pragma unroll
                for (int x = 0; x < 16; ++x) b[x] = (int)((w[x >> 2] >> (8 * (x & 3))) & 255u);
                int S[8], P[16];
                S[7] = b[7];
#pragma unroll
                for (int x = 6; x >= 0; --x) S[x] = op(b[x], S[x + 1]);
                P[8] = b[8];
#pragma unroll
                for (int x = 9; x < 16; ++x) P[x] = op(b[x], P[x - 1]);
                uint32_t o[2] = {0u, 0u};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int a = j + SH;                         // first byte of the window, 0..8
                    const int v = a == 0 ? S[0] : (a == 8 ? P[15] : op(S[a], P[a + 7]));
                    o[j >> 2] |= (uint32_t)v << (8 * (j & 3));
                }
                reinterpret_cast<uint32_t*>(dst)[2 * g] = o[0]; reinterpret_cast<uint32_t*>(dst)[2 * g + 1] = o[1];
            }
            __syncthreads();
        };
        // offsets -3..+4 -> bytes [j + 1, j + 8] of the 16; offsets -4..+3 -> bytes [j, j + 7]
        stage8(bufA, bufB, std::integral_constant<int, 1>{}, std::true_type{});       // erosion   (opening, first stage)
        stage8(bufB, bufA, std::integral_constant<int, 0>{}, std::false_type{});      // dilation  (opening, second stage)
        stage8(bufA, bufB, std::integral_constant<int, 1>{}, std::false_type{});      // dilation  (closing, first stage)
        stage8(bufB, bufA, std::integral_constant<int, 0>{}, std::true_type{});       // erosion   (closing, second stage)
        if ((reinterpret_cast<uintptr_t>(levels + t0) & 7) == 0) {
            // eight levels per thread, one 8-byte store (the caller gives the level buffer the phase that makes this hold)
            static_assert(COND_TILE == 8 * 256, "one group of eight per thread");
            const uint2 q = *reinterpret_cast<const uint2*>(bufA + COND_HALO + 8 * threadIdx.x);
            *reinterpret_cast<uint2*>(levels + t0 + 8 * threadIdx.x) = q;
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&h8[((e < 4 ? q.x : q.y) >> (8 * (e & 3))) & 255u], 1u);
        } else {
            for (int x = threadIdx.x; x < COND_TILE; x += 256) {
                const uint8_t v = bufA[x + COND_HALO]; levels[t0 + x] = v; atomicAdd(&h8[v], 1u);
            }
        }
        __syncthreads();
        if (h8[threadIdx.x]) atomicAdd(&hist8[(size_t)blockIdx.y * 256 + threadIdx.x], h8[threadIdx.x]);
        return;
    }
    // stage 0: quantised signal at the real positions of the extended tile
    for (int x = threadIdx.x; x < span; x += 256) {
        const int i = lo + x;
        uint8_t q = 0;
        if (i >= 0 && i < n) {
            double z = ((double)flt[i] - rc.med) / rc.mad;
            z = z * 24.0 + 127.0;
            z = z < 0.0 ? 0.0 : z;
            z = z > 255.0 ? 255.0 : z;
            q = (uint8_t)z;                       // astype(uint8): truncation
        }
        bufA[x] = q;
    }
    __syncthreads();
    // a stage reads position p of the previous stage; positions outside [0, n) are reflections of it
    auto at = [&](const uint8_t* buf, int p) -> int {
        if (p < 0 || p >= n) p = reflect_idx(p, n);
        int x = p - lo;
        x = x < 0 ? 0 : (x >= span ? span - 1 : x);     // only reached for positions no result depends on
        return buf[x];
    };
    auto stage = [&](const uint8_t* src, uint8_t* dst, int k0, int k1, bool take_min) {
        for (int x = threadIdx.x; x < span; x += 256) {
            const int i = lo + x;
            int v = 0;
            if (i >= 0 && i < n) {
                v = at(src, i + k0);
                for (int k = k0 + 1; k <= k1; ++k) { const int w = at(src, i + k); v = take_min ? (w < v ? w : v) : (w > v ? w : v); }
            }
            dst[x] = (uint8_t)v;
        }
        __syncthreads();
    };
    stage(bufA, bufB, -3, 4, true);      // erosion   (opening, first stage)
    stage(bufB, bufA, -4, 3, false);     // dilation  (opening, second stage)
    stage(bufA, bufB, -3, 4, false);     // dilation  (closing, first stage)
    stage(bufB, bufA, -4, 3, true);      // erosion   (closing, second stage)
    for (int x = threadIdx.x; x < COND_TILE; x += 256) {
        const int i = t0 + x;
        if (i >= 0 && i < n) { const uint8_t v = bufA[x + COND_HALO]; levels[i] = v; atomicAdd(&h8[v], 1u); }
    }
    __syncthreads();
    if (h8[threadIdx.x]) atomicAdd(&hist8[(size_t)blockIdx.y * 256 + threadIdx.x], h8[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------------------
// float64 reads (what the reference's unit tests feed, scripts/STRique_test.py): no histogram is exact for them, so the
// order statistics come from a radix selection over the samples themselves and the MAD from numpy's own summation tree.
//   np.median / np.percentile(x, [1, 99]) ('linear', numpy's _lerp) / the medians of the two tails (STRique.py:152-160):
//     ten order statistics in two rounds (the tails' sizes are known after the first), each round eight passes of a
//     most-significant-byte-first radix selection over order-preserving 64-bit keys, all ranks of a round in the same pass;
//   np.mean(|x - median|) (STRique.py:142-143): numpy reduces a contiguous float64 array in chunks of 8192 elements (the
//     ufunc buffer), each chunk by pairwise summation -- blocks of at most 128 elements with eight running sums, blocks
//     combined by recursive halving (numpy/_core/src/umath/loops_utils.h.src) -- and adds the chunk sums in order.  A full
//     chunk is 64 blocks of 128: one block per lane, combined by a butterfly (a + b is commutative, the tree is the same);
//     the last, partial chunk follows the recursion's uneven cuts.
// One workgroup per (read, signal): signal 0 = the median-filtered samples (median, MAD, c1, h1), 1 = the raw samples
// (c1, h1; modification pass only).  -0.0 is taken as +0.0 (equal under every comparison numpy makes).
#define F64S_THREADS 1024
#define F64S_MAX_RANKS 6
#define F64S_MAX_LEAVES 192

static __device__ __forceinline__ uint64_t f64_key(double x)
{
    if (x == 0.0) x = 0.0;
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static __device__ __forceinline__ double f64_from_key(uint64_t k)
{
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// numpy's pairwise sum of |x[i] - med| over one block (len <= 128)
static __device__ double abs_dev_block(const double* __restrict__ x, double med, int len)
{
    if (len < 8) {
        double r = 0.0;
        for (int i = 0; i < len; ++i) r += fabs(x[i] - med);
        return r;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = fabs(x[j] - med);
    int i = 8;
    for (; i < len - (len % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += fabs(x[i + j] - med);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < len; ++i) res += fabs(x[i] - med);
    return res;
}

__global__ void __launch_bounds__(F64S_THREADS)
f64_stats_kernel(const double* __restrict__ flt_all, const double* __restrict__ raw_all, ReadCond* __restrict__ rc_all,
                 double* __restrict__ chunk_sums_all, const int64_t* __restrict__ chunk_first)
{
    __shared__ uint32_t hist[F64S_MAX_RANKS][256];
    __shared__ uint64_t s_prefix[F64S_MAX_RANKS];
    __shared__ long long s_rem[F64S_MAX_RANKS];
    __shared__ int s_src[F64S_MAX_RANKS];
    __shared__ unsigned long long s_cnt[2];
    __shared__ int s_flag;
    __shared__ int leaf_a[F64S_MAX_LEAVES], leaf_len[F64S_MAX_LEAVES], s_leaves;
    __shared__ double leaf_sum[F64S_MAX_LEAVES];
    const int t = threadIdx.x, which = blockIdx.y;
    ReadCond& rc = rc_all[blockIdx.x];
    const int n = rc.n;
    const double* x = (which == 0 ? flt_all : raw_all) + rc.off;
    const double nan = __builtin_nan("");
    auto finish = [&](double med, double mad, double c1, double h1) {
        if (t != 0) return;
        if (which == 0) {
            rc.med = med; rc.mad = mad; rc.f_c1 = c1; rc.f_h1 = h1;
            const bool ok = isfinite(med) && mad > 0.0 && isfinite(c1) && h1 > 0.0 && isfinite(h1);
            rc.status = ok ? COND_OK : COND_DEGENERATE;
        } else { rc.r_c1 = c1; rc.r_h1 = h1; }
    };
    if (n <= 0) { finish(nan, nan, nan, nan); return; }
    // NaN in, NaN out (np.median / np.percentile)
    if (t == 0) s_flag = 0;
    __syncthreads();
    { int bad = 0;
      for (int i = t; i < n; i += F64S_THREADS) { const double v = x[i]; bad |= !(v == v); }
      if (bad) s_flag = 1; }
    __syncthreads();
    if (s_flag) { finish(nan, nan, nan, nan); return; }

    // ranks[0..k) -> s_prefix[0..k): keys of the order statistics
    auto select = [&](int k) {
        if (t < k) { s_prefix[t] = 0; s_src[t] = 0; }
        __syncthreads();
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            for (int b = t; b < k * 256; b += F64S_THREADS) hist[b >> 8][b & 255] = 0;
            __syncthreads();
            uint64_t pre[F64S_MAX_RANKS]; bool own[F64S_MAX_RANKS];
            for (int j = 0; j < k; ++j) { pre[j] = pass == 0 ? 0 : (s_prefix[j] >> (shift + 8)); own[j] = s_src[j] == j; }
            for (int i = t; i < n; i += F64S_THREADS) {
                const uint64_t key = f64_key(x[i]);
                const uint64_t hi = pass == 0 ? 0 : (key >> (shift + 8));
                const int byte = (int)((key >> shift) & 255);
                // the high bytes of a read's samples are all but constant: the lanes that share the first matching lane's byte
                // add their count with one atomic instead of queueing on one LDS address, the others add for themselves
                for (int j = 0; j < k; ++j) {
                    if (!own[j]) continue;
                    const bool m = hi == pre[j];
                    const unsigned long long act = __ballot(m);
                    if (!act) continue;
                    const int leader = __ffsll((long long)act) - 1;
                    const int lb = __shfl(byte, leader, 64);
                    const unsigned long long same = __ballot(m && byte == lb);
                    if ((t & 63) == leader) atomicAdd(&hist[j][lb], (uint32_t)__popcll(same));
                    else if (m && byte != lb) atomicAdd(&hist[j][byte], 1u);
                }
            }
            __syncthreads();
            if (t < k) {
                const uint32_t* h = hist[s_src[t]];
                long long rem = s_rem[t];
                int b = 0;
                for (; b < 255; ++b) { const long long c = h[b]; if (rem < c) break; rem -= c; }
                s_rem[t] = rem; s_prefix[t] |= (uint64_t)b << shift;
            }
            __syncthreads();
            if (t == 0) for (int j = 0; j < k; ++j) { int src = j; for (int q = 0; q < j; ++q) if ((s_prefix[q] >> shift) == (s_prefix[j] >> shift)) { src = s_src[q]; break; } s_src[j] = src; }
            __syncthreads();
        }
    };
    const double last = (double)(n - 1);
    const double vi_lo = last * 0.01, vi_hi = last * 0.99;
    long long p_lo = (long long)floor(vi_lo), p_hi = (long long)floor(vi_hi);
    long long x_lo = p_lo + 1, x_hi = p_hi + 1;
    double g_lo, g_hi;
    if (vi_lo >= last) { g_lo = vi_lo + 1.0; p_lo = x_lo = (long long)n - 1; } else g_lo = vi_lo - (double)p_lo;
    if (vi_hi >= last) { g_hi = vi_hi + 1.0; p_hi = x_hi = (long long)n - 1; } else g_hi = vi_hi - (double)p_hi;
    if (t == 0) { s_rem[0] = p_lo; s_rem[1] = x_lo; s_rem[2] = p_hi; s_rem[3] = x_hi; s_rem[4] = (n - 1) / 2; s_rem[5] = n / 2; }
    select(which == 0 ? 6 : 4);
    const double q_lo = np_lerp(f64_from_key(s_prefix[0]), f64_from_key(s_prefix[1]), g_lo);
    const double q_hi = np_lerp(f64_from_key(s_prefix[2]), f64_from_key(s_prefix[3]), g_hi);
    const double med = which == 0 ? (f64_from_key(s_prefix[4]) + f64_from_key(s_prefix[5])) / 2 : nan;
    __syncthreads();
    // sizes of the two tails (strict comparisons, STRique.py:155-156)
    if (t == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    { unsigned long long cl = 0, ch = 0;
      for (int i = t; i < n; i += F64S_THREADS) { const double v = x[i]; cl += v < q_lo; ch += v > q_hi; }
      for (int o = 32; o > 0; o >>= 1) { cl += __shfl_xor(cl, o, 64); ch += __shfl_xor(ch, o, 64); }
      if ((t & 63) == 0) { if (cl) atomicAdd(&s_cnt[0], cl); if (ch) atomicAdd(&s_cnt[1], ch); } }
    __syncthreads();
    const long long c_lo = (long long)s_cnt[0], c_hi = (long long)s_cnt[1];
    double c1 = nan, h1 = nan;
    if (c_lo > 0 && c_hi > 0) {
        if (t == 0) { s_rem[0] = (c_lo - 1) / 2; s_rem[1] = c_lo / 2; s_rem[2] = n - c_hi + (c_hi - 1) / 2; s_rem[3] = n - c_hi + c_hi / 2; }
        select(4);
        const double m_lo = (f64_from_key(s_prefix[0]) + f64_from_key(s_prefix[1])) / 2;
        const double m_hi = (f64_from_key(s_prefix[2]) + f64_from_key(s_prefix[3])) / 2;
        c1 = m_lo + (m_hi - m_lo) / 2;
        h1 = (m_hi - m_lo) / 2;
    }
    if (which != 0) { finish(nan, nan, c1, h1); return; }
    // ---- MAD = np.mean(|x - med|)
    double* chunk_sums = chunk_sums_all + chunk_first[blockIdx.x];
    const int n_chunks = (n + 8191) / 8192, full = n / 8192;
    const int wave = t >> 6, lane = t & 63;
    for (int ch = wave; ch < full; ch += F64S_THREADS / 64) {
        double r = abs_dev_block(x + (size_t)ch * 8192 + lane * 128, med, 128);
        for (int o = 1; o < 64; o <<= 1) r += __shfl_xor(r, o, 64);
        if (lane == 0) chunk_sums[ch] = r;
    }
    if (n_chunks > full) {
        // the partial chunk: the recursion's cuts (n2 = len / 2 rounded down to a multiple of 8), blocks left to right
        const int base = full * 8192, len = n - base;
        if (t == 0) {
            int sa[16], sl[16], sp = 0, nl = 0;
            sa[0] = 0; sl[0] = len; sp = 1;
            while (sp > 0) {
                const int a = sa[sp - 1], l = sl[sp - 1]; --sp;
                if (l <= 128) { leaf_a[nl] = a; leaf_len[nl] = l; ++nl; continue; }
                int n2 = l / 2; n2 -= n2 % 8;
                sa[sp] = a + n2; sl[sp] = l - n2; ++sp;      // right half, taken after ...
                sa[sp] = a; sl[sp] = n2; ++sp;               // ... the left one
            }
            s_leaves = nl;
        }
        __syncthreads();
        if (t < s_leaves) leaf_sum[t] = abs_dev_block(x + base + leaf_a[t], med, leaf_len[t]);
        __syncthreads();
        if (t == 0) {
            int sl[16], stage[16], sp = 0, li = 0; double left[16], ret = 0.0;
            sl[0] = len; stage[0] = 0; sp = 1;
            while (sp > 0) {
                const int l = sl[sp - 1];
                if (l <= 128) { ret = leaf_sum[li++]; --sp; continue; }
                int n2 = l / 2; n2 -= n2 % 8;
                if (stage[sp - 1] == 0) { stage[sp - 1] = 1; sl[sp] = n2; stage[sp] = 0; ++sp; }
                else if (stage[sp - 1] == 1) { left[sp - 1] = ret; stage[sp - 1] = 2; sl[sp] = l - n2; stage[sp] = 0; ++sp; }
                else { ret = left[sp - 1] + ret; --sp; }
            }
            chunk_sums[full] = ret;
        }
    }
    __threadfence();
    __syncthreads();
    if (t == 0) {
        double res = 0.0;
        for (int ch = 0; ch < n_chunks; ++ch) res += chunk_sums[ch];
        finish(med, res / (double)n, c1, h1);
    }
}

static inline dim3 tile_grid(int max_n, int n_reads) { return dim3((max_n + 7 + COND_TILE - 1) / COND_TILE, n_reads); }      // + 7: tiles may start up to seven samples in front of a read

int launch_medfilt_hist_i16(hipStream_t s, const int16_t* raw, int16_t* flt, const ReadCond* rc, int n_reads, int max_n,
                            uint32_t* hist_flt, uint32_t* hist_raw, uint32_t* range4)
{
    if (n_reads <= 0 || max_n <= 0) return 0;
    const dim3 hgrid((max_n + HIST_TILE - 1) / HIST_TILE, n_reads);
    // 16 bytes per lane when the raw and the filtered buffer have the same alignment phase (the caller arranges that);
    // its tiles start up to seven samples in front of a read: one more tile covers the longest read
    const bool same_phase = ((reinterpret_cast<uintptr_t>(raw) ^ reinterpret_cast<uintptr_t>(flt)) & 15) == 0 && !strq::opt("STRQ_COND_SCALAR");
    const dim3 vgrid((max_n + 7 + HIST_TILE - 1) / HIST_TILE, n_reads);
    if (hist_flt && same_phase) hipLaunchKernelGGL(medfilt_hist16_vec_kernel, vgrid, dim3(256), 0, s, raw, flt, rc, hist_flt, range4, 4);
    else if (hist_flt) hipLaunchKernelGGL(medfilt_hist16_kernel, hgrid, dim3(256), 0, s, raw, flt, rc, hist_flt, range4, 4);
    else hipLaunchKernelGGL((medfilt_kernel<int16_t>), tile_grid(max_n, n_reads), dim3(256), 0, s, raw, flt, rc, hist_flt, hist_raw);
    if (hist_raw) hipLaunchKernelGGL(hist16_kernel, hgrid, dim3(256), 0, s, raw, rc, hist_raw, range4 ? range4 + 2 : nullptr, 4);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_medfilt_f64(hipStream_t s, const double* raw, double* flt, const ReadCond* rc, int n_reads, int max_n)
{
    if (n_reads <= 0 || max_n <= 0) return 0;
    hipLaunchKernelGGL((medfilt_kernel<double>), tile_grid(max_n, n_reads), dim3(256), 0, s, raw, flt, rc, (uint32_t*)nullptr, (uint32_t*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_f64_stats(hipStream_t s, const double* flt, const double* raw, ReadCond* rc, int n_reads, double* chunk_sums, const int64_t* chunk_first)
{
    if (n_reads <= 0) return 0;
    hipLaunchKernelGGL(f64_stats_kernel, dim3(n_reads, raw ? 2 : 1), dim3(F64S_THREADS), 0, s, flt, raw, rc, chunk_sums, chunk_first);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_hist_stats(hipStream_t s, const uint32_t* hist, int nbins, int bias, ReadCond* rc, int n_reads, PoreStats ps,
                      int which, float* level_val, const uint32_t* range, int range_stride)
{
    if (n_reads <= 0) return 0;
    hipLaunchKernelGGL(hist_stats_kernel, dim3(n_reads), dim3(256), 0, s, hist, nbins, bias, rc, ps, which, level_val, range, range_stride);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_quant_morph_i16(hipStream_t s, const int16_t* flt, uint8_t* levels, const ReadCond* rc, int n_reads, int max_n, uint32_t* hist8)
{
    if (n_reads <= 0 || max_n <= 0) return 0;
    hipLaunchKernelGGL((quant_morph_kernel<int16_t>), tile_grid(max_n, n_reads), dim3(256), 0, s, flt, levels, rc, hist8);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_quant_morph_f64(hipStream_t s, const double* flt, uint8_t* levels, const ReadCond* rc, int n_reads, int max_n, uint32_t* hist8)
{
    if (n_reads <= 0 || max_n <= 0) return 0;
    hipLaunchKernelGGL((quant_morph_kernel<double>), tile_grid(max_n, n_reads), dim3(256), 0, s, flt, levels, rc, hist8);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq
