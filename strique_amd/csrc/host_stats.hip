// Order statistics of float64 reads, on the host (no device code in this file).  A helper behind strq_host_stats for callers
// that want the six scalars without a device (and the numpy-pinned twin the GPU path is tested against): strq_detect_batch
// itself takes them on the GPU (cond_kernels.hip: f64_stats_kernel) since round 5.
//
// int16 reads -- what a fast5 file holds -- get every statistic of the conditioning from exact histograms on the
// GPU (cond_kernels.hip).  float64 reads (what the reference's own unit tests feed, scripts/STRique_test.py) have no
// histogram; their six scalars per read come from here: median and MAD of the median-filtered signal
// (STRique.py:590-592, 142-143) and the (c1, h1) of the 'minmax' map of the filtered and of the raw signal
// (STRique.py:152-160).  The arithmetic follows numpy's, which is what the reference runs:
//   * np.median: the middle order statistic, or (a + b) / 2 of the two middle ones;
//   * np.percentile(x, [1, 99]) with the default 'linear' method: virtual index (n - 1) * q, numpy's _lerp;
//   * np.mean over a contiguous float64 array: the reduction runs over chunks of 8192 elements (the ufunc buffer
//     size), each summed pairwise -- eight running sums over blocks of at most 128 elements, blocks combined by
//     recursive halving (numpy/_core/src/umath/loops_utils.h.src) -- and the chunk sums are added in order.
// Selection instead of a full sort: only the lowest and highest ~1 % of a read and its middle are ever looked at.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <sched.h>
#include <thread>
#include <vector>
#include "../../include/strique_hip.h"
#include "strq_ctx.h"

namespace strq {

static inline double np_lerp_host(double a, double b, double t)
{
    const double d = b - a;
    return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}

// numpy's pairwise sum of |x[i] - med| over n <= 8192 elements
static double pairwise_abs_dev(const double* x, double med, size_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (size_t i = 0; i < n; ++i) r += std::fabs(x[i] - med);
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = std::fabs(x[j] - med);
        size_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += std::fabs(x[i + j] - med);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += std::fabs(x[i] - med);
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_abs_dev(x, med, n2) + pairwise_abs_dev(x + n2, med, n - n2);
}

static double np_mean_abs_dev(const double* x, double med, size_t n)
{
    double res = 0.0;
    for (size_t i = 0; i < n; i += 8192) res += pairwise_abs_dev(x + i, med, std::min<size_t>(8192, n - i));
    return res / (double)n;
}

// The order statistics one signal needs, out of a scratch copy that is reordered in place.
struct Ranks {
    double med, c1, h1;
};

static Ranks order_stats(double* buf, size_t n, bool want_median)
{
    const double nan = std::numeric_limits<double>::quiet_NaN();
    Ranks out{nan, nan, nan};
    if (n == 0) return out;
    for (size_t i = 0; i < n; ++i) if (!(buf[i] == buf[i])) return out;          // NaN in, NaN out (np.percentile / np.median)
    const double last = (double)(n - 1);
    const double vi_lo = last * 0.01, vi_hi = last * 0.99;
    long long p_lo = (long long)std::floor(vi_lo), p_hi = (long long)std::floor(vi_hi);
    long long x_lo = p_lo + 1, x_hi = p_hi + 1;
    double g_lo, g_hi;
    if (vi_lo >= last) { g_lo = vi_lo + 1.0; p_lo = x_lo = (long long)n - 1; } else g_lo = vi_lo - (double)p_lo;
    if (vi_hi >= last) { g_hi = vi_hi + 1.0; p_hi = x_hi = (long long)n - 1; } else g_hi = vi_hi - (double)p_hi;
    // sorted prefix [0, lo_end) and sorted suffix [hi_begin, n); the middle only partitioned around the median
    size_t lo_end = (size_t)x_lo + 1, hi_begin = (size_t)p_hi;
    const size_t k1 = (n - 1) / 2, k2 = n / 2;
    if (n <= 8192 || lo_end + 2 >= hi_begin || k1 < lo_end || k2 + 1 >= hi_begin) {
        std::sort(buf, buf + n);
    } else {
        std::nth_element(buf, buf + lo_end - 1, buf + n);
        std::sort(buf, buf + lo_end);
        std::nth_element(buf + lo_end, buf + hi_begin, buf + n);
        std::sort(buf + hi_begin, buf + n);
        if (want_median) {
            std::nth_element(buf + lo_end, buf + k1, buf + hi_begin);
            if (k2 != k1) { double* m = std::min_element(buf + k1 + 1, buf + hi_begin); std::swap(*m, buf[k2]); }
        }
    }
    if (want_median) out.med = (buf[k1] + buf[k2]) / 2;
    const double q_lo = np_lerp_host(buf[p_lo], buf[x_lo], g_lo);
    const double q_hi = np_lerp_host(buf[p_hi], buf[x_hi], g_hi);
    // strict comparisons (STRique.py:155-156); every element < q_lo lies in the sorted prefix, every element > q_hi in the suffix
    const size_t c_lo = (size_t)(std::lower_bound(buf, buf + std::min(lo_end, n), q_lo) - buf);
    const size_t first_hi = (size_t)(std::upper_bound(buf + std::min(hi_begin, n), buf + n, q_hi) - buf);
    const size_t c_hi = n - first_hi;
    if (c_lo == 0 || c_hi == 0) return out;
    const double m_lo = (buf[(c_lo - 1) / 2] + buf[c_lo / 2]) / 2;
    const double m_hi = (buf[n - c_hi + (c_hi - 1) / 2] + buf[n - c_hi + c_hi / 2]) / 2;
    out.c1 = m_lo + (m_hi - m_lo) / 2;
    out.h1 = (m_hi - m_lo) / 2;
    return out;
}

static inline double med3(double a, double b, double c)
{
    const double lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}

// out[6] = median, MAD, c1, h1 of medfilt(raw, 3); c1, h1 of raw (0, 1 when not wanted)
void host_read_stats(const double* raw, int64_t n, bool want_raw, double* out)
{
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (n <= 0) { for (int i = 0; i < 6; ++i) out[i] = nan; return; }
    std::vector<double> flt((size_t)n), buf((size_t)n);
    for (int64_t i = 0; i < n; ++i)           // scipy.signal.medfilt pads with zeros
        flt[(size_t)i] = med3(i > 0 ? raw[i - 1] : 0.0, raw[i], i + 1 < n ? raw[i + 1] : 0.0);
    std::copy(flt.begin(), flt.end(), buf.begin());
    const Ranks f = order_stats(buf.data(), (size_t)n, true);
    out[0] = f.med;
    out[1] = f.med == f.med ? np_mean_abs_dev(flt.data(), f.med, (size_t)n) : nan;
    out[2] = f.c1; out[3] = f.h1;
    if (want_raw) {
        std::copy(raw, raw + n, buf.begin());
        const Ranks r = order_stats(buf.data(), (size_t)n, false);
        out[4] = r.c1; out[5] = r.h1;
    } else { out[4] = 0.0; out[5] = 1.0; }
}

void host_stats_batch(const double* signals, const int64_t* offsets, int64_t n_reads, bool want_raw, double* out,
                      const double* const* reads)
{
    // the CPUs this process may run on: a rank pinned to its share of the host (strique_amd.dist.pin_rank_cpus, taskset,
    // a container's cpuset) sizes its pool by that share, not by the machine
    int threads = 0;
    { cpu_set_t set; CPU_ZERO(&set); if (sched_getaffinity(0, sizeof(set), &set) == 0) threads = CPU_COUNT(&set); }
    const int machine = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = machine;
    if (threads < 1) threads = 1;
    // one process per GPU under torchrun and not pinned: every rank still takes only its share of the host's cores
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) { const int w = atoi(e); if (w > 1 && threads == machine) threads = threads / w > 0 ? threads / w : 1; }
    if (threads > 64) threads = 64;          // 9 ms per 375 k-sample read and core: 64 threads ~ 7 k reads/s of float64 input
    if (const char* e = strq::opt("STRQ_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) threads = v; }
    if ((int64_t)threads > n_reads) threads = (int)n_reads;
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n_reads) break;
            host_read_stats(reads ? reads[i] : signals + offsets[i], offsets[i + 1] - offsets[i], want_raw, out + 6 * i);
        }
    };
    if (threads <= 1) { work(); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
}

}  // namespace strq

extern "C" int strq_host_stats(const double* signals, const int64_t* offsets, int64_t n_reads, int32_t want_raw, double* out)
{
    if (n_reads < 0 || (n_reads > 0 && (!signals || !offsets || !out))) return STRQ_ERR_ARG;
    for (int64_t i = 0; i < n_reads; ++i) if (offsets[i + 1] < offsets[i]) return STRQ_ERR_ARG;
    strq::host_stats_batch(signals, offsets, n_reads, want_raw != 0, out, nullptr);
    return STRQ_OK;
}
