// Internal: signal conditioning kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace strq {

// model-side constants of pore_model.normalize2model('minmax') (scripts/STRique.py:154,157-160,178-179)
struct PoreStats {
    double M_lo, M_hi;          // medians of the model means below the 1st / above the 99th percentile
    double clip_lo, clip_hi;    // model_min + 0.5, model_max - 0.5
};

enum { COND_OK = 0, COND_DEGENERATE = 1 };

// per-read conditioning state, device resident
struct ReadCond {
    int64_t off;        // offset of the read in the concatenated sample arrays
    int32_t n;          // samples
    int32_t status;     // COND_*
    double med, mad;    // median / mean absolute deviation of the median-filtered signal
    double f_c1, f_h1;  // minmax map of the filtered signal:   x' = (x - c1) / h1 * h2 + c2
    double m_c1, m_h1;  //            ... of the 8-bit morphology signal
    double r_c1, r_h1;  //            ... of the raw signal (modification pass only)
    double h2, c2;      // model side of the map (same for all three)
};

int launch_medfilt_hist_i16(hipStream_t s, const int16_t* raw, int16_t* flt, const ReadCond* rc, int n_reads,
                            int max_n, uint32_t* hist_flt, uint32_t* hist_raw, uint32_t* range4);      // range4: 4 zeroed words per read (occupied bins of flt / raw)
int launch_medfilt_f64(hipStream_t s, const double* raw, double* flt, const ReadCond* rc, int n_reads, int max_n);
// float64 reads: median, MAD, f_c1, f_h1 and the status of every read from its filtered samples, r_c1, r_h1 from the raw ones when `raw` is
// given; chunk_sums: one double per 8192 samples of a read, read i's at chunk_first[i]
int launch_f64_stats(hipStream_t s, const double* flt, const double* raw, ReadCond* rc, int n_reads, double* chunk_sums, const int64_t* chunk_first);
// which: 0 = filtered int16 histogram (fills med, mad, f_*), 1 = 8-bit histogram (fills m_* and level_val),
//        2 = raw int16 histogram (fills r_*)
int launch_hist_stats(hipStream_t s, const uint32_t* hist, int nbins, int bias, ReadCond* rc, int n_reads,
                      PoreStats ps, int which, float* level_val, const uint32_t* range, int range_stride);   // range: from launch_medfilt_hist_i16, or null
int launch_quant_morph_i16(hipStream_t s, const int16_t* flt, uint8_t* levels, const ReadCond* rc, int n_reads,
                           int max_n, uint32_t* hist8);
int launch_quant_morph_f64(hipStream_t s, const double* flt, uint8_t* levels, const ReadCond* rc, int n_reads,
                           int max_n, uint32_t* hist8);

}  // namespace strq
