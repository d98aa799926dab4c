// Upper-bound screen of the flank alignment (not part of the C ABI): see screen_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "align_kernels.h"

#define STRQ_SCREEN_R 14                 // rows per lane of the screen (flanks up to 64 x 14 = 896 rows: STRique's are 870)
#define STRQ_SCREEN_S 6                  // samples per k-mer class
#define STRQ_SCREEN_SEG 4                // pieces (waves) per read
#define STRQ_SCREEN_CHUNK_COLS 128       // columns per reported chunk (64 steps x 2 columns)
#define STRQ_SCREEN_MAX_WINDOWS 4

namespace strq {

// integer frame of the screen (units of 1 / sc of a score)
struct ScreenParams {
    int32_t sc;          // scale: a power of two
    int32_t hh, v;       // -ext_h * sc, -ext_v * sc (exact integers)
    int32_t cadd;        // hh + v: added to every table entry (the two potentials absorb the gap scores)
    int32_t slack;       // float32 rounding slack of the exact DP, scaled (32 * sc)
    int32_t merge_gap;   // candidate chunks closer than this many columns share a window
};

// one piece (wave) of one alignment
struct ScreenTask {
    const uint8_t* levels;       // first level of the piece (column j of the piece <-> levels[j - 1])
    const float* table;          // float32 score table of the alignment (AlignTask::table)
    const int32_t* band_lo;      // its band descriptors
    int32_t* out;                // chunk maxima of this piece: max over the chunk's columns of S[m][j] * sc + m * v (upper bounds)
    int32_t tsize;
    int32_t n, m, k, col_off;    // columns of the piece; flank rows / classes; read column of the piece's column 0
    int32_t n_chunks;
};

// per alignment: the columns the exact DP has to look at
struct ScreenWindows {
    int32_t n_win;               // 0: the screen does not prune this alignment (run the whole read)
    int32_t lo[STRQ_SCREEN_MAX_WINDOWS], hi[STRQ_SCREEN_MAX_WINDOWS];      // read columns, 1-based, inclusive
    float lower_bound;           // the best score of the alignment is at least this
    float upper_bound;           // ... and at most this
    int32_t n_cand;              // candidate chunks
};

// 0 when the parameters allow no screen; else fills sp (max_n: the longest read of the launch).  screen_flank_ok: a flank
// of m rows fits the screen's one strip
int screen_plan(const AlignParams& p, int samples, int max_n, ScreenParams* sp);
static inline bool screen_flank_ok(int m) { return m >= 2 * STRQ_SCREEN_R && m <= 64 * STRQ_SCREEN_R; }
size_t screen_lds_bytes(int tsize);
int launch_screen(hipStream_t stream, const ScreenTask* tasks, int n_groups, int* queue, const ScreenParams& sp,
                  size_t lds_bytes, int tables_per_cu, int n_cu);
// windows of alignment g from the chunk maxima of its pieces; bound_scaled[g]: the score (scaled) above which
// the cold-started pieces of the alignment are exact
int launch_screen_windows(hipStream_t stream, const ScreenTask* tasks, int n_groups, const ScreenParams& sp,
                          const int32_t* bound_scaled, ScreenWindows* out);

}  // namespace strq
