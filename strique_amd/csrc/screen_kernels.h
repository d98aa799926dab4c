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
// the coarse screen (align_screen2 / 3 / 6_kernel): 2, 3 or 6 flank rows per DP row, both flanks of a read in the two halves of one wave
#define STRQ_SCREEN2_CPL 5               // classes per lane: 30 flank rows, no class straddles a lane
#define STRQ_SCREEN2_LPF 29              // lanes per flank: 29 x 5 = 145 classes = 870 rows (STRique's flanks)
#define STRQ_SCREEN2_LANE_B 32           // first lane of the second flank

namespace strq {

// integer frame of the screen (units of 1 / sc of a score)
struct ScreenParams {
    int32_t sc;          // scale: a power of two
    int32_t hh, v;       // -ext_h * sc, -ext_v * sc (exact integers)
    int32_t cadd;        // hh + v: added to every table entry (the two potentials absorb the gap scores)
    int32_t slack;       // float32 rounding slack of the exact DP, scaled (32 * sc)
    int32_t merge_gap;   // candidate chunks closer than this many columns share a window
    int32_t max_cand;    // coarse screen: at most this many candidate chunks per alignment in the first look (0: no limit)
    int32_t margin;      // coarse screen: how far below the best chunk value a chunk is still a candidate (scaled); 0: the fine rule (m + 2 slack)
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
    int32_t lane_last;           // lane whose last register holds the flank's last row ((m - 1) / R for the fine screen)
};

// one piece (wave) of one READ for the coarse screen: both flank alignments of the read
struct Screen2Task {
    const uint8_t* levels;
    const float* table[2];       // float32 score tables of the prefix / suffix alignment
    const int32_t* band_lo[2];
    int32_t* out[2];             // chunk maxima of this piece, per flank
    int32_t tsize[2], k[2];
    int32_t n, col_off, n_chunks;
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
// coarse screen: scores of `merge` (2, 3 or 6) flank rows at once, both flanks of a read per wave.  screen2_plan like screen_plan (entries are
// merge times as large, so the scale is smaller); screen2_flank_ok: the flank fits 29 lanes of 5 classes
int screen2_plan(const AlignParams& p, int samples, int max_n, int merge, ScreenParams* sp);
static inline bool screen2_flank_ok(int m, int k) { return k >= 1 && k <= STRQ_SCREEN2_LPF * STRQ_SCREEN2_CPL && m == k * STRQ_SCREEN_S; }
size_t screen2_lds_bytes(int tsize_a, int tsize_b);
int launch_screen2(hipStream_t stream, const Screen2Task* tasks, int n_groups, int* queue, const ScreenParams& sp,
                   size_t lds_bytes, int groups_per_cu, int n_cu, int merge);
// list / theta_list (nullable, device): only the alignments list[0 .. n_groups), each with the candidate threshold theta_list[idx]
// in chunk units (the coarse screen's second look: every chunk whose bound reaches the score the first look found); out[idx]
int launch_screen_windows(hipStream_t stream, const ScreenTask* tasks, int n_groups, const ScreenParams& sp,
                          const int32_t* bound_scaled, ScreenWindows* out, const int32_t* list = nullptr, const int32_t* theta_list = nullptr);

}  // namespace strq
