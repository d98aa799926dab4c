// Internal device-side structures of the flank-alignment kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define STRQ_CKPT_STEPS 256                  // wavefront checkpoint spacing in steps (multiple of 64)
#define STRQ_COLS_PER_STEP 2                 // a lane computes two DP columns per step (ILP 2)
#define STRQ_CKPT_FIELDS(R) (2 * (R) + 6)    // S[R], H[R], SbotA, VbotA, VbotB, upS, pad, pad
#define STRQ_TRACE_WORDS(R) (((R) + 15) / 16)

namespace strq {

struct AlignParams {   // reference: scripts/STRique.py:507-523 -> src/align_raw.h:84-103
    float open_h, ext_h, open_v, ext_v, dist_offset, dist_min;
};

// read-only description of one strip of one alignment (flank x read)
struct AlignTask {
    const uint8_t* levels;   // n levels of the read (column j <-> levels[j-1])
    const float* table;      // ragged banded score table of the whole flank (row offsets in band_lo; equal classes share a row)
    const uint8_t* table3;   // the same table as 24-bit fixed point (units of 2^-20), 3 bytes per entry, or null
    const int32_t* band_lo;  // k classes of this strip: first level | (levels - 1) << 8 | row offset << 16
    const float* col0;       // m+1: S[row0 + i][0] (column 0 is not free)
    float* ckpt;             // wavefront checkpoints of the forward pass (per strip)
    int32_t* rec;            // m_total: per flank row (j << 1) | is_vertical   (trace pass output)
    const float* bnd_in;     // {S, V} of row `row0` for columns 1..n (from the strip above) or null
    float* bnd_out;          // {S, V} of this strip's last row for the strip below, or null
    const AlignTask* up;     // task of the strip above (trace pass), or null
    int32_t n, m, k, tsize;  // columns; rows / classes of this strip; floats of the whole score table
    int32_t row0, m_total;   // rows above this strip; rows of the whole flank
    int32_t col_off, n_full; // column segment: DP column 0 of this task is column col_off of the read; columns of the whole read
};

struct AlignResult {
    float best;              // forward: max_j S[m][j]
    int32_t j_end;           // forward: leftmost column of that maximum
    int32_t j0;              // trace: column where the path leaves row 0
    int32_t status;
};

static inline int align_num_steps(int n) { return (n + 1) / 2 + 63; }
static inline int align_num_ckpts(int n) { return (align_num_steps(n) - 1) / STRQ_CKPT_STEPS; }

// rows per lane and number of strips for a flank of m rows; 0 if no compiled shape fits
#define STRQ_MAX_STRIPS 64          // flanks up to 64 x 64 x 12 = 49 152 samples (8192 k-mer classes: 8197 nt at 6 samples per k-mer)
int align_plan(int m, int samples, int* rows_per_lane, int* n_strips);
int align_effective_samples(int samples);
// Column segments (several waves per alignment, one score table per workgroup).
// A read of n columns is cut into `segs` pieces; piece k owns the columns (o_k, o_k+1] and starts its DP
// cold (the column-0 rule) `overlap` columns to the left of o_k.  With dist_min >= 0 and negative
// horizontal gap scores every path that scores >= 0 spans at most `overlap` columns, so every last-row
// value >= 0 inside the owned range -- and every cell of the optimal path that ends there -- is computed
// exactly; the best of the pieces (leftmost on ties) is the best of the read.
// Returns 0 when the parameters do not allow it (then segs must be 1).
int align_segment_overlap(const AlignParams& p, int m);
// forward pass of `n_groups` alignments of `segs` tasks each (tasks[g * segs + k]; n == 0: unused piece);
// one workgroup of `segs` waves per table, `tables_per_cu` workgroups per CU.  LH = LV = true, one strip.
// group_list / n_list (nullable): run only the alignments listed on the device (second round, see below).
int launch_align_segments(hipStream_t stream, int R, int S, const AlignTask* tasks, AlignResult* seg_results,
                          int n_groups, int segs, int* queue, const AlignParams& p, int lds_dwords,
                          int tables_per_cu, int n_cu, int packed, const int* group_list = nullptr, const int* n_list = nullptr,
                          bool known_last_row = false);
int align_segments_wpe(int segs, int tables_per_cu);      // waves per SIMD the kernel instance of such a launch is compiled for
// per alignment: best piece -> results[a] (j_end in read columns), pick[a] = pick_base + index of its task.
// The pieces may be cut with less overlap than align_segment_overlap: the result is exact whenever the best
// score reaches align_segment_min_score(p, m, overlap_used); alignments below it are appended to `redo`
// (device list + counter) for a second round with the worst-case overlap (list / n_list select them).
int launch_align_combine(hipStream_t stream, const AlignTask* tasks, const AlignResult* seg_results, int n_align,
                         int segs, AlignResult* results, int32_t* pick, int pick_base = 0, const int* list = nullptr,
                         const int* n_list = nullptr, const float* min_score = nullptr, int* redo = nullptr, int* redo_count = nullptr,
                         unsigned int* redo_total = nullptr);
// results / picks of a side launch (the coarse screen's second look: groups [first[a], first[a + 1]) of four pieces per alignment a) into the
// slots of their alignments: dst[pos[a]] = the best group's result, the leftmost on ties
int launch_align_scatter(hipStream_t stream, const AlignResult* src, const int32_t* pick_src, const int32_t* first, const int32_t* pos, int n,
                         AlignResult* dst, int32_t* pick_dst);
float align_segment_min_score(const AlignParams& p, int m, int overlap_used);
// columns a path that scores at least `score` can span (the overlap a cold-started piece needs to hold every such path)
int align_overlap_for_score(const AlignParams& p, int m, float score);
// phase 0 = forward, 1 = trace.  `queue`: one zero-initialised int per launch.
// trace: task of alignment ti is tasks[pick[ti]] (pick may be null: identity), result slot is results[ti].
// mode (forward only): bit 0 = strip has an input boundary, bit 1 = strip has an output boundary.
// packed: the tasks carry 3-byte tables (table3); lds_floats_per_wave is the LDS slice of a wave in dwords.
int launch_align(hipStream_t stream, int R, int S, const AlignTask* tasks, AlignResult* results,
                 int n_tasks, int* queue, const AlignParams& p, int lds_floats_per_wave,
                 int waves_per_block, int n_blocks, uint64_t* scratch, int phase, int mode, int packed,
                 const int32_t* pick = nullptr);
size_t align_trace_scratch_words_per_wave(int R);

}  // namespace strq
