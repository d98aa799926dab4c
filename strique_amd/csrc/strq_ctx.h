// Internal context of libstrique_hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <functional>
#include <mutex>
#include "align_kernels.h"
#include "viterbi_kernels.h"
#include "screen_kernels.h"
#include "strq_opt.h"

namespace strq {

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// strq_ctx::queue: work-queue heads of the persistent kernels, one zero-initialised int per launch of a call (index 0:
// the table kernel's count of borderline entries, launches from index STRQ_QUEUE_FIRST on)
#define STRQ_QUEUE_BYTES 65536
#define STRQ_QUEUE_FIRST 8
#define STRQ_QUEUE_SLOTS (STRQ_QUEUE_BYTES / 4)

// one sub-batch of alignments for align_core (strq_align_api.hip)
struct AlignCoreIn {
    int nb = 0, samples = 6;
    const uint8_t* d_levels = nullptr;     // device: concatenated level streams
    const int64_t* read_off = nullptr;     // host: offsets of the reads in d_levels
    const float* d_level_val = nullptr;    // device: 256 level values per read
    const int32_t* read = nullptr;         // host, per alignment
    const int* n = nullptr; const int* m = nullptr; const int* k = nullptr; const int* R = nullptr;
    const int* NS = nullptr;               // strips per alignment (1 or 2)
    const float* const* flank = nullptr;   // host: flank template of each alignment
    // called once the score-table kernel of the sub-batch is queued (whatever is to share the GPU with the alignment kernels that
    // follow is launched behind it -- strq_detect_api.hip); a non-zero return aborts the call
    std::function<int()> after_tables;
};
struct AlignCoreOut {
    std::vector<int> order;                // task position -> alignment index
    std::vector<size_t> rec_off;           // per alignment: offset of its record in d_rec
    size_t rec_total = 0;
    AlignTask* d_tasks = nullptr; AlignResult* d_results = nullptr; int32_t* d_rec = nullptr;
    int n_hard = 0;
    int n_launches = 0;                    // forward-kernel launches
    double wave_steps = 0, columns = 0;    // forward work of this sub-batch
    int segs = 1, tables = 0, packed = 0, rows_per_lane = 0;
    int overlap_first = 0, overlap_worst = 0;      // columns the pieces of the last segmented launch were cut with first / in the worst case
};

struct HostModel {
    VitModel h;              // host copy (pointers are device pointers)
    const VitModel* dev = nullptr;
    DevBuf blob;
    DevBuf g2_blob;          // register-resident image of the same model (strq_model_set_positions), if any
    // the baked arrays as they were handed to strq_model_create (strq_model_set_positions lays them out once more)
    int32_t n_states = 0, silent_start = 0, start = 0, end = 0;
    std::vector<int32_t> in_ptr, in_src, emis_kind, count_inc, state_tag;
    std::vector<double> in_logp, emis_a, emis_b, emis_c;
};

}  // namespace strq

struct strq_ctx {
    // strq_set_option: per-context switches (key -> value; an empty value = "unset", whatever the environment says).  Every switch
    // of the library is read through strq::opt(key): this map first, then the process-wide table (strq_set_option(NULL, ...)),
    // then the environment variable of the same name -- so that one process can run A/B legs on the same resident batches.
    std::map<std::string, std::string> options;
    mutable std::mutex options_mu;            // strq_set_option on one thread, strq::opt on another (the upload thread of the context; a second context's host thread never sees this map)
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream[3] = {};          // forward launches of a screened sub-batch beside the first one (align_core), created on first use
    hipEvent_t ev_fork = nullptr, side_join[3] = {};
    hipEvent_t ev[8] = {};
    strq::AlignParams ap{-2.0f, -8.0f, -2.0f, -8.0f, 8.0f, -16.0f};   // src/align_raw.h:51-60
    std::string err;
    float timing[8] = {};
    double counters[8] = {};
    int32_t geometry[8] = {};                 // strq_last_geometry
    int32_t vit_launches[4] = {};             // strq_last_viterbi_launches
    int64_t second_round[2] = {};             // strq_last_second_round: alignments that ran the second forward round / alignments, last batched call
    strq::DevBuf redo_total;                  // device counter: alignments whose first forward round missed its certificate (second look + whole-read second round)
    int64_t look2_served = 0;                 // of them, resolved by the coarse screen's second look (or dropped with an attempt that started over): not whole-read reruns
    double screen_stats[8] = {};              // strq_last_screen
    double overlap[4] = {};                   // strq_last_overlap
    bool screen_ran = false;                  // the last align_core call ran the screen (events 5, 6 bracket it)
    // The screen pays when nearly every alignment gets windows (a read that holds its flank clearly) and costs a pass when not:
    // a sub-batch in which fewer than 90 % did, whose windows hold more than 6 % of the columns, or in which more than one alignment
    // in 4096 is left with over 32 k columns to run (a tail behind thousands of small windows) pauses it for the next eight
    // sub-batches of this context, then it is tried again.
    int screen_pause = 0;
    // the coarse screen in front of it (align_screen3_kernel): paused the same way when it leaves too many columns or certifies too
    // few alignments in its first look; its candidate margin (score units below the best chunk bound) is a constant
    int coarse_pause = 0;
    int coarse_fail = 0, screen_fail = 0;     // consecutive sub-batches on which the coarse / the fine screen did not pay: the pause doubles (8, 16, ... 256)
    float coarse_margin = 384.0f;
    float aborted_fwd_ms = 0, aborted_screen_ms = 0;      // forward / screen time of an attempt align_core stopped and started over (added to the call's times)
    int screen_mode_last = 0;                 // screen of the last align_core call: 0 none, 1 fine, 2 coarse
    int coarse_merge_last = 0;                // ... and the flank rows per DP row of the coarse one
    
    // workspace
    strq::DevBuf levels, level_val, flank_cls, tables, tables3, band_lo, col0, ckpt, rec, tasks, results,
        queue, scratch, lutinfo, hard, misc, vit_x, vit_tasks, vit_bp, vit_path, bnd,
        gen_codes, gen_table, gen_bnd, gen_trace, gen_hard,      // generic align_overlap path
        screen,                                                   // upper-bound screen: tasks, chunk maxima, windows
        ckpt2;                                                    // checkpoints of the coarse screen's second look
    std::vector<strq::HostModel*> models;
    void* detect = nullptr;                   // DetectState (strq_detect_api.hip)
    size_t max_ws_bytes = (size_t)96 << 30;   // cap for checkpoint workspace per sub-batch
    // best flank scores of the previous detect sub-batch as fractions of m * dist_offset (sorted), and its mean read
    // length: the column segments of the next sub-batch are cut with the overlap that is cheapest for that distribution
    std::vector<float> score_fracs; double mean_n = 0;
};

namespace strq {
// marks `c` as that context for the lifetime of the object (every C-ABI entry that takes a context opens one)
struct CtxScope { const strq_ctx* prev; explicit CtxScope(const strq_ctx* c); ~CtxScope(); };
int align_core(strq_ctx* c, const AlignCoreIn& in, AlignCoreOut& out);
int align_core_times(strq_ctx* c, float* t_lut, float* t_fwd, float* t_tr);
int align_validate_flank(strq_ctx* c, const float* f, int64_t m, int samples, int* k_out, int* R_out, int* NS_out);
size_t align_workspace_bytes(int n, int m, int R, int NS);   // checkpoints + strip boundary of one alignment
float host_cell_score(const AlignParams& p, float h, float v);
void detect_state_free(strq_ctx* c);
void host_stats_batch(const double* signals, const int64_t* offsets, int64_t n_reads, bool want_raw, double* out,
                      const double* const* reads = nullptr);   // host_stats.hip; reads: one buffer per read instead of `signals`
}

#define STRQ_HIP(ctx, call)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return STRQ_ERR_DEVICE;                                                            \
        }                                                                                      \
    } while (0)
