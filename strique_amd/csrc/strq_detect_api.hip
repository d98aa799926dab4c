// C ABI: repeatCounter.add_target / detect as a batched device pipeline
// (reference scripts/STRique.py:553-618):
//   conditioning -> score tables -> 2 flank alignments per read -> positions / gate -> HMM Viterbi
// Everything between upload and fetch stays in HBM; the host only sequences kernels, reads back the
// table-width class of each alignment and assembles the result records.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <time.h>
#include <vector>
#include "../../include/strique_hip.h"
#include "strq_ctx.h"
#include "cond_kernels.h"
#include "viterbi_kernels.h"
#include "mod_kernels.h"

using namespace strq;

#define STRQ_DBG(...) do { if (strq::opt("STRQ_DEBUG")) { fprintf(stderr, "[strq] " __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)
static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

namespace strq {

struct ReadGeom {           // per read, written by finalize_kernel
    double score_prefix, score_suffix;
    int64_t prefix_begin, prefix_end, suffix_begin, suffix_end;
    int32_t gate, pad_;
    float best_prefix, best_suffix;      // raw alignment scores (the overlap planning of the next sub-batch reads their distribution)
};

// position of flank row k in the read: argmin_i |a_idx[i] - b_idx[k]| of __detect_range__
// (STRique.py:540-547) evaluated on the compact record: a diagonal row sits on its sample; a row
// inside a vertical run sits between two samples and takes the nearer one, the lower index on a tie.
static __device__ int64_t row_position(const int32_t* rec, int m, int k, int n)
{
    const int32_t r = rec[k];
    const int64_t j = r >> 1;
    if (!(r & 1)) return j - 1;
    int k1 = k; while (k1 > 0 && rec[k1 - 1] == r) --k1;
    int k2 = k; while (k2 < m - 1 && rec[k2 + 1] == r) ++k2;
    const int d_prev = k - k1 + 1, d_next = k2 - k + 1;
    const bool has_prev = j >= 1, has_next = j < n;
    if (has_prev && (!has_next || d_prev <= d_next)) return j - 1;
    return j;
}

struct FinalizeArgs {
    const AlignTask* tasks; const AlignResult* results;
    const int32_t* task_of;        // 2 per read: task position of the prefix / suffix alignment
    const int32_t* trim;           // 2 per read: pre_trim of the prefix flank, post_trim of the suffix flank
    const int32_t* vit_slot;       // per read: index of its VitTask
    const ReadCond* rc;
    const VitModel* const* model_of;   // per read
    const void* flt;               // filtered signal (int16 or double), concatenated
    int is_f64;
    PoreStats ps;
    ReadGeom* geom; VitTask* vit;
    int n_reads;
};

__global__ void finalize_kernel(FinalizeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    const ReadCond rc = a.rc[r];
    ReadGeom g = {};
    VitTask vt = {};
    // the alignments run on whatever the conditioning produced (an all-NaN signal scores dist_min in every
    // cell, as it does in the reference), so the positions are reported for every non-empty read
    if (rc.n > 0) {
        const AlignTask& tp = a.tasks[a.task_of[2 * r]]; const AlignResult& rp = a.results[a.task_of[2 * r]];
        const AlignTask& ts = a.tasks[a.task_of[2 * r + 1]]; const AlignResult& rs = a.results[a.task_of[2 * r + 1]];
        {
            const int64_t b = row_position(tp.rec, tp.m_total, 0, tp.n), e = row_position(tp.rec, tp.m_total, tp.m_total - 1, tp.n);
            g.score_prefix = e > b ? (double)rp.best / (double)(e - b) : 0.0;
            g.best_prefix = rp.best;
            g.prefix_begin = row_position(tp.rec, tp.m_total, a.trim[2 * r], tp.n);
            g.prefix_end = e;
        }
        {
            const int64_t b = row_position(ts.rec, ts.m_total, 0, ts.n), e = row_position(ts.rec, ts.m_total, ts.m_total - 1, ts.n);
            g.score_suffix = e > b ? (double)rs.best / (double)(e - b) : 0.0;
            g.best_suffix = rs.best;
            g.suffix_begin = b;
            g.suffix_end = row_position(ts.rec, ts.m_total, ts.m_total - 1 - a.trim[2 * r + 1], ts.n);
        }
        // the reference's gate (STRique.py:602) looks at the two alignments only: a read whose 8-bit morphology signal
        // normalises while its filtered signal does not (empty percentile tails: NaN constants) still goes to the HMM,
        // as a window of NaN observations -- pomegranate's missing-value rule, see viterbi_kernels.hip
        g.gate = (g.prefix_begin < g.suffix_end && g.score_prefix > 0.0 && g.score_suffix > 0.0) ? 1 : 0;
    }
    vt.model = a.model_of[r];
    if (g.gate) {
        vt.T = g.suffix_end - g.prefix_begin;
        if (a.is_f64) { vt.sig = reinterpret_cast<const double*>(a.flt) + rc.off + g.prefix_begin; vt.src_kind = VIT_SRC_F64_AFFINE; }
        else { vt.sig = reinterpret_cast<const int16_t*>(a.flt) + rc.off + g.prefix_begin; vt.src_kind = VIT_SRC_I16_AFFINE; }
        vt.c1 = rc.f_c1; vt.h1 = rc.f_h1; vt.h2 = rc.h2; vt.c2 = rc.c2; vt.lo = a.ps.clip_lo; vt.hi = a.ps.clip_hi;
    }
    a.geom[r] = g;
    a.vit[a.vit_slot[r]] = vt;
}

struct Target {
    std::vector<float> prefix_ext, suffix_ext;
    int trim_prefix = 0, trim_suffix = 0, samples = 6;
    int kp = 0, Rp = 0, NSp = 1, ks = 0, Rs = 0, NSs = 1;
    int model_id = -1, count_bias = 0;
    int mod_model_id = -1; double mod_min = 0, mod_max = 0;
};

struct Batch {
    int64_t n_reads = 0;
    int dtype = 0;                       // 0 int16, 1 float64
    std::vector<int64_t> off;            // n_reads + 1
    std::vector<int32_t> target;
    std::vector<double> host_stats;      // 6 per read (float64 input only)
    DevBuf raw;                          // all reads, resident
    bool on_host = false;                // samples still (partly) in the caller's memory: uploaded sub-batch by sub-batch
    const char* host_src = nullptr;      // strq_detect_batch: the caller's concatenated buffer
    std::vector<const char*> host_reads; // strq_detect_batch_reads: one buffer per read instead
    void forget_host() { on_host = false; host_src = nullptr; host_reads.clear(); }
    int64_t uploaded = 0;                // reads whose samples are in `raw`
    std::vector<strq_result> results;
    std::vector<std::string> mod;        // modification pattern per read ('-' if none)
    float t_cond = 0, t_lut = 0, t_fwd = 0, t_trace = 0, t_vit = 0, t_total = 0;
    double n_hard = 0;
    int n_fwd_launches = 0;
};

struct DetectState {
    PoreStats ps{0, 0, 0, 0};
    bool have_ps = false;
    std::vector<Target> targets;
    Batch batch;
    DevBuf rc, hist16, hist8, geom, idx, hist_raw, bp, path, modtask, modsig, modlen, pattern, hrange, modpool, f64s;
    hipEvent_t ev[4] = {};
    bool ev_ok = false;
    int64_t part_reads = 0;              // strq_batch_upload_part: reads uploaded so far
    // Two sub-batches are in flight at a time: the Viterbi launches of sub-batch k run on `vit_stream` while the conditioning and the
    // flank alignments of sub-batch k + 1 are queued on the context's stream (the Viterbi launch lasts as long as its longest window --
    // reads whose flanks were mislocated decode 10^5 steps and more -- and most of the GPU idles under that tail).  What a Viterbi launch
    // reads or writes exists twice (filtered signal, tasks, results, order, queue heads); results come back one sub-batch late.
    struct Slot {
        DevBuf flt, vit, vres, order, vq;
        bool active = false;             // forward stage done, results not yet in Batch::results
        bool launch_pending = false;     // ... and its Viterbi launches not yet queued (they go behind the score-table kernel of the next sub-batch)
        struct VL { int shape, first, count, max_states; };
        std::vector<VL> vls;             // the Viterbi launches of the sub-batch: kernel shape, task range
        int vit_mode = 0;                // 0 count, 2 MARK (modification pass follows)
        int64_t r0 = 0; int nr = 0;
        std::vector<int32_t> vit_slot;
        void* pinned = nullptr; size_t pinned_cap = 0;      // ReadGeom[nr], VitResult[nr], ReadCond[nr], unsigned redo
        hipEvent_t fwd_done = nullptr, v0 = nullptr, v1 = nullptr;
    };
    Slot slot[2];
    int next_slot = 0;
    // samples of the NEXT sub-batch travel host -> HBM on a thread of their own while this thread sequences the kernels of the current one
    // (align_core blocks it in three host round trips per sub-batch: an upload from the same thread only started when those were through,
    // and the GPU idled under it -- 250 ms per 4096 reads instead of 175: gpurun_out/r6n)
    std::thread up_thread;
    int up_rc = 0;
    hipStream_t vit_stream = nullptr;
    int levels_shift = 0;                // bytes the level stream of the current sub-batch starts behind the buffer's base (alignment phase)
    hipStream_t copy_stream = nullptr;   // host -> HBM uploads that overlap the kernels of the previous sub-batch
    static constexpr int N_STAGE = 4;    // pinned staging ring of upload_reads
    void* stage[N_STAGE] = {}; hipEvent_t stage_ev[N_STAGE] = {}; bool stage_busy[N_STAGE] = {};
};

static int upload_join(DetectState* d);

static DetectState* dstate(strq_ctx* c)
{
    if (!c->detect) c->detect = new DetectState();
    return static_cast<DetectState*>(c->detect);
}

void detect_state_free(strq_ctx* c)
{
    if (!c->detect) return;
    DetectState* d = static_cast<DetectState*>(c->detect);
    (void)upload_join(d);
    if (d->vit_stream) (void)hipStreamSynchronize(d->vit_stream);
    for (DevBuf* b : {&d->batch.raw, &d->rc, &d->hist16, &d->hist8, &d->geom, &d->idx,
                      &d->hist_raw, &d->bp, &d->path, &d->modtask, &d->modsig, &d->modlen, &d->pattern, &d->hrange, &d->modpool, &d->f64s}) b->release();
    for (auto& sl : d->slot) {
        for (DevBuf* b : {&sl.flt, &sl.vit, &sl.vres, &sl.order, &sl.vq}) b->release();
        if (sl.pinned) (void)hipHostFree(sl.pinned);
        for (hipEvent_t e : {sl.fwd_done, sl.v0, sl.v1}) if (e) (void)hipEventDestroy(e);
    }
    if (d->vit_stream) (void)hipStreamDestroy(d->vit_stream);
    if (d->ev_ok) for (auto& e : d->ev) (void)hipEventDestroy(e);
    if (d->copy_stream) (void)hipStreamDestroy(d->copy_stream);
    for (int i = 0; i < DetectState::N_STAGE; ++i) { if (d->stage[i]) (void)hipHostFree(d->stage[i]); if (d->stage_ev[i]) (void)hipEventDestroy(d->stage_ev[i]); }
    delete d;
    c->detect = nullptr;
}

// Modification pass for the reads of one sub-batch whose target has a modification model.
// The flanked-model Viterbi ran in MARK mode (viterbi_kernels.hip): its result carries the first and
// last sample decoded into the repeat section, which is all detect step 13 (STRique.py:608) needs.
static int run_mod_pass(strq_ctx* c, DetectState* d, DetectState::Slot& sl, int64_t r0, int nr, const ReadCond* rc,
                        const ReadGeom* geom, const VitResult* vres,
                        const std::vector<int32_t>& vit_slot)
{
    Batch& B = d->batch;
    hipStream_t st = c->stream;
    const int esz = B.dtype == 0 ? 2 : 8;
    const int64_t s0 = B.off[r0];
    std::vector<int> who;                      // reads that reach the modification model
    for (int i = 0; i < nr; ++i) {
        const Target& t = d->targets[B.target[r0 + i]];
        if (t.mod_model_id < 0 || !geom[i].gate) continue;
        const VitResult& v = vres[vit_slot[i]];
        if (v.status == 2) { c->err = "modification pass: repeat window of 2^21 samples or more"; return STRQ_ERR_UNSUPPORTED; }
        if (v.status == 0) who.push_back(i);
    }
    const int nm = (int)who.size();
    if (!nm) return STRQ_OK;
    // 1. + 2. the samples decoded into repeat states: one contiguous stretch [enter, leave) of the window,
    //         renormalised from the raw signal and clipped
    std::vector<int64_t> len(nm), first(nm);
    size_t sig_tot = 0; std::vector<size_t> sig_off(nm);
    for (int k = 0; k < nm; ++k) {
        const int i = who[k];
        const VitResult& v = vres[vit_slot[i]];
        const int64_t T = geom[i].suffix_end - geom[i].prefix_begin;
        const int64_t enter = v.dbg[0], leave = v.dbg[1];
        first[k] = enter ? enter - 1 : 0;
        len[k] = enter ? (leave ? leave - 1 : T) - (enter - 1) : 0;
        sig_off[k] = sig_tot; sig_tot += (size_t)len[k];
    }
    STRQ_HIP(c, d->modtask.reserve((size_t)nm * (sizeof(VitTask) + sizeof(VitResult) + 8 + sizeof(ModTask) + sizeof(PatTask)) + 256));
    VitTask* d_tb = d->modtask.as<VitTask>();
    VitResult* d_tr = reinterpret_cast<VitResult*>(d_tb + nm);
    int32_t** d_tp = reinterpret_cast<int32_t**>(d_tr + nm);
    ModTask* d_mt = reinterpret_cast<ModTask*>(d_tp + nm);
    PatTask* d_pt = reinterpret_cast<PatTask*>(d_mt + nm);
    STRQ_HIP(c, d->modsig.reserve(sig_tot * 8 + 64));
    STRQ_HIP(c, d->modlen.reserve((size_t)nm * 16 + 64));
    std::vector<ModTask> mt(nm);
    for (int k = 0; k < nm; ++k) {
        const int i = who[k]; const Target& t = d->targets[B.target[r0 + i]];
        ModTask& m = mt[k];
        m.path = nullptr; m.tag = nullptr;          // contiguous stretch: every sample is kept
        m.raw = d->batch.raw.as<char>() + (size_t)(s0 + rc[i].off + geom[i].prefix_begin + first[k]) * esz;
        m.out = d->modsig.as<double>() + sig_off[k]; m.T = len[k]; m.is_f64 = B.dtype; m.pad_ = 0;
        m.c1 = rc[i].r_c1; m.h1 = rc[i].r_h1; m.h2 = rc[i].h2; m.c2 = rc[i].c2;
        m.clip_lo = d->ps.clip_lo; m.clip_hi = d->ps.clip_hi; m.mod_lo = t.mod_min; m.mod_hi = t.mod_max;
    }
    int64_t* d_len = d->modlen.as<int64_t>();
    STRQ_HIP(c, hipMemcpyAsync(d_mt, mt.data(), (size_t)nm * sizeof(ModTask), hipMemcpyHostToDevice, st));
    if (launch_mod_compact(st, d_mt, nm, d_len)) { c->err = "compaction launch failed"; return STRQ_ERR_DEVICE; }
    // 3. Viterbi on the modification model.  Hub records (one 8-byte record per time step, read back with one
    //    hop per repeat unit) when the model has the hub structure; back-pointers + traceback otherwise.
    std::map<int, std::vector<int>> by_shape;
    bool use_hub = !strq::opt("STRQ_MOD_BACKPOINTERS");
    for (int k = 0; k < nm; ++k) {
        HostModel* hm = c->models[d->targets[B.target[r0 + who[k]]].mod_model_id];
        const int shape = vit_shape_of(hm->h);
        if (shape < 0) { c->err = "modification model does not fit a compiled Viterbi kernel"; return STRQ_ERR_UNSUPPORTED; }
        by_shape[shape].push_back(k);
        if (hm->h.rec_state < 0 || hm->h.epl > 2 || len[k] >= ((int64_t)1 << 31)) use_hub = false;
    }
    std::vector<VitTask> vt2(nm); std::vector<int> slot2(nm); std::vector<int32_t*> tp2(nm);
    size_t bp2 = 0, p2 = 0; std::vector<size_t> bp2_off(nm), p2_off(nm);
    for (int k = 0; k < nm; ++k) {
        HostModel* hm = c->models[d->targets[B.target[r0 + who[k]]].mod_model_id];
        // back-pointers: uint16 per (time step, state); hub records: 8 bytes per time step (in uint16 units: 4)
        bp2_off[k] = bp2; bp2 += use_hub ? (size_t)(len[k] + 1) * 4 : (size_t)(len[k] + 1) * hm->h.n_states;
        p2_off[k] = p2; p2 += (size_t)len[k] + 1;
    }
    STRQ_HIP(c, d->bp.reserve(bp2 * 2 + 64));
    STRQ_HIP(c, d->pattern.reserve((use_hub ? 0 : p2 * 4) + p2 + (size_t)nm * 8 + (size_t)nm * sizeof(HubTask) + 64));
    int32_t* d_path2 = d->pattern.as<int32_t>(); char* d_chars = reinterpret_cast<char*>(d_path2 + (use_hub ? 0 : p2));
    STRQ_HIP(c, hipMemsetAsync(c->queue.p, 0, 1024, st));
    { int sidx = 0, qi = 0;
      for (auto& g : by_shape) {
        const int first = sidx; int mx = 0;
        for (int k : g.second) {
            HostModel* hm = c->models[d->targets[B.target[r0 + who[k]]].mod_model_id];
            VitTask& v = vt2[sidx]; v = VitTask();
            v.model = hm->dev; v.sig = mt[k].out; v.T = len[k]; v.src_kind = VIT_SRC_F64; v.bp = d->bp.as<uint16_t>() + bp2_off[k];
            slot2[k] = sidx; tp2[sidx] = d_path2 + p2_off[k]; mx = std::max(mx, hm->h.n_cells); ++sidx;
        }
        STRQ_HIP(c, hipMemcpyAsync(d_tb + first, vt2.data() + first, (size_t)(sidx - first) * sizeof(VitTask), hipMemcpyHostToDevice, st));
        int* d_order = nullptr;
        if (sidx - first <= 8192) {
            d_order = sl.order.as<int>() + first;
            if (launch_vit_sort(st, d_tb + first, sidx - first, d_order)) { c->err = "sort launch failed"; return STRQ_ERR_DEVICE; }
        }
        if (const int vrc = launch_viterbi(st, g.first, mx, d_tb + first, d_tr + first, sidx - first, c->queue.as<int>() + qi, c->n_cu, use_hub ? 3 : 1, d_order)) {
            // 2 / 3: this kernel shape has no such decode mode -- the caller's input, not a device fault (strq_viterbi_batch maps them the same way)
            c->err = (vrc == 2 || vrc == 3) ? "viterbi: decode mode not available for this model's kernel shape" : "viterbi launch failed";
            return (vrc == 2 || vrc == 3) ? STRQ_ERR_UNSUPPORTED : STRQ_ERR_DEVICE;
        }
        ++qi;
      } }
    int64_t* d_plen = d_len;
    if (use_hub) {
        // 4. pattern strings from the hub records
        std::vector<HubTask> ht(nm);
        for (int k = 0; k < nm; ++k) {
            const int sl = slot2[k];
            ht[sl].rec = reinterpret_cast<const uint64_t*>(d->bp.as<uint16_t>() + bp2_off[k]);
            ht[sl].result = d_tr + sl; ht[sl].out = d_chars + p2_off[k];
        }
        HubTask* d_ht = reinterpret_cast<HubTask*>(d_chars + ((p2 + 15) & ~(size_t)15));
        STRQ_HIP(c, hipMemcpyAsync(d_ht, ht.data(), (size_t)nm * sizeof(HubTask), hipMemcpyHostToDevice, st));
        if (launch_mod_hub_pattern(st, d_ht, nm, d_plen)) { c->err = "pattern launch failed"; return STRQ_ERR_DEVICE; }
    } else {
        STRQ_HIP(c, hipMemcpyAsync(d_tp, tp2.data(), (size_t)nm * 8, hipMemcpyHostToDevice, st));
        if (launch_vit_traceback(st, d_tb, d_tr, d_tp, nm)) { c->err = "traceback launch failed"; return STRQ_ERR_DEVICE; }
        // 4. pattern strings
        std::vector<PatTask> pt(nm);
        for (int k = 0; k < nm; ++k) {
            const int sl = slot2[k];
            HostModel* hm = c->models[d->targets[B.target[r0 + who[k]]].mod_model_id];
            pt[sl].path = tp2[sl]; pt[sl].tag = hm->h.state_tag; pt[sl].out = d_chars + p2_off[k]; pt[sl].T = len[k];
            pt[sl].status = &d_tr[sl].status;
        }
        STRQ_HIP(c, hipMemcpyAsync(d_pt, pt.data(), (size_t)nm * sizeof(PatTask), hipMemcpyHostToDevice, st));
        if (launch_mod_pattern(st, d_pt, nm, d_plen)) { c->err = "pattern launch failed"; return STRQ_ERR_DEVICE; }
    }
    // read-back: the lengths first, then the strings gathered into a dense pool (the sparse buffer has one byte per time step:
    // ~180 MB per 4096 reads of 50 kb, 25 ms through pageable memory, for ~4 MB of strings)
    std::vector<int64_t> plen(nm);
    STRQ_HIP(c, hipMemcpyAsync(plen.data(), d_plen, (size_t)nm * 8, hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipStreamSynchronize(st));
    std::vector<GatherTask> gt(nm); size_t dense = 0;
    for (int k = 0; k < nm; ++k) {
        const int64_t ln = std::max<int64_t>(0, std::min<int64_t>(plen[slot2[k]], len[k] + 1));
        gt[k] = {(int64_t)p2_off[k], (int64_t)dense, ln}; dense += (size_t)ln;
    }
    STRQ_HIP(c, d->modpool.reserve(dense + (size_t)nm * sizeof(GatherTask) + 64));
    GatherTask* d_gt = d->modpool.as<GatherTask>(); char* d_dense = reinterpret_cast<char*>(d_gt + nm);
    STRQ_HIP(c, hipMemcpyAsync(d_gt, gt.data(), (size_t)nm * sizeof(GatherTask), hipMemcpyHostToDevice, st));
    if (launch_mod_gather(st, d_gt, nm, d_chars, d_dense)) { c->err = "gather launch failed"; return STRQ_ERR_DEVICE; }
    std::vector<char> chars(dense + 1);
    if (dense) STRQ_HIP(c, hipMemcpyAsync(chars.data(), d_dense, dense, hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipStreamSynchronize(st));
    for (int k = 0; k < nm; ++k) B.mod[r0 + who[k]] = std::string(chars.data() + gt[k].dst, (size_t)gt[k].len);
    return STRQ_OK;
}

// bytes [pos, pos + len) of the batch (reads back to back) from the caller's memory: one buffer, or one per read
static void host_bytes(const Batch& B, size_t esz, char* dst, size_t pos, size_t len)
{
    if (B.host_reads.empty()) { std::memcpy(dst, B.host_src + pos, len); return; }
    // the read that holds byte `pos`
    size_t r = (size_t)(std::upper_bound(B.off.begin(), B.off.end(), (int64_t)(pos / esz)) - B.off.begin()) - 1;
    while (len > 0) {
        const size_t r_begin = (size_t)B.off[r] * esz, r_end = (size_t)B.off[r + 1] * esz;
        const size_t take = std::min(len, r_end - pos);
        if (take) std::memcpy(dst, B.host_reads[r] + (pos - r_begin), take);
        dst += take; pos += take; len -= take; ++r;
    }
}

static int upload_reads(strq_ctx* c, DetectState* d, int64_t upto);

// waits for the prefetch thread (if any); returns what its upload returned
static int upload_join(DetectState* d)
{
    if (d->up_thread.joinable()) d->up_thread.join();
    const int rc = d->up_rc; d->up_rc = 0;
    return rc;
}

static void upload_prefetch(strq_ctx* c, DetectState* d, int64_t upto)
{
    Batch& B = d->batch;
    if (!B.on_host || upto <= B.uploaded) return;
    d->up_thread = std::thread([c, d, upto] {
        strq::CtxScope scope_(c);
        if (hipSetDevice(c->device) != hipSuccess) { d->up_rc = STRQ_ERR_DEVICE; return; }
        d->up_rc = upload_reads(c, d, upto);
    });
}

// Samples of reads [B.uploaded, upto) from the caller's (pageable) buffer into `raw`.  The runtime's own
// pageable path measures 8.3 GB/s; here the bytes go through a ring of pinned staging buffers: a few host
// threads copy the next piece into a free slot while the DMA engine drains the previous ones on the copy
// stream.  The calling thread blocks here while the kernels already queued on the compute stream keep
// running: that is the overlap of sub-batch k + 1's upload with sub-batch k's kernels.
static int upload_reads(strq_ctx* c, DetectState* d, int64_t upto)
{
    Batch& B = d->batch;
    if (!B.on_host || upto <= B.uploaded) return STRQ_OK;
    if (!d->copy_stream) STRQ_HIP(c, hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
    const size_t esz = B.dtype == 0 ? 2 : 8;
    const size_t b0 = (size_t)B.off[B.uploaded] * esz, b1 = (size_t)B.off[upto] * esz;
    const size_t SLOT = (size_t)32 << 20;
    int n_threads = 6;
    if (const char* e = strq::opt("STRQ_UPLOAD_THREADS")) { const int v = atoi(e); if (v >= 0 && v <= 32) n_threads = v; }
    if (b1 > b0 && n_threads == 0) {          // the runtime's pageable path
        if (B.host_reads.empty()) STRQ_HIP(c, hipMemcpyAsync(B.raw.as<char>() + b0, B.host_src + b0, b1 - b0, hipMemcpyHostToDevice, d->copy_stream));
        else for (int64_t r = B.uploaded; r < upto; ++r) {
            const size_t rb = (size_t)B.off[r] * esz, rl = (size_t)(B.off[r + 1] - B.off[r]) * esz;
            if (rl) STRQ_HIP(c, hipMemcpyAsync(B.raw.as<char>() + rb, B.host_reads[r], rl, hipMemcpyHostToDevice, d->copy_stream));
        }
        STRQ_HIP(c, hipStreamSynchronize(d->copy_stream));
    } else if (b1 > b0) {
        if (!d->stage[0]) {
            for (int i = 0; i < DetectState::N_STAGE; ++i) {
                STRQ_HIP(c, hipHostMalloc(&d->stage[i], SLOT, hipHostMallocDefault));
                STRQ_HIP(c, hipEventCreateWithFlags(&d->stage_ev[i], hipEventDisableTiming));
                d->stage_busy[i] = false;
            }
        }
        int slot = 0;
        for (size_t pos = b0; pos < b1; pos += SLOT, slot = (slot + 1) % DetectState::N_STAGE) {
            const size_t len = std::min(SLOT, b1 - pos);
            if (d->stage_busy[slot]) { STRQ_HIP(c, hipEventSynchronize(d->stage_ev[slot])); d->stage_busy[slot] = false; }
            char* dst = static_cast<char*>(d->stage[slot]);
            const size_t part = ((len + n_threads - 1) / n_threads + 4095) & ~(size_t)4095;
            std::vector<std::thread> th;
            const Batch* Bp = &B;
            for (int t = 1; t < n_threads; ++t) {
                const size_t o = (size_t)t * part;
                if (o < len) th.emplace_back([=] { host_bytes(*Bp, esz, dst + o, pos + o, std::min(part, len - o)); });
            }
            host_bytes(B, esz, dst, pos, std::min(part, len));
            for (auto& t : th) t.join();
            STRQ_HIP(c, hipMemcpyAsync(B.raw.as<char>() + pos, dst, len, hipMemcpyHostToDevice, d->copy_stream));
            STRQ_HIP(c, hipEventRecord(d->stage_ev[slot], d->copy_stream));
            d->stage_busy[slot] = true;
        }
        STRQ_HIP(c, hipStreamSynchronize(d->copy_stream));
        for (int i = 0; i < DetectState::N_STAGE; ++i) d->stage_busy[i] = false;
    }
    B.uploaded = upto;
    return STRQ_OK;
}

static void publish_timing(strq_ctx* c, const Batch& B)
{
    std::fill(c->timing, c->timing + 8, 0.0f);
    c->timing[0] = B.t_lut; c->timing[1] = B.t_fwd; c->timing[2] = B.t_trace; c->timing[5] = B.t_cond; c->timing[6] = B.t_vit;
    c->timing[3] = B.t_lut + B.t_fwd + B.t_trace + B.t_cond + B.t_vit; c->timing[4] = (float)B.n_hard; c->timing[7] = (float)B.n_fwd_launches;
}

// Queues the Viterbi launches of a sub-batch whose forward stage is complete (sort by window length, one persistent launch per kernel
// shape, results to pinned host memory) on `vs`, behind `after` when given.
static int launch_viterbi_of(strq_ctx* c, DetectState* d, DetectState::Slot& sl, hipStream_t vs, hipEvent_t after)
{
    if (!sl.launch_pending) return STRQ_OK;
    sl.launch_pending = false;
    const int nr = sl.nr;
    VitResult* h_vres = reinterpret_cast<VitResult*>(static_cast<ReadGeom*>(sl.pinned) + nr);
    if (after) STRQ_HIP(c, hipStreamWaitEvent(vs, after, 0));
    STRQ_HIP(c, hipMemsetAsync(sl.vq.p, 0, 1024, vs));
    for (auto& v : sl.vls)
        if (v.count <= 8192 && launch_vit_sort(vs, sl.vit.as<VitTask>() + v.first, v.count, sl.order.as<int>() + v.first)) { c->err = "sort launch failed"; return STRQ_ERR_DEVICE; }
    STRQ_HIP(c, hipEventRecord(sl.v0, vs));
    int qi = 0;
    std::memset(c->vit_launches, 0, sizeof(c->vit_launches));
    for (auto& v : sl.vls) {
        ++c->vit_launches[0];
        ++c->vit_launches[(v.shape & ~VIT_SHAPE_SS) == VIT_SHAPE_G2 ? 1 : ((v.shape & ~VIT_SHAPE_SS) == VIT_SHAPE_CSR ? 3 : 2)];
        int* d_order = v.count <= 8192 ? sl.order.as<int>() + v.first : nullptr;
        const int rc2 = launch_viterbi(vs, v.shape, v.max_states, sl.vit.as<VitTask>() + v.first, sl.vres.as<VitResult>() + v.first, v.count,
                                       sl.vq.as<int>() + qi, c->n_cu, sl.vit_mode, d_order, (after && vs != c->stream) ? 4 : 0);
        if (rc2) {
            c->err = (rc2 == 2 || rc2 == 3) ? "viterbi: decode mode not available for this model's kernel shape" : "viterbi launch failed";
            return (rc2 == 2 || rc2 == 3) ? STRQ_ERR_UNSUPPORTED : STRQ_ERR_DEVICE;
        }
        ++qi;
    }
    STRQ_HIP(c, hipMemcpyAsync(h_vres, sl.vres.p, (size_t)nr * sizeof(VitResult), hipMemcpyDeviceToHost, vs));
    STRQ_HIP(c, hipEventRecord(sl.v1, vs));
    return STRQ_OK;
}

// Results of a sub-batch whose Viterbi launches were queued earlier: waits for them, fills Batch::results (and runs the
// modification pass of the sub-batch, which needs the decoded repeat stretch on the host).
static int harvest(strq_ctx* c, DetectState* d, DetectState::Slot& sl, bool under_current = false)
{
    if (!sl.active) return STRQ_OK;
    sl.active = false;
    Batch& B = d->batch;
    { const int lrc = launch_viterbi_of(c, d, sl, d->vit_stream, nullptr); if (lrc) return lrc; }      // nobody came after this sub-batch
    STRQ_HIP(c, hipEventSynchronize(sl.v1));
    const int nr = sl.nr; const int64_t r0 = sl.r0;
    const ReadGeom* geom = static_cast<const ReadGeom*>(sl.pinned);
    const VitResult* vres = reinterpret_cast<const VitResult*>(geom + nr);
    const ReadCond* rc_out = reinterpret_cast<const ReadCond*>(vres + nr);
    bool any_mod = false;
    for (int i = 0; i < nr; ++i) {
        strq_result& o = B.results[r0 + i];
        std::memset(&o, 0, sizeof(o));
        const ReadGeom& g = geom[i];
        const VitResult& v = vres[sl.vit_slot[i]];
        o.status = rc_out[i].status == COND_OK ? 0 : 1;
        o.score_prefix = g.score_prefix; o.score_suffix = g.score_suffix;
        o.prefix_begin = g.prefix_begin; o.prefix_end = g.prefix_end; o.suffix_begin = g.suffix_begin; o.suffix_end = g.suffix_end;
        o.offset = g.prefix_end; o.ticks = std::max<int64_t>(g.suffix_begin - g.prefix_end, 0);
        if (g.gate) c->counters[7] += (double)(g.suffix_end - g.prefix_begin);
        if (g.gate && v.status == 0) {
            o.count = (int32_t)v.counted + d->targets[B.target[r0 + i]].count_bias;
            o.log_p = v.logp;
        }
        any_mod |= d->targets[B.target[r0 + i]].mod_model_id >= 0;
    }
    float ms;
    STRQ_HIP(c, hipEventElapsedTime(&ms, sl.v0, sl.v1)); B.t_vit += ms;
    c->overlap[0] += ms;
    if (under_current) {
        // how much of these launches lay under the alignment kernels of the sub-batch that followed (whose events are the context's
        // current ones): [forward stage start, trace end], and the screen kernel alone
        auto under = [&](hipEvent_t a, hipEvent_t b) -> double {
            float ta = 0, tb = 0;
            if (hipEventElapsedTime(&ta, sl.v0, a) != hipSuccess || hipEventElapsedTime(&tb, sl.v0, b) != hipSuccess) return 0.0;
            return std::max(0.0, std::min((double)ms, (double)tb) - std::max(0.0, (double)ta));
        };
        if (c->screen_ran) c->overlap[1] += under(c->ev[5], c->ev[6]);
        c->overlap[2] += under(c->ev[2], c->ev[4]);
        c->overlap[3] += 1;
    }
    publish_timing(c, B);
    if (any_mod) return run_mod_pass(c, d, sl, r0, nr, rc_out, geom, vres, sl.vit_slot);
    return STRQ_OK;
}

// every sub-batch still in flight, oldest first
static int drain(strq_ctx* c, DetectState* d)
{
    for (int k = 0; k < 2; ++k) { const int rc = harvest(c, d, d->slot[(d->next_slot + k) & 1]); if (rc) return rc; }
    return STRQ_OK;
}

static int run_sub_batch(strq_ctx* c, DetectState* d, int64_t r0, int64_t r1, int64_t next_r1)
{
    Batch& B = d->batch;
    hipStream_t st = c->stream;
    const int nr = (int)(r1 - r0);
    const int esz = B.dtype == 0 ? 2 : 8;
    const int64_t s0 = B.off[r0], tot = B.off[r1] - s0;
    // the slot of this sub-batch (its previous user's results are taken first: normally done a sub-batch ago)
    DetectState::Slot& sl = d->slot[d->next_slot];
    DetectState::Slot& other = d->slot[d->next_slot ^ 1];
    { const int urc = upload_join(d); if (urc) { c->err = "upload of the sub-batch's samples failed (prefetch thread)"; return urc; } }
    { const int hrc = harvest(c, d, sl); if (hrc) return hrc; }
    d->next_slot ^= 1;
    int max_n = 0;
    std::vector<ReadCond> rc(nr);
    std::vector<int64_t> loff(nr + 1);
    for (int i = 0; i < nr; ++i) {
        std::memset(&rc[i], 0, sizeof(ReadCond));
        rc[i].off = B.off[r0 + i] - s0; rc[i].n = (int)(B.off[r0 + i + 1] - B.off[r0 + i]);
        loff[i] = rc[i].off;
        max_n = std::max(max_n, rc[i].n);
        if (B.dtype == 1) {
            rc[i].h2 = (d->ps.M_hi - d->ps.M_lo) / 2; rc[i].c2 = d->ps.M_lo + (d->ps.M_hi - d->ps.M_lo) / 2;
        }
        if (B.dtype == 1 && !B.host_stats.empty()) {
            const double* hs = &B.host_stats[(size_t)(r0 + i) * 6];
            rc[i].med = hs[0]; rc[i].mad = hs[1]; rc[i].f_c1 = hs[2]; rc[i].f_h1 = hs[3]; rc[i].r_c1 = hs[4]; rc[i].r_h1 = hs[5];
            const bool okv = std::isfinite(hs[0]) && hs[1] > 0.0 && std::isfinite(hs[2]) && hs[3] > 0.0 && std::isfinite(hs[3]);
            rc[i].status = okv ? COND_OK : COND_DEGENERATE;
        }
    }
    loff[nr] = tot;
    // both slots are sized together: the first sub-batch on the second slot would otherwise pay a 3 GB hipMalloc in the middle of a run
    for (DetectState::Slot* q : {&sl, &other}) if (q == &sl || !q->active) STRQ_HIP(c, q->flt.reserve((size_t)tot * esz + 64 + 16));
    STRQ_HIP(c, c->levels.reserve((size_t)tot + 64 + 8));
    STRQ_HIP(c, c->level_val.reserve((size_t)nr * 256 * 4));
    STRQ_HIP(c, d->rc.reserve((size_t)nr * sizeof(ReadCond)));
    STRQ_HIP(c, d->hist8.reserve((size_t)nr * 256 * 4));
    if (B.dtype == 0) STRQ_HIP(c, d->hist16.reserve((size_t)nr * 65536 * 4));
    ReadCond* d_rc = d->rc.as<ReadCond>();
    STRQ_HIP(c, hipMemcpyAsync(d_rc, rc.data(), (size_t)nr * sizeof(ReadCond), hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemsetAsync(d->hist8.p, 0, (size_t)nr * 256 * 4, st));
    const char* raw = d->batch.raw.as<char>() + (size_t)s0 * esz;
    // the filtered signal of the sub-batch starts at the same offset inside a 16-byte line as its raw signal, so that the
    // conditioning kernels can move both with aligned 16-byte accesses
    char* const flt_base = sl.flt.as<char>() + (reinterpret_cast<uintptr_t>(raw) & 15);
    bool any_mod = false;
    for (int i = 0; i < nr; ++i) any_mod |= d->targets[B.target[r0 + i]].mod_model_id >= 0;
    // STRQ_SERIAL=1: the Viterbi launches on the context's own stream and their results before the call returns, as up to round 5.
    // (A sub-batch with a modification model is pipelined like any other: its MARK-mode Viterbi launches run under the next sub-batch's
    // alignments; its second pass -- which needs the decoded repeat stretch on the host -- runs when its rows are taken, on the context's
    // stream, which is idle then: the taking thread has just waited for the following sub-batch's forward stage, or is the caller's fetch.)
    const bool serial = strq::opt("STRQ_SERIAL") != nullptr;
    uint32_t* d_hist_raw = nullptr; uint32_t* d_range = nullptr;
    if (B.dtype == 0) {
        STRQ_HIP(c, hipMemsetAsync(d->hist16.p, 0, (size_t)nr * 65536 * 4, st));
        if (any_mod) {
            STRQ_HIP(c, d->hist_raw.reserve((size_t)nr * 65536 * 4));
            STRQ_HIP(c, hipMemsetAsync(d->hist_raw.p, 0, (size_t)nr * 65536 * 4, st));
            d_hist_raw = d->hist_raw.as<uint32_t>();
        }
        STRQ_HIP(c, d->hrange.reserve((size_t)nr * 16));
        STRQ_HIP(c, hipMemsetAsync(d->hrange.p, 0, (size_t)nr * 16, st));
        d_range = d->hrange.as<uint32_t>();
    }
    // Viterbi tasks are grouped by kernel shape over the whole sub-batch (windows of all models with one
    // shape share a launch)
    std::map<int, std::vector<int>> by_shape;
    std::vector<const VitModel*> model_of(nr);
    for (int i = 0; i < nr; ++i) {
        HostModel* hm = c->models[d->targets[B.target[r0 + i]].model_id];
        model_of[i] = hm->dev;
        const int shape = vit_shape_for(hm->h, any_mod ? 2 : 0);
        if (shape < 0) { c->err = "model does not fit a compiled Viterbi kernel"; return STRQ_ERR_UNSUPPORTED; }
        by_shape[shape].push_back(i);
    }
    std::vector<int32_t>& vit_slot = sl.vit_slot;
    vit_slot.assign(nr, 0);
    std::vector<DetectState::Slot::VL>& vls = sl.vls;
    vls.clear();
    { int k = 0;
      for (auto& g : by_shape) {
        int mx = 0;
        for (int i : g.second) mx = std::max(mx, c->models[d->targets[B.target[r0 + i]].model_id]->h.n_cells);
        vls.push_back({g.first, k, (int)g.second.size(), mx});
        for (int i : g.second) vit_slot[i] = k++;
      } }
    const size_t idx_ints = (size_t)nr * 5;        // task_of (2 per read), trim (2 per read), vit_slot
    STRQ_HIP(c, d->idx.reserve(idx_ints * 4 + (size_t)nr * 8 + 64));
    int32_t* d_task_of = d->idx.as<int32_t>(); int32_t* d_trim = d_task_of + 2 * (size_t)nr; int32_t* d_slot = d_trim + 2 * (size_t)nr;
    const VitModel** d_model_of = reinterpret_cast<const VitModel**>(d->idx.as<char>() + ((idx_ints * 4 + 15) & ~(size_t)15));
    STRQ_HIP(c, hipMemcpyAsync(d_model_of, model_of.data(), (size_t)nr * 8, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemcpyAsync(d_slot, vit_slot.data(), (size_t)nr * 4, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, d->geom.reserve((size_t)nr * sizeof(ReadGeom)));
    for (DetectState::Slot* q : {&sl, &other}) {
        if (q != &sl && q->active) continue;          // (in flight: its buffers are in use and large enough for what it holds)
        STRQ_HIP(c, q->vit.reserve((size_t)nr * sizeof(VitTask)));
        STRQ_HIP(c, q->vres.reserve((size_t)nr * sizeof(VitResult)));
        STRQ_HIP(c, q->order.reserve((size_t)nr * 4 + 64));
        STRQ_HIP(c, q->vq.reserve(1024));
        const size_t need = (size_t)nr * (sizeof(ReadGeom) + sizeof(VitResult) + sizeof(ReadCond)) + 64;
        if (need > q->pinned_cap) {
            if (q->pinned) { STRQ_HIP(c, hipHostFree(q->pinned)); q->pinned = nullptr; q->pinned_cap = 0; }
            STRQ_HIP(c, hipHostMalloc(&q->pinned, need + need / 8, hipHostMallocDefault));
            q->pinned_cap = need + need / 8;
        }
        if (!q->fwd_done) {
            STRQ_HIP(c, hipEventCreateWithFlags(&q->fwd_done, hipEventDisableTiming));
            STRQ_HIP(c, hipEventCreate(&q->v0)); STRQ_HIP(c, hipEventCreate(&q->v1));
        }
    }
    {
        if (!d->vit_stream) {
            // the older sub-batch's Viterbi launches go first where both streams have workgroups to place
            int lo = 0, hi = 0;
            STRQ_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
            const char* e = strq::opt("STRQ_VIT_PRIORITY");
            STRQ_HIP(c, hipStreamCreateWithPriority(&d->vit_stream, hipStreamNonBlocking, (e && atoi(e) == 0) ? lo : hi));
        }
    }
    ReadGeom* h_geom = static_cast<ReadGeom*>(sl.pinned);
    VitResult* h_vres = reinterpret_cast<VitResult*>(h_geom + nr);
    ReadCond* h_rc = reinterpret_cast<ReadCond*>(h_vres + nr);
    unsigned int* h_redo = reinterpret_cast<unsigned int*>(h_rc + nr);

    // Conditioning, the two flank alignments and the positions / gate of the reads, in `parts` pieces: a
    // sub-batch whose samples are still in the caller's buffer is uploaded piece by piece, each piece's
    // kernels running under the upload of the next (only the first piece's upload is exposed); a resident
    // or prefetched sub-batch is one piece.  The Viterbi launches below always cover the whole sub-batch.
    int parts = 1;
    if (B.on_host && B.uploaded < r1 && nr >= 1024) { parts = 2; if (const char* e = strq::opt("STRQ_UPLOAD_PARTS")) { const int v = atoi(e); if (v >= 1 && v <= 16) parts = v; } }
    for (int part = 0; part < parts; ++part) {
        const int i0 = (int)((int64_t)nr * part / parts), i1 = (int)((int64_t)nr * (part + 1) / parts), np_ = i1 - i0;
        if (np_ <= 0) continue;
        { const int urc = upload_reads(c, d, r0 + i1); if (urc) return urc; }
        // this sub-batch's samples are in HBM (or queued): the next sub-batch's follow on their own thread from here on
        if (part == parts - 1) upload_prefetch(c, d, next_r1);
        int max_n = 0;
        for (int i = i0; i < i1; ++i) max_n = std::max(max_n, rc[i].n);
        if (part == 0) STRQ_HIP(c, hipEventRecord(d->ev[0], st));
        // ---- conditioning (STRique.py:590-597)
        int bad = 0;
        // levels: the same sample phase as the raw / filtered signal, so that a tile's eight-level groups are 8-byte aligned
        d->levels_shift = (int)((reinterpret_cast<uintptr_t>(raw) & 15) >> (esz == 2 ? 1 : 4));
        uint8_t* levels = c->levels.as<uint8_t>() + d->levels_shift;
        float* level_val = c->level_val.as<float>() + (size_t)i0 * 256;
        uint32_t* hist8 = d->hist8.as<uint32_t>() + (size_t)i0 * 256;
        if (B.dtype == 0) {
            uint32_t* h16 = d->hist16.as<uint32_t>() + (size_t)i0 * 65536;
            uint32_t* hraw = d_hist_raw ? d_hist_raw + (size_t)i0 * 65536 : nullptr;
            uint32_t* rng = d_range + (size_t)i0 * 4;
            bad |= launch_medfilt_hist_i16(st, reinterpret_cast<const int16_t*>(raw), reinterpret_cast<int16_t*>(flt_base), d_rc + i0, np_, max_n, h16, hraw, rng);
            bad |= launch_hist_stats(st, h16, 65536, -32768, d_rc + i0, np_, d->ps, 0, nullptr, rng, 4);
            if (any_mod) bad |= launch_hist_stats(st, hraw, 65536, -32768, d_rc + i0, np_, d->ps, 2, nullptr, rng + 2, 4);
            bad |= launch_quant_morph_i16(st, reinterpret_cast<const int16_t*>(flt_base), levels, d_rc + i0, np_, max_n, hist8);
        } else {
            bad |= launch_medfilt_f64(st, reinterpret_cast<const double*>(raw), reinterpret_cast<double*>(flt_base), d_rc + i0, np_, max_n);
            if (B.host_stats.empty()) {
                // median, MAD and the two 'minmax' maps of every read: one double of scratch per 8192 samples (numpy's mean)
                std::vector<int64_t> first((size_t)np_ + 1, 0);
                for (int i = 0; i < np_; ++i) first[(size_t)i + 1] = first[(size_t)i] + (rc[i0 + i].n + 8191) / 8192;
                const size_t first_bytes = (((size_t)np_ + 1) * 8 + 255) & ~(size_t)255;
                STRQ_HIP(c, d->f64s.reserve(first_bytes + (size_t)first[(size_t)np_] * 8 + 256));
                STRQ_HIP(c, hipMemcpyAsync(d->f64s.p, first.data(), ((size_t)np_ + 1) * 8, hipMemcpyHostToDevice, st));
                bad |= launch_f64_stats(st, reinterpret_cast<const double*>(flt_base), any_mod ? reinterpret_cast<const double*>(raw) : nullptr, d_rc + i0, np_,
                                        reinterpret_cast<double*>(d->f64s.as<char>() + first_bytes), d->f64s.as<int64_t>());
            }
            bad |= launch_quant_morph_f64(st, reinterpret_cast<const double*>(flt_base), levels, d_rc + i0, np_, max_n, hist8);
        }
        bad |= launch_hist_stats(st, hist8, 256, 0, d_rc + i0, np_, d->ps, 1, level_val, nullptr, 0);
        if (bad) { c->err = "conditioning launch failed"; return STRQ_ERR_DEVICE; }
        if (part == 0) STRQ_HIP(c, hipEventRecord(d->ev[1], st));

        // ---- the two flank alignments of every read
        const int na = 2 * np_;
        std::vector<int32_t> a_read(na); std::vector<int> n(na), m(na), k(na), R(na), NS(na); std::vector<const float*> fl(na);
        std::vector<int32_t> trim(na);
        int S = 6;
        for (int j = 0; j < np_; ++j) {
            const Target& t = d->targets[B.target[r0 + i0 + j]];
            S = t.samples;
            a_read[2 * j] = a_read[2 * j + 1] = j;
            n[2 * j] = n[2 * j + 1] = rc[i0 + j].n;
            m[2 * j] = (int)t.prefix_ext.size(); k[2 * j] = t.kp; R[2 * j] = t.Rp; NS[2 * j] = t.NSp; fl[2 * j] = t.prefix_ext.data(); trim[2 * j] = t.trim_prefix;
            m[2 * j + 1] = (int)t.suffix_ext.size(); k[2 * j + 1] = t.ks; R[2 * j + 1] = t.Rs; NS[2 * j + 1] = t.NSs; fl[2 * j + 1] = t.suffix_ext.data(); trim[2 * j + 1] = t.trim_suffix;
        }
        AlignCoreIn ci; AlignCoreOut co;
        ci.nb = na; ci.samples = S; ci.d_levels = levels; ci.read_off = loff.data() + i0; ci.d_level_val = level_val;
        ci.read = a_read.data(); ci.n = n.data(); ci.m = m.data(); ci.k = k.data(); ci.R = R.data(); ci.NS = NS.data(); ci.flank = fl.data();
        // (Launched behind this sub-batch's SCREEN instead -- under the exact pass and the trace, whose launches leave most of the GPU idle --
        // the eight Viterbi waves per CU take the LDS and registers those launches need: exact pass 69 instead of 19 ms, 180 against 175 ms per
        // step on clean reads, 400 against 338 on empirical ones: gpurun_out/r6r.)
        if (part == 0) ci.after_tables = [&]() -> int {
            // The sub-batch before this one: its Viterbi launches start when this sub-batch's conditioning and score tables are through
            // -- a few ms of HBM-bound streaming kernels that crawl next to a GPU full of Viterbi waves (gpurun_out/r6d: 66 ms instead
            // of 5.7 ms; hist_stats_kernel alone keeps 60 KB of LDS per workgroup), and queued before the table kernel the Viterbi
            // launch kept the next screen from being dispatched until it had ended (gpurun_out/r6h against r6i, measured, both
            // priorities).  The alignment kernels that follow share the SIMDs with the Viterbi waves.
            const bool queued_now = other.launch_pending;
            const int lrc = launch_viterbi_of(c, d, other, d->vit_stream, c->ev[1]); if (lrc) return lrc;
            // ... and they are dispatched after them: persistent workgroups that fill every CU for the length of the screen would
            // otherwise win the race now and then, and the Viterbi workgroups (eight waves of 192 VGPRs each: a whole CU's worth at
            // once) could not be placed before the screen has ended -- the serial order again
            if (queued_now) STRQ_HIP(c, hipStreamWaitEvent(st, other.v0, 0));
            return STRQ_OK;
        };
        const int rcode = align_core(c, ci, co);
        if (rcode) return rcode;
        B.n_hard += co.n_hard; B.n_fwd_launches += co.n_launches;
        c->counters[0] += co.wave_steps; c->counters[1] += co.columns; c->counters[2] += na;
        c->counters[3] = co.segs; c->counters[4] = co.tables; c->counters[5] = co.packed; c->counters[6] = co.rows_per_lane;

        // ---- positions, gate, Viterbi tasks
        std::vector<int32_t> task_of(na);
        for (int pos = 0; pos < na; ++pos) task_of[co.order[pos]] = pos;
        STRQ_HIP(c, hipMemcpyAsync(d_task_of + 2 * (size_t)i0, task_of.data(), (size_t)na * 4, hipMemcpyHostToDevice, st));
        STRQ_HIP(c, hipMemcpyAsync(d_trim + 2 * (size_t)i0, trim.data(), (size_t)na * 4, hipMemcpyHostToDevice, st));
        FinalizeArgs fa;
        fa.tasks = co.d_tasks; fa.results = co.d_results; fa.task_of = d_task_of + 2 * (size_t)i0; fa.trim = d_trim + 2 * (size_t)i0; fa.vit_slot = d_slot + i0;
        fa.rc = d_rc + i0; fa.model_of = d_model_of + i0; fa.flt = flt_base; fa.is_f64 = B.dtype; fa.ps = d->ps;
        fa.geom = d->geom.as<ReadGeom>() + i0; fa.vit = sl.vit.as<VitTask>(); fa.n_reads = np_;
        hipLaunchKernelGGL(finalize_kernel, dim3((np_ + 127) / 128), dim3(128), 0, st, fa);
        STRQ_HIP(c, hipGetLastError());
    }
    // positions and conditioning status of the sub-batch to the host (pinned: the copies do not block), then the fork: everything the
    // Viterbi launches read is final behind `fwd_done`
    STRQ_HIP(c, hipMemcpyAsync(h_geom, d->geom.p, (size_t)nr * sizeof(ReadGeom), hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipMemcpyAsync(h_rc, d_rc, (size_t)nr * sizeof(ReadCond), hipMemcpyDeviceToHost, st));
    *h_redo = 0;
    if (c->redo_total.p) STRQ_HIP(c, hipMemcpyAsync(h_redo, c->redo_total.p, 4, hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipEventRecord(sl.fwd_done, st));
    sl.vit_mode = any_mod ? 2 : 0;
    sl.active = true; sl.launch_pending = true; sl.r0 = r0; sl.nr = nr;
    // The Viterbi launches of this sub-batch: now on the context's stream (serial order), or -- two sub-batches in flight -- behind the
    // conditioning of the NEXT sub-batch (launch_viterbi_of from there), at the latest when somebody asks for the rows.  Conditioning is a
    // few ms of HBM-bound streaming kernels that crawl next to a GPU full of Viterbi waves (gpurun_out/r6d: 66 ms instead of 5.7 ms); the
    // flank-alignment kernels that follow share the SIMDs with them at little cost.
    if (serial) { const int lrc = launch_viterbi_of(c, d, sl, st, nullptr); if (lrc) return lrc; }
    // the forward stage of this sub-batch is complete here (its Viterbi launches need not be)
    STRQ_HIP(c, hipEventSynchronize(sl.fwd_done));
    c->second_round[0] = (int64_t)*h_redo - c->look2_served; c->second_round[1] += 2 * (int64_t)nr;
    // score distribution of this sub-batch for the overlap planning of the next one (align_core)
    if (c->ap.dist_offset > 0.0f) {
        c->score_fracs.clear(); double sum_n = 0;
        for (int i = 0; i < nr; ++i) {
            if (rc[i].n <= 0) continue;
            const Target& t = d->targets[B.target[r0 + i]];
            c->score_fracs.push_back(h_geom[i].best_prefix / ((float)t.prefix_ext.size() * c->ap.dist_offset));
            c->score_fracs.push_back(h_geom[i].best_suffix / ((float)t.suffix_ext.size() * c->ap.dist_offset));
            sum_n += rc[i].n;
        }
        std::sort(c->score_fracs.begin(), c->score_fracs.end());
        c->mean_n = c->score_fracs.empty() ? 0.0 : sum_n / (double)(c->score_fracs.size() / 2);
    }
    float ms;
    STRQ_HIP(c, hipEventElapsedTime(&ms, d->ev[0], d->ev[1])); B.t_cond += ms;
    const int rcode = align_core_times(c, &B.t_lut, &B.t_fwd, &B.t_trace);      // of the last piece when the sub-batch ran in pieces
    if (rcode) return rcode;
    // results: of the sub-batch before this one (its Viterbi launches ran under this sub-batch's forward stage) -- or, serially, of this one
    if (strq::opt("STRQ_DEBUG") && !serial && other.active && !other.launch_pending && c->screen_ran) {
        float a = 0, b = 0, e = 0;
        (void)hipEventSynchronize(other.v1);
        (void)hipEventElapsedTime(&a, other.v0, c->ev[5]); (void)hipEventElapsedTime(&e, other.v0, c->ev[6]); (void)hipEventElapsedTime(&b, other.v0, other.v1);
        STRQ_DBG("overlap: Viterbi launches of the previous sub-batch start at 0, end at %.1f ms; this sub-batch's screen runs from %.1f to %.1f ms", b, a, e);
    }
    if (serial) return harvest(c, d, sl);
    return harvest(c, d, other, /*under_current=*/true);
}

}  // namespace strq

extern "C" {

int strq_set_pore_stats(strq_ctx* c, double tail_lo, double tail_hi, double model_min, double model_max)
{
    if (!c) return STRQ_ERR_ARG;
    DetectState* d = dstate(c);
    d->ps.M_lo = tail_lo; d->ps.M_hi = tail_hi; d->ps.clip_lo = model_min + .5; d->ps.clip_hi = model_max - .5;
    d->have_ps = true;
    return STRQ_OK;
}

int strq_target_add(strq_ctx* c, const float* prefix_ext, int64_t m_prefix, const float* suffix_ext, int64_t m_suffix,
                    int32_t trim_prefix, int32_t trim_suffix, int32_t samples, int32_t hmm_model_id, int32_t count_bias,
                    int32_t* target_id)
{
    strq::CtxScope scope_(c);
    if (!c) return STRQ_ERR_ARG;
    if (!prefix_ext || !suffix_ext || !target_id || hmm_model_id < 0 || hmm_model_id >= (int32_t)c->models.size() ||
        trim_prefix < 0 || trim_suffix < 0 || trim_prefix >= m_prefix || trim_suffix >= m_suffix) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    DetectState* d = dstate(c);
    Target t;
    int rc = align_validate_flank(c, prefix_ext, m_prefix, samples, &t.kp, &t.Rp, &t.NSp); if (rc) return rc;
    rc = align_validate_flank(c, suffix_ext, m_suffix, samples, &t.ks, &t.Rs, &t.NSs); if (rc) return rc;
    t.prefix_ext.assign(prefix_ext, prefix_ext + m_prefix); t.suffix_ext.assign(suffix_ext, suffix_ext + m_suffix);
    t.trim_prefix = trim_prefix; t.trim_suffix = trim_suffix; t.samples = align_effective_samples(samples); t.model_id = hmm_model_id; t.count_bias = count_bias;
    d->targets.push_back(t);
    *target_id = (int32_t)d->targets.size() - 1;
    return STRQ_OK;
}

int strq_target_set_mod(strq_ctx* c, int32_t target_id, int32_t mod_model_id, double mod_min, double mod_max)
{
    if (!c) return STRQ_ERR_ARG;
    DetectState* d = dstate(c);
    if (target_id < 0 || target_id >= (int32_t)d->targets.size() || mod_model_id < 0 || mod_model_id >= (int32_t)c->models.size()) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    d->targets[target_id].mod_model_id = mod_model_id; d->targets[target_id].mod_min = mod_min; d->targets[target_id].mod_max = mod_max;
    return STRQ_OK;
}

int strq_batch_fetch_mod(strq_ctx* c, char* pool, int64_t pool_cap, int64_t* off)
{
    if (!c || !off) return STRQ_ERR_ARG;
    DetectState* d = dstate(c);
    { const int rc = drain(c, d); if (rc) return rc; }
    int64_t pos = 0;
    for (size_t i = 0; i < d->batch.mod.size(); ++i) {
        off[i] = pos;
        const std::string& m = d->batch.mod[i];
        if (pool) { if (pos + (int64_t)m.size() > pool_cap) { c->err = "pattern pool too small"; return STRQ_ERR_ARG; } std::memcpy(pool + pos, m.data(), m.size()); }
        pos += (int64_t)m.size();
    }
    off[d->batch.mod.size()] = pos;
    return STRQ_OK;
}

static int batch_prepare(strq_ctx* c, int64_t n_reads, const void* signals, int32_t dtype, const int64_t* offsets,
                         const int32_t* target_id, const double* host_stats, bool lazy, const void* const* reads = nullptr)
{
    if (!c) return STRQ_ERR_ARG;
    DetectState* d = dstate(c);
    if (n_reads < 0 || (n_reads > 0 && ((!signals && !reads) || !offsets || !target_id)) || (dtype != 0 && dtype != 1)) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    if (!d->have_ps) { c->err = "strq_set_pore_stats has not been called"; return STRQ_ERR_ARG; }
    STRQ_HIP(c, hipSetDevice(c->device));
    { const int rc = drain(c, d); if (rc) return rc; }          // sub-batches of the previous batch still in flight
    Batch& B = d->batch;
    B.n_reads = n_reads; B.dtype = dtype;
    B.off.assign(offsets, offsets + n_reads + 1);
    B.target.assign(target_id, target_id + n_reads);
    for (int64_t i = 0; i < n_reads; ++i) {
        if (B.target[i] < 0 || B.target[i] >= (int32_t)d->targets.size()) { c->err = "unknown target id"; return STRQ_ERR_ARG; }
        if (B.off[i + 1] < B.off[i] || B.off[i + 1] - B.off[i] > ((int64_t)1 << 30)) { c->err = "bad offsets"; return STRQ_ERR_ARG; }
    }
    B.host_stats.clear();
    // float64 reads have no exact histogram: their order statistics come from a radix selection on the GPU (cond_kernels.hip:
    // f64_stats_kernel) unless the caller hands over its own (numpy's) six scalars per read
    if (dtype == 1 && host_stats) B.host_stats.assign(host_stats, host_stats + n_reads * 6);
    const size_t bytes = (size_t)(n_reads ? B.off[n_reads] : 0) * (dtype == 0 ? 2 : 8);
    STRQ_HIP(c, B.raw.reserve(bytes + 64));
    B.forget_host();
    B.host_src = static_cast<const char*>(signals); B.uploaded = 0; B.on_host = true;
    if (reads) B.host_reads.assign(reinterpret_cast<const char* const*>(reads), reinterpret_cast<const char* const*>(reads) + n_reads);
    if (!lazy) {
        const int rc = upload_reads(c, d, n_reads);      // resident batch: everything now
        if (rc) return rc;
        B.forget_host();
    }
    B.results.assign((size_t)n_reads, strq_result());
    B.mod.assign((size_t)n_reads, std::string("-"));
    if (!d->ev_ok) { for (auto& e : d->ev) STRQ_HIP(c, hipEventCreate(&e)); d->ev_ok = true; }
    return STRQ_OK;
}

int strq_batch_upload(strq_ctx* c, int64_t n_reads, const void* signals, int32_t dtype, const int64_t* offsets,
                      const int32_t* target_id, const double* host_stats)
{
    strq::CtxScope scope_(c);
    return batch_prepare(c, n_reads, signals, dtype, offsets, target_id, host_stats, false);
}

int strq_batch_upload_part(strq_ctx* c, int64_t total_reads, int64_t total_samples, int64_t first_read, int64_t n_reads,
                           const void* signals, int32_t dtype, const int64_t* offsets, const int32_t* target_id)
{
    if (!c) return STRQ_ERR_ARG;
    strq::CtxScope scope_(c);
    DetectState* d = dstate(c);
    if (total_reads < 0 || total_samples < 0 || first_read < 0 || n_reads < 0 || first_read + n_reads > total_reads || dtype != 0 ||
        (n_reads > 0 && (!signals || !offsets || !target_id))) { c->err = "bad argument (strq_batch_upload_part takes int16 reads)"; return STRQ_ERR_ARG; }
    if (!d->have_ps) { c->err = "strq_set_pore_stats has not been called"; return STRQ_ERR_ARG; }
    STRQ_HIP(c, hipSetDevice(c->device));
    Batch& B = d->batch;
    if (first_read == 0) {
        // a new resident batch: device memory for all of it now, the parts follow in order
        { const int rc = drain(c, d); if (rc) return rc; }
        B.forget_host();
        B.n_reads = total_reads; B.dtype = dtype;
        B.off.assign((size_t)total_reads + 1, 0); B.target.assign((size_t)total_reads, 0);
        B.host_stats.clear();
        STRQ_HIP(c, B.raw.reserve((size_t)total_samples * 2 + 64));      // total_samples is a hint: the buffer grows (below) when the parts hold more
        B.results.assign((size_t)total_reads, strq_result());
        B.mod.assign((size_t)total_reads, std::string("-"));
        B.uploaded = 0; B.on_host = false;
        d->part_reads = 0;
        if (!d->ev_ok) { for (auto& e : d->ev) STRQ_HIP(c, hipEventCreate(&e)); d->ev_ok = true; }
    }
    if (B.n_reads != total_reads || d->part_reads != first_read) { c->err = "parts of a resident batch must follow each other, first_read = reads uploaded so far"; return STRQ_ERR_ARG; }
    if (n_reads == 0) return STRQ_OK;
    const int64_t base = B.off[(size_t)first_read];
    for (int64_t i = 0; i < n_reads; ++i) {
        const int64_t len = offsets[i + 1] - offsets[i];
        if (target_id[i] < 0 || target_id[i] >= (int32_t)d->targets.size()) { c->err = "unknown target id"; return STRQ_ERR_ARG; }
        if (len < 0 || len > ((int64_t)1 << 30)) { c->err = "bad offsets"; return STRQ_ERR_ARG; }
        B.off[(size_t)(first_read + i + 1)] = base + (offsets[i + 1] - offsets[0]);
        B.target[(size_t)(first_read + i)] = target_id[i];
    }
    {
        const size_t need = (size_t)B.off[(size_t)(first_read + n_reads)] * 2 + 64;
        if (need > B.raw.cap) {          // more samples than announced: a larger buffer, the parts uploaded so far move over
            DevBuf bigger;
            STRQ_HIP(c, bigger.reserve(need + need / 2));
            if (base > 0) {
                const hipError_t e = hipMemcpy(bigger.p, B.raw.p, (size_t)base * 2, hipMemcpyDeviceToDevice);
                if (e != hipSuccess) { bigger.release(); c->err = std::string("hipMemcpy (growing the resident batch): ") + hipGetErrorString(e); return STRQ_ERR_DEVICE; }
            }
            B.raw.release();
            B.raw = bigger;
        }
    }
    for (int64_t i = first_read + n_reads; i < total_reads; ++i) B.off[(size_t)i + 1] = B.off[(size_t)(first_read + n_reads)];      // reads not yet uploaded: empty
    // the staging ring of upload_reads addresses the caller's buffer by the batch's own byte positions
    B.host_src = static_cast<const char*>(signals) + (size_t)offsets[0] * 2 - (size_t)base * 2;
    B.host_reads.clear(); B.uploaded = first_read; B.on_host = true;
    const int rc = upload_reads(c, d, first_read + n_reads);
    B.forget_host();
    if (rc) return rc;
    d->part_reads = first_read + n_reads;
    return STRQ_OK;
}

int strq_batch_run(strq_ctx* c)
{
    if (!c) return STRQ_ERR_ARG;
    return strq_batch_run_range(c, 0, dstate(c)->batch.n_reads);
}

int strq_batch_run_range(strq_ctx* c, int64_t first, int64_t last)
{
    strq::CtxScope scope_(c);
    if (!c) return STRQ_ERR_ARG;
    DetectState* d = dstate(c);
    Batch& B = d->batch;
    if (first < 0 || last < first || last > B.n_reads) { c->err = "read range outside the uploaded batch"; return STRQ_ERR_ARG; }
    STRQ_HIP(c, hipSetDevice(c->device));
    B.t_cond = B.t_lut = B.t_fwd = B.t_trace = B.t_vit = 0; B.n_hard = 0; B.n_fwd_launches = 0;
    std::fill(c->counters, c->counters + 8, 0.0);
    std::fill(c->overlap, c->overlap + 4, 0.0);
    c->second_round[0] = c->second_round[1] = 0; c->look2_served = 0;
    for (double& v : c->screen_stats) v = 0;
    STRQ_HIP(c, c->redo_total.reserve(64));
    STRQ_HIP(c, hipMemsetAsync(c->redo_total.p, 0, 64, c->stream));
    // partition into sub-batches first, so that the upload of piece k + 1 can overlap the kernels of piece k
    std::vector<int64_t> cuts(1, first);
    int64_t r0 = first;
    const int64_t n_end = last;
    while (r0 < n_end) {
        int64_t r1 = r0; size_t ck = 0; int64_t samples = 0;
        bool mod_batch = false;
        // Sub-batch size: 16 reads (32 alignments) per CU.  The alignments of a batch are about equally
        // long, so the forward DP proceeds in rounds: 32 per CU is four full rounds of the eight waves the
        // 24-bit tables allow (and, with six float32 waves -- two alone on their SIMD at 60.6 ms per
        // alignment, four sharing one at 73 ms -- 6 x 60.6 = 5 x 73 ends without a ragged tail as well;
        // 4608 reads measured 437 ms against 365 ms for 4096).
        const int64_t full = std::min<int64_t>(16 * (int64_t)c->n_cu, 8192);      // 8192: task limit of vit_sort_kernel
        int64_t full_env = 0;
        if (const char* e = strq::opt("STRQ_SUBBATCH_READS")) full_env = atoll(e);      // testing: force small sub-batches
        for (int64_t r = r0; r < n_end && r < r0 + full; ++r) mod_batch |= d->targets[B.target[r]].mod_model_id >= 0;
        const int64_t cap = full_env > 0 ? std::min<int64_t>(full_env, 8192) : full;      // (the modification pass keeps back-pointers of the small dual model only: ~3 MB per 50 kb read)
        (void)mod_batch;
        while (r1 < n_end && r1 - r0 < cap) {
            const Target& t = d->targets[B.target[r1]];
            const int n = (int)(B.off[r1 + 1] - B.off[r1]);
            const size_t need = align_workspace_bytes(n, 0, t.Rp, t.NSp) + align_workspace_bytes(n, 0, t.Rs, t.NSs);
            if (r1 > r0 && (ck + need > c->max_ws_bytes || samples + n > ((int64_t)3 << 30))) break;
            ck += need; samples += n; ++r1;
        }
        cuts.push_back(r1);
        r0 = r1;
    }
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
        const double t1 = now_s();
        // samples not yet in HBM (first sub-batch of strq_detect_batch) are uploaded piece by piece inside
        const int rc = run_sub_batch(c, d, cuts[k], cuts[k + 1], k + 2 < cuts.size() ? cuts[k + 2] : cuts[k + 1]);
        if (rc) { (void)upload_join(d); return rc; }          // (the prefetch thread reads the caller's buffers: never left running)
        STRQ_DBG("sub-batch %zu: reads %ld..%ld  %.1f ms", k, (long)cuts[k], (long)cuts[k + 1], (now_s() - t1) * 1e3);
    }
    { const int urc = upload_join(d); if (urc) { c->err = "upload of the batch's samples failed (prefetch thread)"; return urc; } }
    B.forget_host();      // the caller's buffers are not referenced after the call
    publish_timing(c, B);
    return STRQ_OK;
}

int strq_batch_fetch(strq_ctx* c, strq_result* out)
{
    if (!c || !out) return STRQ_ERR_ARG;
    strq::CtxScope scope_(c);
    DetectState* d = dstate(c);
    STRQ_HIP(c, hipSetDevice(c->device));
    { const int rc = drain(c, d); if (rc) return rc; }
    std::memcpy(out, d->batch.results.data(), d->batch.results.size() * sizeof(strq_result));
    return STRQ_OK;
}

int strq_batch_fetch_range(strq_ctx* c, int64_t first, int64_t last, strq_result* out)
{
    if (!c || !out) return STRQ_ERR_ARG;
    strq::CtxScope scope_(c);
    DetectState* d = dstate(c);
    if (first < 0 || last < first || last > d->batch.n_reads) { c->err = "read range outside the uploaded batch"; return STRQ_ERR_ARG; }
    STRQ_HIP(c, hipSetDevice(c->device));
    // only the sub-batches in flight that hold reads of the range are waited for (oldest first): a caller that runs range k + 1
    // before it fetches range k never waits for the Viterbi launches of k + 1
    for (int k = 0; k < 2; ++k) {
        DetectState::Slot& sl = d->slot[(d->next_slot + k) & 1];
        if (sl.active && sl.r0 < last && sl.r0 + sl.nr > first) { const int rc = harvest(c, d, sl); if (rc) return rc; }
    }
    std::memcpy(out, d->batch.results.data() + first, (size_t)(last - first) * sizeof(strq_result));
    return STRQ_OK;
}

int strq_detect_batch(strq_ctx* c, int64_t n_reads, const void* signals, int32_t dtype, const int64_t* offsets,
                      const int32_t* target_id, const double* host_stats, strq_result* out)
{
    strq::CtxScope scope_(c);
    // signals stay in the caller's buffer and are uploaded one sub-batch ahead of the kernels
    int rc = batch_prepare(c, n_reads, signals, dtype, offsets, target_id, host_stats, true);
    if (rc) return rc;
    rc = strq_batch_run(c);
    dstate(c)->batch.forget_host();
    if (rc) return rc;
    return strq_batch_fetch(c, out);
}

int strq_detect_batch_reads(strq_ctx* c, int64_t n_reads, const void* const* reads, const int64_t* lengths, int32_t dtype,
                            const int32_t* target_id, const double* host_stats, strq_result* out)
{
    strq::CtxScope scope_(c);
    // one buffer per read (what a caller holding a list of arrays has): no concatenated copy on the host, the staging
    // threads gather straight from the reads
    if (!c) return STRQ_ERR_ARG;
    if (n_reads < 0 || (n_reads > 0 && (!reads || !lengths))) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    std::vector<int64_t> off((size_t)n_reads + 1, 0);
    for (int64_t i = 0; i < n_reads; ++i) {
        if (lengths[i] < 0 || (lengths[i] > 0 && !reads[i])) { c->err = "bad argument"; return STRQ_ERR_ARG; }
        off[(size_t)i + 1] = off[(size_t)i] + lengths[i];
    }
    int rc = batch_prepare(c, n_reads, nullptr, dtype, off.data(), target_id, host_stats, true, reads);
    if (rc) return rc;
    rc = strq_batch_run(c);
    dstate(c)->batch.forget_host();
    if (rc) return rc;
    return strq_batch_fetch(c, out);
}

// conditioning outputs of the last sub-batch (parity tests of STRique.py:590-597): levels of read
// `read` (index inside the last sub-batch), its 256 level values and the scalars.
int strq_debug_conditioning(strq_ctx* c, int64_t read, uint8_t* levels, int64_t n, float* level_val, double* scalars10)
{
    if (!c) return STRQ_ERR_ARG;
    DetectState* d = dstate(c);
    { const int drc = drain(c, d); if (drc) return drc; }
    ReadCond rc;
    STRQ_HIP(c, hipMemcpy(&rc, d->rc.as<ReadCond>() + read, sizeof(rc), hipMemcpyDeviceToHost));
    if (levels) STRQ_HIP(c, hipMemcpy(levels, c->levels.as<uint8_t>() + d->levels_shift + rc.off, (size_t)std::min<int64_t>(n, rc.n), hipMemcpyDeviceToHost));
    if (level_val) STRQ_HIP(c, hipMemcpy(level_val, c->level_val.as<float>() + read * 256, 1024, hipMemcpyDeviceToHost));
    if (scalars10) {
        const double v[10] = {rc.med, rc.mad, rc.f_c1, rc.f_h1, rc.m_c1, rc.m_h1, rc.r_c1, rc.r_h1, rc.h2, rc.c2};
        std::memcpy(scalars10, v, sizeof(v));
    }
    return STRQ_OK;
}

}  // extern "C"
