// C ABI: context, alignment parameters, align_overlap and its batched form.
// Host-side orchestration only; the arithmetic lives in align_kernels.hip / lut_kernels.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <climits>
#include <cstring>
#include <cstdlib>
#include <time.h>
#include <unistd.h>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <numeric>
#include <vector>
#include "../../include/strique_hip.h"
#include "strq_ctx.h"
#include "align_kernels.h"
#include "lut_kernels.h"
#include "align_generic.h"

using namespace strq;

#define STRQ_DBG(...) do { if (strq::opt("STRQ_DEBUG")) { fprintf(stderr, "[strq] " __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)

namespace strq {

// ---- switches (strq_opt.h)
static thread_local const strq_ctx* t_cur_ctx = nullptr;
static std::mutex g_opt_mu;
static std::map<std::string, std::string> g_opts;
CtxScope::CtxScope(const strq_ctx* c) : prev(t_cur_ctx) { t_cur_ctx = c; }
CtxScope::~CtxScope() { t_cur_ctx = prev; }
// The value is COPIED under the lock into one of a few thread-local strings (a strq_set_option on another thread may rehash or free
// the entry the moment the lock is released; several contexts with a host thread each is a supported way to use the library): the
// pointer returned stays valid until the calling thread has asked for eight more switches.
const char* opt(const char* key)
{
    static thread_local std::string ring[8];
    static thread_local unsigned ring_pos = 0;
    auto keep = [&](const std::string& v) -> const char* {
        if (v.empty()) return nullptr;
        std::string& slot = ring[ring_pos++ & 7u];
        slot = v;
        return slot.c_str();
    };
    if (t_cur_ctx) {
        std::lock_guard<std::mutex> lk(t_cur_ctx->options_mu);
        auto it = t_cur_ctx->options.find(key);
        if (it != t_cur_ctx->options.end()) return keep(it->second);
    }
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        auto it = g_opts.find(key);
        if (it != g_opts.end()) return keep(it->second);
    }
    return getenv(key);
}

// Score<float,Distance>::score, evaluated with the host libm exactly like the reference
// (src/score_distance.h:117-122)
float host_cell_score(const AlignParams& p, float h, float v)
{
    const float d = h > v ? h - v : v - h;
    const float s = p.dist_offset - (float)std::pow((double)d, 1.2);
    return s > p.dist_min ? s : p.dist_min;
}

// S[i][0] of the DP: column 0 is not free
void host_col0(const AlignParams& p, int m, float* out)
{
    const float NINF = -3.4028234663852886e38f / 2;
    out[0] = 0.0f;
    float vprev = NINF, sprev = 0.0f;
    for (int i = 1; i <= m; ++i) {
        const float ext = vprev + p.ext_v, opn = sprev + p.open_v;
        const float v = ext >= opn ? ext : opn;
        out[i] = v; vprev = v; sprev = v;
    }
}

struct BatchIn {
    int64_t n_align, n_reads;
    const uint8_t* levels; const int64_t* read_off; const float* level_val;
    const int32_t* align_read; const float* flank; const int64_t* flank_off; int32_t samples;
};
struct BatchOut { float* score; int64_t* j_end; int64_t* j0; int32_t* rec; };

// One sub-batch of alignments whose reads (levels, level values) are already in HBM.
// Leaves AlignTask / AlignResult / rec on the device in task order (`out.order[pos]` = index of
// the alignment handled by task `pos`).
static int align_core_once(strq_ctx* c, const AlignCoreIn& in, AlignCoreOut& out, bool allow_bail, bool* bailed);

// One attempt -- or two: a sub-batch on which the coarse screen's first look certifies less than two thirds of the alignments is not
// worth its second look (on such reads the chunks that reach the score found cover most of the read: gpurun_out/r5r, r6d -- 650 ms for a
// forward stage the fine screen does in 253).  The attempt stops there, the coarse screen pauses, and the sub-batch starts over with the
// fine screen: ~150 ms lost to the retry instead of ~390.
int align_core(strq_ctx* c, const AlignCoreIn& in, AlignCoreOut& out)
{
    double saved[8];
    std::memcpy(saved, c->screen_stats, sizeof(saved));
    bool bailed = false;
    const int rc = align_core_once(c, in, out, true, &bailed);
    if (rc || !bailed) return rc;
    std::memcpy(c->screen_stats, saved, sizeof(saved));
    return align_core_once(c, in, out, false, &bailed);
}

static int align_core_once(strq_ctx* c, const AlignCoreIn& in, AlignCoreOut& out, bool allow_bail, bool* bailed)
{
    const int nb = in.nb, S = in.samples;
    hipStream_t st = c->stream;
    size_t cls_tot = 0, col0_tot = 0, rec_tot = 0, tab_tot = 0, bnd_floats = 0, desc_tot = 0;
    // cls_off / col0_off: flank class values and column 0 of the DP are shared by all alignments of
    // the same flank / flank length (a batch has a handful of distinct ones); desc_off: per-alignment
    // band descriptors
    std::vector<size_t> cls_off(nb), col0_off(nb), bnd_off(nb);
    std::map<const float*, size_t> cls_of_flank; std::map<int, size_t> col0_of_m;
    std::vector<int> cls_first, col0_first;        // first alignment that uses each shared array
    out.rec_off.assign(nb, 0);
    // Score-table jobs.  A flank of at most STRQ_LUT_MAX_K classes has one table; a longer one (more than
    // 948 samples at 6 per k-mer) runs as several strips of 64 x R rows and gets one table per strip, so
    // that a table -- full width in the worst case -- always fits the LDS of the kernel that builds it.
    std::vector<int> J0(nb), NJ(nb);               // first job / number of jobs of an alignment
    std::vector<int> job_align, job_k0, job_k;
    std::vector<size_t> tab_off, desc_off;
    int max_k = 0;
    for (int i = 0; i < nb; ++i) {
        auto fc = cls_of_flank.find(in.flank[i]);
        if (fc == cls_of_flank.end()) { fc = cls_of_flank.emplace(in.flank[i], cls_tot).first; cls_tot += in.k[i]; cls_first.push_back(i); }
        cls_off[i] = fc->second;
        auto f0 = col0_of_m.find(in.m[i]);
        if (f0 == col0_of_m.end()) { f0 = col0_of_m.emplace(in.m[i], col0_tot).first; col0_tot += in.m[i] + 1; col0_first.push_back(i); }
        col0_off[i] = f0->second;
        J0[i] = (int)job_align.size();
        NJ[i] = in.k[i] > STRQ_LUT_MAX_K ? in.NS[i] : 1;
        for (int sidx = 0; sidx < NJ[i]; ++sidx) {
            int k0 = 0, kn = in.k[i];
            if (NJ[i] > 1) {
                const int row0 = sidx * 64 * in.R[i], rows = std::min(64 * in.R[i], in.m[i] - row0);
                k0 = row0 / S; kn = (row0 + rows - 1) / S - k0 + 1;
            }
            job_align.push_back(i); job_k0.push_back(k0); job_k.push_back(kn);
            desc_off.push_back(desc_tot); desc_tot += kn;
            tab_off.push_back(tab_tot); tab_tot += STRQ_TABLE_SLOT_FLOATS(kn);
            max_k = std::max(max_k, kn);
        }
        out.rec_off[i] = rec_tot; rec_tot += in.m[i];
        bnd_off[i] = bnd_floats; if (in.NS[i] > 1) bnd_floats += (size_t)(in.NS[i] - 1) * 2 * ((size_t)in.n[i] + 2);
    }
    const int nj = (int)job_align.size();
    out.rec_total = rec_tot;
    std::vector<float> h_cls(cls_tot), h_col0(col0_tot);
    for (int i : cls_first) {
        const float* f = in.flank[i];
        for (int kk = 0; kk < in.k[i]; ++kk) h_cls[cls_off[i] + kk] = f[(size_t)kk * S];
    }
    for (int i : col0_first) host_col0(c->ap, in.m[i], &h_col0[col0_off[i]]);
    STRQ_HIP(c, c->flank_cls.reserve(cls_tot * 4));
    STRQ_HIP(c, c->col0.reserve(col0_tot * 4));
    STRQ_HIP(c, c->band_lo.reserve(desc_tot * 4));
    STRQ_HIP(c, c->tables.reserve(tab_tot * 4));
    STRQ_HIP(c, c->tables3.reserve(tab_tot * 3 + (size_t)nj * 8 + 64));
    STRQ_HIP(c, c->bnd.reserve(bnd_floats * 4 + 256));
    STRQ_HIP(c, c->rec.reserve(rec_tot * 4 + 256));
    STRQ_HIP(c, c->lutinfo.reserve((size_t)nj * (sizeof(LutJob) + sizeof(LutInfo))));
    const int hard_cap = 1 << 16;
    STRQ_HIP(c, c->hard.reserve((size_t)hard_cap * (sizeof(HardEntry) + 4) + 64));
    STRQ_HIP(c, c->queue.reserve(STRQ_QUEUE_BYTES));
    STRQ_HIP(c, hipMemcpyAsync(c->flank_cls.p, h_cls.data(), cls_tot * 4, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemcpyAsync(c->col0.p, h_col0.data(), col0_tot * 4, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemsetAsync(c->queue.p, 0, STRQ_QUEUE_BYTES, st));

    // ---- score tables
    std::vector<LutJob> jobs(nj);
    LutJob* d_jobs = c->lutinfo.as<LutJob>();
    LutInfo* d_info = reinterpret_cast<LutInfo*>(d_jobs + nj);
    for (int j = 0; j < nj; ++j) {
        const int i = job_align[j];
        jobs[j].level_val = in.d_level_val + (size_t)in.read[i] * 256;
        jobs[j].cls_val = c->flank_cls.as<float>() + cls_off[i] + job_k0[j];
        jobs[j].table = c->tables.as<float>() + tab_off[j];
        jobs[j].table3 = c->tables3.as<uint8_t>() + tab_off[j] * 3 + (size_t)j * 8 - (tab_off[j] * 3 + (size_t)j * 8) % 4;      // 4-byte aligned slot
        jobs[j].band_lo = c->band_lo.as<int32_t>() + desc_off[j];
        jobs[j].k = job_k[j]; jobs[j].pad_ = 0;
    }
    STRQ_HIP(c, hipMemcpyAsync(d_jobs, jobs.data(), (size_t)nj * sizeof(LutJob), hipMemcpyHostToDevice, st));
    HardEntry* d_hard = c->hard.as<HardEntry>();
    float* d_hard_vals = reinterpret_cast<float*>(d_hard + hard_cap);
    int* d_hard_count = c->queue.as<int>();
    STRQ_HIP(c, hipEventRecord(c->ev[0], st));
    if (launch_lut_build(st, d_jobs, d_info, nj, max_k, d_hard, d_hard_count, hard_cap, c->ap)) { c->err = "lut launch failed"; return STRQ_ERR_DEVICE; }
    STRQ_HIP(c, hipEventRecord(c->ev[1], st));
    if (in.after_tables) { const int arc = in.after_tables(); if (arc) return arc; }
    std::vector<LutInfo> info(nj);
    int hard_count = 0;
    STRQ_HIP(c, hipMemcpyAsync(info.data(), d_info, (size_t)nj * sizeof(LutInfo), hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipMemcpyAsync(&hard_count, d_hard_count, 4, hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipStreamSynchronize(st));
    STRQ_DBG("lut done nb=%d hard=%d floats0=%d", nb, hard_count, info[0].total);
    if (strq::opt("STRQ_DEBUG")) { int hist[9] = {0}; for (int i = 0; i < nj; ++i) hist[std::min(8, info[i].need / 8)]++; STRQ_DBG("band need histogram (x8 levels): %d %d %d %d %d %d %d %d %d", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8]); }
    if (strq::opt("STRQ_DEBUG")) { long tot = 0; int mx = 0; for (int i = 0; i < nj; ++i) { tot += info[i].total; mx = std::max(mx, info[i].total); } STRQ_DBG("table floats: mean %.0f max %d (k=%d)", (double)tot / nj, mx, in.k[0]); }
    if (hard_count > hard_cap) { c->err = "too many borderline table entries"; return STRQ_ERR_DEVICE; }
    bool any_rebuild = false;
    for (int j = 0; j < nj; ++j) any_rebuild |= info[j].n_hard < 0;
    std::vector<float> h_lval;      // level values of the reads involved in host work (fetched lazily)
    auto level_vals_of = [&](int read, float* dst) -> int {
        STRQ_HIP(c, hipMemcpy(dst, in.d_level_val + (size_t)read * 256, 256 * 4, hipMemcpyDeviceToHost));
        return STRQ_OK;
    };
    if (hard_count > 0) {
        std::vector<HardEntry> he(hard_count); std::vector<float> hv(hard_count);
        STRQ_HIP(c, hipMemcpy(he.data(), d_hard, (size_t)hard_count * sizeof(HardEntry), hipMemcpyDeviceToHost));
        float lv[256]; int last_read = -1;
        for (int i = 0; i < hard_count; ++i) {
            const int rd = in.read[job_align[he[i].job]];
            if (rd != last_read) { const int rc = level_vals_of(rd, lv); if (rc) return rc; last_read = rd; }
            hv[i] = host_cell_score(c->ap, lv[he[i].level], h_cls[cls_off[job_align[he[i].job]] + job_k0[he[i].job] + he[i].k]);
        }
        STRQ_HIP(c, hipMemcpyAsync(d_hard_vals, hv.data(), (size_t)hard_count * 4, hipMemcpyHostToDevice, st));
        if (launch_lut_patch(st, d_jobs, d_hard, d_hard_vals, hard_count)) { c->err = "patch launch failed"; return STRQ_ERR_DEVICE; }
        STRQ_HIP(c, hipStreamSynchronize(st));
    }
    if (any_rebuild) for (int j = 0; j < nj; ++j) if (info[j].n_hard < 0) {
        // whole table from the host libm, full width
        const int kk = job_k[j], i = job_align[j];
        std::vector<float> tab((size_t)kk * 256); std::vector<int32_t> blo(kk);
        float lv[256];
        { const int rc = level_vals_of(in.read[i], lv); if (rc) return rc; }
        for (int x = 0; x < kk; ++x) {
            float* row = &tab[(size_t)x * 256];
            for (int q = 0; q < 256; ++q) row[q] = host_cell_score(c->ap, lv[q], h_cls[cls_off[i] + job_k0[j] + x]);
            blo[x] = (int32_t)((255u << 8) | ((uint32_t)(x * 256) << 16));     // levels 0..255, row offset x * 256
        }
        STRQ_HIP(c, hipMemcpy(jobs[j].table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        STRQ_HIP(c, hipMemcpy(jobs[j].band_lo, blo.data(), (size_t)kk * 4, hipMemcpyHostToDevice));
        info[j].total = kk * 256;
    }
    // LDS slice an alignment needs: its table, or the largest of its strips' tables
    std::vector<int> tab_total(nb, 0);
    for (int j = 0; j < nj; ++j) tab_total[job_align[j]] = std::max(tab_total[job_align[j]], info[j].total);
    out.n_hard = hard_count;

    // ---- tasks.  Alignments are grouped by (rows per lane, strips, score tables per CU), longest first.
    // Collapsed single-strip alignments (STRique's parameters) run as column segments: `segs` waves per
    // alignment share one table (align_kernels.h), so a CU holds tables_per_cu x segs waves.
    // Layout of the task array: per launch, `segs` consecutive tasks per alignment (in result order);
    // the first strips of two-strip alignments follow at the end.  seg_results is indexed like the
    // tasks; results / pick / heads are indexed by alignment position.
    const bool collapsed = S == 6 && c->ap.open_h == c->ap.ext_h && c->ap.open_v == c->ap.ext_v;      // the collapsed / segmented / packed kernels exist for samples = 6
    int seg_want = 0, max_tables = 8, max_waves = 16;      // seg_want 0: chosen below from the read lengths
    if (const char* e = strq::opt("STRQ_SEG")) { const int v = atoi(e); if (v >= 1 && v <= 4) seg_want = v; }
    if (const char* e = strq::opt("STRQ_TABLES")) { const int v = atoi(e); if (v >= 1 && v <= 8) max_tables = v; }
    if (const char* e = strq::opt("STRQ_WPB")) { const int v = atoi(e); if (v >= 1 && v <= 8) max_tables = v; }      // older name
    if (const char* e = strq::opt("STRQ_MAX_WAVES")) { const int v = atoi(e); if (v >= 4 && v <= 16) max_waves = v; }
    bool allow_pack = collapsed && !strq::opt("STRQ_NO_PACK");
    // overlap the pieces are cut with first (the worst case is ~15 k columns for an 870-row flank); see the piece planning below
    int ov_cap = 8192; bool ov_fixed = false;
    if (const char* e = strq::opt("STRQ_OVERLAP")) { const int v = atoi(e); ov_cap = v > 0 ? v : (1 << 30); ov_fixed = true; }      // 0: always the worst-case overlap
    // Without the variable: 8192 columns for the first sub-batch, afterwards the overlap that would have been cheapest
    // for the previous sub-batch -- extra columns per piece against the share of alignments whose best score would
    // not certify that overlap and which therefore run twice (any choice is exact; this one only sets the cost).
    // redo_weight: what an alignment that misses the certificate costs, in whole-read passes (1 for the exact pass's own second
    // round; the screen's pieces are cut more carefully -- an alignment whose score lies below their cold-start bound gets no
    // windows and runs its whole read in a launch of its own, a ~20 ms tail behind thousands of windows: gpurun_out/r4z)
    std::map<std::pair<int, int>, int> ov_of_m;
    auto overlap_for = [&](int m, int ov_worst, int redo_weight = 1) -> int {
        if (ov_fixed || c->score_fracs.size() < 64 || c->mean_n <= 0) return std::min(ov_worst, ov_cap);
        auto it = ov_of_m.find(std::make_pair(m, redo_weight));
        if (it != ov_of_m.end()) return it->second;
        const std::vector<float>& f = c->score_fracs;
        double best_cost = 0; int best_ov = std::min(ov_worst, ov_cap);
        for (int ov = 1024; ; ov += 512) {
            if (ov > ov_worst) ov = ov_worst;
            const float need = align_segment_min_score(c->ap, m, ov) / ((float)m * c->ap.dist_offset) + 0.01f;      // 1 % margin on last batch's scores
            const double redo = (double)(std::lower_bound(f.begin(), f.end(), need) - f.begin()) / (double)f.size();
            const double cost = 3.0 * ov / c->mean_n + redo_weight * redo * (1.0 + 3.0 * ov_worst / c->mean_n);
            if (best_cost == 0 || cost < best_cost) { best_cost = cost; best_ov = ov; }
            if (ov >= ov_worst) break;
        }
        ov_of_m[std::make_pair(m, redo_weight)] = best_ov;
        STRQ_DBG("overlap for %d-row flanks: %d columns (worst case %d; previous sub-batch: median score fraction %.3f, mean read %.0f samples)", m, best_ov, ov_worst, f[f.size() / 2], c->mean_n);
        return best_ov;
    };
    // Launch geometry, per alignment.  Measured on MI355X (ms per 8192 alignments per 1000 columns computed,
    // 870-row flanks, tools/dp_sweep.py): one wave per table 0.94 (24-bit tables, 8 waves per CU) / 1.02 (float32,
    // 6 waves); two waves per table 0.86 (24-bit, 16 waves) / 0.79 (float32, 12 waves); four waves per table
    // 0.80 / 0.69 (float32, 16 waves).  Three waves per table place unevenly on the four SIMDs (0.85 and worse).
    // Every extra wave recomputes `overlap` columns, so short reads prefer fewer pieces: a batch of mixed read
    // lengths runs as up to three launches (4, 2 and 1 waves per alignment).
    static const double rate[5][2] = {{0, 0}, {1.017, 0.941}, {0.79, 0.862}, {0.85, 0.85}, {0.69, 0.80}};
    auto tables_for = [&](int dwords, int segs) {
        int t = std::min(max_tables, (160 * 1024 - 64) / (std::max(dwords, 1) * 4));
        if (segs > 1) t = std::min(t, max_waves / segs);
        return t;
    };
    auto packed_dwords = [](int entries) { return (((2 * entries + 3) & ~3) + entries + 3) / 4; };
    struct Piece { int col_off, n; size_t ck; };
    auto cut = [](int n, int segs, int ov, Piece* out) {     // returns the number of pieces used
        int use = segs;
        while (use > 1 && (ov <= 0 || (long)n < (long)(use + 1) * ov)) --use;      // every piece owns >= `overlap` columns
        const long len = use > 1 ? ((long)n + (long)(use - 1) * ov + use - 1) / use : n;      // columns each piece computes
        long o = 0;                                                                        // columns owned so far
        for (int k = 0; k < segs; ++k) {
            if (k >= use) { out[k].col_off = 0; out[k].n = 0; continue; }
            const long start = k == 0 ? 0 : o - ov;
            long end = k == use - 1 ? n : start + len; if (end > n) end = n;
            out[k].col_off = (int)start; out[k].n = (int)(end - start);
            o = end;
        }
        return use;
    };

    // ---- upper-bound screen (screen_kernels.hip): where in the read the exact DP has to look.
    // Result per alignment: up to four column windows and a lower bound of its best score -- or nothing, then the whole read runs.
    std::vector<ScreenWindows> wins(nb);
    for (auto& w : wins) { w.n_win = 0; w.lower_bound = 0; }
    c->screen_ran = false;
    STRQ_HIP(c, hipEventRecord(c->ev[2], st));          // the forward time of the sub-batch includes the screen
    ScreenParams sp;
    int scr_max_n = 0;
    for (int i = 0; i < nb; ++i) scr_max_n = std::max(scr_max_n, in.n[i]);
    const bool scr_forced = strq::opt("STRQ_SCREEN_ALWAYS") != nullptr;      // tests: no pause
    // STRQ_SCREEN_MODE: "fine" = the row-exact screen only, "coarse" = the merged-row screen whenever the sub-batch allows it,
    // default: coarse until it stops paying (then fine, then none), each retried after eight sub-batches
    const char* mode_opt = strq::opt("STRQ_SCREEN_MODE");
    const bool mode_fine = mode_opt && std::strcmp(mode_opt, "fine") == 0, mode_coarse = mode_opt && std::strcmp(mode_opt, "coarse") == 0;
    c->screen_mode_last = 0;
    bool did_coarse = false;
    // flank rows per DP row of the coarse screen (2, 3 or 6: STRQ_SCREEN2_MERGE).  Measured on configs[2] (gpurun_out/r5j): screen 114.4 / 94.7 /
    // 96.6 ms per 4096 reads, forward stage 131.6 / 115.8 / 130.9 ms -- with six rows per DP row the 10 table reads of a step bind, and
    // the looser bound sends 14 % of the alignments into the second look (2.4 % at two, 6 % at three)
    const int coarse_merge = 3;
    std::vector<int> g2_of(nb, -1);                 // alignment -> its index in the coarse screen's task views
    ScreenTask* d_st2 = nullptr; int32_t* d_bound2 = nullptr; ScreenParams sp2; std::memset(&sp2, 0, sizeof(sp2));
    double coarse_cols = 0, coarse_all = 0;
    // Two attempts at a screen that takes both flank alignments of a read per wave (screen_kernels.hip: screen2_body): the coarse one
    // (merged rows, its own candidate rules and second look) unless it is paused or switched off, then -- if that did not run and the
    // fine screen is not paused -- the same kernel without merging: the fine screen's bound and rules (merge_now = 1).  A sub-batch
    // that does not hold both alignments of its reads (strq_align_batch with single alignments) falls through to the one-flank kernel.
    for (int attempt = 0; attempt < 2 && !did_coarse; ++attempt) {
    int merge_now = coarse_merge;
    if (attempt == 0) {
        if (!collapsed || mode_fine || strq::opt("STRQ_NO_SCREEN")) continue;
        if (c->coarse_pause > 0 && !mode_coarse && !scr_forced) { --c->coarse_pause; continue; }
    } else {
        if (!collapsed || strq::opt("STRQ_NO_SCREEN") || (c->screen_pause > 0 && !scr_forced)) continue;
        merge_now = 1;
    }
    const bool fine_rules = merge_now == 1;
    if (screen2_plan(c->ap, S, scr_max_n, merge_now, &sp)) {
        // ---- reads whose two flank alignments are both in this sub-batch
        // reads from which the pass pays (gpurun_out/r5z: the coarse screen wins from ~30 k samples on -- 5000-nt reads 79.8 k against 70.7 k
        // reads/s, 3000-nt reads 110.6 k against 114.7 k; the fine bound costs twice as much and keeps round 4's 64 k)
        int min_n = fine_rules ? 65536 : 28672, scr_groups = 6;
        if (const char* e = strq::opt("STRQ_SCREEN_MIN_N")) min_n = atoi(e);
        if (const char* e = strq::opt("STRQ_SCREEN2_GROUPS")) { const int v = atoi(e); if (v >= 1 && v <= 8) scr_groups = v; }
        std::map<int, std::vector<int>> by_read;
        for (int i = 0; i < nb; ++i)
            if (in.NS[i] == 1 && NJ[i] == 1 && in.n[i] >= min_n && screen2_flank_ok(in.m[i], in.k[i])) by_read[in.read[i]].push_back(i);
        std::vector<std::pair<int, int>> pairs;
        for (auto& kv : by_read) if (kv.second.size() == 2) pairs.emplace_back(kv.second[0], kv.second[1]);
        std::stable_sort(pairs.begin(), pairs.end(), [&](const std::pair<int, int>& x, const std::pair<int, int>& y) { return in.n[x.first] > in.n[y.first]; });
        const int ngr = (int)pairs.size();
        if (ngr > 0 && 2 * ngr >= (nb * 9) / 10) {
            constexpr int SSEG = STRQ_SCREEN_SEG;
            sp.margin = (int32_t)std::lround((double)c->coarse_margin * (merge_now == 3 ? 1.3 : 1.0) * sp.sc);
            if (const char* e = strq::opt("STRQ_SCREEN2_MARGIN")) sp.margin = (int32_t)std::lround(atof(e) * sp.sc);
            sp.max_cand = 8;          // (8, 16, 32 candidates and margins of 300 ... 700 score units measure within 1.5 % of each other: gpurun_out/r5n)
            if (const char* e = strq::opt("STRQ_SCREEN2_MAX_CAND")) sp.max_cand = atoi(e);
            if (fine_rules) { sp.margin = 0; sp.max_cand = 0; }          // the fine screen's rule: everything within m + 2 slack of the best chunk
            std::vector<Screen2Task> t2((size_t)ngr * SSEG);
            std::vector<ScreenTask> stasks((size_t)2 * ngr * SSEG);
            std::vector<int32_t> bound((size_t)2 * ngr);
            std::vector<size_t> out_off((size_t)2 * ngr * SSEG, 0);
            size_t out_words = 0, lds_bytes = 0; double steps = 0;
            for (int g = 0; g < ngr; ++g) {
                const int a[2] = {pairs[g].first, pairs[g].second};
                const int n = in.n[a[0]];
                const int m_big = std::max(in.m[a[0]], in.m[a[1]]);
                const int ov_worst = align_segment_overlap(c->ap, m_big), ov_s = overlap_for(m_big, ov_worst, 16);
                Piece pc[SSEG];
                const int used = cut(n, SSEG, ov_s, pc);
                lds_bytes = std::max(lds_bytes, screen2_lds_bytes(info[J0[a[0]]].total, info[J0[a[1]]].total));
                for (int f = 0; f < 2; ++f)
                    bound[(size_t)2 * g + f] = used <= 1 ? INT32_MIN / 2 : (ov_s >= ov_worst ? 0 : (int32_t)std::ceil((double)align_segment_min_score(c->ap, in.m[a[f]], ov_s) * sp.sc));
                for (int w = 0; w < SSEG; ++w) {
                    Screen2Task t; std::memset(&t, 0, sizeof(t));
                    t.levels = in.d_levels + in.read_off[in.read[a[0]]] + pc[w].col_off;
                    t.n = pc[w].n; t.col_off = pc[w].col_off;
                    t.n_chunks = pc[w].n > 0 ? (align_num_steps(pc[w].n) + 63) / 64 : 0;
                    for (int f = 0; f < 2; ++f) {
                        t.table[f] = jobs[J0[a[f]]].table; t.band_lo[f] = jobs[J0[a[f]]].band_lo; t.tsize[f] = info[J0[a[f]]].total; t.k[f] = in.k[a[f]];
                        ScreenTask v; std::memset(&v, 0, sizeof(v));
                        v.n = pc[w].n; v.m = in.m[a[f]]; v.k = in.k[a[f]]; v.col_off = pc[w].col_off; v.n_chunks = t.n_chunks;
                        v.lane_last = f * STRQ_SCREEN2_LANE_B + (in.k[a[f]] - 1) / STRQ_SCREEN2_CPL;
                        out_off[((size_t)2 * g + f) * SSEG + w] = out_words; out_words += (size_t)t.n_chunks;
                        stasks[((size_t)2 * g + f) * SSEG + w] = v;
                    }
                    if (pc[w].n > 0) steps += align_num_steps(pc[w].n);
                    t2[(size_t)g * SSEG + w] = t;
                }
            }
            if (lds_bytes > 160 * 1024 - 64) { c->err = "score tables do not fit LDS (coarse screen)"; return STRQ_ERR_UNSUPPORTED; }
            scr_groups = std::max(1, std::min(scr_groups, (int)((160 * 1024 - 64) / lds_bytes)));
            const size_t t2_bytes = (t2.size() * sizeof(Screen2Task) + 255) & ~(size_t)255, task_bytes = (stasks.size() * sizeof(ScreenTask) + 255) & ~(size_t)255;
            const size_t bound_bytes = ((size_t)2 * ngr * 4 + 255) & ~(size_t)255, win_bytes = ((size_t)2 * ngr * sizeof(ScreenWindows) + 255) & ~(size_t)255;
            STRQ_HIP(c, c->screen.reserve(t2_bytes + task_bytes + bound_bytes + win_bytes + out_words * 4 + 256));
            char* base = c->screen.as<char>();
            Screen2Task* d_t2 = reinterpret_cast<Screen2Task*>(base);
            ScreenTask* d_st = reinterpret_cast<ScreenTask*>(base + t2_bytes);
            int32_t* d_bound = reinterpret_cast<int32_t*>(base + t2_bytes + task_bytes);
            ScreenWindows* d_win = reinterpret_cast<ScreenWindows*>(base + t2_bytes + task_bytes + bound_bytes);
            int32_t* d_out = reinterpret_cast<int32_t*>(base + t2_bytes + task_bytes + bound_bytes + win_bytes);
            for (int g = 0; g < ngr; ++g) for (int f = 0; f < 2; ++f) for (int w = 0; w < SSEG; ++w) {
                int32_t* o = d_out + out_off[((size_t)2 * g + f) * SSEG + w];
                stasks[((size_t)2 * g + f) * SSEG + w].out = o; t2[(size_t)g * SSEG + w].out[f] = o;
            }
            STRQ_HIP(c, hipMemcpyAsync(d_t2, t2.data(), t2.size() * sizeof(Screen2Task), hipMemcpyHostToDevice, st));
            STRQ_HIP(c, hipMemcpyAsync(d_st, stasks.data(), stasks.size() * sizeof(ScreenTask), hipMemcpyHostToDevice, st));
            STRQ_HIP(c, hipMemcpyAsync(d_bound, bound.data(), bound.size() * 4, hipMemcpyHostToDevice, st));
            STRQ_HIP(c, hipEventRecord(c->ev[5], st));
            if (launch_screen2(st, d_t2, ngr, c->queue.as<int>() + STRQ_QUEUE_FIRST - 1, sp, lds_bytes, scr_groups, c->n_cu, merge_now)) { c->err = "coarse screen launch failed"; return STRQ_ERR_DEVICE; }
            STRQ_HIP(c, hipEventRecord(c->ev[6], st));
            if (launch_screen_windows(st, d_st, 2 * ngr, sp, d_bound, d_win)) { c->err = "screen windows launch failed"; return STRQ_ERR_DEVICE; }
            std::vector<ScreenWindows> hw((size_t)2 * ngr);
            STRQ_HIP(c, hipMemcpyAsync(hw.data(), d_win, hw.size() * sizeof(ScreenWindows), hipMemcpyDeviceToHost, st));
            STRQ_HIP(c, hipStreamSynchronize(st));
            c->screen_ran = true; did_coarse = true; c->screen_mode_last = fine_rules ? 1 : 2; c->coarse_merge_last = merge_now;
            d_st2 = d_st; d_bound2 = d_bound; sp2 = sp;
            for (int g2 = 0; g2 < 2 * ngr; ++g2) g2_of[g2 & 1 ? pairs[(size_t)g2 / 2].second : pairs[(size_t)g2 / 2].first] = g2;
            if (const char* path = strq::opt("STRQ_SCREEN_DUMP")) {
                std::vector<int32_t> ho(out_words);
                STRQ_HIP(c, hipMemcpy(ho.data(), d_out, out_words * 4, hipMemcpyDeviceToHost));
                if (FILE* fp = fopen(path, "wb")) {
                    const int32_t hdr[8] = {2 * ngr, sp.sc, sp.hh, sp.v, 2, SSEG, sp.slack, merge_now};
                    fwrite(hdr, 4, 8, fp);
                    for (int g2 = 0; g2 < 2 * ngr; ++g2) {
                        const int al = g2 & 1 ? pairs[(size_t)g2 / 2].second : pairs[(size_t)g2 / 2].first;
                        const int32_t gh[4] = {al, al, bound[(size_t)g2], stasks[(size_t)g2 * SSEG].lane_last};
                        fwrite(gh, 4, 4, fp);
                        for (int w = 0; w < SSEG; ++w) {
                            const ScreenTask& t = stasks[(size_t)g2 * SSEG + w];
                            const int32_t th[4] = {t.col_off, t.n, t.m, t.n_chunks};
                            fwrite(th, 4, 4, fp);
                            fwrite(ho.data() + (t.out - d_out), 4, (size_t)t.n_chunks, fp);
                        }
                        fwrite(&hw[(size_t)g2], sizeof(ScreenWindows), 1, fp);
                    }
                    fclose(fp);
                }
            }
            const bool no_prune = strq::opt("STRQ_SCREEN_NO_PRUNE") != nullptr;
            if (const char* e = strq::opt("STRQ_SCREEN_TEST_RAISE")) { const float up = (float)atof(e); for (auto& w : hw) w.lower_bound += up; }
            int windowed = 0, heavy = 0, below_bound = 0; double cols = 0, all = 0;
            for (int g2 = 0; g2 < 2 * ngr; ++g2) {
                const int al = g2 & 1 ? pairs[(size_t)g2 / 2].second : pairs[(size_t)g2 / 2].first;
                const ScreenWindows& w = hw[(size_t)g2];
                c->screen_stats[1] += 1; c->screen_stats[7] += w.n_cand;
                double mine = in.n[al];
                if (w.n_win > 0 && !no_prune) {
                    wins[al] = w; c->screen_stats[2] += 1; ++windowed; mine = 0;
                    for (int k = 0; k < w.n_win; ++k) { c->screen_stats[4] += w.hi[k] - w.lo[k] + 1; mine += w.hi[k] - w.lo[k] + 1 + 4096; }
                } else {
                    c->screen_stats[3] += 1;
                    // best bound below the score the pieces' cold start was planned for (the previous sub-batch scored higher): a planning
                    // transient, not something the screen is to blame for -- its whole read does not count against the screen
                    if (w.n_cand == 0 && !ov_fixed) { ++below_bound; mine = 0; }
                }
                all += in.n[al]; cols += mine; heavy += mine > 4 * 32768.0;          // more than four pieces of 32 k columns (the cut below)
            }
            coarse_cols = cols; coarse_all = all;
            c->screen_stats[3] += nb - 2 * ngr;          // alignments of the sub-batch the coarse screen does not take: their whole reads run
            c->screen_stats[5] += steps; c->screen_stats[6] = sp.sc;
            // does it pay?  The coarse pass costs ~0.36 of the float32 pass over whole reads, the fine bound ~0.72 (0.85 on the one-flank kernel), the
            // exact pass over a share x of the columns (windows + their cold-start overlaps: what `cols` counts) ~1.3 x: the coarse path beats the
            // fine one while x < ~0.3 (short reads sit at 0.15 - 0.25 from the overlaps alone), the fine bound beats no screen while x < ~0.2.
            // Reads on which the coarse screen's two looks leave more than that, or whose first look certifies less than two thirds -- its bound is 5 - 10 % above the exact scores where events are short, and the
            // background of such reads is that close to the flank: the chunks that reach the score found then cover most of the read
            // (gpurun_out/r5r: 96 % of the alignments of the empirical-noise reads miss the first look's certificate) -- go to the fine
            // screen for a while
            // (heavy alignments are cut into four pieces below and run next to the small windows on the second stream: a few per
            // cent of them cost their own work, not a tail)
            if (fine_rules) {
                // the fine screen's own verdict (below, for the one-flank kernel): nearly every alignment windows, a few per cent of the columns
                if (2 * ngr >= 64 && !no_prune && !scr_forced) {
                    if (below_bound * 10 > 2 * ngr && !ov_fixed) c->score_fracs.clear();
                    else if (windowed < 0.9 * (2 * ngr - (ov_fixed ? 0 : below_bound)) || cols > 0.15 * all || (long)heavy * 100 > 2L * 2 * ngr) { c->screen_pause = std::min(256, 8 << std::min(c->screen_fail, 5)); ++c->screen_fail; }
                    else c->screen_fail = 0;
                }
            } else if (2 * ngr >= 64 && !no_prune && !mode_coarse && !scr_forced) {
                // (alignments whose best bound lies below the score the pieces' cold start was sized for get no windows: the overlap
                // was planned on the previous sub-batch's scores and this one scores lower -- no reason to pause, the plan follows)
                if (below_bound * 10 > 2 * ngr && !ov_fixed) c->score_fracs.clear();
                else if (windowed < 0.9 * (2 * ngr - (ov_fixed ? 0 : below_bound)) || cols > 0.30 * all || (long)heavy * 100 > 2L * 2 * ngr) { c->coarse_pause = std::min(256, 8 << std::min(c->coarse_fail, 5)); ++c->coarse_fail; }
                else c->coarse_fail = 0;
            }
            STRQ_DBG("%s screen (both flanks per wave): %d reads, scale %d, margin %.0f, %d groups per CU, LDS %zu bytes: %d of %d alignments with windows (%d below the cold-start bound), %.2f %% of the columns, %d heavy -> pause %d",
                     fine_rules ? "fine" : "coarse", ngr, sp.sc, (double)sp.margin / sp.sc, scr_groups, lds_bytes, windowed, 2 * ngr, below_bound, 100.0 * cols / std::max(1.0, all), heavy, fine_rules ? c->screen_pause : c->coarse_pause);
        }
    }
    }
    if (did_coarse) {
    } else if (collapsed && c->screen_pause > 0 && !scr_forced) {
        --c->screen_pause; c->screen_stats[3] += nb;      // counted as whole-read alignments
    } else if (collapsed && screen_plan(c->ap, S, scr_max_n, &sp)) {
        // reads below ~64 k samples: the pieces' overlaps eat what the cheaper pass saves (STRQ_SCREEN_MIN_N: tests)
        int min_n = 65536, scr_tables = 6;
        if (const char* e = strq::opt("STRQ_SCREEN_MIN_N")) min_n = atoi(e);
        if (const char* e = strq::opt("STRQ_SCREEN_TABLES")) { const int v = atoi(e); if (v >= 1 && v <= 8) scr_tables = v; }
        std::vector<int> sel;
        for (int i = 0; i < nb; ++i)
            if (in.NS[i] == 1 && NJ[i] == 1 && in.n[i] >= min_n && screen_flank_ok(in.m[i]) && in.k[i] * S == in.m[i]) sel.push_back(i);
        std::stable_sort(sel.begin(), sel.end(), [&](int x, int y) { return in.n[x] > in.n[y]; });
        const int ng = (int)sel.size();
        if (ng > 0) {
            constexpr int SSEG = STRQ_SCREEN_SEG;
            std::vector<ScreenTask> stasks((size_t)ng * SSEG);
            std::vector<int32_t> bound((size_t)ng);
            std::vector<size_t> out_off((size_t)ng * SSEG, 0);
            size_t out_words = 0, lds_bytes = 0; double steps = 0;
            for (int g = 0; g < ng; ++g) {
                const int a = sel[g], n = in.n[a], m = in.m[a];
                const int ov_worst = align_segment_overlap(c->ap, m), ov_s = overlap_for(m, ov_worst, 16);
                Piece pc[SSEG];
                const int used = cut(n, SSEG, ov_s, pc);
                // what the cold start of the pieces costs: below this score a piece's values are not bounds of the whole matrix
                bound[g] = used <= 1 ? INT32_MIN / 2 : (ov_s >= ov_worst ? 0 : (int32_t)std::ceil((double)align_segment_min_score(c->ap, m, ov_s) * sp.sc));
                lds_bytes = std::max(lds_bytes, screen_lds_bytes(info[J0[a]].total));
                for (int w = 0; w < SSEG; ++w) {
                    ScreenTask t; std::memset(&t, 0, sizeof(t));
                    t.levels = in.d_levels + in.read_off[in.read[a]] + pc[w].col_off;
                    t.table = jobs[J0[a]].table; t.band_lo = jobs[J0[a]].band_lo; t.tsize = info[J0[a]].total;
                    t.n = pc[w].n; t.m = m; t.k = in.k[a]; t.col_off = pc[w].col_off; t.lane_last = (m - 1) / STRQ_SCREEN_R;
                    t.n_chunks = pc[w].n > 0 ? (align_num_steps(pc[w].n) + 63) / 64 : 0;
                    out_off[(size_t)g * SSEG + w] = out_words;
                    out_words += (size_t)t.n_chunks;
                    if (pc[w].n > 0) steps += align_num_steps(pc[w].n);
                    stasks[(size_t)g * SSEG + w] = t;
                }
            }
            if (lds_bytes > 160 * 1024 - 64) { c->err = "score table does not fit LDS (screen)"; return STRQ_ERR_UNSUPPORTED; }
            scr_tables = std::max(1, std::min(scr_tables, (int)((160 * 1024 - 64) / lds_bytes)));
            const size_t task_bytes = (stasks.size() * sizeof(ScreenTask) + 255) & ~(size_t)255;
            const size_t bound_bytes = ((size_t)ng * 4 + 255) & ~(size_t)255, win_bytes = ((size_t)ng * sizeof(ScreenWindows) + 255) & ~(size_t)255;
            STRQ_HIP(c, c->screen.reserve(task_bytes + bound_bytes + win_bytes + out_words * 4 + 256));
            char* base = c->screen.as<char>();
            ScreenTask* d_st = reinterpret_cast<ScreenTask*>(base);
            int32_t* d_bound = reinterpret_cast<int32_t*>(base + task_bytes);
            ScreenWindows* d_win = reinterpret_cast<ScreenWindows*>(base + task_bytes + bound_bytes);
            int32_t* d_out = reinterpret_cast<int32_t*>(base + task_bytes + bound_bytes + win_bytes);
            for (size_t x = 0; x < stasks.size(); ++x) stasks[x].out = d_out + out_off[x];
            STRQ_HIP(c, hipMemcpyAsync(d_st, stasks.data(), stasks.size() * sizeof(ScreenTask), hipMemcpyHostToDevice, st));
            STRQ_HIP(c, hipMemcpyAsync(d_bound, bound.data(), (size_t)ng * 4, hipMemcpyHostToDevice, st));
            STRQ_HIP(c, hipEventRecord(c->ev[5], st));
            if (launch_screen(st, d_st, ng, c->queue.as<int>() + STRQ_QUEUE_FIRST - 1, sp, lds_bytes, scr_tables, c->n_cu)) { c->err = "screen launch failed"; return STRQ_ERR_DEVICE; }
            STRQ_HIP(c, hipEventRecord(c->ev[6], st));
            if (launch_screen_windows(st, d_st, ng, sp, d_bound, d_win)) { c->err = "screen windows launch failed"; return STRQ_ERR_DEVICE; }
            std::vector<ScreenWindows> hw((size_t)ng);
            STRQ_HIP(c, hipMemcpyAsync(hw.data(), d_win, hw.size() * sizeof(ScreenWindows), hipMemcpyDeviceToHost, st));
            STRQ_HIP(c, hipStreamSynchronize(st));
            c->screen_ran = true; c->screen_mode_last = 1; c->coarse_merge_last = 0;
            if (const char* path = strq::opt("STRQ_SCREEN_DUMP")) {
                // tests: the chunk maxima as the kernel wrote them (tests/test_gpu_screen.py checks them against the exact last row)
                std::vector<int32_t> ho(out_words);
                STRQ_HIP(c, hipMemcpy(ho.data(), d_out, out_words * 4, hipMemcpyDeviceToHost));
                if (FILE* fp = fopen(path, "wb")) {
                    const int32_t hdr[8] = {ng, sp.sc, sp.hh, sp.v, 1, SSEG, sp.slack, 0};
                    fwrite(hdr, 4, 8, fp);
                    for (int g = 0; g < ng; ++g) {
                        const int32_t gh[4] = {sel[g], sel[g], bound[g], (in.m[sel[g]] - 1) / STRQ_SCREEN_R};
                        fwrite(gh, 4, 4, fp);
                        for (int w = 0; w < SSEG; ++w) {
                            const ScreenTask& t = stasks[(size_t)g * SSEG + w];
                            const int32_t th[4] = {t.col_off, t.n, t.m, t.n_chunks};
                            fwrite(th, 4, 4, fp);
                            fwrite(ho.data() + (t.out - d_out), 4, (size_t)t.n_chunks, fp);
                        }
                        fwrite(&hw[(size_t)g], sizeof(ScreenWindows), 1, fp);
                    }
                    fclose(fp);
                }
            }
            const bool no_prune = strq::opt("STRQ_SCREEN_NO_PRUNE") != nullptr;      // tests: run the screen, then the whole reads
            if (const char* e = strq::opt("STRQ_SCREEN_TEST_RAISE")) {
                // tests: claim a lower bound no alignment reaches -- the certificate must fail and the second round (whole reads) must deliver
                const float up = (float)atof(e);
                for (auto& w : hw) w.lower_bound += up;
            }
            for (int g = 0; g < ng; ++g) {
                const ScreenWindows& w = hw[(size_t)g];
                c->screen_stats[1] += 1; c->screen_stats[7] += w.n_cand;
                if (w.n_win > 0 && !no_prune) {
                    wins[sel[g]] = w; c->screen_stats[2] += 1;
                    for (int k = 0; k < w.n_win; ++k) c->screen_stats[4] += w.hi[k] - w.lo[k] + 1;
                } else c->screen_stats[3] += 1;
            }
            c->screen_stats[5] += steps; c->screen_stats[6] = sp.sc;
            {
                // does it pay?  The screen costs ~0.9 of the float32 pass over whole reads, the windows run at a lower rate than whole
                // reads do: it wins when nearly every alignment gets windows and they hold a few per cent of the columns at most
                // -- and when hardly any alignment is left with a large share of its read to run: a few such alignments behind
                // thousands of windows are a tail of their own (one piece of 100 k columns takes a lone wave ~17 ms; reads at
                // `realism` 1: forward stage 94.6 ms per 1024 reads against 74.2 without the screen, gpurun_out/r4ad)
                int windowed = 0, heavy = 0, below_bound = 0; double cols = 0, all = 0;
                for (int g = 0; g < ng; ++g) {
                    const ScreenWindows& w = wins[sel[g]];
                    windowed += w.n_win > 0;
                    below_bound += hw[(size_t)g].n_win == 0 && hw[(size_t)g].n_cand == 0;
                    all += in.n[sel[g]];
                    double mine = 0;
                    if (w.n_win > 0) for (int k = 0; k < w.n_win; ++k) mine += w.hi[k] - w.lo[k] + 1 + 4096;
                    else mine = in.n[sel[g]];
                    cols += mine; heavy += mine > 4 * 32768.0;          // more than four pieces of 32 k columns (the cut below)
                }
                if (ng >= 64 && !no_prune) {
                    // (no windows because the best bound lies below the score the pieces' cold start was sized for: the overlap was
                    // planned on the previous sub-batch's scores and this one scores lower -- the plan follows, no pause)
                    if (below_bound * 10 > ng && !ov_fixed) c->score_fracs.clear();
                    else if (windowed < 0.9 * (ng - (ov_fixed ? 0 : below_bound)) || cols > 0.10 * all || (long)heavy * 100 > 2L * ng) { c->screen_pause = std::min(256, 8 << std::min(c->screen_fail, 5)); ++c->screen_fail; }
                    else c->screen_fail = 0;
                }
                STRQ_DBG("screen verdict: %d of %d with windows (%d below the cold-start bound), %.2f %% of the columns inside them, %d heavy alignments -> pause %d", windowed, ng, below_bound, 100.0 * cols / std::max(1.0, all), heavy, c->screen_pause);
            }
            STRQ_DBG("screen: %d alignments, scale %d, %d tables per CU, LDS %zu bytes; windows for %.0f of %.0f alignments so far", ng, sp.sc, scr_tables, lds_bytes, c->screen_stats[2], c->screen_stats[1]);
        }
    }
    std::vector<char> packed(nb, 0), segmentable(nb, 0);
    std::vector<int> overlap(nb, 0), segs_of(nb, 1);
    const bool force_pack = strq::opt("STRQ_PACK") != nullptr;
    int class_count[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < nb; ++i) {
        segmentable[i] = collapsed && in.NS[i] == 1;
        overlap[i] = segmentable[i] ? align_segment_overlap(c->ap, in.m[i]) : 0;
        const bool can_pack = allow_pack && in.NS[i] == 1 && info[J0[i]].packed && info[J0[i]].n_hard == 0;
        int best_s = 1, best_p = can_pack ? 1 : 0;
        if (segmentable[i]) {
            const double l = overlap_for(in.m[i], overlap[i]);
            double best_cost = 0;
            for (int sgs : {1, 2, 3, 4}) {
                if (seg_want ? sgs != seg_want : sgs == 3) continue;
                if (sgs > 1 && !seg_want && (l <= 0 || in.n[i] < (sgs + 1) * l)) continue;
                for (int pk = 0; pk < 2; ++pk) {
                    if ((pk && !can_pack) || (force_pack && can_pack && !pk)) continue;
                    const double cost = (in.n[i] + (sgs - 1) * l) * rate[sgs][pk];
                    if (best_cost == 0 || cost < best_cost) { best_cost = cost; best_s = sgs; best_p = pk; }
                }
            }
        }
        if (wins[i].n_win > 0) {
            // An alignment left with a lot of columns (a read that does not hold its flank clearly: candidates all over it) is cut
            // further, its widest window in two, until it has four pieces or none is wider than 16 k columns -- pieces start cold
            // anyway (align_overlap_for_score), so a cut costs one more overlap, and a lone wave over 300 k columns would be a
            // 25 ms tail behind thousands of small windows.
            ScreenWindows& w = wins[i];
            while (w.n_win < STRQ_SCREEN_MAX_WINDOWS) {
                int widest = 0;
                for (int k2 = 1; k2 < w.n_win; ++k2) if (w.hi[k2] - w.lo[k2] > w.hi[widest] - w.lo[widest]) widest = k2;
                if (w.hi[widest] - w.lo[widest] < 16384) break;
                const int mid = w.lo[widest] + (w.hi[widest] - w.lo[widest]) / 2;
                for (int k2 = w.n_win; k2 > widest + 1; --k2) { w.lo[k2] = w.lo[k2 - 1]; w.hi[k2] = w.hi[k2 - 1]; }
                w.lo[widest + 1] = mid + 1; w.hi[widest + 1] = w.hi[widest]; w.hi[widest] = mid;
                ++w.n_win;
            }
            // one piece per window (compiled piece counts: 1, 2, 4; 3 with STRQ_SEG=3), float32 tables
            const int nw = w.n_win;
            best_s = seg_want >= nw ? seg_want : (nw <= 2 ? nw : 4); best_p = 0;
        }
        segs_of[i] = best_s; packed[i] = (char)best_p;
        ++class_count[best_s];
    }
    if (!seg_want && !c->screen_ran) {
        // (Not behind the screen: there the small classes are the few alignments that run their whole read next to thousands
        // of windows -- one wave over 375 k columns would be a 60 ms tail, gpurun_out/r4y.)
        // a length class too small to fill the chip once joins the class with fewer waves per alignment
        int few = 2 * c->n_cu;
        if (const char* e = strq::opt("STRQ_CLASS_MIN")) few = atoi(e);      // tests: keep small length classes apart
        // (an alignment with more windows than its new launch has pieces runs its whole read instead)
        if (class_count[4] && class_count[4] < few) { for (int i = 0; i < nb; ++i) if (segs_of[i] == 4) { segs_of[i] = 2; if (wins[i].n_win > 2) wins[i].n_win = 0; } class_count[2] += class_count[4]; class_count[4] = 0; }
        if (class_count[2] && class_count[2] < few && class_count[1]) { for (int i = 0; i < nb; ++i) if (segs_of[i] == 2) { segs_of[i] = 1; if (wins[i].n_win > 1) wins[i].n_win = 0; } class_count[1] += class_count[2]; class_count[2] = 0; }
    }
    // key: rows per lane, strips, waves per alignment (descending), -tables per CU, 0 = packed / 1 = float32 (packed first among equals)
    std::map<std::tuple<int, int, int, int, int>, std::vector<int>> groups;
    for (int i = 0; i < nb; ++i) {
        const int w = tables_for(packed[i] ? packed_dwords(tab_total[i]) : tab_total[i], segs_of[i]);
        if (w < 1) { c->err = "score table does not fit LDS"; return STRQ_ERR_UNSUPPORTED; }
        groups[std::make_tuple(in.R[i], in.NS[i], -segs_of[i], -w, packed[i] ? 0 : 1)].push_back(i);
    }
    // Every launch ends with a ragged tail (alignments take tens of ms each), so a group that would not
    // keep its waves busy for a few rounds joins the next group with fewer tables per CU (larger LDS
    // slices); groups are ordered by descending tables per CU within (R, strips, waves per alignment).
    {
        int min_rounds = 4;
        if (const char* e = strq::opt("STRQ_MIN_ROUNDS")) min_rounds = atoi(e);
        for (auto it = groups.begin(); it != groups.end();) {
            auto nx = std::next(it);
            const int w = -std::get<3>(it->first);
            // a packed alignment also has its float32 table, so it can join a float32 launch -- not the other way round
            const bool same_kind = nx != groups.end() && std::get<0>(nx->first) == std::get<0>(it->first) && std::get<1>(nx->first) == std::get<1>(it->first)
                                   && std::get<2>(nx->first) == std::get<2>(it->first)
                                   && !(std::get<4>(it->first) == 1 && std::get<4>(nx->first) == 0);
            if (same_kind && (long)it->second.size() < (long)min_rounds * w * c->n_cu) {
                nx->second.insert(nx->second.end(), it->second.begin(), it->second.end());
                it = groups.erase(it);
            } else ++it;
        }
    }
    out.order.clear(); out.order.reserve(nb);
    struct Launch { int R, NS, tables, first, count, first_task, first_up, lds_floats, packed, segs; bool known_last_row; };
    std::vector<Launch> launches;
    int n_up = 0; size_t n_tasks = 0;
    for (auto& g : groups) {
        auto& v = g.second;
        auto work = [&](int x) {      // columns the forward pass of the alignment computes, roughly
            if (wins[x].n_win <= 0) return (long)in.n[x];
            long w = 0;
            for (int k = 0; k < wins[x].n_win; ++k) w += wins[x].hi[k] - wins[x].lo[k] + 1 + 4096;
            return w;
        };
        std::stable_sort(v.begin(), v.end(), [&](int x, int y) { return work(x) > work(y); });
        const int NS = std::get<1>(g.first);
        const int pk = std::get<4>(g.first) == 0;
        int lds_floats = 0;      // LDS slice of a table in dwords
        for (int i : v) lds_floats = std::max(lds_floats, pk ? packed_dwords(tab_total[i]) : tab_total[i]);
        const int segs = -std::get<2>(g.first);
        const int tables = std::min(-std::get<3>(g.first), tables_for(lds_floats, segs));       // members that joined from a smaller-slice group
        if (tables < 1) { c->err = "score table does not fit LDS"; return STRQ_ERR_UNSUPPORTED; }
        // every flank of the launch is STRique's own 870 rows at 14 per lane: round 3's kernel body.  (Flanks that fill their last lane,
        // m % 14 == 0, would also find their instance there, but it measures 14 % slower than the same case of the switched loop:
        // gpurun_out/r4d/flank_sweep.md against r4b.)
        bool known = std::get<0>(g.first) == 14;
        for (int i : v) known = known && in.m[i] == 870;
        launches.push_back({std::get<0>(g.first), NS, tables, (int)out.order.size(), (int)v.size(), (int)n_tasks, 0, lds_floats, pk, segs, known});
        out.order.insert(out.order.end(), v.begin(), v.end());
        n_tasks += (size_t)v.size() * segs;
        if (NS > 1) n_up += (int)v.size() * (NS - 1);
    }
    { size_t up = n_tasks; for (auto& L : launches) if (L.NS > 1) { L.first_up = (int)up; up += (size_t)L.count * (L.NS - 1); } }      // upper strips: level-major per launch
    // Piece boundaries and checkpoint areas.  Two geometries per segmented launch: the pieces that run first
    // are cut with a short overlap (`ov_fast` columns), which is exact whenever the alignment's best score
    // reaches align_segment_min_score -- true for every read that contains the flank; the combine kernel
    // lists the alignments that do not, and those run a second time with the worst-case overlap (`safe`
    // pieces, same checkpoint areas).  Task array: [fast pieces | upper strips | heads | safe pieces].
    const size_t safe0 = n_tasks + n_up + (size_t)nb;
    std::vector<Piece> pieces(n_tasks), safe(n_tasks);
    std::vector<float> min_score(nb, -INFINITY);            // by alignment position
    std::vector<char> two_round(launches.size(), 0);
    std::vector<size_t> ck_up(nb, 0);
    size_t ck_floats = 0;
    for (size_t li = 0; li < launches.size(); ++li) {
        auto& L = launches[li];
        for (int x = 0; x < L.count; ++x) {
            const int pos = L.first + x, i = out.order[pos];
            const int n = in.n[i], R = in.R[i], ov = overlap[i];
            const size_t per_ckpt = (size_t)STRQ_CKPT_FIELDS(R) * 64;
            Piece* pf = &pieces[(size_t)L.first_task + (size_t)x * L.segs];
            Piece* ps = &safe[(size_t)L.first_task + (size_t)x * L.segs];
            cut(n, L.segs, ov, ps);
            const int ov_fast = overlap_for(in.m[i], ov);
            if (wins[i].n_win > 0) {
                // the screen's windows, each started cold as far to the left as a path with the screen's lower bound can reach.
                // Certified like the short overlaps: the best score found must reach that lower bound -- then every column
                // outside the windows is excluded by its upper bound and every path that matters lies inside a piece;
                // an alignment that does not reach it runs its whole read in the second round.
                const ScreenWindows& w = wins[i];
                const int ov_w = std::min(ov > 0 ? ov : (1 << 30), align_overlap_for_score(c->ap, in.m[i], w.lower_bound));
                for (int k = 0; k < L.segs; ++k) {
                    if (k < w.n_win) { const int start = std::max(0, w.lo[k] - 1 - ov_w); pf[k].col_off = start; pf[k].n = w.hi[k] - start; }
                    else { pf[k].col_off = 0; pf[k].n = 0; }
                }
                min_score[pos] = std::max(w.lower_bound, align_segment_min_score(c->ap, in.m[i], ov_w));
                two_round[li] = 1;
            } else if (L.segs > 1 && ov_fast < ov && cut(n, L.segs, ov_fast, pf) > 1) {
                min_score[pos] = align_segment_min_score(c->ap, in.m[i], ov_fast);
                two_round[li] = 1;
            } else {
                for (int k = 0; k < L.segs; ++k) pf[k] = ps[k];
            }
            for (int k = 0; k < L.segs; ++k) {
                pf[k].ck = ps[k].ck = ck_floats;
                ck_floats += (size_t)align_num_ckpts(std::max(pf[k].n, ps[k].n)) * per_ckpt;
            }
            if (L.NS > 1) { ck_up[i] = ck_floats; ck_floats += (size_t)(L.NS - 1) * align_num_ckpts(n) * per_ckpt; }
        }
    }
    bool any_two_round = false;
    for (char t : two_round) any_two_round |= t != 0;
    const size_t n_all_tasks = safe0 + (any_two_round ? n_tasks : 0);
    STRQ_HIP(c, c->ckpt.reserve(ck_floats * 4 + 256));
    const int r2_cap = did_coarse ? std::max(1024, std::min(nb, 8192)) : 0;      // alignments the coarse screen's second look can take: room for four pieces each
    const size_t r2_0 = n_all_tasks;                                  // its tasks, behind everything else
    STRQ_HIP(c, c->tasks.reserve((n_all_tasks + (size_t)r2_cap * 4) * sizeof(AlignTask)));
    // results: [fast pieces | per alignment | safe pieces], then pick, min_score, redo lists and counters
    STRQ_HIP(c, c->results.reserve((2 * n_tasks + nb) * sizeof(AlignResult) + (size_t)nb * 12 + launches.size() * 4 + 256));
    std::vector<AlignTask> tasks(n_all_tasks);      // pieces, upper strips, one head per alignment, safe pieces
    AlignTask* d_tasks = c->tasks.as<AlignTask>();
    AlignTask* d_heads = d_tasks + n_tasks + n_up;
    AlignResult* d_seg = c->results.as<AlignResult>();
    AlignResult* d_res = d_seg + n_tasks;
    AlignResult* d_seg_safe = d_res + nb;
    int32_t* d_pick = reinterpret_cast<int32_t*>(d_seg_safe + n_tasks);
    float* d_min_score = reinterpret_cast<float*>(d_pick + nb);
    int* d_redo = reinterpret_cast<int*>(d_min_score + nb);
    int* d_redo_count = d_redo + nb;                // one counter per launch
    for (auto& L : launches) {
        for (int x = 0; x < L.count; ++x) {
            const int pos = L.first + x, i = out.order[pos];
            const int R = in.R[i], M = in.m[i];
            AlignTask base; std::memset(&base, 0, sizeof(base));
            base.levels = in.d_levels + in.read_off[in.read[i]];
            base.rec = c->rec.as<int32_t>() + out.rec_off[i];
            base.n = in.n[i]; base.n_full = in.n[i]; base.m_total = M;
            auto strip = [&](int row0, int rows, size_t ck, int sidx) {
                AlignTask t = base;
                const int j = NJ[i] > 1 ? J0[i] + sidx : J0[i];            // the table that covers this strip's classes
                const int k0 = row0 / S, k1 = (row0 + rows - 1) / S;
                t.row0 = row0; t.m = rows; t.k = k1 - k0 + 1; t.tsize = info[j].total;
                t.table = jobs[j].table; t.table3 = jobs[j].table3; t.band_lo = jobs[j].band_lo + (k0 - job_k0[j]);
                t.col0 = c->col0.as<float>() + col0_off[i] + row0;
                t.ckpt = c->ckpt.as<float>() + ck;
                return t;
            };
            tasks[n_tasks + n_up + pos] = strip(0, M, 0, 0);       // head: what finalize reads (rec, m_total, n)
            if (L.NS == 1) {
                for (int k = 0; k < L.segs; ++k) {
                    const size_t ti = (size_t)L.first_task + (size_t)x * L.segs + k;
                    AlignTask t = strip(0, M, pieces[ti].ck, 0);
                    t.levels += pieces[ti].col_off; t.n = pieces[ti].n; t.col_off = pieces[ti].col_off;
                    tasks[ti] = t;
                    if (any_two_round) {
                        AlignTask u = strip(0, M, safe[ti].ck, 0);
                        u.levels += safe[ti].col_off; u.n = safe[ti].n; u.col_off = safe[ti].col_off;
                        tasks[safe0 + ti] = u;
                    }
                }
            } else {
                // strips top to bottom: the upper ones (level-major behind the pieces) hand their last row {S, V} to the
                // next through HBM; the bottom strip sits in the launch's piece slot and holds the result
                const int rows_s = 64 * R;
                const size_t ck_strip = (size_t)align_num_ckpts(in.n[i]) * STRQ_CKPT_FIELDS(R) * 64;
                float* bnd = c->bnd.as<float>() + bnd_off[i];
                const size_t bnd_stride = 2 * ((size_t)in.n[i] + 2);
                for (int sidx = 0; sidx < L.NS; ++sidx) {
                    const bool last = sidx == L.NS - 1;
                    const int row0 = sidx * rows_s, rows = last ? M - row0 : rows_s;
                    const size_t slot = last ? (size_t)L.first_task + x : (size_t)L.first_up + (size_t)sidx * L.count + x;
                    AlignTask t = strip(row0, rows, last ? pieces[(size_t)L.first_task + x].ck : ck_up[i] + (size_t)sidx * ck_strip, sidx);
                    if (sidx > 0) { t.bnd_in = bnd + (size_t)(sidx - 1) * bnd_stride; t.up = d_tasks + L.first_up + (size_t)(sidx - 1) * L.count + x; }
                    if (!last) t.bnd_out = bnd + (size_t)sidx * bnd_stride;
                    tasks[slot] = t;
                    if (last && any_two_round) tasks[safe0 + slot] = t;
                }
            }
        }
    }
    STRQ_HIP(c, hipMemcpyAsync(d_tasks, tasks.data(), tasks.size() * sizeof(AlignTask), hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemsetAsync(d_seg, 0, (2 * n_tasks + nb) * sizeof(AlignResult), st));
    if (any_two_round) {
        STRQ_HIP(c, hipMemcpyAsync(d_min_score, min_score.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
        STRQ_HIP(c, hipMemsetAsync(d_redo_count, 0, launches.size() * 4, st));
    }
    size_t scratch_words = 0;
    for (auto& L : launches) scratch_words = std::max(scratch_words, align_trace_scratch_words_per_wave(L.R));
    const int trace_wpb = 8;
    STRQ_HIP(c, c->scratch.reserve(scratch_words * 8 * (size_t)c->n_cu * trace_wpb));
    int qi = STRQ_QUEUE_FIRST;
    out.n_launches = 0; out.wave_steps = 0; out.columns = 0;
    for (size_t t = 0; t < n_tasks + n_up; ++t) if (tasks[t].n > 0) { out.wave_steps += align_num_steps(tasks[t].n); out.columns += tasks[t].n; }
    if (!launches.empty()) {
        const Launch& L = launches.back(); out.segs = L.segs; out.tables = L.tables; out.packed = L.packed; out.rows_per_lane = L.R;
        const int i = out.order[L.first];
        out.overlap_worst = overlap[i]; out.overlap_first = L.segs > 1 ? overlap_for(in.m[i], overlap[i]) : 0;
        const bool seg_kernel = L.NS == 1 && collapsed;
        const int32_t g[8] = {L.segs, L.tables, seg_kernel ? align_segments_wpe(L.segs, L.tables) : 0, L.R, L.packed, out.overlap_first, out.overlap_worst, (int32_t)launches.size()};
        std::memcpy(c->geometry, g, sizeof(g));
    }
    int max_ns = 1;
    for (auto& L : launches) max_ns = std::max(max_ns, L.NS);
    // one queue head per launch: forward (per strip level), second round, trace
    if ((size_t)STRQ_QUEUE_FIRST + launches.size() * ((size_t)max_ns + 2) + 16 > (size_t)STRQ_QUEUE_SLOTS) {
        c->err = "too many distinct (flank shape, table size) groups in one batch"; return STRQ_ERR_UNSUPPORTED;
    }
    // Behind the screen a sub-batch is thousands of windows (one launch, one wave per alignment) and a handful of alignments that
    // run large parts of their reads (launches of two / four waves per alignment): a ~20 ms tail if the launches follow each other.
    // The launch with the most alignments runs on a second stream, next to the others (gpurun_out/r4aa: degraded reads).
    // Round 5: every launch but the first on a stream of its own (up to three side streams): behind the coarse screen there are three
    // launches of comparable length (alignments with one, two and three or four windows).
    bool side = false; int n_side = 0;
    std::vector<hipStream_t> lstream(launches.size(), st);
    if (c->screen_ran && launches.size() > 1 && max_ns == 1 && collapsed && !strq::opt("STRQ_ONE_STREAM")) {
        side = true;
        if (!c->ev_fork) STRQ_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        STRQ_HIP(c, hipEventRecord(c->ev_fork, st));
        // the launch with the most alignments goes to the side stream (every launch but the first on a stream of its own measured the
        // same: gpurun_out/r5n)
        size_t biggest = 0;
        for (size_t li = 1; li < launches.size(); ++li) if (launches[li].count > launches[biggest].count) biggest = li;
        int used = 0;
        {
            const int k = 0;
            if (!c->side_stream[k]) { STRQ_HIP(c, hipStreamCreateWithFlags(&c->side_stream[k], hipStreamNonBlocking)); STRQ_HIP(c, hipEventCreateWithFlags(&c->side_join[k], hipEventDisableTiming)); }
            STRQ_HIP(c, hipStreamWaitEvent(c->side_stream[k], c->ev_fork, 0));
            lstream[biggest] = c->side_stream[k]; ++used;
        }
        n_side = std::min(used, 3);
    }
    for (int level = 0; level < max_ns; ++level) {          // top strips first, then the strips below them
        for (size_t lidx = 0; lidx < launches.size(); ++lidx) {
            auto& L = launches[lidx];
            hipStream_t lst = lstream[lidx];
            if (level >= L.NS) continue;
            STRQ_DBG("forward launch R=%d strips=%d slice dwords=%d packed=%d level=%d count=%d tables/CU=%d segments=%d", L.R, L.NS, L.lds_floats, L.packed, level, L.count, L.tables, L.segs);
            int rc;
            if (L.NS == 1 && collapsed) {
                rc = launch_align_segments(lst, L.R, S, d_tasks + L.first_task, d_seg + L.first_task, L.count, L.segs, c->queue.as<int>() + qi,
                                           c->ap, L.lds_floats, L.tables, c->n_cu, L.packed, nullptr, nullptr, L.known_last_row);
            } else {
                const bool last = level == L.NS - 1;
                const AlignTask* dt = last ? d_tasks + L.first_task : d_tasks + L.first_up + (size_t)level * L.count;
                const int mode = (level > 0 ? 1 : 0) | (last ? 0 : 2);
                rc = launch_align(st, L.R, S, dt, d_seg + L.first_task, L.count, c->queue.as<int>() + qi, c->ap, L.lds_floats, L.tables, c->n_cu,
                                  c->scratch.as<uint64_t>(), 0, mode, L.packed);
            }
            if (rc) { c->err = "align launch failed"; return STRQ_ERR_DEVICE; }
            ++qi; ++out.n_launches;
        }
    }
    if (side) {
        for (int k = 0; k < n_side; ++k) {
            STRQ_HIP(c, hipEventRecord(c->side_join[k], c->side_stream[k]));
            STRQ_HIP(c, hipStreamWaitEvent(st, c->side_join[k], 0));
        }
    }
    for (size_t li = 0; li < launches.size(); ++li) {
        auto& L = launches[li];
        int* redo = d_redo + L.first; int* cnt = d_redo_count + li;
        if (launch_align_combine(st, d_tasks + L.first_task, d_seg + L.first_task, L.count, L.segs, d_res + L.first, d_pick + L.first, L.first_task,
                                 nullptr, nullptr, two_round[li] ? d_min_score + L.first : nullptr, redo, cnt,
                                 c->redo_total.p ? c->redo_total.as<unsigned int>() : nullptr)) { c->err = "combine launch failed"; return STRQ_ERR_DEVICE; }
    }
    // ---- the coarse screen's second look.  Its first look hands the exact pass at most sp.max_cand chunks per alignment, taken
    // with a margin; an alignment whose best score found (B1, a real path's score) does not reach the certificate of that
    // selection gets every chunk whose bound reaches B1 -- nothing else can beat B1 -- as new windows, four cold-started pieces,
    // and the result of that pass is final (it contains the first look's optimum).  One host round trip, only for those
    // alignments; what does not fit here (no windows above the cold-start bound, more than r2_cap alignments) keeps the
    // whole-read second round below.
    std::vector<std::vector<int>> redo_keep(launches.size());
    std::vector<int> redo_keep_n(launches.size(), 0);
    if (did_coarse && any_two_round && r2_cap > 0 && !strq::opt("STRQ_SCREEN2_NO_SECOND_LOOK")) {
        std::vector<int> cnt(launches.size()), hredo((size_t)nb);
        std::vector<AlignResult> hres((size_t)nb);
        STRQ_HIP(c, hipMemcpyAsync(cnt.data(), d_redo_count, launches.size() * 4, hipMemcpyDeviceToHost, st));
        STRQ_HIP(c, hipMemcpyAsync(hredo.data(), d_redo, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
        STRQ_HIP(c, hipMemcpyAsync(hres.data(), d_res, (size_t)nb * sizeof(AlignResult), hipMemcpyDeviceToHost, st));
        STRQ_HIP(c, hipStreamSynchronize(st));
        struct Look2 { int pos, i, g2; float b1; };
        std::vector<Look2> l2;
        bool any_redo = false;
        for (size_t li = 0; li < launches.size(); ++li) {
            if (!two_round[li]) continue;
            const Launch& L = launches[li];
            for (int x = 0; x < cnt[li]; ++x) {
                const int a = hredo[(size_t)L.first + x], pos = L.first + a, i = out.order[pos];
                any_redo = true;
                if (g2_of[i] >= 0 && wins[i].n_win > 0 && (int)l2.size() < r2_cap && L.NS == 1) l2.push_back({pos, i, g2_of[i], hres[(size_t)pos].best});
                else redo_keep[li].push_back(a);
            }
        }
        {
            // more than a third of the alignments missed the first look's certificate: stop here (align_core starts the sub-batch over)
            size_t n_redo = l2.size();
            for (auto& v : redo_keep) n_redo += v.size();
            const bool forced = scr_forced || mode_coarse || c->screen_mode_last == 1;
            if (allow_bail && !forced && nb >= 64 && 3 * n_redo > (size_t)nb && !strq::opt("STRQ_SCREEN2_NO_BAIL")) {
                c->coarse_pause = std::min(256, 8 << std::min(c->coarse_fail, 5)); ++c->coarse_fail;
                float ms = 0;
                STRQ_HIP(c, hipEventRecord(c->ev[3], st));
                STRQ_HIP(c, hipStreamSynchronize(st));
                STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->aborted_fwd_ms += ms;
                STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[5], c->ev[6])); c->aborted_screen_ms += ms;
                c->look2_served += (int64_t)n_redo;          // counted by the combine kernel, dropped with this attempt
                STRQ_DBG("coarse screen: %zu of %d alignments missed the first look's certificate -> the sub-batch starts over with the fine screen, pause %d", n_redo, nb, c->coarse_pause);
                *bailed = true;
                return STRQ_OK;
            }
        }
        if (!l2.empty()) {
            const int n2 = (int)l2.size();
            // device scratch: list, thresholds, windows, positions, piece results, alignment results, picks
            const size_t o_list = 0, o_theta = o_list + (size_t)n2 * 4, o_win = (o_theta + (size_t)n2 * 4 + 15) & ~(size_t)15,
                         o_pos = o_win + (size_t)n2 * sizeof(ScreenWindows), o_seg = (o_pos + (size_t)n2 * 4 + 15) & ~(size_t)15,
                         o_res = o_seg + (size_t)n2 * 4 * sizeof(AlignResult), o_pick = o_res + (size_t)n2 * sizeof(AlignResult), o_end = o_pick + (size_t)n2 * 4;
            STRQ_HIP(c, c->misc.reserve(o_end + 256));
            char* mb = c->misc.as<char>();
            std::vector<int32_t> h_list((size_t)n2), h_theta((size_t)n2), h_pos((size_t)n2);
            for (int k2 = 0; k2 < n2; ++k2) {
                const Look2& e = l2[(size_t)k2];
                h_list[(size_t)k2] = e.g2; h_pos[(size_t)k2] = e.pos;
                // chunk value v <-> S sc = v - m |e_v| sc: every chunk below theta2 holds float32 scores below (theta2 + shift + slack) / sc <= B1
                const double b1s = std::floor((double)e.b1 * sp2.sc);
                h_theta[(size_t)k2] = (int32_t)std::max(-2.0e9, std::min(2.0e9, b1s + (double)in.m[e.i] * sp2.v - (double)sp2.slack));
            }
            STRQ_HIP(c, hipMemcpyAsync(mb + o_list, h_list.data(), (size_t)n2 * 4, hipMemcpyHostToDevice, st));
            STRQ_HIP(c, hipMemcpyAsync(mb + o_theta, h_theta.data(), (size_t)n2 * 4, hipMemcpyHostToDevice, st));
            STRQ_HIP(c, hipMemcpyAsync(mb + o_pos, h_pos.data(), (size_t)n2 * 4, hipMemcpyHostToDevice, st));
            if (launch_screen_windows(st, d_st2, n2, sp2, d_bound2, reinterpret_cast<ScreenWindows*>(mb + o_win),
                                      reinterpret_cast<const int32_t*>(mb + o_list), reinterpret_cast<const int32_t*>(mb + o_theta))) { c->err = "screen windows launch failed"; return STRQ_ERR_DEVICE; }
            std::vector<ScreenWindows> w2((size_t)n2);
            STRQ_HIP(c, hipMemcpyAsync(w2.data(), mb + o_win, (size_t)n2 * sizeof(ScreenWindows), hipMemcpyDeviceToHost, st));
            STRQ_HIP(c, hipStreamSynchronize(st));
            // tasks: alignments with the same rows per lane share a launch (STRique's flanks: one launch).  An alignment's windows are
            // cut into pieces of at most 8192 own columns (each started cold like any piece), four pieces per workgroup, as many
            // workgroups ("groups") as that takes: an alignment whose candidates lie all over its read runs on dozens of waves
            // instead of being a 20 ms tail on four
            std::map<int, std::vector<int>> by_R;
            for (int k2 = 0; k2 < n2; ++k2) {
                if (w2[(size_t)k2].n_win > 0) by_R[in.R[l2[(size_t)k2].i]].push_back(k2);
                else {          // no windows above the pieces' cold-start bound: the whole read, as before
                    for (size_t li = 0; li < launches.size(); ++li) { const Launch& L = launches[li]; if (l2[(size_t)k2].pos >= L.first && l2[(size_t)k2].pos < L.first + L.count) redo_keep[li].push_back(l2[(size_t)k2].pos - L.first); }
                }
            }
            const int piece_cols = 8192;          // (4096 / 2048 own columns per piece: forward stage 118.3 / 119.7 ms against 114.9 -- every piece pays its cold start: gpurun_out/r6n)
            std::vector<AlignTask> t2; t2.reserve((size_t)n2 * 4);
            std::vector<int32_t> pos_sorted, grp_first;          // per alignment: its slot, its first group
            size_t ck2 = 0; int slot = 0, n_grp = 0; double cols2 = 0;
            struct G2 { int R, first_grp, n_grp, lds, tables; bool known; };
            std::vector<G2> groups2;
            const size_t r2_task_cap = (size_t)r2_cap * 4;
            for (auto& kv : by_R) {
                G2 g{kv.first, n_grp, 0, 0, 0, kv.first == 14};
                for (int k2 : kv.second) {
                    const Look2& e = l2[(size_t)k2];
                    const ScreenWindows& w = w2[(size_t)k2];
                    const int ov = overlap[e.i];
                    const int ov_w = std::min(ov > 0 ? ov : (1 << 30), align_overlap_for_score(c->ap, in.m[e.i], w.lower_bound));
                    int need = 0;
                    for (int q = 0; q < w.n_win; ++q) need += (w.hi[q] - w.lo[q] + piece_cols) / piece_cols;
                    need = (need + 3) & ~3;
                    if (t2.size() + (size_t)need > r2_task_cap) {          // out of room: the whole read, as before
                        for (size_t li = 0; li < launches.size(); ++li) { const Launch& L = launches[li]; if (e.pos >= L.first && e.pos < L.first + L.count) redo_keep[li].push_back(e.pos - L.first); }
                        continue;
                    }
                    const AlignTask& head = tasks[n_tasks + n_up + (size_t)e.pos];
                    const size_t per_ckpt = (size_t)STRQ_CKPT_FIELDS(in.R[e.i]) * 64;
                    int made = 0;
                    for (int q = 0; q < w.n_win; ++q) {
                        const int len = w.hi[q] - w.lo[q] + 1, np = (len + piece_cols - 1) / piece_cols;
                        for (int j = 0; j < np; ++j) {
                            const int own_lo = w.lo[q] + (int)((long)len * j / np), own_hi = w.lo[q] + (int)((long)len * (j + 1) / np) - 1;
                            AlignTask t = head;
                            const int start = std::max(0, own_lo - 1 - ov_w);
                            t.levels = head.levels + start; t.n = own_hi - start; t.col_off = start;
                            t.ckpt = reinterpret_cast<float*>((uintptr_t)ck2 * 4);          // offset for now: the buffer is sized below
                            ck2 += (size_t)align_num_ckpts(t.n) * per_ckpt;
                            cols2 += t.n; out.wave_steps += align_num_steps(t.n); out.columns += t.n;
                            t2.push_back(t); ++made;
                        }
                        c->screen_stats[4] += len;
                    }
                    for (; made < need; ++made) { AlignTask t = head; t.n = 0; t.col_off = 0; t2.push_back(t); }
                    g.lds = std::max(g.lds, tab_total[e.i]); g.known = g.known && in.m[e.i] == 870;
                    pos_sorted.push_back(e.pos); grp_first.push_back(n_grp); n_grp += need / 4; ++slot;
                }
                g.n_grp = n_grp - g.first_grp;
                if (g.n_grp <= 0) continue;
                g.tables = tables_for(g.lds, 4);
                if (g.tables < 1) { c->err = "score table does not fit LDS"; return STRQ_ERR_UNSUPPORTED; }
                groups2.push_back(g);
            }
            grp_first.push_back(n_grp);
            c->look2_served += slot;
            if (slot > 0) {
                // device scratch of the launch: piece results, group results, picks, group ranges, slots
                const size_t p_seg = 0, p_res = p_seg + (size_t)n_grp * 4 * sizeof(AlignResult), p_pick = p_res + (size_t)n_grp * sizeof(AlignResult),
                             p_first = p_pick + (size_t)n_grp * 4, p_pos = p_first + ((size_t)slot + 1) * 4, p_end = p_pos + (size_t)slot * 4;
                STRQ_HIP(c, c->ckpt2.reserve(ck2 * 4 + 256 + p_end + 256));
                char* rb = c->ckpt2.as<char>() + ((ck2 * 4 + 255) & ~(size_t)255);
                for (size_t x = 0; x < t2.size(); ++x) if (t2[x].n > 0) t2[x].ckpt = c->ckpt2.as<float>() + ((uintptr_t)t2[x].ckpt / 4);
                AlignResult* d_seg2 = reinterpret_cast<AlignResult*>(rb + p_seg); AlignResult* d_res2 = reinterpret_cast<AlignResult*>(rb + p_res);
                int32_t* d_pick2 = reinterpret_cast<int32_t*>(rb + p_pick);
                STRQ_HIP(c, hipMemcpyAsync(d_tasks + r2_0, t2.data(), t2.size() * sizeof(AlignTask), hipMemcpyHostToDevice, st));
                STRQ_HIP(c, hipMemcpyAsync(rb + p_first, grp_first.data(), grp_first.size() * 4, hipMemcpyHostToDevice, st));
                STRQ_HIP(c, hipMemcpyAsync(rb + p_pos, pos_sorted.data(), (size_t)slot * 4, hipMemcpyHostToDevice, st));
                STRQ_HIP(c, hipMemsetAsync(d_seg2, 0, (size_t)n_grp * 4 * sizeof(AlignResult), st));
                for (const G2& g : groups2) {
                    if (launch_align_segments(st, g.R, S, d_tasks + r2_0 + (size_t)g.first_grp * 4, d_seg2 + (size_t)g.first_grp * 4, g.n_grp, 4, c->queue.as<int>() + qi,
                                              c->ap, g.lds, g.tables, c->n_cu, 0, nullptr, nullptr, g.known)) { c->err = "align launch failed"; return STRQ_ERR_DEVICE; }
                    ++qi; ++out.n_launches;
                    if (launch_align_combine(st, d_tasks + r2_0 + (size_t)g.first_grp * 4, d_seg2 + (size_t)g.first_grp * 4, g.n_grp, 4, d_res2 + g.first_grp, d_pick2 + g.first_grp,
                                             (int)r2_0 + g.first_grp * 4)) { c->err = "combine launch failed"; return STRQ_ERR_DEVICE; }
                }
                if (launch_align_scatter(st, d_res2, d_pick2, reinterpret_cast<const int32_t*>(rb + p_first), reinterpret_cast<const int32_t*>(rb + p_pos), slot, d_res, d_pick)) { c->err = "scatter launch failed"; return STRQ_ERR_DEVICE; }
            }
            // does the coarse screen still pay with what the second look had to run?
            const bool forced = scr_forced || mode_coarse || c->screen_mode_last == 1;
            // (also when more than a third of the alignments needed it: on such reads -- short events, a background as high as the flank --
            // the chunks whose bound reaches the score found cover most of the read, and what does not fit the second look's task
            // room runs its whole read in the second round)
            if (coarse_all > 0 && (coarse_cols + cols2 > 0.30 * coarse_all || 3 * n2 > nb) && nb >= 64 && !forced && c->coarse_pause == 0) { c->coarse_pause = std::min(256, 8 << std::min(c->coarse_fail, 5)); ++c->coarse_fail; }
            STRQ_DBG("coarse screen, second look: %d alignments, %.2f %% of the columns (first look %.2f %%) -> pause %d", slot, 100.0 * cols2 / std::max(1.0, coarse_all), 100.0 * coarse_cols / std::max(1.0, coarse_all), c->coarse_pause);
        }
        if (any_redo) {
            // what is left for the whole-read second round
            for (size_t li = 0; li < launches.size(); ++li) {
                if (!two_round[li]) continue;
                redo_keep_n[li] = (int)redo_keep[li].size();
                if (!redo_keep[li].empty()) STRQ_HIP(c, hipMemcpyAsync(d_redo + launches[li].first, redo_keep[li].data(), redo_keep[li].size() * 4, hipMemcpyHostToDevice, st));
            }
            STRQ_HIP(c, hipMemcpyAsync(d_redo_count, redo_keep_n.data(), launches.size() * 4, hipMemcpyHostToDevice, st));
        }
    }
    for (size_t li = 0; li < launches.size(); ++li) {
        auto& L = launches[li];
        int* redo = d_redo + L.first; int* cnt = d_redo_count + li;
        if (!two_round[li]) continue;
        // second round (normally empty): the listed alignments again, cut with the worst-case overlap
        if (launch_align_segments(st, L.R, S, d_tasks + safe0 + L.first_task, d_seg_safe + L.first_task, L.count, L.segs, c->queue.as<int>() + qi,
                                  c->ap, L.lds_floats, L.tables, c->n_cu, L.packed, redo, cnt, L.known_last_row)) { c->err = "align launch failed"; return STRQ_ERR_DEVICE; }
        ++qi;
        if (launch_align_combine(st, d_tasks + safe0 + L.first_task, d_seg_safe + L.first_task, L.count, L.segs, d_res + L.first, d_pick + L.first,
                                 (int)safe0 + L.first_task, redo, cnt, nullptr, nullptr, nullptr)) { c->err = "combine launch failed"; return STRQ_ERR_DEVICE; }
    }
    STRQ_HIP(c, hipEventRecord(c->ev[3], st));
    for (size_t li = 0; li < launches.size();) {
        // the trace pass keeps one table per wave (it re-runs a few blocks of one piece per alignment); pick holds
        // absolute task indices (a fast or a safe piece).  Launches that differ in their forward geometry only (waves per
        // alignment behind a screen: one, two and four windows) are neighbours in result order and share one trace launch
        const Launch& L = launches[li];
        int count = L.count, lds_floats = L.lds_floats;
        size_t lj = li + 1;
        for (; lj < launches.size(); ++lj) {
            const Launch& M = launches[lj];
            if (M.R != L.R || M.NS != L.NS || M.packed != L.packed || L.NS != 1 || M.first != L.first + count) break;
            count += M.count; lds_floats = std::max(lds_floats, M.lds_floats);
        }
        const int wpb = std::min(trace_wpb, (160 * 1024) / (std::max(lds_floats, 1) * 4));
        if (launch_align(st, L.R, S, d_tasks, d_res + L.first, count, c->queue.as<int>() + qi, c->ap, lds_floats, wpb, c->n_cu,
                         c->scratch.as<uint64_t>(), 1, 0, L.packed, d_pick + L.first)) { c->err = "trace launch failed"; return STRQ_ERR_DEVICE; }
        ++qi; li = lj;
    }
    STRQ_HIP(c, hipEventRecord(c->ev[4], st));
    out.d_tasks = d_heads; out.d_results = d_res; out.d_rec = c->rec.as<int32_t>();
    return STRQ_OK;
}

int align_core_times(strq_ctx* c, float* t_lut, float* t_fwd, float* t_tr)
{
    float ms;
    STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); *t_lut += ms;
    STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); *t_fwd += ms + c->aborted_fwd_ms;
    c->screen_stats[0] += c->aborted_screen_ms;          // an attempt that was stopped after the coarse screen's first look (align_core)
    c->aborted_fwd_ms = c->aborted_screen_ms = 0;
    STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[3], c->ev[4])); *t_tr += ms;
    if (c->screen_ran) { STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[5], c->ev[6])); c->screen_stats[0] += ms; }
    return STRQ_OK;
}

size_t align_workspace_bytes(int n, int m, int R, int NS)
{
    // column segments recompute `overlap` columns per extra piece: a quarter more checkpoints plus a constant covers it
    size_t b = (size_t)NS * (align_num_ckpts(n) + align_num_ckpts(n) / 4 + 300) * STRQ_CKPT_FIELDS(R) * 64 * 4;
    if (NS > 1) b += (size_t)(NS - 1) * ((size_t)n + 2) * 8;
    (void)m;
    return b;
}

int align_validate_flank(strq_ctx* c, const float* f, int64_t mm, int samples, int* k_out, int* R_out, int* NS_out)
{
    if (mm < 1 || samples < 1 || mm % samples != 0) { c->err = "flank length must be a positive multiple of `samples`"; return STRQ_ERR_UNSUPPORTED; }
    for (int64_t i = 0; i < mm; ++i)
        if (std::memcmp(&f[i], &f[i - i % samples], 4) != 0) { c->err = "flank is not made of runs of `samples` equal values"; return STRQ_ERR_UNSUPPORTED; }
    const int S = align_effective_samples(samples);       // the run length the kernels work with (k-mer classes of S rows)
    int R = 0, NS = 0;
    if (mm > (1 << 20) || !align_plan((int)mm, S, &R, &NS)) { c->err = "flank shape not covered by the compiled kernels"; return STRQ_ERR_UNSUPPORTED; }
    if (S == 6 && mm / S > STRQ_LUT_MAX_K && NS == 1) { R = 12; NS = (int)((mm + 64 * 12 - 1) / (64 * 12)); }      // one table would not fit: one per strip of 128 classes
    *k_out = (int)(mm / S); *R_out = R; *NS_out = NS;
    return STRQ_OK;
}

static int run_align_batch(strq_ctx* c, const BatchIn& in, const BatchOut& out)
{
    const int S = align_effective_samples(in.samples);
    const int64_t NA = in.n_align;
    std::fill(c->timing, c->timing + 8, 0.0f);
    for (double& v : c->screen_stats) v = 0;
    if (NA == 0) return STRQ_OK;
    std::vector<int> m(NA), k(NA), R(NA), n(NA), NS(NA);
    for (int64_t a = 0; a < NA; ++a) {
        const int64_t mm = in.flank_off[a + 1] - in.flank_off[a];
        const int rd = in.align_read[a];
        if (rd < 0 || rd >= in.n_reads) { c->err = "align_read out of range"; return STRQ_ERR_ARG; }
        const int64_t nn = in.read_off[rd + 1] - in.read_off[rd];
        if (nn < 0 || nn > (int64_t)1 << 30) { c->err = "bad read length"; return STRQ_ERR_ARG; }
        const int rc = align_validate_flank(c, in.flank + in.flank_off[a], mm, in.samples, &k[a], &R[a], &NS[a]);
        if (rc) return rc;
        m[a] = (int)mm; n[a] = (int)nn;
    }
    hipStream_t st = c->stream;
    c->second_round[0] = 0; c->second_round[1] = NA; c->look2_served = 0;          // strq_last_second_round of this call
    STRQ_HIP(c, c->redo_total.reserve(64));
    STRQ_HIP(c, hipMemsetAsync(c->redo_total.p, 0, 64, st));
    const int64_t tot_levels = in.read_off[in.n_reads];
    STRQ_HIP(c, c->levels.reserve((size_t)tot_levels + 64));
    STRQ_HIP(c, c->level_val.reserve((size_t)in.n_reads * 256 * 4));
    STRQ_HIP(c, hipMemcpyAsync(c->levels.p, in.levels, (size_t)tot_levels, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemcpyAsync(c->level_val.p, in.level_val, (size_t)in.n_reads * 256 * 4, hipMemcpyHostToDevice, st));
    int64_t a0 = 0;
    float t_lut = 0, t_fwd = 0, t_tr = 0, n_hard = 0, n_launch = 0;
    std::fill(c->counters, c->counters + 8, 0.0);
    while (a0 < NA) {
        int64_t a1 = a0; size_t ck_bytes = 0;
        while (a1 < NA) {
            const size_t need = align_workspace_bytes(n[a1], m[a1], R[a1], NS[a1]);
            if (a1 > a0 && ck_bytes + need > c->max_ws_bytes) break;
            ck_bytes += need; ++a1;
        }
        const int nb = (int)(a1 - a0);
        AlignCoreIn ci; AlignCoreOut co;
        ci.nb = nb; ci.samples = S; ci.d_levels = c->levels.as<uint8_t>(); ci.read_off = in.read_off;
        ci.d_level_val = c->level_val.as<float>();
        std::vector<const float*> fl(nb);
        for (int i = 0; i < nb; ++i) fl[i] = in.flank + in.flank_off[a0 + i];
        ci.read = in.align_read + a0; ci.n = &n[a0]; ci.m = &m[a0]; ci.k = &k[a0]; ci.R = &R[a0]; ci.NS = &NS[a0]; ci.flank = fl.data();
        int rc = align_core(c, ci, co);
        if (rc) return rc;
        std::vector<AlignResult> res(nb);
        std::vector<int32_t> h_rec(co.rec_total);
        STRQ_HIP(c, hipMemcpyAsync(res.data(), co.d_results, (size_t)nb * sizeof(AlignResult), hipMemcpyDeviceToHost, st));
        if (out.rec) STRQ_HIP(c, hipMemcpyAsync(h_rec.data(), co.d_rec, co.rec_total * 4, hipMemcpyDeviceToHost, st));
        STRQ_HIP(c, hipStreamSynchronize(st));
        for (int pos = 0; pos < nb; ++pos) {
            const int i = co.order[pos]; const int64_t a = a0 + i;
            if (out.score) out.score[a] = res[pos].best;
            if (out.j_end) out.j_end[a] = res[pos].j_end;
            if (out.j0) out.j0[a] = res[pos].j0;
            if (out.rec) std::memcpy(out.rec + in.flank_off[a], &h_rec[co.rec_off[i]], (size_t)m[a] * 4);
        }
        rc = align_core_times(c, &t_lut, &t_fwd, &t_tr);
        if (rc) return rc;
        n_hard += co.n_hard; n_launch += co.n_launches;
        c->counters[0] += co.wave_steps; c->counters[1] += co.columns; c->counters[2] += nb;
        c->counters[3] = co.segs; c->counters[4] = co.tables; c->counters[5] = co.packed; c->counters[6] = co.rows_per_lane;
        a0 = a1;
    }
    c->timing[0] = t_lut; c->timing[1] = t_fwd; c->timing[2] = t_tr; c->timing[3] = t_lut + t_fwd + t_tr; c->timing[4] = n_hard; c->timing[7] = n_launch;
    {
        unsigned int redo = 0;
        STRQ_HIP(c, hipMemcpy(&redo, c->redo_total.p, 4, hipMemcpyDeviceToHost));
        c->second_round[0] = (int64_t)redo - c->look2_served;
    }
    return STRQ_OK;
}

}  // namespace strq

extern "C" {

int strq_abi_version(void) { return 12; }

int strq_set_option(strq_ctx* c, const char* key, const char* value)
{
    if (!key || std::strncmp(key, "STRQ_", 5) != 0) { if (c) c->err = "option keys start with STRQ_"; return STRQ_ERR_ARG; }
    if (c) { std::lock_guard<std::mutex> lk(c->options_mu); if (value) c->options[key] = value; else c->options.erase(key); return STRQ_OK; }
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (value) g_opts[key] = value; else g_opts.erase(key);
    return STRQ_OK;
}

int strq_get_option(const strq_ctx* c, const char* key, char* out, int32_t cap)
{
    if (!key || !out || cap < 1) return STRQ_ERR_ARG;
    CtxScope scope(c);
    const char* v = opt(key);
    out[0] = 0;
    if (!v) return STRQ_OK;
    std::strncpy(out, v, (size_t)cap - 1); out[cap - 1] = 0;
    return STRQ_OK;
}

int strq_ctx_create(int device_id, strq_ctx** out)
{
    if (!out) return STRQ_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return STRQ_ERR_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return STRQ_ERR_DEVICE;
    strq_ctx* c = new strq_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete c; return STRQ_ERR_DEVICE; }
    c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return STRQ_ERR_DEVICE; }
    for (auto& e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return STRQ_ERR_DEVICE; }
    *out = c;
    return STRQ_OK;
}

void strq_ctx_destroy(strq_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    detect_state_free(c);
    for (DevBuf* b : {&c->levels, &c->level_val, &c->flank_cls, &c->tables, &c->tables3, &c->band_lo, &c->col0, &c->ckpt,
                      &c->rec, &c->tasks, &c->results, &c->queue, &c->scratch, &c->lutinfo, &c->hard, &c->misc,
                      &c->vit_x, &c->vit_tasks, &c->vit_bp, &c->vit_path, &c->bnd, &c->gen_codes, &c->gen_table, &c->gen_bnd, &c->gen_trace, &c->gen_hard, &c->redo_total, &c->screen, &c->ckpt2})
        b->release();
    for (HostModel* m : c->models) if (m) { m->blob.release(); delete m; }
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int k = 0; k < 3; ++k) { if (c->side_join[k]) (void)hipEventDestroy(c->side_join[k]); if (c->side_stream[k]) (void)hipStreamDestroy(c->side_stream[k]); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* strq_last_error(const strq_ctx* c) { return c ? c->err.c_str() : "null context"; }

int strq_set_align_params(strq_ctx* c, const float p[6])
{
    if (!c || !p) return STRQ_ERR_ARG;
    c->ap = AlignParams{p[0], p[1], p[2], p[3], p[4], p[5]};
    return STRQ_OK;
}
int strq_get_align_params(const strq_ctx* c, float p[6])
{
    if (!c || !p) return STRQ_ERR_ARG;
    p[0] = c->ap.open_h; p[1] = c->ap.ext_h; p[2] = c->ap.open_v; p[3] = c->ap.ext_v;
    p[4] = c->ap.dist_offset; p[5] = c->ap.dist_min;
    return STRQ_OK;
}

int strq_device_synchronize(strq_ctx* c)
{
    if (!c) return STRQ_ERR_ARG;
    STRQ_HIP(c, hipSetDevice(c->device));
    STRQ_HIP(c, hipDeviceSynchronize());
    return STRQ_OK;
}

int strq_last_timing(const strq_ctx* c, float ms[8])
{
    if (!c || !ms) return STRQ_ERR_ARG;
    std::memcpy(ms, c->timing, sizeof(c->timing));
    return STRQ_OK;
}

int strq_last_counters(const strq_ctx* c, double out[8])
{
    if (!c || !out) return STRQ_ERR_ARG;
    std::memcpy(out, c->counters, sizeof(c->counters));
    return STRQ_OK;
}

int strq_debug_screen_plan(const float params[6], int32_t samples, int32_t max_n, int32_t out[6])
{
    if (!params || !out) return 0;
    const AlignParams p{params[0], params[1], params[2], params[3], params[4], params[5]};
    ScreenParams sp;
    if (!screen_plan(p, samples, max_n, &sp)) return 0;
    out[0] = sp.sc; out[1] = sp.hh; out[2] = sp.v; out[3] = sp.cadd; out[4] = sp.slack; out[5] = sp.merge_gap;
    return 1;
}

int strq_last_overlap(const strq_ctx* c, double out[4])
{
    if (!c || !out) return STRQ_ERR_ARG;
    for (int i = 0; i < 4; ++i) out[i] = c->overlap[i];
    return STRQ_OK;
}
int strq_last_screen(const strq_ctx* c, double out[8])
{
    if (!c || !out) return STRQ_ERR_ARG;
    for (int i = 0; i < 8; ++i) out[i] = c->screen_stats[i];
    return STRQ_OK;
}

int strq_last_screen_mode(const strq_ctx* c, int32_t out[8])
{
    if (!c || !out) return STRQ_ERR_ARG;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[0] = c->screen_mode_last; out[1] = c->coarse_pause; out[2] = c->screen_pause; out[3] = (int32_t)c->coarse_margin;
    out[5] = (int32_t)std::min<int64_t>(c->look2_served, INT32_MAX);
    out[4] = c->screen_mode_last ? c->coarse_merge_last : 0;
    return STRQ_OK;
}

int strq_last_second_round(const strq_ctx* c, int64_t out[2])
{
    if (!c || !out) return STRQ_ERR_ARG;
    out[0] = c->second_round[0]; out[1] = c->second_round[1];
    return STRQ_OK;
}

int strq_last_viterbi_launches(const strq_ctx* c, int32_t out[4])
{
    if (!c || !out) return STRQ_ERR_ARG;
    std::memcpy(out, c->vit_launches, sizeof(c->vit_launches));
    return STRQ_OK;
}

int strq_last_geometry(const strq_ctx* c, int32_t out[8])
{
    if (!c || !out) return STRQ_ERR_ARG;
    std::memcpy(out, c->geometry, sizeof(c->geometry));
    return STRQ_OK;
}

int strq_align_batch(strq_ctx* c, int64_t n_align, int64_t n_reads, const uint8_t* levels,
                     const int64_t* read_off, const float* level_val, const int32_t* align_read,
                     const float* flank, const int64_t* flank_off, int32_t samples,
                     float* score, int64_t* j_end, int64_t* j0, int32_t* rec)
{
    strq::CtxScope scope_(c);
    if (!c) return STRQ_ERR_ARG;
    if (n_align < 0 || n_reads < 0 || (n_align > 0 && (!levels || !read_off || !level_val || !align_read || !flank || !flank_off)) || samples < 1) {
        c->err = "bad argument"; return STRQ_ERR_ARG;
    }
    STRQ_HIP(c, hipSetDevice(c->device));
    BatchIn in{n_align, n_reads, levels, read_off, level_val, align_read, flank, flank_off, samples};
    BatchOut out{score, j_end, j0, rec};
    return run_align_batch(c, in, out);
}

// align_overlap for inputs the 8-bit / 6-run kernels do not cover: any `a`, any `b` (align_generic.hip)
static int align_overlap_generic(strq_ctx* c, const float* a, int64_t n, const float* b, int64_t m, float* score,
                                 std::vector<int32_t>& rec, int64_t* je, int64_t* j0)
{
    if (n > ((int64_t)1 << 30) || m > ((int64_t)1 << 24) || (double)(n + 1) * (double)(m + 1) > 1.7e10) {
        c->err = "align_overlap: (n + 1) x (m + 1) exceeds the 16 GiB trace limit of the generic path"; return STRQ_ERR_UNSUPPORTED;
    }
    hipStream_t st = c->stream;
    // dictionary-encode by bit pattern (so that -0.0 / +0.0 and NaN payloads keep their own rows)
    auto encode = [](const float* x, int64_t len, std::vector<float>& vals, std::vector<uint32_t>& code) {
        std::vector<uint32_t> bits((size_t)len);
        std::memcpy(bits.data(), x, (size_t)len * 4);
        std::vector<uint32_t> uniq(bits);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        code.resize((size_t)len);
        for (int64_t i = 0; i < len; ++i) code[(size_t)i] = (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), bits[(size_t)i]) - uniq.begin());
        vals.resize(uniq.size());
        std::memcpy(vals.data(), uniq.data(), uniq.size() * 4);
    };
    std::vector<float> va, vb; std::vector<uint32_t> ca, cb;
    encode(a, n, va, ca); encode(b, m, vb, cb);
    const size_t na = va.size(), nbv = vb.size();
    if ((double)na * (double)nbv > 2.0e9) { c->err = "align_overlap: more than 2e9 distinct (a, b) value pairs"; return STRQ_ERR_UNSUPPORTED; }
    const size_t tab = std::max<size_t>(1, na * nbv);
    const int hard_cap = 1 << 20;
    STRQ_HIP(c, c->gen_codes.reserve(((size_t)n + (size_t)m + na + nbv) * 4 + 64));
    STRQ_HIP(c, c->gen_table.reserve(tab * 4 + 64));
    STRQ_HIP(c, c->gen_bnd.reserve(((size_t)n + 1) * 16 + ((size_t)m + 1) * 4 + 256));
    STRQ_HIP(c, c->gen_trace.reserve(((size_t)n + 1) * ((size_t)m + 1) + 64));
    STRQ_HIP(c, c->gen_hard.reserve((size_t)hard_cap * (sizeof(GenericHard) + 4) + 64));
    STRQ_HIP(c, c->results.reserve(sizeof(AlignResult) + 64));
    STRQ_HIP(c, c->queue.reserve(STRQ_QUEUE_BYTES));
    uint32_t* d_ca = c->gen_codes.as<uint32_t>(); uint32_t* d_cb = d_ca + n;
    float* d_va = reinterpret_cast<float*>(d_cb + m); float* d_vb = d_va + na;
    float* d_tab = c->gen_table.as<float>();
    float* d_bnd = c->gen_bnd.as<float>();
    GenericHard* d_hard = c->gen_hard.as<GenericHard>(); float* d_hvals = reinterpret_cast<float*>(d_hard + hard_cap);
    int* d_count = c->queue.as<int>();
    if (n) STRQ_HIP(c, hipMemcpyAsync(d_ca, ca.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemcpyAsync(d_cb, cb.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    if (na) STRQ_HIP(c, hipMemcpyAsync(d_va, va.data(), na * 4, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemcpyAsync(d_vb, vb.data(), nbv * 4, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemsetAsync(d_count, 0, 4, st));
    int hard_count = 0;
    if (na) {
        if (launch_generic_table(st, d_va, (int)na, d_vb, (int)nbv, d_tab, d_hard, d_count, hard_cap, c->ap)) { c->err = "table launch failed"; return STRQ_ERR_DEVICE; }
        STRQ_HIP(c, hipMemcpyAsync(&hard_count, d_count, 4, hipMemcpyDeviceToHost, st));
        STRQ_HIP(c, hipStreamSynchronize(st));
        if (hard_count > hard_cap) {       // pathological: evaluate the whole table with the host libm
            std::vector<float> full(tab);
            for (size_t y = 0; y < nbv; ++y) for (size_t x = 0; x < na; ++x) full[y * na + x] = host_cell_score(c->ap, va[x], vb[y]);
            STRQ_HIP(c, hipMemcpy(d_tab, full.data(), tab * 4, hipMemcpyHostToDevice));
        } else if (hard_count > 0) {
            std::vector<GenericHard> he((size_t)hard_count); std::vector<float> hv((size_t)hard_count);
            STRQ_HIP(c, hipMemcpy(he.data(), d_hard, (size_t)hard_count * sizeof(GenericHard), hipMemcpyDeviceToHost));
            for (int i = 0; i < hard_count; ++i) hv[(size_t)i] = host_cell_score(c->ap, va[(size_t)he[(size_t)i].ia], vb[(size_t)he[(size_t)i].ib]);
            STRQ_HIP(c, hipMemcpyAsync(d_hvals, hv.data(), (size_t)hard_count * 4, hipMemcpyHostToDevice, st));
            if (launch_generic_patch(st, d_tab, (int)na, d_hard, d_hvals, hard_count)) { c->err = "patch launch failed"; return STRQ_ERR_DEVICE; }
        }
    }
    c->timing[4] = (float)hard_count;
    std::vector<float> col0((size_t)m + 1);
    host_col0(c->ap, (int)m, col0.data());
    GenericAlignArgs ga;
    ga.code_a = d_ca; ga.code_b = d_cb; ga.table = d_tab;
    ga.bnd_S[0] = d_bnd; ga.bnd_V[0] = d_bnd + (n + 1); ga.bnd_S[1] = d_bnd + 2 * (n + 1); ga.bnd_V[1] = d_bnd + 3 * (n + 1);
    float* d_col0 = d_bnd + 4 * (n + 1);
    STRQ_HIP(c, hipMemcpyAsync(d_col0, col0.data(), ((size_t)m + 1) * 4, hipMemcpyHostToDevice, st));
    ga.col0 = d_col0; ga.trace = c->gen_trace.as<uint8_t>(); ga.result = c->results.as<AlignResult>();
    ga.p = c->ap; ga.n = (int32_t)n; ga.m = (int32_t)m; ga.na = (int32_t)na; ga.nb = (int32_t)nbv;
    if (launch_generic_align(st, ga)) { c->err = "generic align launch failed"; return STRQ_ERR_DEVICE; }
    AlignResult res;
    std::vector<uint8_t> trace(((size_t)n + 1) * ((size_t)m + 1));
    STRQ_HIP(c, hipMemcpyAsync(&res, ga.result, sizeof(res), hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipMemcpyAsync(trace.data(), ga.trace, trace.size(), hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipStreamSynchronize(st));
    // traceback (the state machine of SURVEY.md A.1; column 0 is not free: what is left of the flank there is vertical)
    rec.assign((size_t)m, 0);
    int64_t i = m, j = res.j_end; int state = 0;
    const size_t W = (size_t)n + 1;
    while (i > 0 && j > 0) {
        const uint8_t tr = trace[(size_t)i * W + (size_t)j];
        if (state == 0) {
            const unsigned dsel = tr & 3u;
            if (dsel == 0) { rec[(size_t)i - 1] = (int32_t)(j << 1); --i; --j; }
            else state = (int)dsel;
        } else if (state == 1) { --j; if (!(tr & 4u)) state = 0; }
        else { rec[(size_t)i - 1] = (int32_t)((j << 1) | 1); --i; if (!(tr & 8u)) state = 0; }
    }
    for (; i > 0; --i) rec[(size_t)i - 1] = (int32_t)((j << 1) | 1);
    *score = res.best; *je = res.j_end; *j0 = j;
    return STRQ_OK;
}

int strq_align_overlap(strq_ctx* c, const float* a, int64_t n, const float* b, int64_t m, float* score,
                       uint64_t* a_idx, uint64_t* b_idx, int32_t* rec_out, int64_t* j_end_out, int64_t* j0_out)
{
    strq::CtxScope scope_(c);
    if (!c) return STRQ_ERR_ARG;
    if ((!a && n > 0) || !b || n < 0 || m < 1 || !score) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    STRQ_HIP(c, hipSetDevice(c->device));
    std::fill(c->timing, c->timing + 8, 0.0f);
    std::vector<int32_t> rec((size_t)m);
    int64_t je = 0, j0 = 0;
    // fast path: what repeatCounter.detect passes -- at most 256 distinct values in `a` (an 8-bit signal) and a
    // flank made of runs of 6 equal samples that a compiled kernel shape covers
    std::vector<float> vals(a, a + n);
    std::sort(vals.begin(), vals.end(), [](float x, float y) { return __builtin_bit_cast(uint32_t, x) < __builtin_bit_cast(uint32_t, y); });
    vals.erase(std::unique(vals.begin(), vals.end(), [](float x, float y) { return std::memcmp(&x, &y, 4) == 0; }), vals.end());
    // run length of the flank: the gcd of its runs of equal values (6 for what detect passes)
    int run = 0;
    { int64_t start = 0;
      for (int64_t i = 1; i <= m; ++i)
          if (i == m || std::memcmp(&b[i], &b[start], 4) != 0) { int len = (int)std::min<int64_t>(i - start, 1 << 20); run = run ? std::__gcd(run, len) : len; start = i; } }
    bool fast = vals.size() <= 256 && run >= 1 && !strq::opt("STRQ_GENERIC_ALIGN");
    for (float v : vals) if (v != v) fast = false;
    if (fast) {
        std::sort(vals.begin(), vals.end());
        for (size_t x = 1; x < vals.size(); ++x) if (vals[x] == vals[x - 1]) fast = false;      // -0.0 and +0.0 both present
        int kk, RR, NN;
        if (fast && align_validate_flank(c, b, m, run, &kk, &RR, &NN) != STRQ_OK) fast = false;
    }
    if (fast) {
        std::vector<uint8_t> lv((size_t)n);
        for (int64_t i = 0; i < n; ++i) lv[(size_t)i] = (uint8_t)(std::lower_bound(vals.begin(), vals.end(), a[i]) - vals.begin());
        std::vector<float> lval(256, INFINITY);   // unused levels: score clips to dist_min
        std::copy(vals.begin(), vals.end(), lval.begin());
        const int64_t roff[2] = {0, n}, foff[2] = {0, m};
        const int32_t ar = 0;
        BatchIn in{1, 1, lv.data(), roff, lval.data(), &ar, b, foff, run};
        BatchOut out{score, &je, &j0, rec.data()};
        const int rc = run_align_batch(c, in, out);
        if (rc) return rc;
    } else {
        const int rc = align_overlap_generic(c, a, n, b, m, score, rec, &je, &j0);
        if (rc) return rc;
    }
    if (rec_out) std::memcpy(rec_out, rec.data(), (size_t)m * 4);
    if (j_end_out) *j_end_out = je;
    if (j0_out) *j0_out = j0;
    if (a_idx || b_idx) {
        // view positions as src/align_raw.h:141-146 returns them
        if (a_idx) for (int64_t x = 0; x < j0; ++x) a_idx[x] = (uint64_t)x;
        int64_t ai = j0, col = j0;
        for (int64_t kk = 0; kk < m; ++kk) {
            const int64_t j = rec[(size_t)kk] >> 1; const bool vert = rec[(size_t)kk] & 1;
            const int64_t upto = vert ? j : j - 1;       // samples consumed by horizontal steps first
            for (; ai < upto; ++ai, ++col) if (a_idx) a_idx[ai] = (uint64_t)col;
            if (b_idx) b_idx[kk] = (uint64_t)col;
            if (!vert) { if (a_idx) a_idx[ai] = (uint64_t)col; ++ai; }
            ++col;
        }
        for (; ai < n; ++ai, ++col) if (a_idx) a_idx[ai] = (uint64_t)col;
    }
    return STRQ_OK;
}

}  // extern "C"
