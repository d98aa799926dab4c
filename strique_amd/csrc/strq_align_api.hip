// C ABI: context, alignment parameters, align_overlap and its batched form.
// Host-side orchestration only; the arithmetic lives in align_kernels.hip / lut_kernels.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <time.h>
#include <unistd.h>
#include <map>
#include <numeric>
#include <vector>
#include "../../include/strique_hip.h"
#include "strq_ctx.h"
#include "align_kernels.h"
#include "lut_kernels.h"

using namespace strq;

#define STRQ_DBG(...) do { if (getenv("STRQ_DEBUG")) { fprintf(stderr, "[strq] " __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)

namespace strq {

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

static int run_align_batch(strq_ctx* c, const BatchIn& in, const BatchOut& out)
{
    const int S = in.samples;
    const int64_t NA = in.n_align;
    std::fill(c->timing, c->timing + 8, 0.0f);
    if (NA == 0) return STRQ_OK;
    // ---- validate flanks, derive classes
    std::vector<int> m(NA), k(NA), R(NA), n(NA);
    int max_k = 0;
    for (int64_t a = 0; a < NA; ++a) {
        const int64_t mm = in.flank_off[a + 1] - in.flank_off[a];
        const int rd = in.align_read[a];
        if (rd < 0 || rd >= in.n_reads) { c->err = "align_read out of range"; return STRQ_ERR_ARG; }
        const int64_t nn = in.read_off[rd + 1] - in.read_off[rd];
        if (mm < 1 || mm % S != 0 || nn < 0 || nn > (int64_t)1 << 30) { c->err = "flank length must be a positive multiple of `samples`"; return STRQ_ERR_UNSUPPORTED; }
        const float* f = in.flank + in.flank_off[a];
        for (int64_t i = 0; i < mm; ++i)
            if (std::memcmp(&f[i], &f[i - i % S], 4) != 0) { c->err = "flank is not made of runs of `samples` equal values"; return STRQ_ERR_UNSUPPORTED; }
        m[a] = (int)mm; k[a] = (int)(mm / S); n[a] = (int)nn;
        R[a] = align_pick_rows_per_lane((int)mm, S);
        if (!R[a] || k[a] > STRQ_LUT_MAX_K) { c->err = "flank shape not covered by the compiled kernels"; return STRQ_ERR_UNSUPPORTED; }
        max_k = std::max(max_k, k[a]);
    }
    hipStream_t st = c->stream;
    // ---- reads: levels + level values resident for the whole call
    const int64_t tot_levels = in.read_off[in.n_reads];
    STRQ_HIP(c, c->levels.reserve((size_t)tot_levels + 64));
    STRQ_HIP(c, c->level_val.reserve((size_t)in.n_reads * 256 * 4));
    STRQ_HIP(c, hipMemcpyAsync(c->levels.p, in.levels, (size_t)tot_levels, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, hipMemcpyAsync(c->level_val.p, in.level_val, (size_t)in.n_reads * 256 * 4, hipMemcpyHostToDevice, st));

    // ---- sub-batches bounded by checkpoint memory
    const int n_wave_slots = c->n_cu * 4;
    int64_t a0 = 0;
    float t_lut = 0, t_fwd = 0, t_tr = 0;
    while (a0 < NA) {
        int64_t a1 = a0; size_t ck_bytes = 0;
        while (a1 < NA) {
            const size_t need = (size_t)align_num_ckpts(n[a1]) * STRQ_CKPT_FIELDS(R[a1]) * 64 * 4;
            if (a1 > a0 && ck_bytes + need > c->max_ws_bytes) break;
            ck_bytes += need; ++a1;
        }
        const int nb = (int)(a1 - a0);
        // per-alignment host arrays
        size_t cls_tot = 0, col0_tot = 0, rec_tot = 0, tab_tot = 0;
        std::vector<size_t> cls_off(nb), col0_off(nb), rec_off(nb), tab_off(nb), ck_off(nb);
        { size_t ck = 0;
          for (int i = 0; i < nb; ++i) {
            const int64_t a = a0 + i;
            cls_off[i] = cls_tot; cls_tot += k[a];
            col0_off[i] = col0_tot; col0_tot += m[a] + 1;
            rec_off[i] = rec_tot; rec_tot += m[a];
            tab_off[i] = tab_tot; tab_tot += STRQ_TABLE_SLOT_FLOATS(k[a]);
            ck_off[i] = ck; ck += (size_t)align_num_ckpts(n[a]) * STRQ_CKPT_FIELDS(R[a]) * 64;
          } }
        std::vector<float> h_cls(cls_tot), h_col0(col0_tot);
        for (int i = 0; i < nb; ++i) {
            const int64_t a = a0 + i;
            const float* f = in.flank + in.flank_off[a];
            for (int kk = 0; kk < k[a]; ++kk) h_cls[cls_off[i] + kk] = f[(size_t)kk * S];
            host_col0(c->ap, m[a], &h_col0[col0_off[i]]);
        }
        STRQ_HIP(c, c->flank_cls.reserve(cls_tot * 4));
        STRQ_HIP(c, c->col0.reserve(col0_tot * 4));
        STRQ_HIP(c, c->band_lo.reserve(cls_tot * 4));
        STRQ_HIP(c, c->tables.reserve(tab_tot * 4));
        STRQ_HIP(c, c->ckpt.reserve(ck_bytes + 256));
        STRQ_HIP(c, c->rec.reserve(rec_tot * 4 + 256));
        STRQ_HIP(c, c->tasks.reserve((size_t)nb * sizeof(AlignTask)));
        STRQ_HIP(c, c->results.reserve((size_t)nb * sizeof(AlignResult)));
        STRQ_HIP(c, c->lutinfo.reserve((size_t)nb * (sizeof(LutJob) + sizeof(LutInfo))));
        const int hard_cap = 1 << 16;
        STRQ_HIP(c, c->hard.reserve((size_t)hard_cap * (sizeof(HardEntry) + 4) + 64));
        STRQ_HIP(c, c->queue.reserve(256));
        STRQ_HIP(c, hipMemcpyAsync(c->flank_cls.p, h_cls.data(), cls_tot * 4, hipMemcpyHostToDevice, st));
        STRQ_HIP(c, hipMemcpyAsync(c->col0.p, h_col0.data(), col0_tot * 4, hipMemcpyHostToDevice, st));
        STRQ_HIP(c, hipMemsetAsync(c->queue.p, 0, 256, st));

        // ---- score tables
        std::vector<LutJob> jobs(nb);
        LutJob* d_jobs = c->lutinfo.as<LutJob>();
        LutInfo* d_info = reinterpret_cast<LutInfo*>(d_jobs + nb);
        for (int i = 0; i < nb; ++i) {
            const int64_t a = a0 + i;
            jobs[i].level_val = c->level_val.as<float>() + (size_t)in.align_read[a] * 256;
            jobs[i].cls_val = c->flank_cls.as<float>() + cls_off[i];
            jobs[i].table = c->tables.as<float>() + tab_off[i];
            jobs[i].band_lo = c->band_lo.as<int32_t>() + cls_off[i];
            jobs[i].k = k[a]; jobs[i].pad_ = 0;
        }
        STRQ_HIP(c, hipMemcpyAsync(d_jobs, jobs.data(), (size_t)nb * sizeof(LutJob), hipMemcpyHostToDevice, st));
        HardEntry* d_hard = c->hard.as<HardEntry>();
        float* d_hard_vals = reinterpret_cast<float*>(d_hard + hard_cap);
        int* d_hard_count = c->queue.as<int>() + 32;
        STRQ_HIP(c, hipEventRecord(c->ev[0], st));
        if (launch_lut_build(st, d_jobs, d_info, nb, max_k, d_hard, d_hard_count, hard_cap, c->ap)) { c->err = "lut launch failed"; return STRQ_ERR_DEVICE; }
        STRQ_HIP(c, hipEventRecord(c->ev[1], st));
        STRQ_DBG("lut launched nb=%d max_k=%d", nb, max_k);
        std::vector<LutInfo> info(nb);
        int hard_count = 0;
        STRQ_HIP(c, hipMemcpyAsync(info.data(), d_info, (size_t)nb * sizeof(LutInfo), hipMemcpyDeviceToHost, st));
        STRQ_HIP(c, hipMemcpyAsync(&hard_count, d_hard_count, 4, hipMemcpyDeviceToHost, st));
        STRQ_HIP(c, hipStreamSynchronize(st));
        STRQ_DBG("lut done hard=%d tw0=%d", hard_count, info[0].tw);
        if (hard_count > hard_cap) { c->err = "too many borderline table entries"; return STRQ_ERR_DEVICE; }
        if (hard_count > 0) {
            std::vector<HardEntry> he(hard_count); std::vector<float> hv(hard_count);
            STRQ_HIP(c, hipMemcpy(he.data(), d_hard, (size_t)hard_count * sizeof(HardEntry), hipMemcpyDeviceToHost));
            for (int i = 0; i < hard_count; ++i) {
                const int64_t a = a0 + he[i].job;
                const float lv = in.level_val[(size_t)in.align_read[a] * 256 + he[i].level];
                hv[i] = host_cell_score(c->ap, lv, h_cls[cls_off[he[i].job] + he[i].k]);
            }
            STRQ_HIP(c, hipMemcpyAsync(d_hard_vals, hv.data(), (size_t)hard_count * 4, hipMemcpyHostToDevice, st));
            if (launch_lut_patch(st, d_jobs, d_hard, d_hard_vals, hard_count)) { c->err = "patch launch failed"; return STRQ_ERR_DEVICE; }
            STRQ_HIP(c, hipStreamSynchronize(st));
        }
        for (int i = 0; i < nb; ++i) if (info[i].n_hard < 0) {
            // whole table from the host libm, full width
            const int64_t a = a0 + i; const int kk = k[a];
            std::vector<float> tab((size_t)kk * 259); std::vector<int32_t> blo(kk, -1);
            const float* lv = in.level_val + (size_t)in.align_read[a] * 256;
            for (int x = 0; x < kk; ++x) {
                float* row = &tab[(size_t)x * 259];
                row[0] = c->ap.dist_min; row[257] = c->ap.dist_min; row[258] = c->ap.dist_min;
                for (int q = 0; q < 256; ++q) row[1 + q] = host_cell_score(c->ap, lv[q], h_cls[cls_off[i] + x]);
            }
            STRQ_HIP(c, hipMemcpy(jobs[i].table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            STRQ_HIP(c, hipMemcpy(jobs[i].band_lo, blo.data(), (size_t)kk * 4, hipMemcpyHostToDevice));
            info[i].tw = 258;
        }
        c->timing[4] += (float)hard_count;

        // ---- tasks, grouped by (R, table width class), longest first
        std::vector<AlignTask> tasks(nb);
        std::map<std::pair<int, int>, std::vector<int>> groups;
        for (int i = 0; i < nb; ++i) groups[{R[a0 + i], info[i].tw}].push_back(i);
        std::vector<int> order; order.reserve(nb);
        struct Launch { int R, tw, first, count; };
        std::vector<Launch> launches;
        for (auto& g : groups) {
            auto& v = g.second;
            std::stable_sort(v.begin(), v.end(), [&](int x, int y) { return n[a0 + x] > n[a0 + y]; });
            launches.push_back({g.first.first, g.first.second, (int)order.size(), (int)v.size()});
            order.insert(order.end(), v.begin(), v.end());
        }
        for (int pos = 0; pos < nb; ++pos) {
            const int i = order[pos]; const int64_t a = a0 + i;
            AlignTask& t = tasks[pos];
            t.levels = c->levels.as<uint8_t>() + in.read_off[in.align_read[a]];
            t.table = jobs[i].table; t.band_lo = jobs[i].band_lo;
            t.col0 = c->col0.as<float>() + col0_off[i];
            t.ckpt = c->ckpt.as<float>() + ck_off[i];
            t.rec = c->rec.as<int32_t>() + rec_off[i];
            t.n = n[a]; t.m = m[a]; t.k = k[a]; t.tw = info[i].tw;
        }
        AlignTask* d_tasks = c->tasks.as<AlignTask>();
        AlignResult* d_res = c->results.as<AlignResult>();
        STRQ_HIP(c, hipMemcpyAsync(d_tasks, tasks.data(), (size_t)nb * sizeof(AlignTask), hipMemcpyHostToDevice, st));
        STRQ_HIP(c, hipMemsetAsync(d_res, 0, (size_t)nb * sizeof(AlignResult), st));
        size_t scratch_words = 0;
        for (auto& L : launches) scratch_words = std::max(scratch_words, align_trace_scratch_words_per_wave(L.R));
        STRQ_HIP(c, c->scratch.reserve(scratch_words * 8 * (size_t)n_wave_slots));
        int qi = 0;
        static int* dbg = nullptr;
        if (getenv("STRQ_DEBUG") && !dbg) {
            STRQ_HIP(c, hipHostMalloc((void**)&dbg, 4096, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(dbg, 0, 4096);
            align_set_debug_buffer(dbg);
        }
        STRQ_HIP(c, hipEventRecord(c->ev[2], st));
        for (int phase = 0; phase < 2; ++phase) {
            for (auto& L : launches) {
                int max_kk = 0;
                for (int x = 0; x < L.count; ++x) max_kk = std::max(max_kk, tasks[L.first + x].k);
                const int lds_floats = max_kk * (L.tw + 1);
                int wpb = (160 * 1024) / (lds_floats * 4);
                if (wpb < 1) { c->err = "score table does not fit LDS"; return STRQ_ERR_UNSUPPORTED; }
                if (wpb > 4) wpb = 4;
                const int blocks = c->n_cu;     // one block per CU, `wpb` persistent waves each
                if (launch_align(st, L.R, S, d_tasks + L.first, d_res + L.first, L.count, c->queue.as<int>() + qi,
                                 c->ap, lds_floats, wpb, blocks, c->scratch.as<uint64_t>(), phase)) {
                    c->err = "align launch failed"; return STRQ_ERR_DEVICE;
                }
                ++qi;
            }
            STRQ_HIP(c, hipEventRecord(c->ev[3 + phase], st));
            if (getenv("STRQ_DEBUG")) {
                for (int w = 0; w < 50 && hipStreamQuery(st) == hipErrorNotReady; ++w) {
                    struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr);
                    if (w % 10 == 9) STRQ_DBG("waiting phase %d: task %d n %d m %d t0 %d end %d", phase, dbg[0], dbg[1], dbg[2], dbg[3], dbg[4]);
                }
                if (hipStreamQuery(st) == hipErrorNotReady) { STRQ_DBG("HUNG in phase %d", phase); _exit(3); }
                STRQ_DBG("phase %d done", phase);
            }
        }
        // ---- results
        std::vector<AlignResult> res(nb);
        std::vector<int32_t> h_rec(rec_tot);
        STRQ_HIP(c, hipMemcpyAsync(res.data(), d_res, (size_t)nb * sizeof(AlignResult), hipMemcpyDeviceToHost, st));
        if (out.rec) STRQ_HIP(c, hipMemcpyAsync(h_rec.data(), c->rec.p, rec_tot * 4, hipMemcpyDeviceToHost, st));
        STRQ_HIP(c, hipStreamSynchronize(st));
        for (int pos = 0; pos < nb; ++pos) {
            const int i = order[pos]; const int64_t a = a0 + i;
            if (out.score) out.score[a] = res[pos].best;
            if (out.j_end) out.j_end[a] = res[pos].j_end;
            if (out.j0) out.j0[a] = res[pos].j0;
            if (out.rec) std::memcpy(out.rec + in.flank_off[a], &h_rec[rec_off[i]], (size_t)m[a] * 4);
        }
        float ms;
        STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); t_lut += ms;
        STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); t_fwd += ms;
        STRQ_HIP(c, hipEventElapsedTime(&ms, c->ev[3], c->ev[4])); t_tr += ms;
        a0 = a1;
    }
    c->timing[0] = t_lut; c->timing[1] = t_fwd; c->timing[2] = t_tr; c->timing[3] = t_lut + t_fwd + t_tr;
    return STRQ_OK;
}

}  // namespace strq

extern "C" {

int strq_abi_version(void) { return 1; }

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
    for (DevBuf* b : {&c->levels, &c->level_val, &c->flank_cls, &c->tables, &c->band_lo, &c->col0, &c->ckpt,
                      &c->rec, &c->tasks, &c->results, &c->queue, &c->scratch, &c->lutinfo, &c->hard, &c->misc,
                      &c->vit_x, &c->vit_tasks, &c->vit_bp, &c->vit_path})
        b->release();
    for (HostModel* m : c->models) if (m) { m->blob.release(); delete m; }
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
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

int strq_last_timing(const strq_ctx* c, float ms[8])
{
    if (!c || !ms) return STRQ_ERR_ARG;
    std::memcpy(ms, c->timing, sizeof(c->timing));
    return STRQ_OK;
}

int strq_align_batch(strq_ctx* c, int64_t n_align, int64_t n_reads, const uint8_t* levels,
                     const int64_t* read_off, const float* level_val, const int32_t* align_read,
                     const float* flank, const int64_t* flank_off, int32_t samples,
                     float* score, int64_t* j_end, int64_t* j0, int32_t* rec)
{
    if (!c) return STRQ_ERR_ARG;
    if (n_align < 0 || n_reads < 0 || (n_align > 0 && (!levels || !read_off || !level_val || !align_read || !flank || !flank_off)) || samples < 1) {
        c->err = "bad argument"; return STRQ_ERR_ARG;
    }
    STRQ_HIP(c, hipSetDevice(c->device));
    BatchIn in{n_align, n_reads, levels, read_off, level_val, align_read, flank, flank_off, samples};
    BatchOut out{score, j_end, j0, rec};
    return run_align_batch(c, in, out);
}

int strq_align_overlap(strq_ctx* c, const float* a, int64_t n, const float* b, int64_t m, float* score,
                       uint64_t* a_idx, uint64_t* b_idx, int32_t* rec_out, int64_t* j_end_out, int64_t* j0_out)
{
    if (!c) return STRQ_ERR_ARG;
    if (!a || !b || n < 0 || m < 1 || !score) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    // dictionary-encode `a`: levels are the ranks of its distinct values
    std::vector<float> vals(a, a + n);
    std::sort(vals.begin(), vals.end());
    vals.erase(std::unique(vals.begin(), vals.end(), [](float x, float y) { return std::memcmp(&x, &y, 4) == 0; }), vals.end());
    if (vals.size() > 256) { c->err = "more than 256 distinct values in `a`"; return STRQ_ERR_UNSUPPORTED; }
    for (float v : vals) if (v != v) { c->err = "NaN in `a`"; return STRQ_ERR_UNSUPPORTED; }
    std::vector<uint8_t> lv((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        // -0.0 and +0.0 compare equal under '<' but differ bitwise; resolve by exact bit match
        auto it = std::lower_bound(vals.begin(), vals.end(), a[i]);
        while (std::memcmp(&*it, &a[i], 4) != 0) ++it;
        lv[(size_t)i] = (uint8_t)(it - vals.begin());
    }
    std::vector<float> lval(256, vals.empty() ? 0.0f : vals.back());
    std::copy(vals.begin(), vals.end(), lval.begin());
    const int64_t roff[2] = {0, n}, foff[2] = {0, m};
    const int32_t ar = 0;
    std::vector<int32_t> rec((size_t)m);
    int64_t je = 0, j0 = 0;
    STRQ_HIP(c, hipSetDevice(c->device));
    BatchIn in{1, 1, lv.data(), roff, lval.data(), &ar, b, foff, 6};
    BatchOut out{score, &je, &j0, rec.data()};
    const int rc = run_align_batch(c, in, out);
    if (rc) return rc;
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
