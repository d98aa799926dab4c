// C ABI: baked HMM upload and Viterbi decode (single and batched).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../include/strique_hip.h"
#include "strq_ctx.h"
#include "viterbi_kernels.h"

using namespace strq;

namespace strq {

int build_vit_model(strq_ctx* c, int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                    const int32_t* in_ptr, const int32_t* in_src, const double* in_logp,
                    const int32_t* emis_kind, const double* emis_a, const double* emis_b, const double* emis_c,
                    const int32_t* count_inc, HostModel** out)
{
    const int ne = silent_start, ns = n_states - silent_start;
    if (n_states < 2 || ne < 1 || ns < 2 || start < ne || end < ne || start >= n_states || end >= n_states) {
        c->err = "bad model dimensions"; return STRQ_ERR_ARG;
    }
    const int epl = (ne + 63) / 64, spl = (ns + 63) / 64;
    if (epl > 8 || spl > 4 || n_states >= 65535) { c->err = "model too large for the compiled Viterbi kernels"; return STRQ_ERR_UNSUPPORTED; }
    for (int l = 0; l < n_states; ++l)
        for (int e = in_ptr[l]; e < in_ptr[l + 1]; ++e) {
            const int k = in_src[e];
            if (k < 0 || k >= n_states) { c->err = "edge source out of range"; return STRQ_ERR_ARG; }
            if (l >= ne && k >= l) { c->err = "silent states are not in topological order"; return STRQ_ERR_ARG; }
            if (e > in_ptr[l] && in_src[e - 1] >= k) { c->err = "in-edges must be sorted by source"; return STRQ_ERR_ARG; }
        }
    HostModel* hm = new HostModel();
    VitModel& m = hm->h;
    std::memset(&m, 0, sizeof(m));
    m.n_states = n_states; m.n_emit = ne; m.n_silent = ns; m.start = start; m.end = end; m.epl = epl; m.spl = spl;
    int rows = 0;
    for (int s = 0; s < epl; ++s) {
        int deg = 0;
        for (int lane = 0; lane < 64; ++lane) { const int e = s * 64 + lane; if (e < ne) deg = std::max(deg, in_ptr[e + 1] - in_ptr[e]); }
        m.e_deg[s] = deg; m.e_base[s] = rows; rows += deg;
    }
    for (int s = 0; s < spl; ++s) {
        int deg = 0;
        for (int lane = 0; lane < 64; ++lane) { const int q = s * 64 + lane; if (q < ns) deg = std::max(deg, in_ptr[ne + q + 1] - in_ptr[ne + q]); }
        m.s_deg[s] = deg; m.s_base[s] = rows; rows += deg;
    }
    m.n_edge_rows = rows;
    std::vector<int32_t> src((size_t)rows * 64, n_states);   // padding -> the -inf cell
    std::vector<double> lp((size_t)rows * 64, 0.0);
    auto fill = [&](int state, int base, int lane) {
        for (int e = in_ptr[state], j = 0; e < in_ptr[state + 1]; ++e, ++j) {
            src[(size_t)(base + j) * 64 + lane] = in_src[e];
            lp[(size_t)(base + j) * 64 + lane] = in_logp[e];
        }
    };
    for (int s = 0; s < epl; ++s) for (int lane = 0; lane < 64; ++lane) { const int e = s * 64 + lane; if (e < ne) fill(e, m.e_base[s], lane); }
    for (int s = 0; s < spl; ++s) for (int lane = 0; lane < 64; ++lane) { const int q = s * 64 + lane; if (q < ns) fill(ne + q, m.s_base[s], lane); }
    std::vector<int32_t> kind((size_t)epl * 64, 0); std::vector<double> a((size_t)epl * 64, 0.0), b(a), cc(a);
    for (int e = 0; e < ne; ++e) { kind[e] = emis_kind[e]; a[e] = emis_a[e]; b[e] = emis_b[e]; cc[e] = emis_c[e]; }
    std::vector<int32_t> inc((size_t)n_states + 1, 0);
    if (count_inc) std::copy(count_inc, count_inc + n_states, inc.begin());
    // one device blob
    const size_t o_src = 0, o_lp = o_src + src.size() * 4, o_kind = o_lp + lp.size() * 8, o_a = o_kind + kind.size() * 4 + 4,
                 o_b = o_a + a.size() * 8, o_c = o_b + b.size() * 8, o_inc = o_c + cc.size() * 8, o_m = (o_inc + inc.size() * 4 + 15) & ~(size_t)15,
                 total = o_m + sizeof(VitModel);
    const size_t o_a8 = (o_a + 7) & ~(size_t)7;
    const size_t o_b8 = o_a8 + a.size() * 8, o_c8 = o_b8 + b.size() * 8, o_inc8 = o_c8 + cc.size() * 8;
    const size_t o_m8 = (o_inc8 + inc.size() * 4 + 15) & ~(size_t)15;
    (void)o_b; (void)o_c; (void)o_inc; (void)o_m; (void)total;
    const size_t total8 = o_m8 + sizeof(VitModel);
    if (hm->blob.reserve(total8) != hipSuccess) { delete hm; c->err = "out of device memory"; return STRQ_ERR_NOMEM; }
    char* d = hm->blob.as<char>();
    m.edge_src = reinterpret_cast<const int32_t*>(d + o_src);
    m.edge_logp = reinterpret_cast<const double*>(d + o_lp);
    m.emis_kind = reinterpret_cast<const int32_t*>(d + o_kind);
    m.emis_a = reinterpret_cast<const double*>(d + o_a8);
    m.emis_b = reinterpret_cast<const double*>(d + o_b8);
    m.emis_c = reinterpret_cast<const double*>(d + o_c8);
    m.count_inc = reinterpret_cast<const int32_t*>(d + o_inc8);
    hm->dev = reinterpret_cast<const VitModel*>(d + o_m8);
    std::vector<char> host(total8, 0);
    std::memcpy(&host[o_src], src.data(), src.size() * 4);
    std::memcpy(&host[o_lp], lp.data(), lp.size() * 8);
    std::memcpy(&host[o_kind], kind.data(), kind.size() * 4);
    std::memcpy(&host[o_a8], a.data(), a.size() * 8);
    std::memcpy(&host[o_b8], b.data(), b.size() * 8);
    std::memcpy(&host[o_c8], cc.data(), cc.size() * 8);
    std::memcpy(&host[o_inc8], inc.data(), inc.size() * 4);
    std::memcpy(&host[o_m8], &m, sizeof(VitModel));
    if (hipMemcpy(d, host.data(), total8, hipMemcpyHostToDevice) != hipSuccess) { delete hm; c->err = "model upload failed"; return STRQ_ERR_DEVICE; }
    *out = hm;
    return STRQ_OK;
}

}  // namespace strq

extern "C" {

int strq_model_create(strq_ctx* c, int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                      const int32_t* in_ptr, const int32_t* in_src, const double* in_logp,
                      const int32_t* emis_kind, const double* emis_a, const double* emis_b, const double* emis_c,
                      const int32_t* count_inc, int32_t* model_id)
{
    if (!c) return STRQ_ERR_ARG;
    if (!in_ptr || !in_src || !in_logp || !emis_kind || !emis_a || !emis_b || !emis_c || !model_id) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    STRQ_HIP(c, hipSetDevice(c->device));
    HostModel* hm = nullptr;
    const int rc = build_vit_model(c, n_states, silent_start, start, end, in_ptr, in_src, in_logp, emis_kind, emis_a, emis_b, emis_c, count_inc, &hm);
    if (rc) return rc;
    c->models.push_back(hm);
    *model_id = (int32_t)c->models.size() - 1;
    return STRQ_OK;
}

int strq_viterbi_batch(strq_ctx* c, int32_t model_id, int64_t n_seq, const double* x, const int64_t* x_off,
                       double* logp, int64_t* counted, int32_t* status, int32_t* paths)
{
    if (!c) return STRQ_ERR_ARG;
    if (model_id < 0 || model_id >= (int32_t)c->models.size() || !c->models[model_id] || n_seq < 0 || (n_seq > 0 && (!x || !x_off))) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    if (n_seq == 0) return STRQ_OK;
    STRQ_HIP(c, hipSetDevice(c->device));
    HostModel* hm = c->models[model_id];
    hipStream_t st = c->stream;
    const int64_t tot = x_off[n_seq];
    const int n = hm->h.n_states;
    STRQ_HIP(c, c->vit_x.reserve((size_t)tot * 8 + 64));
    STRQ_HIP(c, hipMemcpyAsync(c->vit_x.p, x, (size_t)tot * 8, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, c->vit_tasks.reserve((size_t)n_seq * (sizeof(VitTask) + sizeof(VitResult) + 8)));
    VitTask* d_tasks = c->vit_tasks.as<VitTask>();
    VitResult* d_res = reinterpret_cast<VitResult*>(d_tasks + n_seq);
    int32_t** d_paths = reinterpret_cast<int32_t**>(d_res + n_seq);
    size_t bp_cells = 0;
    if (paths) { for (int64_t i = 0; i < n_seq; ++i) bp_cells += (size_t)(x_off[i + 1] - x_off[i] + 1) * n; }
    if (paths) { STRQ_HIP(c, c->vit_bp.reserve(bp_cells * 2 + 64)); STRQ_HIP(c, c->vit_path.reserve((size_t)tot * 4 + 64)); }
    // longest first
    std::vector<int64_t> order(n_seq);
    for (int64_t i = 0; i < n_seq; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return x_off[a + 1] - x_off[a] > x_off[b + 1] - x_off[b]; });
    std::vector<VitTask> tasks(n_seq); std::vector<int32_t*> hp(n_seq, nullptr);
    size_t bp_off = 0;
    for (int64_t pos = 0; pos < n_seq; ++pos) {
        const int64_t i = order[pos];
        VitTask& t = tasks[pos];
        std::memset(&t, 0, sizeof(t));
        t.sig = c->vit_x.as<double>() + x_off[i]; t.T = x_off[i + 1] - x_off[i]; t.src_kind = VIT_SRC_F64;
        if (paths) { t.bp = c->vit_bp.as<uint16_t>() + bp_off; bp_off += (size_t)(t.T + 1) * n; hp[pos] = c->vit_path.as<int32_t>() + x_off[i]; }
    }
    STRQ_HIP(c, hipMemcpyAsync(d_tasks, tasks.data(), (size_t)n_seq * sizeof(VitTask), hipMemcpyHostToDevice, st));
    if (paths) STRQ_HIP(c, hipMemcpyAsync(d_paths, hp.data(), (size_t)n_seq * 8, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, c->queue.reserve(256));
    STRQ_HIP(c, hipMemsetAsync(c->queue.p, 0, 256, st));
    STRQ_HIP(c, hipEventRecord(c->ev[0], st));
    int rc = launch_viterbi(st, hm->h, hm->dev, d_tasks, d_res, (int)n_seq, c->queue.as<int>(), c->n_cu, paths ? 1 : 0);
    if (rc) { c->err = "viterbi launch failed"; return rc == 2 || rc == 3 ? STRQ_ERR_UNSUPPORTED : STRQ_ERR_DEVICE; }
    STRQ_HIP(c, hipEventRecord(c->ev[1], st));
    if (paths) {
        if (launch_vit_traceback(st, hm->dev, d_tasks, d_res, d_paths, (int)n_seq)) { c->err = "traceback launch failed"; return STRQ_ERR_DEVICE; }
    }
    STRQ_HIP(c, hipEventRecord(c->ev[2], st));
    std::vector<VitResult> res(n_seq);
    STRQ_HIP(c, hipMemcpyAsync(res.data(), d_res, (size_t)n_seq * sizeof(VitResult), hipMemcpyDeviceToHost, st));
    if (paths) STRQ_HIP(c, hipMemcpyAsync(paths, c->vit_path.p, (size_t)tot * 4, hipMemcpyDeviceToHost, st));
    STRQ_HIP(c, hipStreamSynchronize(st));
    for (int64_t pos = 0; pos < n_seq; ++pos) {
        const int64_t i = order[pos];
        if (logp) logp[i] = res[pos].logp;
        if (counted) counted[i] = res[pos].counted;
        if (status) status[i] = res[pos].status;
    }
    std::fill(c->timing, c->timing + 8, 0.0f);
    STRQ_HIP(c, hipEventElapsedTime(&c->timing[0], c->ev[0], c->ev[1]));
    STRQ_HIP(c, hipEventElapsedTime(&c->timing[1], c->ev[1], c->ev[2]));
    c->timing[3] = c->timing[0] + c->timing[1];
    return STRQ_OK;
}

int strq_viterbi(strq_ctx* c, int32_t model_id, const double* x, int64_t T, double* logp, int64_t* counted,
                 int32_t* status, int32_t* path)
{
    const int64_t off[2] = {0, T};
    return strq_viterbi_batch(c, model_id, 1, x, off, logp, counted, status, path);
}

}  // extern "C"
