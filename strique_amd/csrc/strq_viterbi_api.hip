// C ABI: baked HMM upload and Viterbi decode (single and batched).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <string>
#include "../../include/strique_hip.h"
#include "strq_ctx.h"
#include "viterbi_kernels.h"

using namespace strq;

namespace strq {

// What every builder relies on: in_ptr monotone from 0, sources in range, silent states in topological order, the in-edges
// of a state sorted by ascending source (the order ties are broken in), emitting states Normal (1) or Uniform (2).
// Returns the reason, or null.
static const char* validate_vit_model(int32_t n_states, int32_t silent_start, const int32_t* in_ptr, const int32_t* in_src, const int32_t* emis_kind)
{
    if (in_ptr[0] != 0) return "in_ptr must start at 0";
    for (int l = 0; l < n_states; ++l) if (in_ptr[l + 1] < in_ptr[l]) return "in_ptr must not decrease";
    for (int l = 0; l < n_states; ++l)
        for (int e = in_ptr[l]; e < in_ptr[l + 1]; ++e) {
            const int k = in_src[e];
            if (k < 0 || k >= n_states) return "edge source out of range";
            if (l >= silent_start && k >= l) return "silent states are not in topological order";
            if (e > in_ptr[l] && in_src[e - 1] >= k) return "in-edges must be sorted by source";
        }
    for (int e = 0; e < silent_start; ++e) if (emis_kind[e] != 1 && emis_kind[e] != 2) return "emission kind must be 1 (Normal) or 2 (Uniform)";
    return nullptr;
}

int build_vit_model(strq_ctx* c, int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                    const int32_t* in_ptr, const int32_t* in_src, const double* in_logp,
                    const int32_t* emis_kind, const double* emis_a, const double* emis_b, const double* emis_c,
                    const int32_t* count_inc, const int32_t* state_tag,
                    const int32_t* hint_slot, const int32_t* hint_lane, HostModel** out)
{
    const int ne = silent_start, ns = n_states - silent_start;
    if (n_states < 2 || ne < 1 || ns < 2 || start < ne || end < ne || start >= n_states || end >= n_states) {
        c->err = "bad model dimensions"; return STRQ_ERR_ARG;
    }
    // the edge lists are validated before any size decision: a model too large for the lane layouts goes on to the general
    // kernel's builder, which indexes host arrays with these sources
    if (const char* bad = validate_vit_model(n_states, silent_start, in_ptr, in_src, emis_kind)) { c->err = bad; return STRQ_ERR_ARG; }
    const int epl = (ne + 63) / 64, spl = (ns + 63) / 64;
    if (epl > 8 || spl > 4 || n_states >= 65535) { c->err = "model too large for the compiled Viterbi kernels"; return STRQ_ERR_UNSUPPORTED; }
    HostModel* hm = new HostModel();
    VitModel& m = hm->h;
    std::memset(&m, 0, sizeof(m));
    m.n_states = n_states; m.n_emit = ne; m.n_silent = ns; m.start = start; m.end = end; m.epl = epl; m.spl = spl;
    // emitting states: dealt to slots by descending in-degree (slot 0 takes the 64 busiest, ...)
    auto deg_of = [&](int st) { return in_ptr[st + 1] - in_ptr[st]; };
    std::vector<int> eord(ne);
    for (int i = 0; i < ne; ++i) eord[i] = i;
    std::stable_sort(eord.begin(), eord.end(), [&](int a, int b) { return deg_of(a) > deg_of(b); });
    std::vector<int32_t> own_e((size_t)epl * 64, -1);
    bool hinted = hint_slot && hint_lane;
    if (hinted) {      // caller-provided (slot, lane) of every emitting state: must be a valid injective map
        for (int e = 0; e < ne && hinted; ++e) {
            const int sl = hint_slot[e], ln = hint_lane[e];
            if (sl < 0 || sl >= epl || ln < 0 || ln >= 64 || own_e[sl * 64 + ln] >= 0) hinted = false; else own_e[sl * 64 + ln] = e;
        }
        if (!hinted) std::fill(own_e.begin(), own_e.end(), -1);
    }
    if (!hinted) for (int i = 0; i < ne; ++i) own_e[i] = eord[i];
    else if (epl >= 2) {
        // the few states with more than five in-edges (the first match / insert of the repeat unit) move into free lanes of
        // slot 0, so that the second busy slot gets by with five in-edge registers (kernel shape 5)
        std::vector<int> movers;
        for (int lane = 0; lane < 64; ++lane) { const int st = own_e[64 + lane]; if (st >= 0 && deg_of(st) > 5) movers.push_back(lane); }
        int free0 = 0;
        for (int lane = 0; lane < 64; ++lane) if (own_e[lane] < 0) ++free0;
        if (!movers.empty() && (int)movers.size() <= free0 && movers.size() <= 4) {
            int lane0 = 0;
            for (int lane : movers) {
                while (own_e[lane0] >= 0) ++lane0;
                own_e[lane0] = own_e[64 + lane]; own_e[64 + lane] = -1;
            }
        }
    }
    // silent states: chains.  The chain predecessor of b is its highest-numbered silent predecessor
    // (the last in-edge in evaluation order, so a strict '>' reproduces the tie rule) if that state
    // does not already lead another chain.
    std::vector<int> chain_pred(n_states, -1), chain_succ(n_states, -1); std::vector<double> chain_lp(n_states, 0.0);
    for (int b = ne; b < n_states; ++b) {
        int a = -1; double lpv = 0;
        for (int e = in_ptr[b]; e < in_ptr[b + 1]; ++e) if (in_src[e] >= ne) { a = in_src[e]; lpv = in_logp[e]; }
        if (a >= 0 && chain_succ[a] < 0) { chain_pred[b] = a; chain_succ[a] = b; chain_lp[b] = lpv; }
    }
    std::vector<std::vector<int>> chains;
    for (int b = ne; b < n_states; ++b) if (chain_pred[b] < 0) { std::vector<int> ch; for (int x = b; x >= 0; x = chain_succ[x]) ch.push_back(x); chains.push_back(ch); }
    std::stable_sort(chains.begin(), chains.end(), [](const std::vector<int>& x, const std::vector<int>& y) { return x.size() > y.size(); });
    // Chains zig-zag through the Z silent slots of a lane (the chains back to back; cell p of that sequence: lane p / Z,
    // slot p % Z; the first cell of a chain has no chain edge), so that one lane shift of the kernel's chain sweep carries
    // a value Z positions along (viterbi_kernels.hip).  Z is the slot count of the kernel shape the model runs on: the smallest layout is tried
    // first and widened until the shape that fits it has exactly that many silent slots.
    std::vector<int32_t> own_s, chain_src_v; std::vector<double> chain_lp_v;
    auto lay_out = [&](int Z) -> bool {
        if (ns > 64 * Z) return false;
        own_s.assign((size_t)Z * 64, -1); chain_src_v.assign((size_t)Z * 64, -1); chain_lp_v.assign((size_t)Z * 64, 0.0);
        int p = 0;
        for (auto& ch : chains) {
            for (size_t q = 0; q < ch.size(); ++q, ++p) {
                const int lane = p / Z, slot = p % Z, st = ch[q];
                own_s[(size_t)slot * 64 + lane] = st;
                if (q > 0) { chain_src_v[(size_t)slot * 64 + lane] = ch[q - 1]; chain_lp_v[(size_t)slot * 64 + lane] = chain_lp[st]; }
            }
        }
        m.spl = Z;
        return true;
    };
    auto slot_degrees = [&]() {
        for (int s2 = 0; s2 < 8; ++s2) m.s_deg[s2] = 0;
        for (int s2 = 0; s2 < m.spl; ++s2) {
            int deg = 0;
            for (int lane = 0; lane < 64; ++lane) {
                const int st = own_s[(size_t)s2 * 64 + lane];
                if (st < 0) continue;
                int dg = 0;
                for (int e = in_ptr[st]; e < in_ptr[st + 1]; ++e) if (chain_src_v[(size_t)s2 * 64 + lane] != in_src[e]) ++dg;
                deg = std::max(deg, dg);
            }
            m.s_deg[s2] = deg;
        }
    };
    for (int s2 = 0; s2 < epl; ++s2) {       // the shape test below needs the emitting degrees
        int deg = 0; bool flat = true;
        for (int lane = 0; lane < 64; ++lane) if (own_e[s2 * 64 + lane] >= 0) {
            deg = std::max(deg, deg_of(own_e[s2 * 64 + lane]));
            flat = flat && emis_kind[own_e[s2 * 64 + lane]] != 1;
        }
        m.e_deg[s2] = deg; m.e_flat[s2] = flat ? 1 : 0;
    }
    {
        bool placed = false;
        for (int Z = 1; Z <= 4 && !placed; Z *= 2) {
            if (!lay_out(Z)) continue;
            slot_degrees();
            m.single_stage = 1;      // refined below; the shape does not depend on it
            const int shape = vit_shape_of(m);
            if (shape >= 0 && vit_shape_silent_slots(shape) == Z) placed = true;
        }
        if (!placed) { delete hm; c->err = "model does not fit a compiled Viterbi kernel"; return STRQ_ERR_UNSUPPORTED; }
    }
    const int spl2 = m.spl;
    auto is_chain_edge = [&](int dst, int srcst) {
        for (int i = 0; i < spl2 * 64; ++i) if (own_s[i] == dst) return chain_src_v[i] == srcst;
        return false;
    };
    int rows = 0;
    for (int s = 0; s < epl; ++s) {
        int deg = 0;
        for (int lane = 0; lane < 64; ++lane) if (own_e[s * 64 + lane] >= 0) deg = std::max(deg, deg_of(own_e[s * 64 + lane]));
        m.e_deg[s] = deg; m.e_base[s] = rows; rows += deg;
    }
    std::vector<int> sdeg_state(n_states, 0);
    for (int b = ne; b < n_states; ++b) { int dg = 0; for (int e = in_ptr[b]; e < in_ptr[b + 1]; ++e) if (!is_chain_edge(b, in_src[e])) ++dg; sdeg_state[b] = dg; }
    for (int s = 0; s < spl2; ++s) {
        int deg = 0;
        for (int lane = 0; lane < 64; ++lane) if (own_s[s * 64 + lane] >= 0) deg = std::max(deg, sdeg_state[own_s[s * 64 + lane]]);
        m.s_deg[s] = deg; m.s_base[s] = rows; rows += deg;
    }
    m.single_stage = 1;
    for (int b = ne; b < n_states; ++b) for (int e = in_ptr[b]; e < in_ptr[b + 1]; ++e) if (in_src[e] >= ne && !is_chain_edge(b, in_src[e])) m.single_stage = 0;
    for (int s = 0; s < 8; ++s) if (m.e_deg[s] > 8 || m.s_deg[s] > 8) { delete hm; c->err = "a state has more than 8 in-edges"; return STRQ_ERR_UNSUPPORTED; }
    m.n_edge_rows = rows;
    // LDS cells: the owner (slot, lane) of a state is its cell; the last cell is the -inf cell
    m.n_cells = (epl + spl2) * 64 + 1;
    std::vector<int32_t> cell_of(n_states, -1), cell_state((size_t)m.n_cells, -1);
    for (int i = 0; i < epl * 64; ++i) if (own_e[i] >= 0) { cell_of[own_e[i]] = i; cell_state[i] = own_e[i]; }
    // silent cells are interleaved (cell = first silent cell + lane * spl + slot = position along the zig-zag), so that
    // the emitting states of consecutive lanes read their delete predecessors from consecutive LDS cells
    for (int i = 0; i < spl2 * 64; ++i) if (own_s[i] >= 0) {
        const int cell = epl * 64 + (i % 64) * spl2 + i / 64;
        cell_of[own_s[i]] = cell; cell_state[cell] = own_s[i];
    }
    m.start_cell = cell_of[start]; m.end_cell = cell_of[end];
    std::vector<int32_t> src((size_t)std::max(rows, 1) * 64, m.n_cells - 1);   // padding -> the -inf cell
    std::vector<double> lp((size_t)std::max(rows, 1) * 64, 0.0);
    auto fill = [&](int state, int base, int lane) {
        for (int e = in_ptr[state], j = 0; e < in_ptr[state + 1]; ++e) {
            if (state >= ne && is_chain_edge(state, in_src[e])) continue;
            src[(size_t)(base + j) * 64 + lane] = cell_of[in_src[e]];
            lp[(size_t)(base + j) * 64 + lane] = in_logp[e];
            ++j;
        }
    };
    for (int s = 0; s < epl; ++s) for (int lane = 0; lane < 64; ++lane) if (own_e[s * 64 + lane] >= 0) fill(own_e[s * 64 + lane], m.e_base[s], lane);
    for (int s = 0; s < spl2; ++s) for (int lane = 0; lane < 64; ++lane) if (own_s[s * 64 + lane] >= 0) fill(own_s[s * 64 + lane], m.s_base[s], lane);
    std::vector<int32_t> kind((size_t)epl * 64, 0); std::vector<double> a((size_t)epl * 64, 0.0), b(a), cc(a);
    for (int i = 0; i < epl * 64; ++i) if (own_e[i] >= 0) { const int e = own_e[i]; kind[i] = emis_kind[e]; a[i] = emis_a[e]; b[i] = emis_b[e]; cc[i] = emis_c[e]; }
    m.uni_lo_max = -INFINITY; m.uni_hi_min = INFINITY;
    for (int e = 0; e < ne; ++e) if (emis_kind[e] == 2) { m.uni_lo_max = std::max(m.uni_lo_max, emis_a[e]); m.uni_hi_min = std::min(m.uni_hi_min, emis_b[e]); }
    std::vector<int32_t> inc((size_t)n_states + 1, 0);
    if (count_inc) std::copy(count_inc, count_inc + n_states, inc.begin());
    std::vector<int32_t> tagv((size_t)n_states + 1, 0);
    if (state_tag) std::copy(state_tag, state_tag + n_states, tagv.begin());
    // one device blob
    struct Part { const void* p; size_t bytes; size_t off; };
    std::vector<Part> parts = {
        {lp.data(), lp.size() * 8, 0}, {a.data(), a.size() * 8, 0}, {b.data(), b.size() * 8, 0}, {cc.data(), cc.size() * 8, 0},
        {src.data(), src.size() * 4, 0}, {kind.data(), kind.size() * 4, 0}, {inc.data(), inc.size() * 4, 0},
        {own_e.data(), own_e.size() * 4, 0}, {own_s.data(), own_s.size() * 4, 0},
        {chain_src_v.data(), chain_src_v.size() * 4, 0}, {chain_lp_v.data(), chain_lp_v.size() * 8, 0},
        {tagv.data(), tagv.size() * 4, 0}, {cell_state.data(), cell_state.size() * 4, 0}};
    size_t total = 0;
    for (auto& pt : parts) { pt.off = total; total += (pt.bytes + 15) & ~(size_t)15; }
    const size_t o_m = total; total += sizeof(VitModel);
    if (hm->blob.reserve(total) != hipSuccess) { delete hm; c->err = "out of device memory"; return STRQ_ERR_NOMEM; }
    char* d = hm->blob.as<char>();
    m.edge_logp = reinterpret_cast<const double*>(d + parts[0].off);
    m.emis_a = reinterpret_cast<const double*>(d + parts[1].off);
    m.emis_b = reinterpret_cast<const double*>(d + parts[2].off);
    m.emis_c = reinterpret_cast<const double*>(d + parts[3].off);
    m.edge_src = reinterpret_cast<const int32_t*>(d + parts[4].off);
    m.emis_kind = reinterpret_cast<const int32_t*>(d + parts[5].off);
    m.count_inc = reinterpret_cast<const int32_t*>(d + parts[6].off);
    m.own_e = reinterpret_cast<const int32_t*>(d + parts[7].off);
    m.own_s = reinterpret_cast<const int32_t*>(d + parts[8].off);
    m.chain_src = reinterpret_cast<const int32_t*>(d + parts[9].off);
    m.chain_logp = reinterpret_cast<const double*>(d + parts[10].off);
    m.state_tag = reinterpret_cast<const int32_t*>(d + parts[11].off);
    m.rec_state = -1;
    for (int e = in_ptr[end]; e < in_ptr[end + 1]; ++e) if (in_src[e] < ne && state_tag && state_tag[in_src[e]] == 2) m.rec_state = in_src[e];
    m.silent_counted = 0;
    for (int s2 = ne; s2 < n_states; ++s2) if (count_inc && count_inc[s2] != 0) m.silent_counted = 1;
    m.cell_state = reinterpret_cast<const int32_t*>(d + parts[12].off);
    hm->dev = reinterpret_cast<const VitModel*>(d + o_m);
    std::vector<char> host(total, 0);
    for (auto& pt : parts) std::memcpy(&host[pt.off], pt.p, pt.bytes);
    std::memcpy(&host[o_m], &m, sizeof(VitModel));
    if (hipMemcpy(d, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) { delete hm; c->err = "model upload failed"; return STRQ_ERR_DEVICE; }
    *out = hm;
    return STRQ_OK;
}

// The baked arrays as they are, for viterbi_csr_kernel (models the lane layouts do not cover).
int build_vit_model_csr(strq_ctx* c, int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                        const int32_t* in_ptr, const int32_t* in_src, const double* in_logp,
                        const int32_t* emis_kind, const double* emis_a, const double* emis_b, const double* emis_c,
                        const int32_t* count_inc, const int32_t* state_tag, HostModel** out)
{
    const int ne = silent_start, ns = n_states - silent_start;
    if (n_states > VIT_CSR_MAX_STATES) { c->err = "model too large: more than 4096 states"; return STRQ_ERR_UNSUPPORTED; }
    HostModel* hm = new HostModel();
    VitModel& m = hm->h;
    std::memset(&m, 0, sizeof(m));
    m.n_states = n_states; m.n_emit = ne; m.n_silent = ns; m.start = start; m.end = end; m.epl = 0; m.spl = 0;
    m.csr = 1; m.n_cells = n_states + 1; m.start_cell = start; m.end_cell = end; m.rec_state = -1; m.single_stage = 0;
    // level of a silent state: length of its longest chain of silent predecessors (they are in topological order)
    std::vector<int> level(n_states, 0); int nlev = 0;
    for (int l = ne; l < n_states; ++l) {
        int lv = 0;
        for (int e = in_ptr[l]; e < in_ptr[l + 1]; ++e) if (in_src[e] >= ne) lv = std::max(lv, level[in_src[e]] + 1);
        level[l] = lv; nlev = std::max(nlev, lv + 1);
    }
    std::vector<int32_t> level_ptr((size_t)nlev + 1, 0), level_state((size_t)std::max(ns, 1), 0);
    for (int l = ne; l < n_states; ++l) ++level_ptr[(size_t)level[l] + 1];
    for (int v = 0; v < nlev; ++v) level_ptr[(size_t)v + 1] += level_ptr[v];
    { std::vector<int32_t> pos(level_ptr.begin(), level_ptr.end() - 1);
      for (int l = ne; l < n_states; ++l) level_state[(size_t)pos[level[l]]++] = l; }
    m.n_levels = nlev;
    m.uni_lo_max = -INFINITY; m.uni_hi_min = INFINITY;
    for (int e = 0; e < ne; ++e) if (emis_kind[e] == 2) { m.uni_lo_max = std::max(m.uni_lo_max, emis_a[e]); m.uni_hi_min = std::min(m.uni_hi_min, emis_b[e]); }
    std::vector<int32_t> inc((size_t)n_states + 1, 0), tagv((size_t)n_states + 1, 0);
    if (count_inc) std::copy(count_inc, count_inc + n_states, inc.begin());
    if (state_tag) std::copy(state_tag, state_tag + n_states, tagv.begin());
    const int n_edges = in_ptr[n_states];
    struct Part { const void* p; size_t bytes; size_t off; };
    std::vector<Part> parts = {
        {in_logp, (size_t)n_edges * 8, 0}, {emis_a, (size_t)ne * 8, 0}, {emis_b, (size_t)ne * 8, 0}, {emis_c, (size_t)ne * 8, 0},
        {in_ptr, ((size_t)n_states + 1) * 4, 0}, {in_src, (size_t)n_edges * 4, 0}, {emis_kind, (size_t)ne * 4, 0},
        {inc.data(), inc.size() * 4, 0}, {tagv.data(), tagv.size() * 4, 0},
        {level_ptr.data(), level_ptr.size() * 4, 0}, {level_state.data(), level_state.size() * 4, 0}};
    size_t total = 0;
    for (auto& pt : parts) { pt.off = total; total += (pt.bytes + 15) & ~(size_t)15; }
    const size_t o_m = total; total += sizeof(VitModel);
    if (hm->blob.reserve(total) != hipSuccess) { delete hm; c->err = "out of device memory"; return STRQ_ERR_NOMEM; }
    char* d = hm->blob.as<char>();
    m.csr_in_logp = reinterpret_cast<const double*>(d + parts[0].off);
    m.csr_a = reinterpret_cast<const double*>(d + parts[1].off);
    m.csr_b = reinterpret_cast<const double*>(d + parts[2].off);
    m.csr_c = reinterpret_cast<const double*>(d + parts[3].off);
    m.csr_in_ptr = reinterpret_cast<const int32_t*>(d + parts[4].off);
    m.csr_in_src = reinterpret_cast<const int32_t*>(d + parts[5].off);
    m.csr_kind = reinterpret_cast<const int32_t*>(d + parts[6].off);
    m.count_inc = reinterpret_cast<const int32_t*>(d + parts[7].off);
    m.state_tag = reinterpret_cast<const int32_t*>(d + parts[8].off);
    m.csr_level_ptr = reinterpret_cast<const int32_t*>(d + parts[9].off);
    m.csr_level_state = reinterpret_cast<const int32_t*>(d + parts[10].off);
    hm->dev = reinterpret_cast<const VitModel*>(d + o_m);
    std::vector<char> host(total, 0);
    for (auto& pt : parts) if (pt.bytes) std::memcpy(&host[pt.off], pt.p, pt.bytes);
    std::memcpy(&host[o_m], &m, sizeof(VitModel));
    if (hipMemcpy(d, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) { delete hm; c->err = "model upload failed"; return STRQ_ERR_DEVICE; }
    *out = hm;
    return STRQ_OK;
}

// Register-resident image of a profile chain (VitG2, viterbi_kernels.h) from the position of every emitting state along the
// chain: kind 0 = match-type, 1 = insert-type, pos >= 0.  Silent states take the position after their emitting / silent
// predecessors (a silent state without predecessors -- start -- the position before its match-type successor).  Every
// in-edge must fall into a column of the layout, in ascending column order (= ascending source state, the order ties are
// broken in); the states that need the even-position columns (a broadcast source, an insert-type state fed by the previous
// delete state) decide the parity of the whole chain.  Returns STRQ_ERR_UNSUPPORTED when the model is not such a chain:
// it then keeps running on its lane layout.
struct G2Host {
    std::vector<double> lp, em; std::vector<int32_t> knd, own, inc, tag;
    std::vector<uint64_t> mark_add;      // empty: the tagged states are no contiguous stretch of the chain
    VitG2 G;
    int odd = 0;
};

// host part: the tables of the image (no device involved), or the reason there is none
static bool g2_layout(const HostModel* hm, const int32_t* kind_hint, const int32_t* pos_hint, G2Host& out, std::string& why_out)
{
    const int n = hm->n_states, ne = hm->silent_start;
    const std::vector<int32_t>& in_ptr = hm->in_ptr; const std::vector<int32_t>& in_src = hm->in_src; const std::vector<double>& in_logp = hm->in_logp;
    auto unsupported = [&](const char* why) { why_out = why; return false; };
    for (int b = ne; b < n; ++b) if (!hm->count_inc.empty() && hm->count_inc[b] != 0) return unsupported("a silent state is counted");
    for (int e = 0; e < ne; ++e) {
        if ((kind_hint[e] != 0 && kind_hint[e] != 1) || pos_hint[e] < 0 || pos_hint[e] > 125) return unsupported("an emitting state has no position");
        if (kind_hint[e] == 1 && hm->emis_kind[e] == 1) return unsupported("an insert-type state has a Normal emission");
    }
    const double NEG = -INFINITY;
    std::string why = "?", why_all;
    struct Edge { int src; double lp; };
    for (int off = 0; off < 2; ++off) {
        if (off == 1) why_all = why + "; ";
        std::vector<int> g(n, -1), kind(n, 2);
        for (int e = 0; e < ne; ++e) { g[e] = pos_hint[e] + 1 + off; kind[e] = kind_hint[e]; }
        bool ok = true;
        for (int b = ne; b < n && ok; ++b) {
            int gp = -1;
            for (int e = in_ptr[b]; e < in_ptr[b + 1]; ++e) {
                const int k = in_src[e];
                if (g[k] < 0) { ok = false; why = "a silent state precedes its predecessor"; break; }
                if (gp < 0) gp = g[k] + 1; else if (gp != g[k] + 1) { ok = false; why = "a silent state has predecessors at two positions"; break; }
            }
            if (ok && gp < 0) {      // no predecessors: the position before its match-type successor (else that of its insert-type successor)
                int gm = -1, gi = -1;
                for (int l = 0; l < ne; ++l) for (int e = in_ptr[l]; e < in_ptr[l + 1]; ++e) if (in_src[e] == b) { if (kind[l] == 0) gm = g[l] - 1; else gi = g[l]; }
                gp = gm >= 0 ? gm : gi;
                if (gp < 0) {      // only silent successors: right before the first of them is decided below -- take position 0
                    gp = 0;
                }
            }
            g[b] = gp;
        }
        if (!ok) continue;
        // one state per (kind, position), positions below 128
        std::vector<int> at(3 * 128, -1);
        for (int l = 0; l < n && ok; ++l) {
            if (g[l] < 0 || g[l] > 127) { ok = false; why = "the chain has more than 128 positions"; break; }
            int& slot = at[kind[l] * 128 + g[l]];
            if (slot >= 0) { ok = false; why = "two states of one type at one position"; break; }
            slot = l;
        }
        if (!ok) continue;
        // ---- in-edges per state, then the virtual relay states (see viterbi_kernels.h).  The in-edges of an insert-type state T
        // at position g, in evaluation order, must read   F ++ mid ++ back   with
        //   F    = sources among {I_{g-1}, M_{g-1}}                         (repeat0i, dummy1 <- e1 <- last insert / match before them)
        //   mid  = the regular columns {I_g (T itself), M_g, D_g}
        //   back = [one far emitting state], [D_{g-1}]                      (repeat0i <- s1 <- {dummy1, last prefix delete})
        // F and back go to a virtual delete-type state V at T's own position (free in a profile without delete states): F in V's
        // two gather columns, the far state in V's broadcast column, D_{g-1} as V's chain edge -- V evaluates them in exactly this
        // order -- and T keeps mid ++ [V with 0.0].  V stands LAST in T's tournament although F stands first in T's list: the
        // lane is flagged in `hub_mask`, and there the kernel lets V win a tie against mid when V's own winner came from F.
        std::vector<std::vector<Edge>> edges(n);
        for (int l = 0; l < n; ++l) for (int e = in_ptr[l]; e < in_ptr[l + 1]; ++e) edges[l].push_back({in_src[e], in_logp[e]});
        const int n_real = n;
        uint64_t hub_mask = 0;
        for (int l = 0; l < ne && ok; ++l) {
            if (kind[l] != 1) continue;
            const std::vector<Edge> E = edges[l];
            size_t i = 0;
            std::vector<Edge> F, mid, back;
            while (i < E.size() && E[i].src < ne && E[i].src != l && g[l] - g[E[i].src] == 1) F.push_back(E[i++]);
            while (i < E.size() && (E[i].src == l || (kind[E[i].src] == 0 && g[E[i].src] == g[l]) || (kind[E[i].src] == 2 && g[E[i].src] == g[l]))) mid.push_back(E[i++]);
            if (i < E.size() && E[i].src < ne && !(g[l] - g[E[i].src] == 1 || g[l] == g[E[i].src])) back.push_back(E[i++]);      // a far emitting state
            if (i < E.size() && kind[E[i].src] == 2 && g[l] - g[E[i].src] == 1) back.push_back(E[i++]);                          // D_{g-1}
            if (i != E.size()) { ok = false; why = "an insert-type state with in-edges no relay covers"; break; }
            if (F.empty() && back.empty()) continue;
            if (at[2 * 128 + g[l]] >= 0) { ok = false; why = "an insert-type state with irregular in-edges and a delete state at its position"; break; }
            if (!F.empty() && !mid.empty()) {
                // ties between F and mid are decided by the flag: only the even insert slot evaluates it
                if (g[l] & 1) { ok = false; why = "a relayed insert-type state at an odd position (parity " + std::to_string(off) + ")"; break; }
                hub_mask |= (uint64_t)1 << (g[l] >> 1);
            }
            const int v = (int)edges.size();
            std::vector<Edge> ve(F); ve.insert(ve.end(), back.begin(), back.end());
            mid.push_back({v, 0.0});
            edges[l] = mid;                                   // before the push_back below: it may move the vectors
            edges.push_back(ve); g.push_back(g[l]); kind.push_back(2);
            at[2 * 128 + g[l]] = v;
        }
        if (!ok) continue;
        const int n_all = (int)edges.size();
        std::vector<double> lp((size_t)G2_ROWS * 64, NEG), em((size_t)12 * 64, 0.0);
        std::vector<int32_t> knd((size_t)4 * 64, 0), own((size_t)6 * 64, -1), inc((size_t)4 * 64, 0), tag((size_t)4 * 64, 0);
        int bc_state[2] = {-1, -1};
        for (int l = 0; l < n_all && ok; ++l) {
            const int gl = g[l], par = gl & 1, lane = gl >> 1;
            int last_col = -1;
            for (const Edge& ed : edges[l]) {
                const int k = ed.src, kk = kind[k], dg = gl - g[k];
                int col = -1, row = -1;
                if (kind[l] == 0) {          // match-type
                    row = par ? G2_ROW_MO : G2_ROW_ME;
                    if (kk == 0 && dg == 2) col = 0; else if (kk == 1 && dg == 1) col = 1; else if (kk == 0 && dg == 1) col = 2;
                    else if (kk == 1 && dg == 0) col = 3; else if (k == l) col = 4;
                    else if (kk == 2 && dg == 1) col = par ? 5 : 6;
                    else if (!par && kk == 0 && k < ne && (bc_state[0] < 0 || bc_state[0] == k)) { col = 5; bc_state[0] = k; }
                } else if (kind[l] == 1) {   // insert-type: itself, the match and the delete state of its position
                    row = par ? G2_ROW_IO : G2_ROW_IE;
                    if (k == l) col = 0; else if (kk == 0 && dg == 0) col = 1; else if (kk == 2 && dg == 0) col = 2;
                } else {                     // delete-type: the gather columns, then the chain edge
                    row = par ? G2_ROW_DO : G2_ROW_DE;
                    if (kk == 1 && dg == 1) col = 0;
                    else if (kk == 0 && dg == 1) col = 1;
                    else if (kk == 2 && dg == 1) { row = G2_ROW_CHAIN + par; col = 3; }
                    else if (!par && kk == 1 && k < ne && (bc_state[1] < 0 || bc_state[1] == k)) { col = 2; bc_state[1] = k; }
                }
                if (col < 0) { ok = false; why = "an edge outside the columns of the layout (parity " + std::to_string(off) + ": state " + std::to_string(l) + " at position " + std::to_string(gl) +
                                                 " <- state " + std::to_string(k) + " of type " + std::to_string(kk) + " at position " + std::to_string(g[k]) + ")"; break; }
                if (col <= last_col) { ok = false; why = "in-edges not in column order"; break; }
                last_col = col;
                lp[(size_t)(row + (kind[l] == 2 && col == 3 ? 0 : col)) * 64 + lane] = ed.lp;
            }
            if (!ok) break;
            if (kind[l] < 2) {
                const int slot = kind[l] * 2 + par;
                own[(size_t)slot * 64 + lane] = l; knd[(size_t)slot * 64 + lane] = hm->emis_kind[l];
                em[((size_t)slot * 3 + 0) * 64 + lane] = hm->emis_a[l]; em[((size_t)slot * 3 + 1) * 64 + lane] = hm->emis_b[l]; em[((size_t)slot * 3 + 2) * 64 + lane] = hm->emis_c[l];
                inc[(size_t)slot * 64 + lane] = hm->count_inc.empty() ? 0 : hm->count_inc[l];
                tag[(size_t)slot * 64 + lane] = (!hm->state_tag.empty() && hm->state_tag[l] == 1) ? 1 : 0;
            } else own[(size_t)(4 + par) * 64 + lane] = l < n_real ? l : -2;
        }
        if (!ok) continue;
        if (in_ptr[hm->start + 1] != in_ptr[hm->start]) { why = "the start state has in-edges"; continue; }
        VitG2& G = out.G; std::memset(&G, 0, sizeof(G));
        for (int i = 0; i < 2; ++i) {
            G.bc_slot[i] = 2 * i; G.bc_lane[i] = -1;
            if (bc_state[i] >= 0) { G.bc_slot[i] = kind[bc_state[i]] * 2 + (g[bc_state[i]] & 1); G.bc_lane[i] = g[bc_state[i]] >> 1; }
        }
        G.start_slot = g[hm->start] & 1; G.start_lane = g[hm->start] >> 1; G.end_slot = g[hm->end] & 1; G.end_lane = g[hm->end] >> 1;
        G.hub_mask = hub_mask;
        // one code path per parity of the broadcast sources (a repeat profile of odd length puts them at an odd position)
        int par_bc = -1; bool bad_par = false;
        for (int i = 0; i < 2; ++i) if (bc_state[i] >= 0) { const int pb = g[bc_state[i]] & 1; if (par_bc >= 0 && par_bc != pb) bad_par = true; par_bc = pb; }
        if (bad_par) { why = "broadcast sources at positions of both parities"; continue; }
        out.odd = par_bc == 1 ? 1 : 0;
        // the kernel adds count increments only in the two slots of that parity (STRique counts the two dummy states: the broadcast sources)
        bool inc_elsewhere = false;
        for (int k = 0; k < 4; ++k) if ((k & 1) != out.odd) for (int lane = 0; lane < 64; ++lane) if (inc[(size_t)k * 64 + lane] != 0) inc_elsewhere = true;
        if (inc_elsewhere) { why = "a counted state at a position of the other parity"; continue; }
        // mark decodes: tagged emitting states must fill one stretch of positions [g_lo, g_hi] (every emitting state there is
        // tagged), so that a path -- which only moves forward along the chain, the loop inside the stretch apart -- emits
        // untagged / tagged / untagged in that order and two counters give the two boundaries (G2_MARK_*)
        out.mark_add.clear();
        if (!hm->state_tag.empty()) {
            int g_lo = 1 << 30, g_hi = -1; bool any = false, holes = false;
            for (int e = 0; e < ne; ++e) if (hm->state_tag[e] == 1) { any = true; g_lo = std::min(g_lo, g[e]); g_hi = std::max(g_hi, g[e]); }
            for (int e = 0; e < ne; ++e) if (hm->state_tag[e] != 1 && g[e] >= g_lo && g[e] <= g_hi) holes = true;
            if (any && !holes) {
                out.mark_add.assign((size_t)4 * 64, 0);
                for (int e = 0; e < ne; ++e) {
                    const size_t at_ = (size_t)(kind[e] * 2 + (g[e] & 1)) * 64 + (size_t)(g[e] >> 1);
                    uint64_t a = hm->state_tag[e] == 1 ? 1 : (g[e] > g_hi ? (uint64_t)1 << G2_MARK_BEHIND_SHIFT : 0);
                    if (!hm->count_inc.empty()) a += (uint64_t)(uint32_t)hm->count_inc[e] << G2_MARK_COUNT_SHIFT;
                    out.mark_add[at_] = a;
                }
            }
        }
        out.lp.swap(lp); out.em.swap(em); out.knd.swap(knd); out.own.swap(own); out.inc.swap(inc); out.tag.swap(tag);
        return true;
    }
    why = why_all + why;
    return unsupported(why.c_str());
}

int build_vit_g2(strq_ctx* c, HostModel* hm, const int32_t* kind_hint, const int32_t* pos_hint)
{
    if (hm->h.csr) { c->err = "no register-resident layout: the model runs on the general kernel"; return STRQ_ERR_UNSUPPORTED; }
    G2Host L; std::string why;
    if (!g2_layout(hm, kind_hint, pos_hint, L, why)) { c->err = "no register-resident layout: " + why; return STRQ_ERR_UNSUPPORTED; }
    {
        VitG2& G = L.G;
        std::vector<double>& lp = L.lp; std::vector<double>& em = L.em;
        std::vector<int32_t>& knd = L.knd; std::vector<int32_t>& own = L.own; std::vector<int32_t>& inc = L.inc; std::vector<int32_t>& tag = L.tag;
        struct Part { const void* p; size_t bytes; size_t off; };
        std::vector<Part> parts = {{lp.data(), lp.size() * 8, 0}, {em.data(), em.size() * 8, 0}, {knd.data(), knd.size() * 4, 0},
                                   {own.data(), own.size() * 4, 0}, {inc.data(), inc.size() * 4, 0}, {tag.data(), tag.size() * 4, 0},
                                   {L.mark_add.data(), L.mark_add.size() * 8, 0}};
        size_t total = 0;
        for (auto& pt : parts) { pt.off = total; total += (pt.bytes + 15) & ~(size_t)15; }
        const size_t o_g = total; total += sizeof(VitG2);
        if (hm->g2_blob.reserve(total) != hipSuccess) { c->err = "out of device memory"; return STRQ_ERR_NOMEM; }
        char* d = hm->g2_blob.as<char>();
        G.lp = reinterpret_cast<const double*>(d + parts[0].off); G.em = reinterpret_cast<const double*>(d + parts[1].off);
        G.kind = reinterpret_cast<const int32_t*>(d + parts[2].off); G.own = reinterpret_cast<const int32_t*>(d + parts[3].off);
        G.inc = reinterpret_cast<const int32_t*>(d + parts[4].off); G.tag = reinterpret_cast<const int32_t*>(d + parts[5].off);
        G.mark_add = L.mark_add.empty() ? nullptr : reinterpret_cast<const uint64_t*>(d + parts[6].off);
        std::vector<char> host(total, 0);
        for (auto& pt : parts) std::memcpy(&host[pt.off], pt.p, pt.bytes);
        std::memcpy(&host[o_g], &G, sizeof(VitG2));
        if (hipMemcpy(d, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) { c->err = "model upload failed"; return STRQ_ERR_DEVICE; }
        hm->h.g2 = reinterpret_cast<const VitG2*>(d + o_g); hm->h.g2_odd = L.odd; hm->h.g2_mark = L.mark_add.empty() ? 0 : 1;
        if (hipMemcpy(const_cast<VitModel*>(hm->dev), &hm->h, sizeof(VitModel), hipMemcpyHostToDevice) != hipSuccess) { hm->h.g2 = nullptr; c->err = "model upload failed"; return STRQ_ERR_DEVICE; }
        return STRQ_OK;
    }
}

}  // namespace strq

extern "C" {

int strq_model_create(strq_ctx* c, int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                      const int32_t* in_ptr, const int32_t* in_src, const double* in_logp,
                      const int32_t* emis_kind, const double* emis_a, const double* emis_b, const double* emis_c,
                      const int32_t* count_inc, const int32_t* state_tag,
                      const int32_t* hint_slot, const int32_t* hint_lane, int32_t* model_id)
{
    strq::CtxScope scope_(c);
    if (!c) return STRQ_ERR_ARG;
    if (!in_ptr || !in_src || !in_logp || !emis_kind || !emis_a || !emis_b || !emis_c || !model_id) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    STRQ_HIP(c, hipSetDevice(c->device));
    HostModel* hm = nullptr;
    int rc = build_vit_model(c, n_states, silent_start, start, end, in_ptr, in_src, in_logp, emis_kind, emis_a, emis_b, emis_c, count_inc, state_tag, hint_slot, hint_lane, &hm);
    if (rc == STRQ_ERR_UNSUPPORTED)      // no lane layout: the model runs on the general (slow) kernel
        rc = build_vit_model_csr(c, n_states, silent_start, start, end, in_ptr, in_src, in_logp, emis_kind, emis_a, emis_b, emis_c, count_inc, state_tag, &hm);
    if (rc) return rc;
    hm->n_states = n_states; hm->silent_start = silent_start; hm->start = start; hm->end = end;
    hm->in_ptr.assign(in_ptr, in_ptr + n_states + 1);
    hm->in_src.assign(in_src, in_src + in_ptr[n_states]); hm->in_logp.assign(in_logp, in_logp + in_ptr[n_states]);
    hm->emis_kind.assign(emis_kind, emis_kind + silent_start); hm->emis_a.assign(emis_a, emis_a + silent_start);
    hm->emis_b.assign(emis_b, emis_b + silent_start); hm->emis_c.assign(emis_c, emis_c + silent_start);
    if (count_inc) hm->count_inc.assign(count_inc, count_inc + n_states);
    if (state_tag) hm->state_tag.assign(state_tag, state_tag + n_states);
    c->models.push_back(hm);
    *model_id = (int32_t)c->models.size() - 1;
    return STRQ_OK;
}

// Host-only (no context, no device): the tables strq_model_set_positions would upload, for tests of the layout.
// out_lp[G2_ROWS * 64], out_own[6 * 64], out_meta[10] = {bc_slot0, bc_lane0, bc_slot1, bc_lane1, start_slot, start_lane, end_slot, end_lane, hub_mask lo, hub_mask hi}.
int strq_debug_g2_layout(int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                         const int32_t* in_ptr, const int32_t* in_src, const double* in_logp,
                         const int32_t* emis_kind, const int32_t* count_inc, const int32_t* kind, const int32_t* pos,
                         double* out_lp, int32_t* out_own, int32_t* out_meta, char* why, int32_t why_len)
{
    HostModel hm;
    hm.n_states = n_states; hm.silent_start = silent_start; hm.start = start; hm.end = end;
    hm.in_ptr.assign(in_ptr, in_ptr + n_states + 1); hm.in_src.assign(in_src, in_src + in_ptr[n_states]); hm.in_logp.assign(in_logp, in_logp + in_ptr[n_states]);
    hm.emis_kind.assign(emis_kind, emis_kind + silent_start); hm.emis_a.assign(silent_start, 0.0); hm.emis_b.assign(silent_start, 0.0); hm.emis_c.assign(silent_start, 0.0);
    if (count_inc) hm.count_inc.assign(count_inc, count_inc + n_states);
    for (int e = 0; e < silent_start; ++e) if ((kind[e] != 0 && kind[e] != 1) || pos[e] < 0 || pos[e] > 125) { if (why && why_len > 0) snprintf(why, why_len, "an emitting state has no position"); return STRQ_ERR_UNSUPPORTED; }
    for (int e = 0; e < silent_start; ++e) if (kind[e] == 1 && emis_kind[e] == 1) { if (why && why_len > 0) snprintf(why, why_len, "an insert-type state has a Normal emission"); return STRQ_ERR_UNSUPPORTED; }
    G2Host L; std::string w;
    if (!g2_layout(&hm, kind, pos, L, w)) { if (why && why_len > 0) snprintf(why, why_len, "%s", w.c_str()); return STRQ_ERR_UNSUPPORTED; }
    std::memcpy(out_lp, L.lp.data(), L.lp.size() * 8); std::memcpy(out_own, L.own.data(), L.own.size() * 4);
    const int32_t meta[10] = {L.G.bc_slot[0], L.G.bc_lane[0], L.G.bc_slot[1], L.G.bc_lane[1], L.G.start_slot, L.G.start_lane, L.G.end_slot, L.G.end_lane,
                              (int32_t)(uint32_t)L.G.hub_mask, (int32_t)(uint32_t)(L.G.hub_mask >> 32)};
    std::memcpy(out_meta, meta, sizeof(meta));
    return STRQ_OK;
}

int strq_model_set_positions(strq_ctx* c, int32_t model_id, const int32_t* kind, const int32_t* pos)
{
    strq::CtxScope scope_(c);
    if (!c) return STRQ_ERR_ARG;
    if (model_id < 0 || model_id >= (int32_t)c->models.size() || !c->models[model_id] || !kind || !pos) { c->err = "bad argument"; return STRQ_ERR_ARG; }
    STRQ_HIP(c, hipSetDevice(c->device));
    return build_vit_g2(c, c->models[model_id], kind, pos);
}

int strq_viterbi_batch(strq_ctx* c, int32_t model_id, int64_t n_seq, const double* x, const int64_t* x_off,
                       double* logp, int64_t* counted, int32_t* status, int32_t* paths)
{
    strq::CtxScope scope_(c);
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
        t.model = hm->dev; t.sig = c->vit_x.as<double>() + x_off[i]; t.T = x_off[i + 1] - x_off[i]; t.src_kind = VIT_SRC_F64;
        if (paths) { t.bp = c->vit_bp.as<uint16_t>() + bp_off; bp_off += (size_t)(t.T + 1) * n; hp[pos] = c->vit_path.as<int32_t>() + x_off[i]; }
    }
    STRQ_HIP(c, hipMemcpyAsync(d_tasks, tasks.data(), (size_t)n_seq * sizeof(VitTask), hipMemcpyHostToDevice, st));
    if (paths) STRQ_HIP(c, hipMemcpyAsync(d_paths, hp.data(), (size_t)n_seq * 8, hipMemcpyHostToDevice, st));
    STRQ_HIP(c, c->queue.reserve(1024));
    STRQ_HIP(c, hipMemsetAsync(c->queue.p, 0, 1024, st));
    STRQ_HIP(c, hipEventRecord(c->ev[0], st));
    const int shape = vit_shape_for(hm->h, paths ? 1 : 0);
    if (shape < 0) { c->err = "model does not fit a compiled Viterbi kernel"; return STRQ_ERR_UNSUPPORTED; }
    int rc = launch_viterbi(st, shape, hm->h.n_cells, d_tasks, d_res, (int)n_seq, c->queue.as<int>(), c->n_cu, paths ? 1 : 0);
    if (rc) { c->err = "viterbi launch failed"; return rc == 2 || rc == 3 ? STRQ_ERR_UNSUPPORTED : STRQ_ERR_DEVICE; }
    STRQ_HIP(c, hipEventRecord(c->ev[1], st));
    if (paths) {
        if (launch_vit_traceback(st, d_tasks, d_res, d_paths, (int)n_seq)) { c->err = "traceback launch failed"; return STRQ_ERR_DEVICE; }
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
    strq::CtxScope scope_(c);
    const int64_t off[2] = {0, T};
    return strq_viterbi_batch(c, model_id, 1, x, off, logp, counted, status, path);
}

}  // extern "C"
