// Profile-HMM Viterbi decode on gfx950: one wave64 per observation window.
//
// Replaces pomegranate 0.10.0 HiddenMarkovModel.viterbi() as STRique calls it
// (reference scripts/STRique.py:434 flankedRepeatHMM.count_repeats, :493 repeatModHMM.mod_repeats).
//
// float64 log-space max-plus, same operation order as the CPU oracle (oracle/viterbi_oracle.c):
//   emitting l:  v[t][l] = max_k( v[t-1][k] + A[k][l] ) + e_l(x_t)      first in-edge wins ties
//   silent   l:  v[t][l] = max_k( v[t][k]   + A[k][l] )                  emitting k, then earlier silent k
//
// Mapping: states are spread over the 64 lanes (EPL emitting + SPL silent slots per lane), the
// value vector of the previous/current time step lives in this wave's LDS slice, and every lane
// gathers the predecessors of the states it owns from it (in-edges and their log-probabilities
// stay in registers).  The silent states form a DAG dominated by the delete chains of the two
// flank profiles (~50 states each), and those chains are live on every time step: ahead of the
// decoding front a state is best reached by skipping, so the value of d[i] really is
// d[i-1] + log(delete_delete).  The chains are therefore laid along the lanes -- the chain
// predecessor of the state in lane l sits in lane l-1 -- and relaxed systolically in registers:
//     y[l] = max(y[l], y[l-1] + a[l])          one DPP wave_shr:1 per sweep, no LDS
// until no lane changes.  Each sweep performs exactly the additions of the serial topological pass
// (max commutes with the monotone map y -> fl(y + a)), so the fixed point is bit-identical to the
// oracle's, including the argmax.  The few silent edges that are not chain edges (profile exits,
// model end) go through LDS like the emitting states, in an outer fixed-point loop.
//
// Missing observations: pomegranate >= 0.9 gives a NaN observation log-probability 0 under every distribution, so a
// window of NaNs is decoded on the transition probabilities alone (the reference reaches this when normalize2model
// returns NaNs for the filtered signal but not for the morphology signal, STRique.py:596-597,603).
//
// Repeat counting does not need a traceback: the number of visits of the counted states
// (the two `dummy` states of repeatHMM, STRique.py:374-378) is carried along the best path.
// Back-pointers are written only when the caller wants the state path (modification pass).
#include "strq_opt.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>
#include "viterbi_kernels.h"

namespace strq {

static __device__ __forceinline__ int vit_next_task(int* queue, int lane)
{
    __builtin_amdgcn_wave_barrier();
    int ti = 0;
    if (lane == 0) ti = atomicAdd(queue, 1);
    __builtin_amdgcn_wave_barrier();
    ti = __builtin_amdgcn_readfirstlane(ti);
    __builtin_amdgcn_wave_barrier();
    return ti;
}

static __device__ __forceinline__ double readlane_f64(double v, int l)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)u, l);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

#define VIT_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

// -DSTRQ_VIT_TIMING (diagnostic build, tools/vit_timing.py): shader cycles spent in the four parts of a time step, summed
// over all waves -- [0] emitting phase up to its stores, [1] silent gather + tournament, [2] chain sweeps, [3] stores and loop
// overhead, [4] time steps.  s_memtime drains the wave's LDS / scalar queues, so the split perturbs what it measures (the
// parts add up to more than an uninstrumented step); it shows where a step's time goes, not how long a step takes.
#ifdef STRQ_VIT_TIMING
__device__ unsigned long long vit_timing_acc[8];
#define VIT_CLOCK() __builtin_readcyclecounter()
#endif

// wave_shr:1 -- lane l receives lane l-1; lane 0 receives +0.0 / 0 (bound_ctrl).  The chain sweeps add the
// chain log-probability afterwards, which is -inf for every lane without a chain predecessor (lane 0
// never has one), so the filler value cannot survive.
static __device__ __forceinline__ double dpp_shr1_f64(double v)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, 0x138, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), 0x138, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
static __device__ __forceinline__ int dpp_shr1_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);
}
// (tin > y) ? value of `prev` in lane l-1 : keep -- compare and select with the lane shift folded into the select's first operand
// (v_cndmask_b32 picks src0 when the condition is clear, and only src0 takes a DPP control: hence the inverted compare).
// The s_nop gives the two wait states a DPP read needs after a VALU write of the same register, whatever the compiler put in front.
static __device__ __forceinline__ int sel_shr1_i32(double tin, double y, int prev, int keep)
{
    asm("v_cmp_ngt_f64 vcc, %2, %3\n\ts_nop 0\n\tv_cndmask_b32_dpp %0, %1, %0, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(keep) : "v"(prev), "v"(tin), "v"(y) : "vcc");
    return keep;
}
static __device__ __forceinline__ uint64_t sel_shr1_u64(double tin, double y, uint64_t prev, uint64_t keep)
{
    int klo = (int)(uint32_t)keep, khi = (int)(uint32_t)(keep >> 32);
    asm("v_cmp_ngt_f64 vcc, %4, %5\n\ts_nop 0\n\tv_cndmask_b32_dpp %0, %2, %0, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_dpp %1, %3, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(klo), "+v"(khi) : "v"((int)(uint32_t)prev), "v"((int)(uint32_t)(prev >> 32)), "v"(tin), "v"(y) : "vcc");
    return ((uint64_t)(uint32_t)khi << 32) | (uint32_t)klo;
}
// lane-mask select: the mask lives in an SGPR pair (the result of compares combined with scalar instructions)
static __device__ __forceinline__ int sel_mask_i32(int if0, int if1, uint64_t mask)
{
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
    return r;
}
// max of two non-NaN doubles in one instruction (__builtin_fmax makes the compiler canonicalise a
// loop-carried operand first: a second v_max_f64 per sweep)
static __device__ __forceinline__ double max_f64_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// In-edges per state held in registers: compile-time bounds so that the gather loops are branch free
// and all LDS reads of a time step are issued back to back.  Emitting slots are sorted by in-degree:
// the first half of the slots gets DE_HI edge registers, the second half DE_LO; silent slots get DS
// (without their chain edge).  Padding edges read the -inf cell.
//
// The two value buffers of a wave sit a compile-time distance apart and the time loop is unrolled
// by two, so every LDS access is `register pointer + immediate offset` (no per-step address
// arithmetic); lanes that own no state in a slot write to a cell nobody reads (no exec masking).
template <int EPL, int SPL> struct VitLds {
    static constexpr int TRASH = (EPL + SPL) * 64 + 1;     // model cells: (epl + spl) * 64 + the -inf cell
    static constexpr int BUF = (TRASH + 1) * 16;           // bytes per value buffer
    // waves per CU: two per SIMD.  (Three fit shape (4,2) at 168 VGPRs and raise the throughput of
    // large uniform batches by ~8%, but a 4096-read batch is bound by its longest window, whose
    // per-step latency gets worse -- measured 176 ms vs 164 ms per bench step.)
#ifdef STRQ_VIT_WAVES12
    static constexpr int WAVES = (160 * 1024) / (2 * BUF) >= 12 ? 12 : ((160 * 1024) / (2 * BUF) >= 8 ? 8 : ((160 * 1024) / (2 * BUF) >= 4 ? 4 : 2));
#else
    static constexpr int WAVES = (160 * 1024) / (2 * BUF) >= 8 ? 8 : ((160 * 1024) / (2 * BUF) >= 4 ? 4 : 2);
#endif
};

// SS: every model of the launch is single-stage (no silent state has a silent predecessor outside
// its chain) -- the silent phase is then straight-line code: gather, chain sweeps, store.
//
// MARK (modification pass): the flanked model is one-way -- prefix profile, repeat loop, suffix profile --
// so the samples decoded into `repeat` states (STRique.py:608) are one contiguous stretch of the window.
// Instead of back-pointers for every (time step, state) the best path carries, next to its repeat count,
// the time of its first emission inside the repeat section and of its first emission after it: the 8
// payload bytes of a cell hold  lo = count | enter[11:0] << 20,  hi = enter[21:12] | leave << 10  (windows below
// 2^21 samples): the split keeps every update a 32-bit operation (64-bit shifts are slow on the VALU).
#define VIT_MARK_T_MAX ((int64_t)1 << 21)
//
// HUB (modification model): every path passes through the emitting hub states between two repeat units
// (s0 -> base | modified unit profile -> e0 -> s0 ...), and all states of a unit belong to one branch.  The
// best path carries the time of its last e0 emission and the branch of its current unit; every e0 emission
// at time t stores that pair as record t (8 bytes) -- so the per-unit '0'/'1' string is read back by hopping
// from hub to hub (one hop per repeat unit) instead of tracing a back-pointer per (time step, state).
template <int EPL, int SPL, int DE_HI, int DE_LO, int DS, bool BP, bool SS, bool MARK = false, bool HUB = false>
__global__ void __launch_bounds__((64 * VitLds<EPL, SPL>::WAVES))
viterbi_kernel(const VitTask* __restrict__ tasks, VitResult* __restrict__ results,
               int n_tasks, int* __restrict__ queue, const int* __restrict__ order)
{
    extern __shared__ double lds_d[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int TRASH = VitLds<EPL, SPL>::TRASH, BUF = VitLds<EPL, SPL>::BUF;
    // one 16-byte cell per state: {double value; int count; int pad} -> one ds_read_b128 per in-edge
    char* const vbase = reinterpret_cast<char*>(lds_d) + (size_t)wave * 2 * BUF;
    // DE_HI above 16 packs two degrees: tens = in-edge registers of slot 0, units = of the other busy slots (65: the one
    // or two states of a flanked-repeat model with six in-edges sit in slot 0, the matches with five in the next)
    constexpr int HI0 = DE_HI > 16 ? DE_HI / 10 : DE_HI, HI1 = DE_HI > 16 ? DE_HI % 10 : DE_HI;
    // DE_LO above 10: the units are the degree, and the slots of the second half hold no Normal emission (the inserts of
    // a profile: uniform) -- with clipped observations inside every uniform support their emission is the constant ecf
    constexpr int LO = DE_LO > 10 ? DE_LO % 10 : DE_LO;
    constexpr bool LO_FLAT = DE_LO > 10;
    // the packed shapes are chosen for models without counted silent states (vit_shape_base): the payload of a silent state is
    // then its predecessor's as it is -- one VALU instruction less per chain hop (the lane shift folds into the select)
    constexpr bool SILENT_COUNTED = DE_HI <= 16;
    constexpr int DEMAX = HI0 > LO ? HI0 : LO;
    auto de_of = [](int s) constexpr { return s == 0 ? HI0 : (s < (EPL + 1) / 2 ? HI1 : LO); };
    const double NEGINF = -__builtin_inf();
    constexpr bool WIDE = MARK || HUB;
    using Pay = std::conditional_t<WIDE, uint64_t, int>;      // what rides along the best path
    struct alignas(16) Cell { double v; Pay c; };
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    auto ldcell = [](const char* p, int boff) {      // one ds_read_b128 (a 96-bit read of the three dwords in use measures 40 % slower: profiles/r03_dead_ends.md)
        const v4u q = *reinterpret_cast<const v4u*>(p + boff);
        Cell x; x.v = __builtin_bit_cast(double, ((uint64_t)q.y << 32) | q.x);
        if constexpr (WIDE) x.c = ((uint64_t)q.w << 32) | q.z; else x.c = (int)q.z;
        return x;
    };
    typedef unsigned v3u __attribute__((ext_vector_type(3)));
    auto stcell = [](char* p, int boff, double v, Pay c) {      // one ds_write_b128, or ds_write_b96 when the payload is 32 bits
        const uint64_t u = __builtin_bit_cast(uint64_t, v);
        if constexpr (WIDE) {
            v4u q; q.x = (unsigned)u; q.y = (unsigned)(u >> 32); q.z = (unsigned)c; q.w = (unsigned)((uint64_t)c >> 32);
            *reinterpret_cast<v4u*>(p + boff) = q;
        } else {
            // the fourth dword of a cell is never read as data: no register to fill, and the store moves 12 instead of 16 bytes per lane
            v3u q; q.x = (unsigned)u; q.y = (unsigned)(u >> 32); q.z = (unsigned)c;
            *reinterpret_cast<v3u*>(p + boff) = q;
        }
    };
    auto pay_add = [](Pay v, int inc) -> Pay {       // count += inc (the count field never carries out of the low half)
        if constexpr (HUB) return v;          // the modification model counts nothing
        else if constexpr (MARK) return ((uint64_t)v & 0xFFFFFFFF00000000ull) | (uint32_t)((uint32_t)v + (uint32_t)inc);
        else return v + inc;
    };
    auto spay_add = [&](Pay v, int inc) -> Pay { if constexpr (SILENT_COUNTED) return pay_add(v, inc); else return v; };
    auto shr1_pay = [](Pay v) -> Pay {               // payload of lane l-1
        if constexpr (WIDE) return ((uint64_t)(uint32_t)dpp_shr1_i32((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)dpp_shr1_i32((int)(uint32_t)v);
        else return dpp_shr1_i32(v);
    };
    const VitModel* cur_model = nullptr;
    int n = 0, NP = 0, m_start = 0, m_end = 0, scell0 = 0, dummy = 0, start_state = 0;
    constexpr bool single_stage = SS;
    // everything a lane needs about the states it owns lives in registers (reloaded when the model changes)
    int own_e[EPL], einc[EPL]; bool enorm[EPL]; bool etag[EPL]; bool ehub[EPL], erec[EPL];
    const char* esrc[EPL][DEMAX]; char* edst[EPL];
    double ea[EPL], eb[EPL], ec[EPL], elp[EPL][DEMAX];
    double ebf[EPL], ecf[EPL];     // branch-free emission: ecf - (x - ea)^2 * ebf  (uniform: ebf = 0; padding: ecf = -inf)
    double uni_lo_max = 0.0, uni_hi_min = 0.0;
    int own_s[SPL], sinc[SPL];
    const char* ssrc[SPL][DS]; char* sdst[SPL];
    double slp[SPL][DS], clp[SPL];

    for (;;) {
        const int tq = vit_next_task(queue, lane);
        if (tq >= n_tasks) break;
        const int ti = order ? order[tq] : tq;        // longest observation windows first
#ifdef STRQ_VIT_PRIO
        // experiment: the longest windows of a launch are its critical path -- let their waves issue ahead of the wave they share a SIMD with
        if (tq < n_tasks / STRQ_VIT_PRIO) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
#endif
        const VitTask tk = tasks[ti];
        if (tk.model != cur_model) {
            cur_model = tk.model;
            const VitModel& M = *cur_model;
            n = M.n_states; NP = M.n_cells; m_start = M.start_cell; m_end = M.end_cell;
            scell0 = M.epl * 64; dummy = M.n_cells - 1; start_state = M.start;
            uni_lo_max = M.uni_lo_max; uni_hi_min = M.uni_hi_min;
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                const bool on = s < M.epl;
                own_e[s] = on ? M.own_e[s * 64 + lane] : -1;
                const int kind = on ? M.emis_kind[s * 64 + lane] : 0;
                enorm[s] = kind == 1;
                // padding states: the uniform test x <= -inf never holds -> emission -inf
                ea[s] = kind ? M.emis_a[s * 64 + lane] : 0.0; eb[s] = kind ? M.emis_b[s * 64 + lane] : NEGINF;
                ec[s] = kind ? M.emis_c[s * 64 + lane] : 0.0;
                ebf[s] = kind == 1 ? eb[s] : 0.0; ecf[s] = kind ? ec[s] : NEGINF;
                einc[s] = own_e[s] >= 0 ? M.count_inc[own_e[s]] : 0;
                etag[s] = own_e[s] >= 0 && M.state_tag[own_e[s]] == 1;
                ehub[s] = own_e[s] >= 0 && M.state_tag[own_e[s]] == 2;
                erec[s] = own_e[s] >= 0 && own_e[s] == M.rec_state;
                edst[s] = vbase + 16 * (own_e[s] >= 0 ? s * 64 + lane : TRASH);
#pragma unroll
                for (int j = 0; j < DEMAX; ++j) {
                    const bool ej = on && j < M.e_deg[s];
                    esrc[s][j] = vbase + 16 * (ej ? M.edge_src[(M.e_base[s] + j) * 64 + lane] : dummy);
                    elp[s][j] = ej ? M.edge_logp[(M.e_base[s] + j) * 64 + lane] : 0.0;
                }
            }
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const bool on = s < M.spl;
                own_s[s] = on ? M.own_s[s * 64 + lane] : -1;
                sinc[s] = own_s[s] >= 0 ? M.count_inc[own_s[s]] : 0;
                const bool has_chain = on && M.chain_src[s * 64 + lane] >= 0;
                clp[s] = has_chain ? M.chain_logp[s * 64 + lane] : NEGINF;
                sdst[s] = vbase + 16 * (own_s[s] >= 0 ? scell0 + lane * SPL + s : TRASH);      // chain position p = lane * SPL + s sits in cell scell0 + p
#pragma unroll
                for (int j = 0; j < DS; ++j) {
                    const bool ej = on && j < M.s_deg[s];
                    ssrc[s][j] = vbase + 16 * (ej ? M.edge_src[(M.s_base[s] + j) * 64 + lane] : dummy);
                    slp[s][j] = ej ? M.edge_logp[(M.s_base[s] + j) * 64 + lane] : 0.0;
                }
            }
        }
        const int64_t T = tk.T;
#ifdef STRQ_VIT_TIMING
        unsigned long long tm_emit = 0, tm_gather = 0, tm_sweep = 0, tm_rest = 0, tm_steps = 0, tm_mark = 0;
#endif
#ifdef STRQ_VIT_STATS
        uint32_t stat_sweeps = 0;      // debug build (-DSTRQ_VIT_STATS): chain sweeps of the whole window, reported in place of the count
#endif
        // clipped observations (detect pipeline) that cannot leave any uniform emission's support
        // ... and are numbers: a window normalised with NaN constants (detect on a read whose filtered signal has empty
        // percentile tails, STRique.py:597,603) is all NaN and takes the general emission code below
        const bool fast_em = tk.src_kind != VIT_SRC_F64 && tk.lo >= uni_lo_max && tk.hi <= uni_hi_min && tk.c1 == tk.c1 && tk.h1 == tk.h1;
        for (int i = lane; i < NP; i += 64) { stcell(vbase, 16 * i, NEGINF, 0); stcell(vbase, BUF + 16 * i, NEGINF, 0); }
        VIT_FENCE();
        if (lane == 0) stcell(vbase, 16 * m_start, 0.0, 0);
        VIT_FENCE();

        // Best of the first `cnt` candidates into element 0 (value, carried count, source cell).  Pairwise
        // tournament instead of a serial chain (shorter dependency chains); the right-hand candidate wins
        // only with a strict '>', so among equal values the lowest index -- the first in-edge -- survives.
        auto tournament = [&](auto& cv, auto& cc, auto& ca, int cnt) {
#pragma unroll
            for (int stride = 1; stride < 8; stride *= 2) {
#pragma unroll
                for (int j = 0; j + stride < 8; j += 2 * stride) {
                    if (j + stride < cnt) {
                        const bool gt = cv[j + stride] > cv[j];
                        cc[j] = gt ? cc[j + stride] : cc[j];
                        if (BP) ca[j] = gt ? ca[j + stride] : ca[j];
                        cv[j] = __builtin_fmax(cv[j], cv[j + stride]);
                    }
                }
            }
        };

        // Chains of silent states (the delete states of a profile) zig-zag through the silent slots: the chain
        // predecessor of a cell in slot s > 0 is the cell of the same lane in slot s - 1, that of a cell in slot 0 the
        // cell of the previous lane in the last slot.  One sweep therefore carries a value SPL positions along a chain
        // for one lane shift (a time step needs ~7 positions on the C9orf72 model), every hop the same sequential
        // float64 addition as when each state is evaluated once in topological order.  After a sweep the only values
        // not yet carried on are those that changed in the last slot (the next lane reads them in the next sweep).
        auto chain_sweeps = [&](double (&y)[SPL], Pay (&yc)[SPL], int (&arg)[SPL]) {
            for (;;) {
                bool win_any = false;
#ifdef STRQ_VIT_STATS
                ++stat_sweeps;
#endif
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    double tin; Pay cin;
                    if (s == 0) { tin = dpp_shr1_f64(y[SPL - 1]) + clp[0]; if constexpr (SILENT_COUNTED) cin = pay_add(shr1_pay(yc[SPL - 1]), sinc[0]); }
                    else { tin = y[s - 1] + clp[s]; cin = spay_add(yc[s - 1], sinc[s]); }
                    const bool win = tin > y[s];     // the chain edge is the last in-edge: strict (clp = -inf without one)
                    if (s == 0 && !SILENT_COUNTED) {      // payload of the previous lane's last slot, taken as it is
                        if constexpr (WIDE) yc[0] = sel_shr1_u64(tin, y[0], yc[SPL - 1], yc[0]); else yc[0] = sel_shr1_i32(tin, y[0], yc[SPL - 1], yc[0]);
                    } else yc[s] = win ? cin : yc[s];
                    y[s] = max_f64_raw(y[s], tin);
                    if (BP) arg[s] = win ? scell0 + lane * SPL + s - 1 : arg[s];     // the chain predecessor's cell
                    if (s == SPL - 1) win_any = win;     // wins in the earlier slots were carried on inside this sweep already
                }
                if (!__any(win_any)) break;
            }
        };

        // Silent states of the buffer at byte offset OFF to their fixed point.  PIN: keep start at 0 (t = 0).
        auto relax_silent = [&](auto pin_c, auto off_c, int64_t trow) {
            constexpr bool PIN = decltype(pin_c)::value;
            constexpr int OFF = decltype(off_c)::value;
            if constexpr (SS) {
                double y[SPL]; Pay yc[SPL]; int arg[SPL];
                Cell spc[SPL][DS];
#pragma unroll
                for (int s = 0; s < SPL; ++s)
#pragma unroll
                    for (int j = 0; j < DS; ++j) spc[s][j] = ldcell(ssrc[s][j], OFF);
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    double cv[DS]; Pay cc[DS]; int ca[DS];
#pragma unroll
                    for (int j = 0; j < DS; ++j) { cv[j] = spc[s][j].v + slp[s][j]; cc[j] = spc[s][j].c; ca[j] = BP ? (int)(ssrc[s][j] - vbase) >> 4 : 0; }
                    tournament(cv, cc, ca, DS);
                    double best = cv[0]; Pay bc = cc[0]; int a = ca[0];
                    if (PIN && own_s[s] == start_state) { best = 0.0; bc = spay_add((Pay)0, -sinc[s]); a = dummy; }
                    y[s] = best; yc[s] = spay_add(bc, sinc[s]); arg[s] = a;
                }
#ifdef STRQ_VIT_TIMING
                const unsigned long long tc_a = VIT_CLOCK();
#endif
                chain_sweeps(y, yc, arg);
#ifdef STRQ_VIT_TIMING
                const unsigned long long tc_b = VIT_CLOCK();
                tm_sweep += tc_b - tc_a; tm_mark = tc_a;
#endif
#pragma unroll
                for (int s = 0; s < SPL; ++s) stcell(sdst[s], OFF, y[s], yc[s]);
                VIT_FENCE();
                if (BP) {
#pragma unroll
                    for (int s = 0; s < SPL; ++s)
                        if (own_s[s] >= 0 && tk.bp) tk.bp[(size_t)trow * n + own_s[s]] = (uint16_t)arg[s];
                }
                return;
            }
            double y[SPL], base_prev[SPL]; Pay yc[SPL]; int arg[SPL];
#pragma unroll
            for (int s = 0; s < SPL; ++s) { y[s] = NEGINF; yc[s] = 0; arg[s] = dummy; base_prev[s] = __builtin_nan(""); }
            for (int outer = 0;; ++outer) {
                // (A) best non-chain in-edge of every silent state: emitting predecessors (final for
                //     this time step) and silent predecessors that are not chain neighbours
                bool base_changed = false;
                double base[SPL]; Pay basec[SPL]; int basea[SPL];
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    double best = NEGINF; Pay bc = 0; int a = dummy;
#pragma unroll
                    for (int j = 0; j < DS; ++j) {
                        const Cell pc = ldcell(ssrc[s][j], OFF);
                        const double c = pc.v + slp[s][j];
                        const bool gt = c > best;          // strict: the first of equal candidates wins
                        bc = gt ? pc.c : bc;
                        if (BP) a = gt ? (int)(ssrc[s][j] - vbase) >> 4 : a;
                        best = __builtin_fmax(best, c);
                    }
                    if (PIN && own_s[s] == start_state) { best = 0.0; bc = spay_add((Pay)0, -sinc[s]); a = dummy; }
                    base[s] = best; basec[s] = spay_add(bc, sinc[s]); basea[s] = a;
                    if (!(best == base_prev[s]) && !(best != best)) base_changed = true;
                }
                bool changed = false;
                if (single_stage || outer == 0 || __any(base_changed)) {
                    // (B) chains.  Within a time step every quantity only grows, so the sweep continues
                    //     from the current values; a local candidate that ties with a chain token wins
                    //     (it precedes the chain edge in evaluation order).
#pragma unroll
                    for (int s = 0; s < SPL; ++s) {
                        if (base[s] >= y[s]) { y[s] = base[s]; yc[s] = basec[s]; arg[s] = basea[s]; }
                        base_prev[s] = base[s];
                    }
                    chain_sweeps(y, yc, arg);
                    if (!single_stage) {
#pragma unroll
                        for (int s = 0; s < SPL; ++s) {
                            if (own_s[s] >= 0) {
                                const Cell oc = ldcell(sdst[s], OFF);
                                if (!(oc.v == y[s]) || oc.c != yc[s]) changed = true;
                            }
                        }
                    }
                }
                if (!single_stage && !__any(changed)) break;
                VIT_FENCE();
#pragma unroll
                for (int s = 0; s < SPL; ++s) stcell(sdst[s], OFF, y[s], yc[s]);
                VIT_FENCE();
                if (single_stage) break;     // nothing downstream of the chains inside this time step
            }
            if (BP) {
#pragma unroll
                for (int s = 0; s < SPL; ++s)
                    if (own_s[s] >= 0 && tk.bp) tk.bp[(size_t)trow * n + own_s[s]] = (uint16_t)arg[s];
            }
        };

        // one observation: emitting states from the buffer at RD into the buffer at WR, then its silent states
        // FAST: the window's clipped observations cannot leave a uniform emission's support (and are numbers): decided once
        // per window, outside the time loop -- inside it the choice costs ~20 scalar instructions of exec-mask bookkeeping per step
        auto step = [&](auto rd_c, auto fast_c, double x, int64_t t) {
            constexpr int RD = decltype(rd_c)::value, WR = RD ? 0 : BUF;
            constexpr bool FAST = decltype(fast_c)::value;
            double nv[EPL]; Pay nc[EPL]; int na[EPL];
#ifdef STRQ_VIT_TIMING
            const unsigned long long tc0 = VIT_CLOCK();
#endif

            const uint32_t tt1 = (uint32_t)(t + 1);                       // wave-uniform: scalar registers
            const uint32_t mark_e_lo = (tt1 & 0xFFFu) << 20, mark_e_hi = tt1 >> 12, mark_l_hi = tt1 << 10;
            (void)mark_e_lo; (void)mark_e_hi; (void)mark_l_hi;
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                Cell pcs[DEMAX];          // all reads of the slot in flight before the first use
#pragma unroll
                for (int j = 0; j < DEMAX; ++j)
                    if (j < de_of(s)) pcs[j] = ldcell(esrc[s][j], RD);
                double cv[DEMAX]; Pay cc[DEMAX]; int ca[DEMAX];
#pragma unroll
                for (int j = 0; j < DEMAX; ++j) {
                    if (j < de_of(s)) { cv[j] = pcs[j].v + elp[s][j]; cc[j] = pcs[j].c; ca[j] = BP ? (int)(esrc[s][j] - vbase) >> 4 : 0; }
                }
                tournament(cv, cc, ca, de_of(s));
                const double best = cv[0]; const Pay bc = cc[0]; const int a = ca[0];
                double em;
                if constexpr (FAST) {          // every observation of this window lies inside all uniform emissions
                    if constexpr (LO_FLAT) {
                        if (s >= (EPL + 1) / 2) em = ecf[s];
                        else { const double d = x - ea[s]; em = ecf[s] - (d * d) * ebf[s]; }
                    } else {
                        const double d = x - ea[s];
                        em = ecf[s] - (d * d) * ebf[s];
                    }
                } else {
                    const double d = x - ea[s];
                    const double en = ec[s] - (d * d) * eb[s];
                    const double eu = (x >= ea[s] && x <= eb[s]) ? ec[s] : NEGINF;
                    em = enorm[s] ? en : eu;
                    // a missing observation (NaN) has log-probability 0 under every distribution (pomegranate 0.10
                    // NormalDistribution / UniformDistribution._log_probability, [recalled]); slots without a state stay at -inf
                    if (x != x) em = own_e[s] >= 0 ? 0.0 : NEGINF;
                }
                nv[s] = best + em; na[s] = a;
                if constexpr (MARK) {
                    // first emission inside the repeat section / first emission after it, at time t + 1
                    uint32_t lo = (uint32_t)bc + (uint32_t)einc[s], hi = (uint32_t)((uint64_t)bc >> 32);
                    const bool entered = ((lo >> 20) | (hi & 0x3FFu)) != 0, left = (hi >> 10) != 0;
                    const bool set_e = etag[s] && !entered, set_l = !etag[s] && entered && !left;
                    lo |= set_e ? mark_e_lo : 0u;
                    hi |= set_e ? mark_e_hi : (set_l ? mark_l_hi : 0u);
                    nc[s] = ((uint64_t)hi << 32) | lo;
                } else if constexpr (HUB) {
                    uint32_t lo = (uint32_t)bc, hi = (uint32_t)((uint64_t)bc >> 32);      // last e0 time, branch of the current unit
                    if (!ehub[s]) hi = etag[s] ? 1u : 0u;
                    if (erec[s] && tk.bp) {
                        reinterpret_cast<uint64_t*>(tk.bp)[t + 1] = ((uint64_t)hi << 32) | lo;      // record of this e0 emission
                        lo = tt1;
                    }
                    nc[s] = ((uint64_t)hi << 32) | lo;
                } else nc[s] = bc + einc[s];
            }
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                stcell(edst[s], WR, nv[s], nc[s]);
                if (BP) { if (own_e[s] >= 0 && tk.bp) tk.bp[(size_t)(t + 1) * n + own_e[s]] = (uint16_t)na[s]; }
            }
            if (!single_stage) {      // silent cells feed other silent states only in multi-stage models
#pragma unroll
                for (int s = 0; s < SPL; ++s) stcell(sdst[s], WR, NEGINF, 0);
            }
            VIT_FENCE();
#ifdef STRQ_VIT_TIMING
            const unsigned long long tc1 = VIT_CLOCK();
            tm_emit += tc1 - tc0;
#endif
#if defined(STRQ_X_ST) || defined(STRQ_X_RD) || defined(STRQ_X_VALU)
            // sensitivity probes (tools/build_variant.sh, never in the product build): redundant work of one kind per time step --
            // what a step's time does in response tells which unit it waits for (profiles/r03_vit_sensitivity.md)
#ifdef STRQ_X_ST
#pragma unroll
            for (int k = 0; k < STRQ_X_ST; ++k)      // the cell just stored, stored again (same bytes)
                asm volatile("ds_write_b96 %0, %1 offset:%2" :: "v"((unsigned)(uintptr_t)edst[k & 3]), "v"(v3u{(unsigned)__builtin_bit_cast(uint64_t, nv[k & 3]), (unsigned)(__builtin_bit_cast(uint64_t, nv[k & 3]) >> 32), (unsigned)nc[k & 3]}), "n"(WR) : "memory");
#endif
#ifdef STRQ_X_RD
#pragma unroll
            for (int k = 0; k < STRQ_X_RD; ++k)      // reads nobody uses, into registers the kernel does not allocate
                asm volatile("ds_read_b128 v[248:251], %0 offset:%1" :: "v"((unsigned)(uintptr_t)edst[k & 3]), "n"(RD) : "v248", "v249", "v250", "v251", "memory");
#endif
#ifdef STRQ_X_VALU
            {
                double acc0 = x, acc1 = x, acc2 = x, acc3 = x;
#pragma unroll
                for (int k = 0; k < STRQ_X_VALU; k += 4) {
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc0) : "v"(nv[0]));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc1) : "v"(nv[1]));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc2) : "v"(nv[2]));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc3) : "v"(nv[3]));
                }
            }
#endif
#endif
            relax_silent(std::false_type{}, std::integral_constant<int, WR>{}, t + 1);
#ifdef STRQ_VIT_TIMING
            const unsigned long long tc2 = VIT_CLOCK();
            tm_gather += tm_mark - tc1;           // silent gather + tournament: up to the first sweep
            tm_rest += tc2 - tm_mark;             // (includes the sweeps; subtracted on the host)
            ++tm_steps;
#endif
        };

        relax_silent(std::true_type{}, std::integral_constant<int, 0>{}, 0);

        auto run_window = [&](auto fast_c) {
        double xchunk = 0.0;
        for (int64_t t0 = 0; t0 < T; t0 += 64) {
            // observations t0 .. t0+63, one per lane
            {
                const int64_t idx = t0 + lane;
                double xv = 0.0;
                if (idx < T) {
                    if (tk.src_kind == VIT_SRC_F64) xv = reinterpret_cast<const double*>(tk.sig)[idx];
                    else {
                        double sv = tk.src_kind == VIT_SRC_I16_AFFINE ? (double)reinterpret_cast<const int16_t*>(tk.sig)[idx]
                                                                      : reinterpret_cast<const double*>(tk.sig)[idx];
                        sv = (sv - tk.c1) / tk.h1;
                        sv = sv * tk.h2 + tk.c2;
                        sv = sv < tk.lo ? tk.lo : sv;          // np.clip
                        sv = sv > tk.hi ? tk.hi : sv;
                        xv = sv;
                    }
                }
                xchunk = xv;
            }
            const int send = (int)((T - t0) < 64 ? (T - t0) : 64);
            // chunks start at even t, so the buffer roles alternate A->B, B->A inside every pair
            for (int s0 = 0; s0 < send; s0 += 2) {
                step(std::integral_constant<int, 0>{}, fast_c, readlane_f64(xchunk, s0), t0 + s0);
                if (s0 + 1 < send) step(std::integral_constant<int, BUF>{}, fast_c, readlane_f64(xchunk, s0 + 1), t0 + s0 + 1);
            }
        }
        };
        if (__builtin_amdgcn_readfirstlane((int)fast_em) != 0) run_window(std::true_type{});
        else run_window(std::false_type{});
        const Cell fin = ldcell(vbase + 16 * m_end, (T & 1) ? BUF : 0);
        const double lp = fin.v;
        VitResult r; r.logp = lp; r.status = (lp > NEGINF) ? 0 : 1; r.pad_ = 0;
        r.dbg[0] = r.dbg[1] = r.dbg[2] = r.dbg[3] = 0;
        if constexpr (HUB) {
            r.counted = 0;
            r.dbg[0] = (uint32_t)fin.c;          // time of the last e0 emission on the best path (= T when the path ends properly)
        } else if constexpr (MARK) {
            const uint32_t plo = (uint32_t)fin.c, phi = (uint32_t)((uint64_t)fin.c >> 32);
            r.counted = (lp > NEGINF) ? (int64_t)(plo & 0xFFFFFu) : 0;
            r.dbg[0] = (plo >> 20) | ((phi & 0x3FFu) << 12);       // time (1-based) of the first repeat-section emission, 0 = none
            r.dbg[1] = phi >> 10;                                  // time of the first emission after the repeat section, 0 = none
            if (tk.T >= VIT_MARK_T_MAX) r.status = 2;              // window too long for the packed marks
        } else {
            r.counted = (lp > NEGINF) ? (int64_t)fin.c : 0;
        }
#ifdef STRQ_VIT_STATS
        r.counted = stat_sweeps;
#endif
        results[ti] = r;     // every lane stores the same value
#ifdef STRQ_VIT_TIMING
        if (lane == 0) {
            atomicAdd(&vit_timing_acc[0], tm_emit); atomicAdd(&vit_timing_acc[1], tm_gather); atomicAdd(&vit_timing_acc[2], tm_sweep);
            atomicAdd(&vit_timing_acc[3], tm_rest - tm_sweep); atomicAdd(&vit_timing_acc[4], tm_steps);
        }
#endif
        VIT_FENCE();
    }
}

// ------------------------------------------------------------------------------------------
// Register-resident profile chain (VitG2, viterbi_kernels.h): one wave per window, two chain positions per lane, the value
// vector of the previous time step in VGPRs.  What the lane layouts above fetch through LDS -- 21 ds_read_b128 and 6
// ds_write_b96 per time step, the unit the step was found to wait for (profiles/r03_vit_sensitivity.md) -- is here the lane's
// own registers, one wave_shr:1 DPP shift of four previous values, and two v_readlane broadcasts for the edges that close the
// repeat loop.  Same recurrence, same candidate order (ascending source state, first of equal candidates wins), same chain
// sweeps as viterbi_kernel; count and mark modes (no back-pointers: models keep their lane layout for those).
template <int N, typename P>
static __device__ __forceinline__ void g2_tournament(double (&cv)[8], P (&cc)[8])
{
#pragma unroll
    for (int stride = 1; stride < 8; stride *= 2) {
#pragma unroll
        for (int j = 0; j + stride < 8; j += 2 * stride) {
            if (j + stride < N) {
                const bool gt = cv[j + stride] > cv[j];          // strict: the lower index -- the earlier in-edge -- survives a tie
                cc[j] = gt ? cc[j + stride] : cc[j];
                cv[j] = __builtin_fmax(cv[j], cv[j + stride]);
            }
        }
    }
}

// LX: the three values a lane needs from lane - 1 on every time step (its odd-position match, insert and delete state) travel
// through LDS -- one 16-byte cell per lane and kind, written by the owner, read by the right neighbour -- instead of three DPP
// shifts of three registers each: the kernel is VALU-bound without it (177 VALU instructions per step), and the new Mo / Io
// read for the delete gather of this step are next step's shifted previous values for free.  The rarely used neighbours (the
// even match for the skip edges of a repeat profile) and the two broadcast sources stay on DPP / v_readlane.
#define G2_LDS_CELLS 65              // cell 0 stays -inf: lane 0's left neighbour
#define G2_LDS_WAVE_BYTES (5 * G2_LDS_CELLS * 16)
// Both parities of the chain in one kernel: a repeat profile of odd length (CGG, CAG) puts the two broadcast sources at odd
// positions, and which of the two code paths a window takes is decided per task -- wave-uniform, outside the time loop -- from
// its model (VitModel::g2_odd).  A sub-batch that mixes such targets with even ones (C9orf72 + FMR1 + HTT) is one launch.
template <bool MARK, int WAVES, int LXL>
__global__ void __launch_bounds__(64 * WAVES)
viterbi_g2_kernel(const VitTask* __restrict__ tasks, VitResult* __restrict__ results,
                  int n_tasks, int* __restrict__ queue, const int* __restrict__ order)
{
    extern __shared__ double lds_d[];
    const int lane = threadIdx.x & 63;
    const double NEGINF = -__builtin_inf();
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned v3u __attribute__((ext_vector_type(3)));
    // LXL 1: Mo, Io, Do through LDS; 2: also the even match (skip edges) and the two broadcast sources (uniform-address reads)
    constexpr bool LX = LXL >= 1, LX2 = LXL >= 2;
    // this wave's cells: [Mo | Io | Do | Me | Ie][65]; the owner lane l writes cell l + 1, lane l reads cell l
    char* const xbase = reinterpret_cast<char*>(lds_d) + (size_t)(threadIdx.x >> 6) * G2_LDS_WAVE_BYTES;
    char* const xst = xbase + 16 * (lane + 1);
    const char* const xld = xbase + 16 * lane;
    using Pay = std::conditional_t<MARK, uint64_t, int>;
    auto shr1_pay = [](Pay v) -> Pay {
        if constexpr (MARK) return ((uint64_t)(uint32_t)dpp_shr1_i32((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)dpp_shr1_i32((int)(uint32_t)v);
        else return dpp_shr1_i32(v);
    };
    auto readlane_pay = [](Pay v, int l) -> Pay {
        if constexpr (MARK) return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
        else return __builtin_amdgcn_readlane(v, l);
    };
    auto xstore = [&](int kind, double v, Pay c) {          // ds_write_b96 / b128 at a compile-time offset
        const uint64_t u = __builtin_bit_cast(uint64_t, v);
        if constexpr (MARK) { v4u q; q.x = (unsigned)u; q.y = (unsigned)(u >> 32); q.z = (unsigned)c; q.w = (unsigned)((uint64_t)c >> 32); *reinterpret_cast<v4u*>(xst + kind * (G2_LDS_CELLS * 16)) = q; }
        else { v3u q; q.x = (unsigned)u; q.y = (unsigned)(u >> 32); q.z = (unsigned)c; *reinterpret_cast<v3u*>(xst + kind * (G2_LDS_CELLS * 16)) = q; }
    };
    auto xload = [&](int kind, double& v, Pay& c) {         // ds_read_b128
        const v4u q = *reinterpret_cast<const v4u*>(xld + kind * (G2_LDS_CELLS * 16));
        v = __builtin_bit_cast(double, ((uint64_t)q.y << 32) | q.x);
        if constexpr (MARK) c = ((uint64_t)q.w << 32) | q.z; else c = (int)q.z;
    };
    auto xload_at = [&](int kind, int src_lane, double& v, Pay& c) {      // the cell of one lane, read by all (same address: one LDS cycle)
        const v4u q = *reinterpret_cast<const v4u*>(xbase + kind * (G2_LDS_CELLS * 16) + 16 * (src_lane + 1));
        v = __builtin_bit_cast(double, ((uint64_t)q.y << 32) | q.x);
        if constexpr (MARK) c = ((uint64_t)q.w << 32) | q.z; else c = (int)q.z;
    };
    const VitModel* cur_model = nullptr;
    const VitG2* G = nullptr;
    double la[7], lb[6], lc[3], ld[3], sg0[3], sg1[2], clp[2];
    double ea[2], ebf[2], ecf[4];
    int einc[4]; uint64_t madd[4];
    double uni_lo_max = 0.0, uni_hi_min = 0.0;
    bool odd_model = false;
    uint64_t hub_mask = 0;
    int bc0_lane = 0, bc1_lane = 0, start_slot = 0, start_lane = 0, end_slot = 0, end_lane = 0;

    for (;;) {
        const int tq = vit_next_task(queue, lane);
        if (tq >= n_tasks) break;
        const int ti = order ? order[tq] : tq;        // longest observation windows first
        const VitTask tk = tasks[ti];
        if (tk.model != cur_model) {
            cur_model = tk.model;
            G = cur_model->g2;
            odd_model = cur_model->g2_odd != 0;
            { const uint64_t hm = G->hub_mask;          // into SGPRs: the model pointer came through a vector load
              hub_mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hm >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hm); }
            uni_lo_max = cur_model->uni_lo_max; uni_hi_min = cur_model->uni_hi_min;
            const double* lp = G->lp;
#pragma unroll
            for (int j = 0; j < 7; ++j) la[j] = lp[(G2_ROW_ME + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 6; ++j) lb[j] = lp[(G2_ROW_MO + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 3; ++j) { lc[j] = lp[(G2_ROW_IE + j) * 64 + lane]; ld[j] = lp[(G2_ROW_IO + j) * 64 + lane]; sg0[j] = lp[(G2_ROW_DE + j) * 64 + lane]; }
#pragma unroll
            for (int j = 0; j < 2; ++j) { sg1[j] = lp[(G2_ROW_DO + j) * 64 + lane]; clp[j] = lp[(G2_ROW_CHAIN + j) * 64 + lane]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kind = G->kind[k * 64 + lane];
                const double a = G->em[(k * 3 + 0) * 64 + lane], b = G->em[(k * 3 + 1) * 64 + lane], c = G->em[(k * 3 + 2) * 64 + lane];
                if (k < 2) { ea[k] = kind ? a : 0.0; ebf[k] = kind == 1 ? b : 0.0; }
                ecf[k] = kind ? c : NEGINF;          // branch-free emission  ecf - (x - ea)^2 * ebf  (uniform: ebf = 0; no state: -inf)
                einc[k] = G->inc[k * 64 + lane];
                if constexpr (MARK) madd[k] = G->mark_add[k * 64 + lane]; else madd[k] = 0;
            }
            bc0_lane = G->bc_lane[0] < 0 ? 0 : G->bc_lane[0];
            bc1_lane = G->bc_lane[1] < 0 ? 0 : G->bc_lane[1];
            start_slot = G->start_slot; start_lane = G->start_lane; end_slot = G->end_slot; end_lane = G->end_lane;
        }
        const int64_t T = tk.T;
        const bool fast_em = tk.src_kind != VIT_SRC_F64 && tk.lo >= uni_lo_max && tk.hi <= uni_hi_min && tk.c1 == tk.c1 && tk.h1 == tk.h1;

        // chain sweeps: De (slot 0) takes from lane - 1's Do, Do (slot 1) from the lane's own De -- as in viterbi_kernel
        auto chain_sweeps = [&](double (&y)[2], Pay (&yc)[2]) {
            // The first STRQ_G2_PRESWEEPS sweeps run without the convergence test: a sweep that changes nothing is harmless, a time step of
            // the benchmark's windows needs ~3 productive ones anyway, and every test is a compare -> scalar branch round trip the wave
            // waits for.  Measured (gpurun_out/r4o, A/B on one box): 66.4 ms per 4096 configs[2] windows with 0, 64.5 with 1, 63.7 with 2,
            // 63.8 with 3 unconditional sweeps.
#ifndef STRQ_G2_PRESWEEPS
#define STRQ_G2_PRESWEEPS 2
#endif
#pragma unroll
            for (int pre = 0; pre < STRQ_G2_PRESWEEPS; ++pre) {
                double tin = dpp_shr1_f64(y[1]) + clp[0];
                if constexpr (MARK) yc[0] = sel_shr1_u64(tin, y[0], yc[1], yc[0]); else yc[0] = sel_shr1_i32(tin, y[0], yc[1], yc[0]);
                y[0] = max_f64_raw(y[0], tin);
                tin = y[0] + clp[1];
                const bool win = tin > y[1];
                y[1] = max_f64_raw(y[1], tin);
                yc[1] = win ? yc[0] : yc[1];
            }
            for (;;) {
                double tin = dpp_shr1_f64(y[1]) + clp[0];
                if constexpr (MARK) yc[0] = sel_shr1_u64(tin, y[0], yc[1], yc[0]); else yc[0] = sel_shr1_i32(tin, y[0], yc[1], yc[0]);
                y[0] = max_f64_raw(y[0], tin);
                tin = y[0] + clp[1];
                const bool win = tin > y[1];
                y[1] = max_f64_raw(y[1], tin);
                yc[1] = win ? yc[0] : yc[1];
                if (!__any(win)) break;
            }
        };

        double pv[4]; Pay pc[4]; double dv[2]; Pay dc[2];      // Me, Mo, Ie, Io of the previous time step; De, Do
#pragma unroll
        for (int k = 0; k < 4; ++k) { pv[k] = NEGINF; pc[k] = 0; }
#pragma unroll
        for (int s = 0; s < 2; ++s) { dv[s] = (lane == start_lane && s == start_slot) ? 0.0 : NEGINF; dc[s] = 0; }
        chain_sweeps(dv, dc);          // t = 0: the silent states reachable from start (start itself has no chain edge: it stays 0)
        double rMo = NEGINF, rIo = NEGINF, rDo = NEGINF; Pay qMo = 0, qIo = 0, qDo = 0;      // LX: lane - 1's Mo, Io, Do of the previous time step
        double rMe = NEGINF, rB0 = NEGINF; Pay qMe = 0, qB0 = 0;      // LX2: lane - 1's Me, the broadcast source of the match-type states
        // hub lanes (VitG2::hub_mask) whose even delete slot -- a virtual relay -- took its value of the previous time step from one
        // of its gather columns: there the relay wins a tie against the insert-type state's own columns (the relayed sources
        // stand in front of them in the baked model's in-edge order; a relay whose winner is its broadcast / chain source stands behind)
        uint64_t front_won = 0;
        if constexpr (LX) {
            if (lane < 5) { const uint64_t u = __builtin_bit_cast(uint64_t, NEGINF); v4u q; q.x = (unsigned)u; q.y = (unsigned)(u >> 32); q.z = 0; q.w = 0; *reinterpret_cast<v4u*>(xbase + lane * (G2_LDS_CELLS * 16)) = q; }
            VIT_FENCE();
            xstore(2, dv[1], dc[1]);
            VIT_FENCE();
            xload(2, rDo, qDo);
        }

        auto step = [&](auto fast_c, auto odd_c, double x, int64_t t) {
            constexpr bool FAST = decltype(fast_c)::value;
            // ODD: the two broadcast sources sit at an odd position (a repeat profile of odd length), and insert-type states at
            // odd positions may be fed by the match / insert of the position before (the dummy state behind such a profile)
            constexpr bool ODD = decltype(odd_c)::value;
            constexpr int B0 = ODD ? 1 : 0, B1 = ODD ? 3 : 2;
            (void)t;
            // previous values of lane - 1 (lane 0 receives 0.0: every column that uses them is -inf there)
            double sMe; Pay cMe;
            if constexpr (LX2) { sMe = rMe; cMe = qMe; } else { sMe = dpp_shr1_f64(pv[0]); cMe = shr1_pay(pc[0]); }
            double sMo, sIo, sDo; Pay cMo, cIo, cDo;
            if constexpr (LX) { sMo = rMo; sIo = rIo; sDo = rDo; cMo = qMo; cIo = qIo; cDo = qDo; }
            else { sMo = dpp_shr1_f64(pv[1]); sIo = dpp_shr1_f64(pv[3]); sDo = dpp_shr1_f64(dv[1]); cMo = shr1_pay(pc[1]); cIo = shr1_pay(pc[3]); cDo = shr1_pay(dc[1]); }
            // the broadcast source of the match-type states (previous time step); the one of the delete-type states is read below, from this step's values
            double b0v; Pay b0c;
            if constexpr (LX2) { b0v = rB0; b0c = qB0; }
            else { b0v = readlane_f64(pv[B0], bc0_lane); b0c = readlane_pay(pc[B0], bc0_lane); }
            double nv[4]; Pay nc[4];
            double best[4]; Pay bcnt[4];
            auto tour_me = [&]() {
                double cv[8]; Pay cc[8];
                cv[0] = sMe + la[0]; cc[0] = cMe;  cv[1] = sIo + la[1]; cc[1] = cIo;  cv[2] = sMo + la[2]; cc[2] = cMo;
                cv[3] = pv[2] + la[3]; cc[3] = pc[2];  cv[4] = pv[0] + la[4]; cc[4] = pc[0];  cv[5] = b0v + la[5]; cc[5] = b0c;
                cv[6] = sDo + la[6]; cc[6] = cDo;
                g2_tournament<7>(cv, cc); best[0] = cv[0]; bcnt[0] = cc[0];
            };
            auto tour_mo = [&]() {
                double cv[8]; Pay cc[8];
                cv[0] = sMo + lb[0]; cc[0] = cMo;  cv[1] = pv[2] + lb[1]; cc[1] = pc[2];  cv[2] = pv[0] + lb[2]; cc[2] = pc[0];
                cv[3] = pv[3] + lb[3]; cc[3] = pc[3];  cv[4] = pv[1] + lb[4]; cc[4] = pc[1];  cv[5] = dv[0] + lb[5]; cc[5] = dc[0];
                g2_tournament<6>(cv, cc); best[1] = cv[0]; bcnt[1] = cc[0];
            };
            // insert-type states: themselves, the match and the delete state of their position (what else feeds one in the
            // baked model reaches it through a virtual delete state, viterbi_kernels.h)
            auto tour_ie = [&]() {
                const double c0 = pv[2] + lc[0], c1 = pv[0] + lc[1], c2 = dv[0] + lc[2];
                const bool g1 = c1 > c0;
                const double w = __builtin_fmax(c0, c1); const Pay pw = g1 ? pc[0] : pc[2];
                // the delete column wins on '>' everywhere and on '==' in the flagged hub lanes
                const uint64_t win = __builtin_amdgcn_ballot_w64(c2 > w) | (__builtin_amdgcn_ballot_w64(c2 == w) & front_won);
                best[2] = __builtin_fmax(w, c2);
                if constexpr (MARK) bcnt[2] = ((uint64_t)(uint32_t)sel_mask_i32((int)(uint32_t)(pw >> 32), (int)(uint32_t)(dc[0] >> 32), win) << 32) | (uint32_t)sel_mask_i32((int)(uint32_t)pw, (int)(uint32_t)dc[0], win);
                else bcnt[2] = sel_mask_i32(pw, dc[0], win);
            };
            auto tour_io = [&]() {
                double cv[8]; Pay cc[8];
                cv[0] = pv[3] + ld[0]; cc[0] = pc[3];  cv[1] = pv[1] + ld[1]; cc[1] = pc[1];  cv[2] = dv[1] + ld[2]; cc[2] = dc[1];
                g2_tournament<3>(cv, cc); best[3] = cv[0]; bcnt[3] = cc[0];
            };
            // emission and payload of slot k
            auto finish = [&](auto k_c) {
                constexpr int k = decltype(k_c)::value;
                double em;
                if constexpr (FAST) {
                    if constexpr (k < 2) { const double d = x - ea[k]; em = ecf[k] - (d * d) * ebf[k]; }
                    else em = ecf[k];          // insert-type states carry no Normal emission (checked when the image is built)
                } else {
                    // general emission (observations outside a uniform support, NaN): parameters from memory, every step (rare path)
                    const volatile int32_t* kp = G->kind; const volatile double* ep = G->em;
                    const int kind = kp[k * 64 + lane];
                    const double a = ep[(k * 3 + 0) * 64 + lane], b = ep[(k * 3 + 1) * 64 + lane], c = ep[(k * 3 + 2) * 64 + lane];
                    const double d = x - a;
                    const double en = c - (d * d) * b;
                    const double eu = (x >= a && x <= b) ? c : NEGINF;
                    em = kind == 1 ? en : (kind == 2 ? eu : NEGINF);
                    if (x != x) em = kind ? 0.0 : NEGINF;          // missing observation: log-probability 0 under every distribution
                }
                nv[k] = best[k] + em;
                const Pay bc = bcnt[k];
                if constexpr (MARK) nc[k] = bc + madd[k];          // the three counters of G2_MARK_* in one 64-bit addition
                else nc[k] = (k == B0 || k == B1) ? bc + einc[k] : bc;          // counted states sit in the slots of the broadcast sources (checked when the image is built)
            };
            using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
            using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
            // silent states of this time step: De from lane - 1's new Io, Mo and the broadcast source B1; Do from the lane's own new Ie, Me; then the chains
            double y[2]; Pay yc[2];
            double gather01;      // what the even delete slot's two gather columns gave, before the broadcast source and the chain
            {
                double nI, nM, nB; Pay cI, cM, cB;      // lane - 1's new Io, Mo; the new value of B1
                if constexpr (LX) {
                  {
                    tour_me(); tour_mo(); tour_ie(); tour_io(); finish(K0{}); finish(K1{}); finish(K2{}); finish(K3{});
                    xstore(0, nv[1], nc[1]); xstore(1, nv[3], nc[3]);
                    if constexpr (LX2) { xstore(3, nv[0], nc[0]); if constexpr (!ODD) xstore(4, nv[2], nc[2]); }
                    VIT_FENCE();
                    xload(0, nM, cM); xload(1, nI, cI);
                    rMo = nM; rIo = nI; qMo = cM; qIo = cI;          // ... which are next step's shifted previous values
                    if constexpr (LX2) {
                        xload_at(ODD ? 1 : 4, bc1_lane, nB, cB);
                        xload(3, rMe, qMe); xload_at(ODD ? 0 : 3, bc0_lane, rB0, qB0);      // for the next time step
                    } else { nB = readlane_f64(nv[B1], bc1_lane); cB = readlane_pay(nc[B1], bc1_lane); }
                  }
                } else {
                    tour_me(); tour_mo(); tour_ie(); tour_io(); finish(K0{}); finish(K1{}); finish(K2{}); finish(K3{});
                    nI = dpp_shr1_f64(nv[3]); nM = dpp_shr1_f64(nv[1]); cI = shr1_pay(nc[3]); cM = shr1_pay(nc[1]);
                    nB = readlane_f64(nv[B1], bc1_lane); cB = readlane_pay(nc[B1], bc1_lane);
                }
                const double tI = nI + sg0[0], tM = nM + sg0[1], tB = nB + sg0[2];
                const bool gt = tM > tI;
                const double v01 = __builtin_fmax(tI, tM); const Pay c01 = gt ? cM : cI;
                const bool gb = tB > v01;
                y[0] = __builtin_fmax(v01, tB); yc[0] = gb ? cB : c01;
                gather01 = v01;
            }
            {
                const double tI = nv[2] + sg1[0], tM = nv[0] + sg1[1];
                const bool gt = tM > tI;
                y[1] = __builtin_fmax(tI, tM); yc[1] = gt ? nc[0] : nc[2];
            }
            chain_sweeps(y, yc);
            // the broadcast source wins only on '>' and the chain only raises a value: the slot kept a gather column's value iff it still equals it
            front_won = __builtin_amdgcn_ballot_w64(y[0] == gather01) & hub_mask;
            if constexpr (LX) {
                VIT_FENCE();
                xstore(2, y[1], yc[1]);
                VIT_FENCE();
                xload(2, rDo, qDo);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { pv[k] = nv[k]; pc[k] = nc[k]; }
            dv[0] = y[0]; dv[1] = y[1]; dc[0] = yc[0]; dc[1] = yc[1];
        };

        auto run_window = [&](auto fast_c, auto odd_c) {
            double xchunk = 0.0;
            for (int64_t t0 = 0; t0 < T; t0 += 64) {
                {
                    const int64_t idx = t0 + lane;
                    double xv = 0.0;
                    if (idx < T) {
                        if (tk.src_kind == VIT_SRC_F64) xv = reinterpret_cast<const double*>(tk.sig)[idx];
                        else {
                            double sv = tk.src_kind == VIT_SRC_I16_AFFINE ? (double)reinterpret_cast<const int16_t*>(tk.sig)[idx]
                                                                          : reinterpret_cast<const double*>(tk.sig)[idx];
                            sv = (sv - tk.c1) / tk.h1;
                            sv = sv * tk.h2 + tk.c2;
                            sv = sv < tk.lo ? tk.lo : sv;          // np.clip
                            sv = sv > tk.hi ? tk.hi : sv;
                            xv = sv;
                        }
                    }
                    xchunk = xv;
                }
                const int send = (int)((T - t0) < 64 ? (T - t0) : 64);
                for (int s0 = 0; s0 < send; ++s0) step(fast_c, odd_c, readlane_f64(xchunk, s0), t0 + s0);
            }
        };
        const bool fast_u = __builtin_amdgcn_readfirstlane((int)fast_em) != 0, odd_u = __builtin_amdgcn_readfirstlane((int)odd_model) != 0;
        if (fast_u) { if (odd_u) run_window(std::true_type{}, std::true_type{}); else run_window(std::true_type{}, std::false_type{}); }
        else { if (odd_u) run_window(std::false_type{}, std::true_type{}); else run_window(std::false_type{}, std::false_type{}); }

        double lp; Pay fc;
        if (__builtin_amdgcn_readfirstlane(end_slot)) { lp = readlane_f64(dv[1], end_lane); fc = readlane_pay(dc[1], end_lane); }
        else { lp = readlane_f64(dv[0], end_lane); fc = readlane_pay(dc[0], end_lane); }
        VitResult r; r.logp = lp; r.status = (lp > NEGINF) ? 0 : 1; r.pad_ = 0;
        r.dbg[0] = r.dbg[1] = r.dbg[2] = r.dbg[3] = 0;
        if constexpr (MARK) {
            const uint64_t f = (uint64_t)fc;
            const int64_t tagged = (int64_t)(f & 0x1FFFFFu), behind = (int64_t)(f >> G2_MARK_BEHIND_SHIFT);
            r.counted = (lp > NEGINF) ? (int64_t)((f >> G2_MARK_COUNT_SHIFT) & 0x3FFFFFu) : 0;
            // the same two numbers the lane-layout mark kernels report: time (1-based) of the first repeat-section emission and of
            // the first emission after the section, 0 = none
            r.dbg[0] = tagged > 0 ? (uint32_t)(T - behind - tagged + 1) : 0u;
            r.dbg[1] = (tagged > 0 && behind > 0) ? (uint32_t)(T - behind + 1) : 0u;
            if (tk.T >= VIT_MARK_T_MAX) r.status = 2;
        } else {
            r.counted = (lp > NEGINF) ? (int64_t)fc : 0;
        }
        results[ti] = r;     // every lane stores the same value
    }
}

static int launch_viterbi_g2(hipStream_t stream, const VitTask* tasks, VitResult* results, int n_tasks, int* queue, int n_cu, int want_bp, const int* order, int waves_hint)
{
    if (want_bp != 0 && want_bp != 2) return 2;
    // one workgroup per CU: eight waves (two per SIMD, 2 x 192 VGPRs) when the launch has the GPU to itself; four (one per SIMD) when it
    // shares the CUs with the next sub-batch's flank-alignment kernels, whose waves then find 320 VGPRs per SIMD instead of 128
    // (gpurun_out/r6i: 175 against 180 ms per step of 4096 reads; alone, four waves take 83 ms against 64)
    int nw = waves_hint == 4 ? 4 : 8, lx = 2;
    if (const char* e = strq::opt("STRQ_VIT_G2_WAVES")) { const int v = atoi(e); if (v == 12 || v == 8 || v == 4) nw = v; }      // experiments
    if (const char* e = strq::opt("STRQ_VIT_G2_LDS")) { const int v = atoi(e); if (v >= 0 && v <= 2) lx = v; }
    const dim3 grid(n_cu), block(64 * nw);
    const size_t lds = lx ? (size_t)nw * G2_LDS_WAVE_BYTES : 0;
#define G2_GO2(MK_, W_, LX_) hipLaunchKernelGGL((viterbi_g2_kernel<MK_, W_, LX_>), grid, block, lds, stream, tasks, results, n_tasks, queue, order)
#define G2_GO(MK_, W_) do { if (lx == 2) G2_GO2(MK_, W_, 2); else if (lx == 1) G2_GO2(MK_, W_, 1); else G2_GO2(MK_, W_, 0); } while (0)
    if (want_bp == 2) { if (nw == 12) G2_GO(true, 12); else if (nw == 4) G2_GO(true, 4); else G2_GO(true, 8); }
    else { if (nw == 12) G2_GO(false, 12); else if (nw == 4) G2_GO(false, 4); else G2_GO(false, 8); }
#undef G2_GO
#undef G2_GO2
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Any baked model (up to VIT_CSR_MAX_STATES states, any in-degree): the recurrence of the oracle written down as it stands.
// One workgroup of 256 threads per window, one 16-byte {value, payload} cell per state in two LDS buffers; emitting states
// are dealt to the threads, every state walks its in-edges in baked order with a strict '>' (the first of equal
// candidates wins, like the lane kernels and pomegranate); silent states go level by level -- a state's level is the
// length of its longest chain of silent predecessors, so the states of one level only read emitting cells and lower
// levels -- with a workgroup barrier in between.  Not a throughput path (a flank profile has ~50 levels: ~100 barriers
// per time step); it exists so that a target whose HMM exceeds the lane layouts (repeat units beyond ~50 nt, states with
// more than eight in-edges) runs instead of being refused -- the reference takes whatever pomegranate takes
// (scripts/STRique.py:553-579).  MODE 0: count, 1: back-pointers (predecessor state per (time step, state)), 2: MARK.
template <int MODE>
__global__ void __launch_bounds__(256)
viterbi_csr_kernel(const VitTask* __restrict__ tasks, VitResult* __restrict__ results,
                   int n_tasks, int* __restrict__ queue, const int* __restrict__ order)
{
    extern __shared__ double lds_d[];
    __shared__ int next_task;
    struct alignas(16) Cell { double v; uint64_t c; };
    const double NEGINF = -__builtin_inf();
    constexpr bool MARK = MODE == 2, BP = MODE == 1;
    auto count_add = [](uint64_t v, int inc) -> uint64_t { return (v & 0xFFFFFFFF00000000ull) | (uint32_t)((uint32_t)v + (uint32_t)inc); };
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) next_task = atomicAdd(queue, 1);
        __syncthreads();
        const int tq = next_task;
        if (tq >= n_tasks) break;
        const int ti = order ? order[tq] : tq;
        const VitTask tk = tasks[ti];
        const VitModel& M = *tk.model;
        const int n = M.n_states, ne = M.n_emit, start = M.start, nlev = M.n_levels;
        Cell* buf0 = reinterpret_cast<Cell*>(lds_d); Cell* buf1 = buf0 + n;
        const int32_t* in_ptr = M.csr_in_ptr; const int32_t* in_src = M.csr_in_src; const double* in_lp = M.csr_in_logp;
        for (int i = threadIdx.x; i < 2 * n; i += 256) { buf0[i].v = NEGINF; buf0[i].c = 0; }
        __syncthreads();
        if (threadIdx.x == 0) buf0[start].v = 0.0;
        __syncthreads();
        // silent states of one time step, level by level, in `cur`; t = 0 keeps the start state at 0
        auto silent = [&](Cell* cur, bool pin, int64_t trow) {
            for (int lev = 0; lev < nlev; ++lev) {
                for (int i = M.csr_level_ptr[lev] + threadIdx.x; i < M.csr_level_ptr[lev + 1]; i += 256) {
                    const int l = M.csr_level_state[i];
                    if (pin && l == start) { if (BP && tk.bp) tk.bp[(size_t)trow * n + l] = 0xFFFF; continue; }
                    double best = NEGINF; uint64_t bc = 0; int arg = 0xFFFF;
                    for (int e = in_ptr[l]; e < in_ptr[l + 1]; ++e) {
                        const int k = in_src[e];
                        const double c = cur[k].v + in_lp[e];
                        if (c > best) { best = c; bc = cur[k].c; arg = k; }
                    }
                    cur[l].v = best; cur[l].c = count_add(bc, M.count_inc[l]);
                    if (BP && tk.bp) tk.bp[(size_t)trow * n + l] = (uint16_t)arg;
                }
                __syncthreads();
            }
        };
        silent(buf0, true, 0);
        const int64_t T = tk.T;
        for (int64_t t = 0; t < T; ++t) {
            Cell* prev = (t & 1) ? buf1 : buf0; Cell* cur = (t & 1) ? buf0 : buf1;
            double x;
            if (tk.src_kind == VIT_SRC_F64) x = reinterpret_cast<const double*>(tk.sig)[t];
            else {
                double sv = tk.src_kind == VIT_SRC_I16_AFFINE ? (double)reinterpret_cast<const int16_t*>(tk.sig)[t] : reinterpret_cast<const double*>(tk.sig)[t];
                sv = (sv - tk.c1) / tk.h1;
                sv = sv * tk.h2 + tk.c2;
                sv = sv < tk.lo ? tk.lo : sv;
                sv = sv > tk.hi ? tk.hi : sv;
                x = sv;
            }
            const uint32_t tt1 = (uint32_t)(t + 1);
            const uint32_t mark_e_lo = (tt1 & 0xFFFu) << 20, mark_e_hi = tt1 >> 12, mark_l_hi = tt1 << 10;
            (void)mark_e_lo; (void)mark_e_hi; (void)mark_l_hi;
            for (int l = threadIdx.x; l < ne; l += 256) {
                double best = NEGINF; uint64_t bc = 0; int arg = 0xFFFF;
                for (int e = in_ptr[l]; e < in_ptr[l + 1]; ++e) {
                    const int k = in_src[e];
                    const double c = prev[k].v + in_lp[e];
                    if (c > best) { best = c; bc = prev[k].c; arg = k; }
                }
                double em;
                if (x != x) em = 0.0;                                   // missing observation (viterbi_kernel)
                else if (M.csr_kind[l] == 1) { const double d = x - M.csr_a[l]; em = M.csr_c[l] - (d * d) * M.csr_b[l]; }
                else em = (x >= M.csr_a[l] && x <= M.csr_b[l]) ? M.csr_c[l] : NEGINF;
                uint64_t nc;
                if constexpr (MARK) {
                    const bool etag = M.state_tag[l] == 1;
                    uint32_t lo = (uint32_t)bc + (uint32_t)M.count_inc[l], hi = (uint32_t)(bc >> 32);
                    const bool entered = ((lo >> 20) | (hi & 0x3FFu)) != 0, left = (hi >> 10) != 0;
                    const bool set_e = etag && !entered, set_l = !etag && entered && !left;
                    lo |= set_e ? mark_e_lo : 0u;
                    hi |= set_e ? mark_e_hi : (set_l ? mark_l_hi : 0u);
                    nc = ((uint64_t)hi << 32) | lo;
                } else nc = count_add(bc, M.count_inc[l]);
                cur[l].v = best + em; cur[l].c = nc;
                if (BP && tk.bp) tk.bp[(size_t)(t + 1) * n + l] = (uint16_t)arg;
            }
            __syncthreads();
            silent(cur, false, t + 1);
        }
        if (threadIdx.x == 0) {
            const Cell fin = ((T & 1) ? buf1 : buf0)[M.end];
            const double lp = fin.v;
            VitResult r; r.logp = lp; r.status = (lp > NEGINF) ? 0 : 1; r.pad_ = 0;
            r.dbg[0] = r.dbg[1] = r.dbg[2] = r.dbg[3] = 0;
            if constexpr (MARK) {
                const uint32_t plo = (uint32_t)fin.c, phi = (uint32_t)(fin.c >> 32);
                r.counted = (lp > NEGINF) ? (int64_t)(plo & 0xFFFFFu) : 0;
                r.dbg[0] = (plo >> 20) | ((phi & 0x3FFu) << 12);
                r.dbg[1] = phi >> 10;
                if (tk.T >= VIT_MARK_T_MAX) r.status = 2;
            } else r.counted = (lp > NEGINF) ? (int64_t)(uint32_t)fin.c : 0;
            results[ti] = r;
        }
    }
}

static int launch_viterbi_csr(hipStream_t stream, int max_cells, const VitTask* tasks, VitResult* results, int n_tasks,
                              int* queue, int n_cu, int want_bp, const int* order)
{
    if (max_cells - 1 > VIT_CSR_MAX_STATES || want_bp == 3) return 2;
    const size_t lds = (size_t)2 * (size_t)(max_cells - 1) * 16;
    int per_cu = (int)((160 * 1024 - 64) / (lds ? lds : 1)); if (per_cu > 4) per_cu = 4; if (per_cu < 1) per_cu = 1;
    const dim3 grid(per_cu * n_cu), block(256);
#define VIT_CSR_GO(MODE_)                                                                                                 \
    do {                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)viterbi_csr_kernel<MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((viterbi_csr_kernel<MODE_>), grid, block, lds, stream, tasks, results, n_tasks, queue, order);  \
    } while (0)
    if (want_bp == 2) VIT_CSR_GO(2); else if (want_bp == 1) VIT_CSR_GO(1); else VIT_CSR_GO(0);
#undef VIT_CSR_GO
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// order[] = task indices by descending T (bitonic sort in LDS, one workgroup; n <= 8192)
__global__ void __launch_bounds__(1024)
vit_sort_kernel(const VitTask* __restrict__ tasks, int n, int* __restrict__ order)
{
    __shared__ long long key[8192];
    int np = 1; while (np < n) np <<= 1;
    for (int i = threadIdx.x; i < np; i += blockDim.x)
        key[i] = i < n ? (((long long)(0x7fffffff - (int)(tasks[i].T > 0x7fffffff ? 0x7fffffff : tasks[i].T))) << 32) | (unsigned)i : 0x7fffffffffffffffll;
    __syncthreads();
    for (int k = 2; k <= np; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const long long a = key[i], b = key[l];
                    if ((a > b) == up) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += blockDim.x) order[i] = (int)(key[i] & 0xffffffff);
}

int launch_vit_sort(hipStream_t stream, const VitTask* tasks, int n, int* order)
{
    if (n <= 0) return 0;
    if (n > 8192) return 2;
    hipLaunchKernelGGL(vit_sort_kernel, dim3(1), dim3(1024), 0, stream, tasks, n, order);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// One wave per task: follow the back-pointers from (T, end) and write the emitting state of every
// observation.  The walk is a serial pointer chase, so the rows it is about to visit are staged
// through LDS in blocks (coalesced dword loads) and every hop costs two LDS reads instead of a
// dependent global load; all lanes walk in lock-step (broadcast reads), the finished stretch of the
// path is written back coalesced.
#define VIT_TB_ROWS_U16 6144      // LDS staging per wave: back-pointer rows
#define VIT_TB_CELLS 1024         // cell -> state map
#define VIT_TB_MAXROWS 256
__global__ void __launch_bounds__(256)
vit_traceback_kernel(const VitTask* __restrict__ tasks, const VitResult* __restrict__ results,
                     int32_t* const* __restrict__ paths, int n_tasks)
{
    __shared__ uint32_t rows_all[4][VIT_TB_ROWS_U16 / 2 + 2];
    __shared__ uint16_t cs_all[4][VIT_TB_CELLS];
    __shared__ int32_t pb_all[4][VIT_TB_MAXROWS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n_tasks) return;
    const VitTask& tk = tasks[i];
    const VitModel& M = *tk.model;
    if (results[i].status != 0 || !tk.bp || !paths[i]) return;
    const int n = M.n_states, ne = M.n_emit, start = M.start, ncell = M.n_cells;
    const bool ident = M.csr != 0;          // viterbi_csr_kernel writes predecessor states, not cells
    if (n > VIT_TB_ROWS_U16 || (!ident && ncell > VIT_TB_CELLS)) return;      // cannot happen with the compiled kernel shapes / the state limit of the csr kernel
    uint32_t* rows32 = rows_all[wave]; const uint16_t* rows = reinterpret_cast<const uint16_t*>(rows32);
    uint16_t* cs = cs_all[wave]; int32_t* pb = pb_all[wave];
    if (!ident) for (int c = lane; c < ncell; c += 64) { const int st = M.cell_state[c]; cs[c] = (uint16_t)(st < 0 ? 0xFFFF : st); }
    int RB = (VIT_TB_ROWS_U16 - 2) / n; if (RB > VIT_TB_MAXROWS) RB = VIT_TB_MAXROWS;
    int32_t* path = paths[i];
    int64_t t = tk.T; int l = M.end;
    int guard = 0; bool bad = false;
    while (!(t == 0 && l == start) && !bad) {
        const int64_t tb = t - RB + 1 > 0 ? t - RB + 1 : 0;          // stage rows tb .. t
        const int nrow = (int)(t - tb + 1);
        const uintptr_t A = reinterpret_cast<uintptr_t>(tk.bp + (size_t)tb * n);
        const uintptr_t A4 = A & ~(uintptr_t)3;
        const int shift = (int)((A - A4) >> 1);
        const int ndw = (nrow * n + shift + 1) / 2;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (int x = lane; x < ndw; x += 64) rows32[x] = reinterpret_cast<const uint32_t*>(A4)[x];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int64_t t_hi = t;                                       // path entries tb-1 .. t_hi-1 may be produced here
        while (t >= tb && !(t == 0 && l == start)) {
            const int pcell = rows[(int)(t - tb) * n + l + shift];
            const int prev = ident ? (pcell < n ? pcell : 0xFFFF) : (pcell < ncell ? cs[pcell] : 0xFFFF);
            if (prev == 0xFFFF || prev >= n) { bad = true; break; }
            if (l < ne) { if (lane == 0) pb[(int)(t - tb)] = l; --t; guard = 0; }      // path[t-1] <- l, kept at slot (t-1) - (tb-1)
            else if (++guard > n) { bad = true; break; }
            l = prev;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // entries for observations t .. t_hi-1 (0-based path index = time - 1 -> slot time - tb)
        for (int64_t tt = t + 1 + lane; tt <= t_hi; tt += 64) path[tt - 1] = pb[(int)(tt - tb)];
    }
}

// kernel shapes: (EPL, SPL, DE_HI, DE_LO, DS)
static int vit_shape_base(const VitModel& mh)
{
    const int e = mh.epl, s = mh.spl;
    int hi = 0, lo = 0, ds = 0;
    for (int i = 0; i < e; ++i) { if (i < (e + 1) / 2) hi = hi > mh.e_deg[i] ? hi : mh.e_deg[i]; else lo = lo > mh.e_deg[i] ? lo : mh.e_deg[i]; }
    for (int i = 0; i < s; ++i) ds = ds > mh.s_deg[i] ? ds : mh.s_deg[i];
    // flanked-repeat models: six-edge states in slot 0, two in-edges per delete state besides its chain
    // (their kernels leave the count increments of silent states out: STRique counts the emitting dummy states, STRique.py:341-342,375-377)
    if (e <= 4 && s <= 2 && e > 2 && mh.e_deg[0] <= 6 && mh.e_deg[1] <= 5 && lo <= 3 && ds <= 2 && !mh.silent_counted) {
        bool flat = true;
        for (int i = (e + 1) / 2; i < e; ++i) flat = flat && mh.e_flat[i];
        if (flat && e == 4) return 7;          // ... and only uniform emissions (the inserts) in the last two slots
        return 5;
    }
    if (e <= 4 && s <= 2 && e > 2 && hi <= 6 && lo <= 3 && ds <= 3) return 0;      // flanked-repeat models
    if (e <= 1 && s <= 1 && hi <= 5 && ds <= 1) return 6;                          // STRique's dual base / mCpG model: 26 + 2 states, at most five in-edges
    if (e <= 1 && s <= 1 && hi <= 8 && ds <= 4) return 1;                          // modification models
    if (e <= 2 && s <= 2 && hi <= 8 && lo <= 8 && ds <= 4) return 2;
    if (e <= 4 && s <= 4 && hi <= 8 && lo <= 8 && ds <= 8) return 3;
    if (e <= 8 && s <= 4 && hi <= 8 && lo <= 8 && ds <= 8) return 4;
    return -1;
}

// silent slots per lane of a kernel shape (the template's SPL)
int vit_shape_silent_slots(int shape)
{
    static const int spl[8] = {2, 1, 2, 4, 4, 2, 1, 2};
    const int b = shape & ~VIT_SHAPE_SS;
    return b >= 0 && b < 8 ? spl[b] : 0;
}

int vit_shape_for(const VitModel& mh, int want_bp)
{
    const bool no_g2 = strq::opt("STRQ_VIT_NO_G2") != nullptr;      // A/B: the lane layout for every mode
    if (mh.g2 && !no_g2 && (want_bp == 0 || (want_bp == 2 && mh.g2_mark))) return VIT_SHAPE_G2;          // either parity of the chain: decided per window inside the kernel
    return vit_shape_of(mh);
}

int vit_shape_of(const VitModel& mh)
{
    if (mh.csr) return VIT_SHAPE_CSR;
    const int b = vit_shape_base(mh);
    return b < 0 ? b : (b | (mh.single_stage ? VIT_SHAPE_SS : 0));
}

template <int E_, int S_, int H_, int L_, int D_>
static int vit_launch_shape(hipStream_t stream, int max_cells, const VitTask* tasks, VitResult* results, int n_tasks,
                            int* queue, int n_cu, int want_bp, int single_stage, const int* order)
{
    if (max_cells > VitLds<E_, S_>::TRASH) return 3;
    // per wave: two buffers of 16-byte {value, count} cells; waves of a block are independent
    int nw = VitLds<E_, S_>::WAVES;
    if (const char* e = strq::opt("STRQ_VIT_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= nw) nw = v; }      // experiments: fewer waves per CU
    const size_t lds = (size_t)nw * 2 * VitLds<E_, S_>::BUF;
    const dim3 grid(n_cu), block(64 * nw);
#define VIT_GO(BP_, SS_, MK_, HB_)                                                                                        \
    do {                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)viterbi_kernel<E_, S_, H_, L_, D_, BP_, SS_, MK_, HB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((viterbi_kernel<E_, S_, H_, L_, D_, BP_, SS_, MK_, HB_>), grid, block, lds, stream, tasks, results, n_tasks, queue, order);       \
    } while (0)
    // want_bp: 0 = count only, 1 = back-pointers, 2 = repeat-section marks carried along the best path, 3 = hub records
    if (want_bp == 3) {
        if constexpr (E_ <= 2) { if (single_stage) VIT_GO(false, true, false, true); else VIT_GO(false, false, false, true); }
        else return 2;
    }
    else if (want_bp == 2) { if (single_stage) VIT_GO(false, true, true, false); else VIT_GO(false, false, true, false); }
    else if (want_bp) { if (single_stage) VIT_GO(true, true, false, false); else VIT_GO(true, false, false, false); }
    else { if (single_stage) VIT_GO(false, true, false, false); else VIT_GO(false, false, false, false); }
#undef VIT_GO
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// `shape` as returned by vit_shape_of: kernel shape | VIT_SHAPE_SS for single-stage models
int launch_viterbi(hipStream_t stream, int shape, int max_cells, const VitTask* tasks, VitResult* results,
                   int n_tasks, int* queue, int n_cu, int want_bp, const int* order, int waves_hint)
{
    const int ss = (shape & VIT_SHAPE_SS) ? 1 : 0;
    if ((shape & ~VIT_SHAPE_SS) == VIT_SHAPE_CSR) return launch_viterbi_csr(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, order);
    if ((shape & ~VIT_SHAPE_SS) == VIT_SHAPE_G2)
        return launch_viterbi_g2(stream, tasks, results, n_tasks, queue, n_cu, want_bp, order, waves_hint);
    switch (shape & ~VIT_SHAPE_SS) {
        case 0: return vit_launch_shape<4, 2, 6, 3, 3>(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, ss, order);
        case 1: return vit_launch_shape<1, 1, 8, 8, 4>(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, ss, order);
        case 2: return vit_launch_shape<2, 2, 8, 8, 4>(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, ss, order);
        case 3: return vit_launch_shape<4, 4, 8, 8, 8>(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, ss, order);
        case 4: return vit_launch_shape<8, 4, 8, 8, 8>(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, ss, order);
        case 7: return vit_launch_shape<4, 2, 65, 13, 2>(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, ss, order);
        case 6: return vit_launch_shape<1, 1, 5, 5, 1>(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, ss, order);
        case 5: return vit_launch_shape<4, 2, 65, 3, 2>(stream, max_cells, tasks, results, n_tasks, queue, n_cu, want_bp, ss, order);
        default: return 2;
    }
}

int launch_vit_traceback(hipStream_t stream, const VitTask* tasks, const VitResult* results,
                         int32_t* const* paths, int n_tasks)
{
    if (n_tasks <= 0) return 0;
    hipLaunchKernelGGL(vit_traceback_kernel, dim3((n_tasks + 3) / 4), dim3(256), 0, stream, tasks, results, paths, n_tasks);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq

#ifdef STRQ_VIT_TIMING
// diagnostic build only: read and clear the per-phase cycle sums (tools/vit_timing.py)
extern "C" int strq_debug_vit_timing(unsigned long long out[8])
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(strq::vit_timing_acc), 64) != hipSuccess) return 2;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(strq::vit_timing_acc), z, 64) == hipSuccess ? 0 : 2;
}
#endif
