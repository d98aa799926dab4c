// Upper-bound screen of the flank alignment on gfx950 (MI355X).
//
// The reference aligns each flank against the WHOLE read (scripts/STRique.py:538-548 -> src/align_raw.h:106-158): 870 rows x
// ~375 k columns x 2 flanks per 50 kb read, of which the optimal path crosses ~1 k columns.  This kernel does not replace that
// DP -- align_kernels.hip still computes it, in float32, bit for bit -- it tells it where to look: a cheaper DP over the same
// matrix whose last-row values are PROVEN upper bounds of the float32 ones, reported as one maximum per 128 columns.  The
// host then runs the exact DP over the few column windows whose bound reaches the best bound minus the (known) slack of the
// bound, and every other column is excluded by inequality, not by heuristic:
//
//   * scores are rounded UP to multiples of 1/sc (sc = 1024 for STRique's parameters and reads below ~1.9 M samples) and the
//     DP runs in 32-bit integers.  A max-plus DP is monotone in its scores, and integer arithmetic does not round: the
//     result bounds the real-arithmetic DP of the float32 table from above by less than m / sc (one rounding per diagonal
//     step), and that one is within `slack` (16 score units for STRique's parameters) of the float32 DP;
//   * collapsed recurrence (open == extend, as STRique configures): S = max(diag + s, left + e_h, up + e_v).  Stored is
//     T = S + i |e_v| + j |e_h|: both gap terms vanish from the recurrence, T = max3(diag + s'', left, up) with
//     s'' = s + |e_v| + |e_h| baked into the (16-bit) table -- one v_add_u32 and one max3 per cell where the float32
//     kernel needs three adds and a max3.  The transformation is exact in integers; in float32 it would change roundings,
//     which is why the exact kernel cannot use it;
//   * rows below the flank (the rest of lane (m-1)/R) score 0 <= s'': with T monotone along rows and columns they copy the
//     last flank row, so its value is read from the lane's LAST register whatever m is;
//   * pieces of a read start cold like the exact kernel's (align_kernels.h): their values are exact (as bounds) wherever they
//     reach the score the overlap was sized for, and below it otherwise -- the windows kernel only prunes above that score.
//
// (A first version ran both flanks of a read in the halves of 16-bit pairs -- v_pk_add_u16 / v_pk_max_u16, 141 instructions per
// step for two alignments.  gfx950 issues the packed 16-bit integer instructions at half rate, like v_pk_add_f32: 96 ms per
// 2048 reads where the float32 pass takes 130; tools/ubench_pk16.hip)
#include "strq_opt.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <cmath>
#include "screen_kernels.h"

namespace strq {

namespace {

constexpr int R = STRQ_SCREEN_R, S = STRQ_SCREEN_S, SEG = STRQ_SCREEN_SEG;
constexpr int G = 2, C = 3;              // gcd(R, S); k-mer classes a lane can touch
static_assert(R == 14 && S == 6, "class expansion below is written for 14 rows per lane, 6 samples per class");

__device__ __forceinline__ int dpp_shr1(int v, int fill)      // lane l receives lane l - 1; lane 0 keeps `fill`
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ int med3i(int a, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi));
    return r;
}
// Maximum of three stored values.  They are integers in [STRQ_SCREEN_BIAS, 2^31 - 2^24): as bit patterns, positive NORMAL
// float32 numbers, whose order is the order of the integers -- so this is v_max3_f32, which gfx950 issues at the full rate
// (v_max3_i32 / v_max_u32 and the packed 16-bit instructions take about twice as long: tools/ubench_pk16.hip).
#define STRQ_SCREEN_BIAS 0x00800000
#ifndef STRQ_SCREEN_WPE
#define STRQ_SCREEN_WPE 6          // waves per SIMD the kernel is compiled for (caps its registers): six tables of four waves per CU
#endif
__device__ __forceinline__ int max3i(int a, int b, int c)
{
    int r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int sel_mask(int if0, int if1, uint64_t mask)      // wave-uniform lane mask in an SGPR pair
{
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
    return r;
}

struct LaneConst { int lo2[C], hi2[C], off[C]; };

struct Screen {
    const char* lds;
    const LaneConst& lc;
    const uint64_t (&pm)[3];
    const int lane, n, hh;
    int T[R], SbotA, upS, potB, cmax;
    int qq;                            // levels (x2) of the two columns of the step about to run
    int scA[C], scB[C];                // their class scores, fetched one step ahead

    __device__ __forceinline__ void fetch(int q2, int (&sc)[C])
    {
#pragma unroll
        for (int c = 0; c < C; ++c)
            sc[c] = *reinterpret_cast<const uint16_t*>(lds + med3i(q2, lc.lo2[c], lc.hi2[c]) + lc.off[c]);
    }
    __device__ __forceinline__ void advance(int qsrc, int snext, int& qn, int (&nA)[C], int (&nB)[C])
    {
        qn = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(qsrc, snext), qq, 0x138, 0xF, 0xF, false);
        fetch(qn & 0xffff, nA);
        fetch((int)((unsigned)qn >> 16), nB);
    }
    __device__ __forceinline__ void prime(int qcur)
    {
        qq = 0;
        int qn, nA[C], nB[C];
        advance(qcur, 0, qn, nA, nB);
        qq = qn;
#pragma unroll
        for (int c = 0; c < C; ++c) { scA[c] = nA[c]; scB[c] = nB[c]; }
    }
    // rows of the lane <- class scores (row r of a lane with phase p belongs to class slot (p + r) / S)
    __device__ __forceinline__ void expand(const int (&sc)[C], int (&rs)[R])
    {
#pragma unroll
        for (int r = 0; r < R; r += G) {
            const int base = r / S, x = (r % S) / G;
            int v;
            if (x == 0 || base + 1 >= C) v = sc[base < C ? base : C - 1];
            else v = sel_mask(sc[base], sc[base + 1], pm[x]);
#pragma unroll
            for (int g = 0; g < G && r + g < R; ++g) rs[r + g] = v;
        }
    }

    // One step = the lane's two columns.  Nothing is copied from one step to the next (round 5, as in the coarse screen below): column B
    // of row r - 1 is computed right after column A of row r and takes the register of the old T[r - 1], and the class scores of the
    // NEXT step are fetched into the registers of this step's as soon as the rows' scores have been expanded from them.
    template <bool PRED>
    __device__ __forceinline__ void step(int t, int qsrc, int snext)
    {
        const int qn = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(qsrc, snext), qq, 0x138, 0xF, 0xF, false);
        // potentials j |e_h| of this step's two columns: what the free top row holds there
        const int potA = potB + hh, potBn = potA + hh;
        // the row above the lane: lane - 1's bottom cells (lane 0: the top row)
        const int upA = dpp_shr1(SbotA, potA);
        const int upB = dpp_shr1(T[R - 1], potBn);
        int rsA[R], rsB[R];
        expand(scA, rsA);
        expand(scB, rsB);
        fetch(qn & 0xffff, scA);
        fetch((int)((unsigned)qn >> 16), scB);
        bool act = true, okB = true;
        if constexpr (PRED) { const int jB = 2 * (t - lane), jA = jB - 1; act = (jA >= 1) && (jA <= n); okB = jB <= n; }
        if (act) {
            int ta_prev = upA, ta_prev2 = upS, tb_prev2 = upB;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int diagA = r == 0 ? upS : T[r - 1];
                const int ta = max3i(diagA + rsA[r], T[r], ta_prev);
                if (r > 0) {
                    const int tb = max3i(ta_prev2 + rsB[r - 1], ta_prev, tb_prev2);
                    T[r - 1] = tb; tb_prev2 = tb;
                }
                ta_prev2 = ta_prev; ta_prev = ta;
            }
            const int tb = max3i(ta_prev2 + rsB[R - 1], ta_prev, tb_prev2);
            T[R - 1] = tb;
            SbotA = ta_prev;
            upS = upB;
            // last row (this lane's last register, see the header) minus the column potential: S[m][j] * sc + m |e_v| (+ bias)
            const int cA = ta_prev - potA + STRQ_SCREEN_BIAS, cB = okB ? tb - potBn + STRQ_SCREEN_BIAS : cA;
            cmax = max3i(cmax, cA, cB);
        }
        potB = potBn;
        qq = qn;
    }
};

__device__ __forceinline__ int load_chunk(const ScreenTask& tk, int chunk, int lane)
{
    const int idx = (chunk * 64 + lane) * 2;
    int a = 0, b = 0;
    if (idx < tk.n) a = tk.levels[idx];
    if (idx + 1 < tk.n) b = tk.levels[idx + 1];
    return (a * 2) | ((b * 2) << 16);
}

}  // namespace

size_t screen_lds_bytes(int tsize)
{
    // the table as 16-bit entries, padded to a dword, and one zero entry (the rows below the flank)
    return (size_t)((tsize + 1) & ~1) * 2 + 4;
}

__global__ void __attribute__((amdgpu_flat_work_group_size(64 * STRQ_SCREEN_SEG, 64 * STRQ_SCREEN_SEG), amdgpu_waves_per_eu(STRQ_SCREEN_WPE, STRQ_SCREEN_WPE)))
align_screen_kernel(const ScreenTask* __restrict__ tasks, int n_groups, int* __restrict__ queue, ScreenParams sp)
{
    extern __shared__ uint32_t lds_all[];
    __shared__ int next_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* ldsb = reinterpret_cast<const char*>(lds_all);
    for (;;) {
        __syncthreads();                       // every wave is done with the previous table
        if (threadIdx.x == 0) next_group = atomicAdd(queue, 1);
        __syncthreads();
        const int gi = __builtin_amdgcn_readfirstlane(next_group);
        if (gi >= n_groups) break;
        const ScreenTask& t0 = tasks[(size_t)gi * SEG];
        const int ts = t0.tsize, zero_idx = (ts + 1) & ~1;      // in 16-bit entries
        {
            // ceil(s * sc) + cadd, as 16-bit entries (s * sc is exact: sc is a power of two; the table holds 0 <= s <= dist_offset)
            uint16_t* dst = reinterpret_cast<uint16_t*>(lds_all);
            const float scf = (float)sp.sc;
            for (int i = threadIdx.x; i < ts; i += 64 * SEG) dst[i] = (uint16_t)((int)ceilf(t0.table[i] * scf) + sp.cadd);
            if (threadIdx.x == 0) { dst[zero_idx] = 0; dst[zero_idx + 1] = 0; }
        }
        __syncthreads();
        const ScreenTask& tk = tasks[(size_t)gi * SEG + wave];
        if (tk.n <= 0) continue;               // unused piece of a short read (wave-uniform)

        LaneConst lc; uint64_t pm[3];
        {
            const int row0 = lane * R, kbase = row0 / S, phase = row0 % S;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int k = kbase + c;
                if (k < tk.k) {
                    const uint32_t d = (uint32_t)tk.band_lo[k];
                    const int lo = (int)(d & 255u), w1 = (int)((d >> 8) & 255u), off = (int)(d >> 16);
                    lc.lo2[c] = lo * 2; lc.hi2[c] = (lo + w1) * 2;
                    lc.off[c] = (off - lo) * 2;
                } else {      // below the flank: the zero entry
                    lc.lo2[c] = 0; lc.hi2[c] = 0; lc.off[c] = zero_idx * 2;
                }
            }
            pm[0] = 0;
#pragma unroll
            for (int x = 1; x < 3; ++x) {
                const uint64_t b = __ballot(phase >= S - x * G);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
                pm[x] = ((uint64_t)hi << 32) | lo;
            }
        }
        const int lM = (tk.m - 1) / R;
        Screen f{ldsb, lc, pm, lane, tk.n, sp.hh};
        {
            // column 0 (cold start): S[i][0] = i * e_v, i.e. T = 0 (+ bias) in every row; the top row holds the column potential
#pragma unroll
            for (int r = 0; r < R; ++r) f.T[r] = STRQ_SCREEN_BIAS;
            f.SbotA = STRQ_SCREEN_BIAS; f.upS = STRQ_SCREEN_BIAS;
            f.potB = STRQ_SCREEN_BIAS - 2 * lane * sp.hh;       // the lane's column jB = 2 (t - lane) at t = 0 (used once j >= 1)
            f.cmax = STRQ_SCREEN_BIAS;
        }
        const int nsteps = (tk.n + 1) / 2 + 63;
        int qcur = load_chunk(tk, 0, lane);
        f.prime(qcur);
        for (int t0s = 0; t0s < nsteps; t0s += 64) {
            const int qnext = load_chunk(tk, t0s / 64 + 1, lane);
            const bool full = (t0s >= 63) && (2 * (t0s + 64) <= tk.n);
            const int send = nsteps - t0s < 64 ? nsteps - t0s : 64;
            if (full) {
                for (int s = 0; s < 63; ++s) f.template step<false>(t0s + s + 1, qcur, s + 1);
                f.template step<false>(t0s + 64, qnext, 0);
            } else {
                for (int s = 0; s < send; ++s) {
                    const int qsrc = s == 63 ? qnext : qcur, snext = (s + 1) & 63;
                    f.template step<true>(t0s + s + 1, qsrc, snext);
                }
            }
            // the chunk's maximum of the last row (lane lM)
            const int cm = __builtin_amdgcn_readlane(f.cmax, lM) - STRQ_SCREEN_BIAS;
            if (lane == 0) tk.out[t0s / 64] = cm;
            f.cmax = STRQ_SCREEN_BIAS;
            qcur = qnext;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The COARSE screen: merged rows, both flanks of a read in one wave.
//
// The six rows of a k-mer class are identical (generate_signal repeats every level `samples` = 6 times, scripts/STRique.py:
// 186-194).  MERGE = 2, 3 or 6 identical neighbouring rows become ONE DP row whose diagonal step gains all their scores at one
// column: a path of the fine matrix that takes the diagonals of those rows at columns j1 < j2 < ... gains s(j1) + s(j2) + ...
// <= MERGE max s(j), and the merged row may take the best of those columns (rows and columns of the remaining path stay ordered,
// horizontal moves are free in the T potential), so the merged DP's last row bounds the fine one's from above -- by more than the
// fine screen's m / sc (the merged row pays for one column where the rows need MERGE), which is why its candidates are taken with a
// margin (ScreenParams::margin), limited in number, and certified by the exact pass -- with a second look for what the first one
// cannot certify (strq_align_api.hip).
// A lane owns five whole classes = 5 x RPC merged rows (RPC = 6 / MERGE rows per class: no class selects); 145 classes = 29 lanes,
// and the two flanks of a read (prefix / suffix alignment, the same columns) sit in lanes 0 .. 28 and 32 .. 60 of one wave: lane 32
// takes the free top row instead of lane 31's bottom cells, the packed levels travel through all 64 lanes.
// One wave-step = 2 columns x both flanks.
namespace {

constexpr int CPL = STRQ_SCREEN2_CPL, LPF = STRQ_SCREEN2_LPF, LB = STRQ_SCREEN2_LANE_B;

struct Lane2Const { int lo2[CPL], hi2[CPL], off[CPL]; };

template <int RPC>
struct Screen2 {
    static constexpr int R2 = RPC * CPL;
    const char* lds;
    const Lane2Const& lc;
    const uint64_t top_mask;           // lanes 0 and LB: their row above is the free top row
    const int lane, n, hh;
    int T[R2], SbotA, upS, potB, cmax;
    int qq;
    int scA[CPL], scB[CPL];

    __device__ __forceinline__ int score(int q2, int c) const
    {
        return *reinterpret_cast<const uint16_t*>(lds + med3i(q2, lc.lo2[c], lc.hi2[c]) + lc.off[c]);
    }
    __device__ __forceinline__ void prime(int qcur)
    {
        qq = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(qcur, 0), 0, 0x138, 0xF, 0xF, false);
#pragma unroll
        for (int c = 0; c < CPL; ++c) { scA[c] = score(qq & 0xffff, c); scB[c] = score((int)((unsigned)qq >> 16), c); }
    }

    // One step = the lane's two columns.  Written so that nothing is copied from one step to the next: column B of row r - 1 is
    // computed right after column A of row r and takes the register of the old T[r - 1] (dead from there on), and the scores of a
    // class are fetched for the NEXT step as soon as its rows are done, into the registers they just left (the LDS latency hides
    // behind the rest of the step).  99 VALU instructions per step at RPC = 3 (30 cells), of which 47 issue at the half rate
    // (31 v_max3_f32, 10 v_med3_i32, 3 DPP moves, 2 selects, 1 v_readlane): 4.7 cycles each + 2.4 for the others is the step's
    // 357 SIMD cycles (tools/runs_r05: hoisting the additions out of the chain or other waves-per-SIMD settings move it by < 1 %).
    template <bool PRED>
    __device__ __forceinline__ void step(int t, int qsrc, int snext)
    {
        const int qn = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(qsrc, snext), qq, 0x138, 0xF, 0xF, false);
        const int qa = qn & 0xffff, qb = (int)((unsigned)qn >> 16);
        const int potA = potB + hh, potBn = potA + hh;
        // the row above the lane: lane - 1's bottom cells; lanes 0 and LB: the top row (the column potential)
        const int upA = sel_mask(dpp_shr1(SbotA, potA), potA, top_mask);
        const int upB = sel_mask(dpp_shr1(T[R2 - 1], potBn), potBn, top_mask);
        bool act = true, okB = true;
        if constexpr (PRED) { const int jB = 2 * (t - lane), jA = jB - 1; act = (jA >= 1) && (jA <= n); okB = jB <= n; }
        int ta_prev = upA, ta_prev2 = upS;          // column A of rows r - 1, r - 2 (row -1: the row above the lane)
        int tb_prev2 = upB;                         // column B of row r - 2
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            if (act) {
#pragma unroll
                for (int r = RPC * c; r < RPC * c + RPC; ++r) {
                    const int diagA = r == 0 ? upS : T[r - 1];
                    const int ta = max3i(diagA + scA[c], T[r], ta_prev);
                    if (r > 0) {
                        // column B of row r - 1: diag = column A of row r - 2, left = column A of row r - 1, up = column B of row r - 2
                        const int tb = max3i(ta_prev2 + scB[(r - 1) / RPC], ta_prev, tb_prev2);
                        T[r - 1] = tb; tb_prev2 = tb;
                    }
                    ta_prev2 = ta_prev; ta_prev = ta;
                }
                if (c == CPL - 1) {
                    const int tb = max3i(ta_prev2 + scB[c], ta_prev, tb_prev2);
                    T[R2 - 1] = tb;
                    SbotA = ta_prev;
                    upS = upB;
                    const int cA = ta_prev - potA + STRQ_SCREEN_BIAS, cB = okB ? tb - potBn + STRQ_SCREEN_BIAS : cA;
                    cmax = max3i(cmax, cA, cB);
                }
            }
            // the class's scores for the next step (scB[c] serves the last row of class c one row later: fetch B one class behind)
            scA[c] = score(qa, c);
            if (c > 0) scB[c - 1] = score(qb, c - 1);
            if (c == CPL - 1) scB[c] = score(qb, c);
        }
        potB = potBn;
        qq = qn;
    }
};

__device__ __forceinline__ int load_chunk2(const Screen2Task& tk, int chunk, int lane)
{
    const int idx = (chunk * 64 + lane) * 2;
    int a = 0, b = 0;
    if (idx < tk.n) a = tk.levels[idx];
    if (idx + 1 < tk.n) b = tk.levels[idx + 1];
    return (a * 2) | ((b * 2) << 16);
}

template <int RPC>
__device__ __forceinline__ void screen2_body(const Screen2Task* __restrict__ tasks, int n_groups, int* __restrict__ queue, const ScreenParams& sp)
{
    constexpr int R2 = RPC * CPL, MERGE = S / RPC;
    extern __shared__ uint32_t lds_all[];
    __shared__ int next_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* ldsb = reinterpret_cast<const char*>(lds_all);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) next_group = atomicAdd(queue, 1);
        __syncthreads();
        const int gi = __builtin_amdgcn_readfirstlane(next_group);
        if (gi >= n_groups) break;
        const Screen2Task& t0 = tasks[(size_t)gi * SEG];
        int base[2], zero_idx[2];          // in 16-bit entries
        base[0] = 0; zero_idx[0] = (t0.tsize[0] + 1) & ~1;
        base[1] = zero_idx[0] + 2; zero_idx[1] = base[1] + ((t0.tsize[1] + 1) & ~1);
        {
            // MERGE (ceil(s * sc) + hh + v): what the merged row gains on a diagonal step (all its rows' scores and potentials)
            uint16_t* dst = reinterpret_cast<uint16_t*>(lds_all);
            const float scf = (float)sp.sc;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                for (int i = threadIdx.x; i < t0.tsize[f]; i += 64 * SEG) dst[base[f] + i] = (uint16_t)(MERGE * ((int)ceilf(t0.table[f][i] * scf) + sp.cadd));
                if (threadIdx.x == 0) { dst[zero_idx[f]] = 0; dst[zero_idx[f] + 1] = 0; }
            }
        }
        __syncthreads();
        const Screen2Task& tk = tasks[(size_t)gi * SEG + wave];
        if (tk.n <= 0) continue;

        Lane2Const lc;
        const int f = lane >= LB ? 1 : 0, lf = lane - f * LB;
        {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int k = lf * CPL + c;
                if (lf < LPF && k < tk.k[f]) {
                    const uint32_t d = (uint32_t)tk.band_lo[f][k];
                    const int lo = (int)(d & 255u), w1 = (int)((d >> 8) & 255u), off = (int)(d >> 16);
                    lc.lo2[c] = lo * 2; lc.hi2[c] = (lo + w1) * 2;
                    lc.off[c] = (base[f] + off - lo) * 2;
                } else {
                    lc.lo2[c] = 0; lc.hi2[c] = 0; lc.off[c] = zero_idx[f] * 2;
                }
            }
        }
        const uint64_t top_mask = 1ull | (1ull << LB);
        // lanes whose last register holds a flank's last row (rows below a flank score 0 and copy it)
        const int lMa = (tk.k[0] - 1) / CPL, lMb = LB + (tk.k[1] - 1) / CPL;
        Screen2<RPC> s2{ldsb, lc, top_mask, lane, tk.n, sp.hh};
        {
#pragma unroll
            for (int r = 0; r < R2; ++r) s2.T[r] = STRQ_SCREEN_BIAS;
            s2.SbotA = STRQ_SCREEN_BIAS; s2.upS = STRQ_SCREEN_BIAS;
            s2.potB = STRQ_SCREEN_BIAS - 2 * lane * sp.hh;
            s2.cmax = STRQ_SCREEN_BIAS;
        }
        const int nsteps = (tk.n + 1) / 2 + 63;
        int qcur = load_chunk2(tk, 0, lane);
        s2.prime(qcur);
        for (int t0s = 0; t0s < nsteps; t0s += 64) {
            const int qnext = load_chunk2(tk, t0s / 64 + 1, lane);
            const bool full = (t0s >= 63) && (2 * (t0s + 64) <= tk.n);
            const int send = nsteps - t0s < 64 ? nsteps - t0s : 64;
            if (full) {
                for (int s = 0; s < 63; ++s) s2.template step<false>(t0s + s + 1, qcur, s + 1);
                s2.template step<false>(t0s + 64, qnext, 0);
            } else {
                for (int s = 0; s < send; ++s) {
                    const int qsrc = s == 63 ? qnext : qcur, snext = (s + 1) & 63;
                    s2.template step<true>(t0s + s + 1, qsrc, snext);
                }
            }
            const int cma = __builtin_amdgcn_readlane(s2.cmax, lMa) - STRQ_SCREEN_BIAS;
            const int cmb = __builtin_amdgcn_readlane(s2.cmax, lMb) - STRQ_SCREEN_BIAS;
            if (lane == 0) { tk.out[0][t0s / 64] = cma; tk.out[1][t0s / 64] = cmb; }
            s2.cmax = STRQ_SCREEN_BIAS;
            qcur = qnext;
        }
    }
}

}  // namespace

size_t screen2_lds_bytes(int tsize_a, int tsize_b)
{
    // both tables as 16-bit entries, each padded to a dword and followed by a zero pair (the rows below a flank)
    return (size_t)(((tsize_a + 1) & ~1) + 2 + ((tsize_b + 1) & ~1) + 2) * 2;
}

// MERGE = 3: two DP rows per class, ten per lane.  (Rounds 5 measured MERGE 2 and 6 as well -- 114.4 and 96.6 ms per 4096 reads against
// 94.7, with 2.4 % / 14 % of the alignments in the second look against 6 %: DESIGN.md 4.2e -- those instances are gone.)
#define STRQ_SCREEN2_ATTR(WPE_) __attribute__((amdgpu_flat_work_group_size(64 * STRQ_SCREEN_SEG, 64 * STRQ_SCREEN_SEG), amdgpu_waves_per_eu(WPE_, WPE_)))
__global__ void STRQ_SCREEN2_ATTR(6) align_screen3_kernel(const Screen2Task* __restrict__ tasks, int n_groups, int* __restrict__ queue, ScreenParams sp) { screen2_body<2>(tasks, n_groups, queue, sp); }
// ... and no merging at all: the fine screen's bound (one DP row per flank row, 30 per lane) for both flanks of a read in one wave, without the class
// selects of align_screen_kernel -- what the fine screen runs on whenever a sub-batch holds both alignments of its reads (every detect call)
__global__ void STRQ_SCREEN2_ATTR(4) align_screen1_kernel(const Screen2Task* __restrict__ tasks, int n_groups, int* __restrict__ queue, ScreenParams sp) { screen2_body<6>(tasks, n_groups, queue, sp); }

// Candidate windows of every alignment.
// A chunk's value v bounds the float32 last-row values S of its columns: S * sc <= max(v', bound) + slack, v' = v - m * v_gap
// (bound: the cold-start score of the pieces).  The best chunk value vmax is reached by a real path whose float32 score is at
// least (vmax' - m) / sc - slack / sc.  Columns of chunks with max(v', bound) + slack < vmax' - m - slack cannot hold the optimum.
__global__ void __launch_bounds__(256)
screen_windows_kernel(const ScreenTask* __restrict__ tasks, int n_groups, ScreenParams sp,
                      const int32_t* __restrict__ bound_scaled, ScreenWindows* __restrict__ out,
                      const int32_t* __restrict__ list, const int32_t* __restrict__ theta_list)
{
    // one wave per alignment: coalesced reads of the chunk values, lane 0 merges the (few) candidates in column order.
    // list / theta_list (second look of the coarse screen): alignment list[idx] with the candidate threshold theta_list[idx]
    // (chunk units) instead of the best chunk value minus the margin; the result goes to out[idx].
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= n_groups) return;
    const int g = list ? list[idx] : idx;
    const ScreenTask* tg = tasks + (size_t)g * SEG;
    const int m = tg[0].m, lM = tg[0].lane_last;
    const int shift = -m * sp.v;            // v' = v + shift
    int vmax = 0;
    for (int w = 0; w < SEG; ++w) {
        if (tg[w].n <= 0) continue;
        for (int c = lane; c < tg[w].n_chunks; c += 64) { const int v = tg[w].out[c]; vmax = v > vmax ? v : vmax; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(vmax, d, 64); vmax = o > vmax ? o : vmax; }
    ScreenWindows r;
    r.n_win = 0; r.n_cand = 0;
    for (int k = 0; k < STRQ_SCREEN_MAX_WINDOWS; ++k) { r.lo[k] = 0; r.hi[k] = 0; }
    // fine screen: a chunk value is less than m above the exact one (one rounding per diagonal step); coarse screen: sp.margin
    const int drop = (sp.margin > 0 ? sp.margin : m) + 2 * sp.slack;
    int theta = theta_list ? theta_list[idx] : vmax - drop;          // in chunk units (before the shift)
    if (!theta_list && sp.max_cand > 0) {
        // at most max_cand candidate chunks in the first look: the smallest threshold >= theta with no more chunks above it
        // (bisection on the value; an alignment whose bound profile is flat -- a read that holds its flank no better than its
        // background -- would otherwise hand hundreds of chunks to the exact pass; what the raised threshold leaves out is the
        // second look's business: coarse screen, strq_align_api.hip)
        auto count_at = [&](int t) {
            int n = 0;
            for (int w = 0; w < SEG; ++w) {
                if (tg[w].n <= 0) continue;
                for (int c = lane; c < tg[w].n_chunks; c += 64) n += tg[w].out[c] >= t;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
            return n;
        };
        if (count_at(theta) > sp.max_cand) {
            int lo = theta, hi = vmax;              // count(lo) > max_cand >= count(hi + 1)
            while (hi - lo > 1) {
                const int mid = lo + (hi - lo) / 2;
                if (count_at(mid) > sp.max_cand) lo = mid; else hi = mid;
            }
            theta = hi;
        }
    }
    r.upper_bound = (float)(vmax + shift + sp.slack) / (float)sp.sc;
    // candidates of a threshold, in column order, merged into windows: returns how many separate windows they need; with `store`
    // fills r (at most STRQ_SCREEN_MAX_WINDOWS: the last one takes everything from there on)
    auto scan = [&](int th, bool store) -> int {
        int nw = 0, last_hi = -(1 << 30), ncand = 0;
        for (int w = 0; w < SEG; ++w) {
            if (tg[w].n <= 0) continue;
            for (int c0 = 0; c0 < tg[w].n_chunks; c0 += 64) {
                const int c1 = c0 + lane;
                const bool cand = c1 < tg[w].n_chunks && tg[w].out[c1] >= th;
                uint64_t mask = __ballot(cand);
                while (mask) {              // wave-uniform: every lane walks the same candidates, lane 0's copy is stored
                    const int bit = __builtin_ctzll(mask); mask &= mask - 1;
                    const int c = c0 + bit;
                    ++ncand;
                    // columns of the chunk (lane lM, steps 64 c + 1 .. 64 c + 64), in read coordinates
                    int lo = 128 * c - 2 * lM + 1, hi = 128 * c - 2 * lM + 128;
                    if (lo < 1) lo = 1;
                    if (hi > tg[w].n) hi = tg[w].n;
                    if (hi < lo) continue;
                    lo += tg[w].col_off; hi += tg[w].col_off;
                    // pieces and chunks come in ascending column order, the overlap zones of a piece repeat columns of the one before
                    if (nw > 0 && lo <= last_hi + sp.merge_gap) {
                        if (hi > last_hi) last_hi = hi;
                        if (store) {
                            const int k = nw <= STRQ_SCREEN_MAX_WINDOWS ? nw - 1 : STRQ_SCREEN_MAX_WINDOWS - 1;
                            if (hi > r.hi[k]) r.hi[k] = hi;
                            if (lo < r.lo[k]) r.lo[k] = lo;
                        }
                    } else {
                        ++nw; last_hi = hi;
                        if (store) {
                            if (nw <= STRQ_SCREEN_MAX_WINDOWS) { r.lo[nw - 1] = lo; r.hi[nw - 1] = hi; }
                            else if (hi > r.hi[STRQ_SCREEN_MAX_WINDOWS - 1]) r.hi[STRQ_SCREEN_MAX_WINDOWS - 1] = hi;      // more separate candidates than windows: the last one takes the rest
                        }
                    }
                }
            }
        }
        if (store) r.n_cand = ncand;
        return nw;
    };
    if (!theta_list && sp.max_cand > 0 && scan(theta, false) > STRQ_SCREEN_MAX_WINDOWS) {
        // first look of the coarse screen: the best four windows only (what a raised threshold leaves out is the second look's business)
        int lo = theta, hi = vmax;
        while (hi - lo > 1) {
            const int mid = lo + (hi - lo) / 2;
            if (scan(mid, false) > STRQ_SCREEN_MAX_WINDOWS) lo = mid; else hi = mid;
        }
        theta = hi;
    }
    {
        // every column of a chunk below theta has a float32 score below (theta + shift + slack) / sc =: lower_bound; the exact pass
        // certifies its windows by reaching it.  Rounded UP to float32: a bound a hair too high costs a spurious second round, one a
        // hair too low would let an excluded column tie with the certificate
        const double lb = (double)(theta + shift + sp.slack) / (double)sp.sc;
        float lbf = (float)lb;
        if ((double)lbf < lb) {          // the next float32 up
            int32_t bits = __float_as_int(lbf);
            bits = lbf > 0.0f ? bits + 1 : (lbf < 0.0f ? bits - 1 : 1);
            lbf = __int_as_float(bits);
        }
        r.lower_bound = lbf;
    }
    // prune only above the cold-start bound of the pieces, and only when the lower bound is a score worth the name
    const bool ok = theta + shift > 0 && theta + shift > bound_scaled[g];
    if (ok) {
        int nw = scan(theta, true);
        if (nw > STRQ_SCREEN_MAX_WINDOWS) nw = STRQ_SCREEN_MAX_WINDOWS;
        // ascending and disjoint (a candidate in a piece's overlap zone may have pulled a window's start to the left of the
        // window before it): the combine kernel breaks score ties by piece order
        for (int a = 1; a < nw; ++a)
            for (int b = a; b > 0 && r.lo[b] < r.lo[b - 1]; --b) {
                const int tl = r.lo[b], th = r.hi[b]; r.lo[b] = r.lo[b - 1]; r.hi[b] = r.hi[b - 1]; r.lo[b - 1] = tl; r.hi[b - 1] = th;
            }
        int kept = nw > 0 ? 1 : 0;
        for (int a = 1; a < nw; ++a) {
            if (r.lo[a] <= r.hi[kept - 1] + 1) { if (r.hi[a] > r.hi[kept - 1]) r.hi[kept - 1] = r.hi[a]; }
            else { r.lo[kept] = r.lo[a]; r.hi[kept] = r.hi[a]; ++kept; }
        }
        for (int a = kept; a < STRQ_SCREEN_MAX_WINDOWS; ++a) { r.lo[a] = 0; r.hi[a] = 0; }
        r.n_win = kept;
    }
    if (lane == 0) out[idx] = r;
}

int screen_plan(const AlignParams& p, int samples, int max_n, ScreenParams* sp)
{
    if (strq::opt("STRQ_NO_SCREEN")) return 0;
    if (samples != S) return 0;
    if (!(p.open_h == p.ext_h && p.open_v == p.ext_v)) return 0;
    if (!(p.dist_min >= 0.0f) || !(p.ext_h < 0.0f) || !(p.ext_v < 0.0f) || !(p.dist_offset > 0.0f) || !(p.dist_min <= p.dist_offset)) return 0;
    // largest power-of-two scale at which T = S + i |e_v| + j |e_h| stays below 2^31 and a table entry below 2^16
    for (int sc = 1024; sc >= 16; sc >>= 1) {
        const double hh = -(double)p.ext_h * sc, v = -(double)p.ext_v * sc, smax = std::ceil((double)p.dist_offset * sc);
        if (hh != (double)(long)hh || v != (double)(long)v) continue;
        if (hh < 1 || smax + v + hh > 65000.0) continue;
        const double top = 64.0 * R * (smax + v) + ((double)max_n + 256.0) * hh + smax + v + hh + (double)STRQ_SCREEN_BIAS;
        if (top > 2.0e9) continue;          // below 0x7f800000: every stored value is the bit pattern of a finite float32
        sp->sc = sc; sp->hh = (int)hh; sp->v = (int)v; sp->cadd = (int)(hh + v);
        // float32 rounding of the exact DP against real arithmetic: one addition per step of a path, each off by at most 2^-24 of a
        // value below the next power of two above 64 R dist_offset; a path that matters has at most 64 R rows and
        // 64 R dist_offset / |e_h| horizontal steps (14.9 score units for STRique's parameters)
        {
            const double rows = 64.0 * R, maxval = rows * (double)p.dist_offset;
            double p2 = 1.0; while (p2 < maxval) p2 *= 2.0;
            const double adds = rows * (1.0 + (double)p.dist_offset / -(double)p.ext_h);
            const double e = adds * p2 / 16777216.0;
            if (!(e < 1.0e4)) continue;
            sp->slack = ((int)std::ceil(e) + 1) * sc;
        }
        sp->merge_gap = 3072; sp->margin = 0; sp->max_cand = 0;
        return 1;
    }
    return 0;
}

int screen2_plan(const AlignParams& p, int samples, int max_n, int merge, ScreenParams* sp)
{
    // the fine frame at a smaller scale: a table entry is merge (ceil(s sc) + hh + v) and has to fit 16 bits
    if (merge != 1 && merge != 3) return 0;
    if (!screen_plan(p, samples, max_n, sp)) return 0;
    while (sp->sc >= 16) {
        const double smax = std::ceil((double)p.dist_offset * sp->sc);
        if ((double)merge * (smax + sp->cadd) <= 65000.0) return 1;
        if ((sp->hh & 1) || (sp->v & 1)) return 0;
        sp->slack = (sp->slack / sp->sc) * (sp->sc / 2);
        sp->sc /= 2; sp->hh /= 2; sp->v /= 2; sp->cadd = sp->hh + sp->v;
    }
    return 0;
}

int launch_screen2(hipStream_t stream, const Screen2Task* tasks, int n_groups, int* queue, const ScreenParams& sp,
                   size_t lds_bytes, int groups_per_cu, int n_cu, int merge)
{
    if (n_groups <= 0) return 0;
    if (merge != 3 && merge != 1) return 2;
    auto kern = merge == 3 ? align_screen3_kernel : align_screen1_kernel;
    if (merge == 1 && groups_per_cu > 4) groups_per_cu = 4;          // ... for four
    if (groups_per_cu > 6) groups_per_cu = 6;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kern, dim3(groups_per_cu * n_cu), dim3(64 * SEG), lds_bytes, stream, tasks, n_groups, queue, sp);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_screen(hipStream_t stream, const ScreenTask* tasks, int n_groups, int* queue, const ScreenParams& sp,
                  size_t lds_bytes, int tables_per_cu, int n_cu)
{
    if (n_groups <= 0) return 0;
    (void)hipFuncSetAttribute((const void*)align_screen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(align_screen_kernel, dim3(tables_per_cu * n_cu), dim3(64 * SEG), lds_bytes, stream, tasks, n_groups, queue, sp);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_screen_windows(hipStream_t stream, const ScreenTask* tasks, int n_groups, const ScreenParams& sp,
                          const int32_t* bound_scaled, ScreenWindows* out, const int32_t* list, const int32_t* theta_list)
{
    if (n_groups <= 0) return 0;
    hipLaunchKernelGGL(screen_windows_kernel, dim3((n_groups + 3) / 4), dim3(256), 0, stream, tasks, n_groups, sp, bound_scaled, out, list, theta_list);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq
