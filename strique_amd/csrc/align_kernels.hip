// Flank alignment on gfx950 (MI355X): semi-global affine-gap DP of a short flank template
// (vertical, M rows) against a whole raw-signal read (horizontal, N columns).
//
// Replaces: align_raw<float,float>::semiglobal  (reference src/align_raw.h:106-158),
//           Score<float,Distance>::score + gap accessors (src/score_distance.h:115-122,140-226),
//           i.e. the SeqAn2 globalAlignment(AlignConfig<true,false,false,true>, AffineGaps) call.
//
// Mapping (persistent waves pulling tasks, longest read first, from an atomic queue):
//   * one wave64 per alignment; a flank of up to 64*R rows is one strip (R <= 15 rows per lane, the
//     normal case: 870 rows = 58 lanes x 15).  Taller flanks are cut into two strips that run in
//     consecutive launches, the bottom row (S, V) of the upper strip being streamed through HBM to
//     the lower one (8 B per column) -- measured no faster than one strip, so only used when needed;
//   * waves per CU are limited by LDS (one score table of 20-26 KB per wave: six waves); a wave alone
//     on its SIMD issues one VALU instruction per ~5 cycles, two waves sharing a SIMD ~6 each;
//   * lane l owns rows [l*R, l*R+R) of its strip in registers (R rows per lane);
//   * the wave marches an anti-diagonal wavefront, two DP columns per lane per step (two
//     independent dependency chains -> ILP 2 inside one wave): at step t lane l computes
//     columns 2(t-l)-1 and 2(t-l).  Cross-lane traffic per step is the bottom cells of lane
//     l-1 and the packed levels of the two columns, each one DPP wave_shr:1;
//   * per-cell scores come from a per-alignment banded table staged in LDS
//     (row = k-mer class of the flank, column = 8-bit level of the read sample): the
//     pow(|h-v|,1.2) of the reference never runs in the DP loop; the table reads of a step are
//     issued one step ahead, so their latency hides behind the arithmetic;
//   * no per-cell trace is written by the forward pass.  It stores the wavefront registers
//     every STRQ_CKPT_STEPS steps; the trace pass re-runs only the blocks of steps the optimal
//     path crosses, with the full affine tie-break semantics, and walks a 4-bit trace back.
//     Re-computation is deterministic IEEE f32 add/max in the same order => identical path.
//
// All arithmetic is float32 add / max exactly in the order of the CPU oracle
// (oracle/align_oracle.c); compile with -ffp-contract=off.
#include "strq_opt.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "align_kernels.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace strq {

#define STRQ_NINF (-3.4028234663852886e38f / 2)

#define STRQ_STR2(x) #x
#define STRQ_STR(x) STRQ_STR2(x)
// tie rules -- keep identical to oracle/align_oracle.c
#define STRQ_TIE_EXT(ext, opn)  ((ext) >= (opn))
#define STRQ_TIE_H_OVER_V(h, v) ((h) >= (v))
#define STRQ_TIE_D_OVER_G(d, g) ((d) >= (g))

static __device__ __forceinline__ float dpp_shr1_f(float v, float fill)
{
    // wave_shr:1 -- lane l receives lane l-1; lane 0 keeps `fill` (bound_ctrl = 0 keeps old)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                              __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
static __device__ __forceinline__ float dpp_shr1_f0(float v)
{
    // ... lane 0 receives +0.0 (bound_ctrl): no register to preload with the fill value
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
static __device__ __forceinline__ int dpp_shr1_i(int v, int fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xF, 0xF, false);
}
template <class T> static __device__ __forceinline__ const T* uniform_ptr(const T* p)
{
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}
static __device__ __forceinline__ int med3i(int a, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi));
    return r;
}
// per-lane select with a wave-uniform 64-bit lane mask held in an SGPR pair
static __device__ __forceinline__ float sel_mask(float if0, float if1, uint64_t mask)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
    return r;
}

// Next task of a persistent wave.  The wave barriers keep the compiler from threading the
// `lane == 0` branch into neighbouring code (which would run readfirstlane with lane 0 masked off).
static __device__ __forceinline__ int next_task(int* queue, int lane)
{
    __builtin_amdgcn_wave_barrier();
    int ti = 0;
    if (lane == 0) ti = atomicAdd(queue, 1);
    __builtin_amdgcn_wave_barrier();
    ti = __builtin_amdgcn_readfirstlane(ti);
    __builtin_amdgcn_wave_barrier();
    return ti;
}

template <int R, int S> struct Shape {
    static constexpr int gcd_(int a, int b) { return b == 0 ? a : gcd_(b, a % b); }
    static constexpr int G = gcd_(R, S);                 // lane phases are multiples of G
    static constexpr int C = (S - G + R - 1) / S + 1;    // k-mer classes a lane can touch
    static constexpr int NMASK = S / G;                  // phase thresholds (index 0 unused)
    static_assert(C <= 6, "too many classes per lane");
};

// Wavefront state of one lane after a step (columns jA = 2(t-l)-1, jB = 2(t-l)).
template <int R> struct Lane {
    float S[R];      // S[row][jB]
    float H[R];      // H[row][jB]
    float SbotA;     // S[bottom row][jA]
    float VbotA;     // V[bottom row][jA]
    float VbotB;     // V[bottom row][jB]
    float upS;       // S[top row - 1][jB]   (diagonal input of the next step's column A)
};

struct LaneConst {
    int off[6];   // LDS byte offset of the class row minus lo4
    int lo4[6];   // band_lo * 4
    int hi4[6];   // last stored level * 4
    int off2[6];  // packed tables: LDS byte offset of the class row in the low-byte plane minus lo (lo4 / hi4 / off are then in level / 16-bit units)
};

// PK: the table holds 24-bit fixed-point scores (units of 2^-20; the table kernel packs a table only
// when every entry is exactly representable) and the whole DP runs in those units: scaling by a
// power of two commutes with every float32 rounding, so the results are the same bits with a shifted
// exponent.  Three bytes per entry in two planes -- the high 16 bits, then the low 8 bits -- so that
// every LDS access is naturally aligned (unaligned 32-bit LDS reads measure 3x slower on gfx950).
#define STRQ_PK_SCALE 1048576.0f
template <int R, int S, bool PK>
static __device__ __forceinline__ void fetch_scores(const char* lds, const LaneConst& lc, int q4,
                                                    float (&sc)[Shape<R, S>::C])
{
#pragma unroll
    for (int c = 0; c < Shape<R, S>::C; ++c) {
        if constexpr (PK) {
            const int ci = med3i(q4, lc.lo4[c], lc.hi4[c]);                      // clamped level
            const uint32_t h = *reinterpret_cast<const uint16_t*>(lds + (ci << 1) + lc.off[c]);
            const uint32_t l = *reinterpret_cast<const uint8_t*>(lds + ci + lc.off2[c]);
            sc[c] = (float)((h << 8) | l);
        } else {
            const int t = med3i(q4, lc.lo4[c], lc.hi4[c]) + lc.off[c];
            sc[c] = *reinterpret_cast<const float*>(lds + t);
        }
    }
}

// expand the <=C class scores of a column to the R rows of the lane.
// row r of a lane with phase p belongs to class slot (p + r) / S; p is a multiple of G, so rows
// split into groups of G with at most one lane-mask select per group (none when R % S == 0).
template <int R, int S>
static __device__ __forceinline__ void expand_scores(const float (&sc)[Shape<R, S>::C],
                                                     const uint64_t (&pm)[Shape<R, S>::NMASK],
                                                     float (&rs)[R])
{
    constexpr int G = Shape<R, S>::G, C = Shape<R, S>::C;
#pragma unroll
    for (int r = 0; r < R; r += G) {
        const int base = r / S, x = (r % S) / G;
        float v;
        if (x == 0 || base + 1 >= C) v = sc[base < C ? base : C - 1];
        else v = sel_mask(sc[base], sc[base + 1], pm[x]);
#pragma unroll
        for (int g = 0; g < G && r + g < R; ++g) rs[r + g] = v;
    }
}

// runtime-indexed read of x[rM] (rM wave-uniform, in an SGPR)
template <int R>
static __device__ __forceinline__ float pick_row(const float (&x)[R], int rM)
{
    typedef float vec_t __attribute__((ext_vector_type(R <= 16 ? 16 : 32)));
    vec_t v;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = x[r];
    return v[rM];
}

struct TraceWords { uint64_t a[2], b[2]; };

// Trace codes, one nibble per cell: bit 0 the diagonal won (D >= G), bit 1 H won over V, bit 2 H extended, bit 3 V extended.
// A cell's four tie-rule compares are shifted into a 32-bit accumulator as they are made -- acc = 2 * acc + (a >= b), one
// v_cmp + one v_addc_co_u32 (the compare's lane mask is the carry) -- instead of four selects and the ors that merged them:
// eight rows per accumulator, two accumulators per 64-bit trace word, the first row of a half in its highest nibble.
#ifndef STRQ_TRACE_SELECT
static __device__ __forceinline__ void trace_push_ge(uint32_t& acc, float a, float b)
{
    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
#endif
// position (bit offset in the 64-bit word r / 16) of the nibble of row r
template <int R> static __device__ __forceinline__ int trace_shift(int r)
{
#ifndef STRQ_TRACE_SELECT
    const int in_word = r % 16, half = in_word / 8;
    const int rows_word = R - 16 * (r / 16) < 16 ? R - 16 * (r / 16) : 16;
    const int rows_half = rows_word - 8 * half < 8 ? rows_word - 8 * half : 8;
    return 32 * half + 4 * (rows_half - 1 - (in_word % 8));
#else
    return 4 * (r % 16);
#endif
}

// Two DP columns (jA, jB) for this lane.
// LH / LV: open == extend in that direction, where the affine recurrence collapses exactly
// (H[i][j-1] <= S[i][j-1] always and x -> x+e is monotone in IEEE arithmetic, so
// max(H+e, S+e) == S+e bit for bit).  KEEP additionally materialises the H / V values the
// collapsed form does not carry (needed when the state is checkpointed).  TRACE computes the
// full recurrence with the oracle's tie rules and returns the 4-bit trace codes.
template <int R, bool LH, bool LV, bool KEEP, bool TRACE, int RMC = -1>
static __device__ __forceinline__ void dp_step2(Lane<R>& st, const float (&rsA)[R], const float (&rsB)[R],
                                                float upA, float upB, float upVA, float upVB,
                                                const AlignParams& p, TraceWords* tw,
                                                int rM, float* candA, float* candB)
{
    float SA[R], SB[R];
    float HA[R], HB[R];
    float vA = upVA, vB = upVB;
    if constexpr (TRACE) { tw->a[0] = tw->a[1] = tw->b[0] = tw->b[1] = 0; }
    uint32_t accA[4] = {0u, 0u, 0u, 0u}, accB[4] = {0u, 0u, 0u, 0u};
    (void)accA; (void)accB;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // ---- column A, row r
        {
            const float diag = r == 0 ? st.upS : st.S[r - 1];
            const float up = r == 0 ? upA : SA[r > 0 ? r - 1 : 0];
            const float D = diag + rsA[r];
            if constexpr (TRACE) {
                const float hext = st.H[r] + p.ext_h, hopn = st.S[r] + p.open_h;
                const float vext = vA + p.ext_v, vopn = up + p.open_v;
#ifndef STRQ_TRACE_SELECT
                // every tie rule is a '>=' and the winner of each is the larger value: the values come from max, the rules from the compares
                uint32_t& acc = accA[(r / 16) * 2 + (r % 16) / 8];
                trace_push_ge(acc, vext, vopn); trace_push_ge(acc, hext, hopn);
                const float Hn = __builtin_fmaxf(hext, hopn), Vn = __builtin_fmaxf(vext, vopn);
                trace_push_ge(acc, Hn, Vn);
                const float Gm = __builtin_fmaxf(Hn, Vn);
                trace_push_ge(acc, D, Gm);
                SA[r] = __builtin_fmaxf(D, Gm); HA[r] = Hn; vA = Vn;
#else
                const bool he = STRQ_TIE_EXT(hext, hopn);
                const float Hn = he ? hext : hopn;
                const bool ve = STRQ_TIE_EXT(vext, vopn);
                const float Vn = ve ? vext : vopn;
                const bool gh = STRQ_TIE_H_OVER_V(Hn, Vn);
                const float Gm = gh ? Hn : Vn;
                const bool dd = STRQ_TIE_D_OVER_G(D, Gm);
                SA[r] = dd ? D : Gm; HA[r] = Hn; vA = Vn;
                const uint64_t code = (dd ? 1u : 0u) | (gh ? 2u : 0u) | (he ? 4u : 0u) | (ve ? 8u : 0u);
                tw->a[r / 16] |= code << (4 * (r % 16));
#endif
            } else {
                float Hn, Vn;
                if constexpr (LH) Hn = st.S[r] + p.ext_h;
                else Hn = __builtin_fmaxf(st.H[r] + p.ext_h, st.S[r] + p.open_h);
                if constexpr (LV) Vn = up + p.ext_v;
                else Vn = __builtin_fmaxf(vA + p.ext_v, up + p.open_v);
                SA[r] = __builtin_fmaxf(__builtin_fmaxf(D, Hn), Vn);
                HA[r] = Hn; vA = Vn;
            }
        }
        // ---- column B, row r
        {
            const float diag = r == 0 ? upA : SA[r > 0 ? r - 1 : 0];
            const float up = r == 0 ? upB : SB[r > 0 ? r - 1 : 0];
            const float D = diag + rsB[r];
            if constexpr (TRACE) {
                const float hext = HA[r] + p.ext_h, hopn = SA[r] + p.open_h;
                const float vext = vB + p.ext_v, vopn = up + p.open_v;
#ifndef STRQ_TRACE_SELECT
                uint32_t& acc = accB[(r / 16) * 2 + (r % 16) / 8];
                trace_push_ge(acc, vext, vopn); trace_push_ge(acc, hext, hopn);
                const float Hn = __builtin_fmaxf(hext, hopn), Vn = __builtin_fmaxf(vext, vopn);
                trace_push_ge(acc, Hn, Vn);
                const float Gm = __builtin_fmaxf(Hn, Vn);
                trace_push_ge(acc, D, Gm);
                SB[r] = __builtin_fmaxf(D, Gm); HB[r] = Hn; vB = Vn;
#else
                const bool he = STRQ_TIE_EXT(hext, hopn);
                const float Hn = he ? hext : hopn;
                const bool ve = STRQ_TIE_EXT(vext, vopn);
                const float Vn = ve ? vext : vopn;
                const bool gh = STRQ_TIE_H_OVER_V(Hn, Vn);
                const float Gm = gh ? Hn : Vn;
                const bool dd = STRQ_TIE_D_OVER_G(D, Gm);
                SB[r] = dd ? D : Gm; HB[r] = Hn; vB = Vn;
                const uint64_t code = (dd ? 1u : 0u) | (gh ? 2u : 0u) | (he ? 4u : 0u) | (ve ? 8u : 0u);
                tw->b[r / 16] |= code << (4 * (r % 16));
#endif
            } else {
                float Hn, Vn;
                if constexpr (LH) Hn = SA[r] + p.ext_h;
                else Hn = __builtin_fmaxf(HA[r] + p.ext_h, SA[r] + p.open_h);
                if constexpr (LV) Vn = up + p.ext_v;
                else Vn = __builtin_fmaxf(vB + p.ext_v, up + p.open_v);
                SB[r] = __builtin_fmaxf(__builtin_fmaxf(D, Hn), Vn);
                HB[r] = Hn; vB = Vn;
            }
        }
    }
#ifndef STRQ_TRACE_SELECT
    if constexpr (TRACE) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            tw->a[x] = (uint64_t)accA[2 * x] | ((uint64_t)accA[2 * x + 1] << 32);
            tw->b[x] = (uint64_t)accB[2 * x] | ((uint64_t)accB[2 * x + 1] << 32);
        }
    }
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) st.S[r] = SB[r];
    if constexpr (TRACE || !LH || KEEP) {
#pragma unroll
        for (int r = 0; r < R; ++r) st.H[r] = HB[r];
    }
    st.SbotA = SA[R - 1];
    if constexpr (TRACE || !LV || KEEP) { st.VbotA = vA; st.VbotB = vB; }
    st.upS = upB;
    if constexpr (RMC >= 0) { *candA = SA[RMC]; *candB = SB[RMC]; }                // last flank row in a register known at compile time
    else if (candA) { *candA = pick_row<R>(SA, rM); *candB = pick_row<R>(SB, rM); }   // last flank row in an interior register
}

template <int R, int S, bool PK>
static __device__ __forceinline__ void load_lane_consts(const AlignTask& tk, int lane, int lds_base,
                                                        LaneConst& lc, uint64_t (&pm)[Shape<R, S>::NMASK])
{
    const int row0 = tk.row0 + lane * R;                 // global (0-based) index of the lane's first row
    const int kbase = row0 / S - tk.row0 / S, phase = row0 % S;
    const int off0 = 0;
#pragma unroll
    for (int c = 0; c < Shape<R, S>::C; ++c) {
        int k = kbase + c; if (k > tk.k - 1) k = tk.k - 1;
        const uint32_t d = (uint32_t)tk.band_lo[k];
        const int lo = (int)(d & 255u), w1 = (int)((d >> 8) & 255u), off = (int)(d >> 16);
        if constexpr (PK) {
            lc.lo4[c] = lo; lc.hi4[c] = lo + w1;
            lc.off[c] = lds_base + (off - lo) * 2;
            lc.off2[c] = lds_base + ((2 * tk.tsize + 3) & ~3) + off - lo;
        } else {
            lc.lo4[c] = lo * 4;
            lc.hi4[c] = (lo + w1) * 4;
            lc.off[c] = lds_base + (off - off0) * 4 - lo * 4;
            lc.off2[c] = 0;
        }
    }
#pragma unroll
    for (int x = 0; x < Shape<R, S>::NMASK; ++x) {
        const uint64_t b = __ballot(phase >= S - x * Shape<R, S>::G);
        // keep the mask in an SGPR pair whatever the compiler thinks of the surrounding control flow
        const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
        pm[x] = ((uint64_t)hi << 32) | lo;
    }
}

// stage the banded score table of one alignment into this wave's LDS slice
template <bool PK>
static __device__ __forceinline__ void stage_table(const AlignTask& tk, float* lds, int lane)
{
    // the whole table (rows of equal classes are shared, so a strip's rows are not contiguous)
    if constexpr (PK) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(tk.table3);
        uint32_t* dst = reinterpret_cast<uint32_t*>(lds);
        const int nd = (((2 * tk.tsize + 3) & ~3) + tk.tsize + 3) / 4;      // 16-bit plane (padded to a dword), 8-bit plane
        for (int i = lane; i < nd; i += 64) dst[i] = src[i];
    } else {
        for (int i = lane; i < tk.tsize; i += 64) lds[i] = tk.table[i];
    }
}

template <int R, bool PK>
static __device__ __forceinline__ void init_lane(const AlignTask& tk, int lane, Lane<R>& st)
{
    constexpr float SC = PK ? STRQ_PK_SCALE : 1.0f;
    const int row0 = lane * R;   // DP row of register r is row0 + r + 1
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int i = row0 + r + 1; if (i > tk.m) i = tk.m;
        st.S[r] = tk.col0[i] * SC;
        st.H[r] = STRQ_NINF;
    }
    { int i = row0 + R; if (i > tk.m) i = tk.m; st.SbotA = st.VbotA = st.VbotB = tk.col0[i] * SC; }  // V[i][0] == S[i][0]
    { int i = row0;     if (i > tk.m) i = tk.m; st.upS = tk.col0[i] * SC; }    // S[row0][0]; col0[0] == 0
}

// packed levels (x4) of columns 2*(64*chunk+lane)+1 and +2
template <bool PK>
static __device__ __forceinline__ int load_chunk(const AlignTask& tk, int chunk, int lane)
{
    constexpr int U = PK ? 1 : 4;
    const int idx = (chunk * 64 + lane) * 2;
    int a = 0, b = 0;
    if (idx < tk.n) a = tk.levels[idx];
    if (idx + 1 < tk.n) b = tk.levels[idx + 1];
    return (a * U) | ((b * U) << 16);
}

// boundary {S, V} of the row above the strip for columns 2*(64*chunk+lane)+1 and +2
struct Bnd4 { float sA, vA, sB, vB; };
static __device__ __forceinline__ Bnd4 load_bnd(const AlignTask& tk, int chunk, int lane)
{
    const int idx = (chunk * 64 + lane) * 2;          // column idx+1 -> entry idx
    Bnd4 b{0.0f, STRQ_NINF, 0.0f, STRQ_NINF};
    if (idx < tk.n) { b.sA = tk.bnd_in[2 * idx]; b.vA = tk.bnd_in[2 * idx + 1]; }
    if (idx + 1 < tk.n) { b.sB = tk.bnd_in[2 * idx + 2]; b.vB = tk.bnd_in[2 * idx + 3]; }
    return b;
}

template <int R>
static __device__ __forceinline__ void save_ckpt(float* c, int lane, const Lane<R>& st)
{
#pragma unroll
    for (int r = 0; r < R; ++r) c[r * 64 + lane] = st.S[r];
#pragma unroll
    for (int r = 0; r < R; ++r) c[(R + r) * 64 + lane] = st.H[r];
    c[(2 * R + 0) * 64 + lane] = st.SbotA;
    c[(2 * R + 1) * 64 + lane] = st.VbotA;
    c[(2 * R + 2) * 64 + lane] = st.VbotB;
    c[(2 * R + 3) * 64 + lane] = st.upS;
}
template <int R>
static __device__ __forceinline__ void load_ckpt(const float* c, int lane, Lane<R>& st)
{
#pragma unroll
    for (int r = 0; r < R; ++r) st.S[r] = c[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < R; ++r) st.H[r] = c[(R + r) * 64 + lane];
    st.SbotA = c[(2 * R + 0) * 64 + lane];
    st.VbotA = c[(2 * R + 1) * 64 + lane];
    st.VbotB = c[(2 * R + 2) * 64 + lane];
    st.upS = c[(2 * R + 3) * 64 + lane];
}

// ------------------------------------------------------------------------------------------
// forward pass: best score of the last flank row, its column, and wavefront checkpoints
// ------------------------------------------------------------------------------------------
// MODE: which boundaries the strip has.  bit 0: input from the strip above (else the free top row),
//       bit 1: output to the strip below (else this strip holds the last flank row and tracks the best).
// RMC >= 0: the last flank row sits in register RMC of its lane, known at compile time (no runtime-indexed pick)
template <int R, int S, bool LH, bool LV, int MODE, bool RM_LAST, bool PK, int RMC = -1>
struct Forward {
    static constexpr bool HAS_IN = (MODE & 1) != 0, HAS_OUT = (MODE & 2) != 0;
    const AlignTask& tk;
    const AlignParams& p;
    const char* ldsb;
    const LaneConst& lc;
    const uint64_t (&pm)[Shape<R, S>::NMASK];
    const int lane, rM;
    Lane<R> st;
    float best, bestA; int bestt;
    int qq;                                   // packed levels (x4) of the two columns of the step about to run
    float scA[Shape<R, S>::C], scB[Shape<R, S>::C];   // their class scores, fetched one step ahead
    Bnd4 bcur;

    // levels of the next step enter at lane 0 (entry `snext` of the chunk register `qsrc`); its scores
    // are read from LDS now and consumed one step later, so the LDS latency hides behind the DP arithmetic
    __device__ __forceinline__ void advance(int qsrc, int snext, int& qn, float (&nA)[Shape<R, S>::C], float (&nB)[Shape<R, S>::C])
    {
        qn = dpp_shr1_i(qq, __builtin_amdgcn_readlane(qsrc, snext));
        fetch_scores<R, S, PK>(ldsb, lc, qn & 0xffff, nA);
        fetch_scores<R, S, PK>(ldsb, lc, (int)((unsigned)qn >> 16), nB);
    }
    __device__ __forceinline__ void prime(int qcur)
    {
        qq = 0;
        int qn; float nA[Shape<R, S>::C], nB[Shape<R, S>::C];
        advance(qcur, 0, qn, nA, nB);
        qq = qn;
#pragma unroll
        for (int c = 0; c < Shape<R, S>::C; ++c) { scA[c] = nA[c]; scB[c] = nB[c]; }
    }

    // PRED: lanes may be idle (before their first / after their last column).
    // s: position of this step inside its 64-step chunk; (qsrc, snext): where the next step's levels come from.
    // RMX: the register of its lane the last flank row sits in, when the caller knows it at compile time (forward_one
    // switches its steady-state loop on (m - 1) % R); -1: read through the runtime index rM
    template <bool PRED, bool KEEP, int RMX = RMC>
    __device__ __forceinline__ void step(int t, int s, int qsrc, int snext)
    {
        int qn; float nA[Shape<R, S>::C], nB[Shape<R, S>::C];
        advance(qsrc, snext, qn, nA, nB);
        float fA = 0.0f, fB = 0.0f, fVA = STRQ_NINF, fVB = STRQ_NINF;
        if constexpr (HAS_IN) {
            fA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bcur.sA), s));
            fB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bcur.sB), s));
            if constexpr (!LV) {
                fVA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bcur.vA), s));
                fVB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bcur.vB), s));
            }
        }
#ifdef STRQ_DP_ZFILL
        // experiment (13 % SLOWER, profiles/r03_dead_ends.md 7): without a strip above, lane 0 takes the DPP's own zero fill -- two v_mov less per step
        const float upA = HAS_IN ? dpp_shr1_f(st.SbotA, fA) : dpp_shr1_f0(st.SbotA);
        const float upB = HAS_IN ? dpp_shr1_f(st.S[R - 1], fB) : dpp_shr1_f0(st.S[R - 1]);
#else
        const float upA = dpp_shr1_f(st.SbotA, fA);
        const float upB = dpp_shr1_f(st.S[R - 1], fB);
#endif
        float upVA = STRQ_NINF, upVB = STRQ_NINF;
        if constexpr (!LV) { upVA = dpp_shr1_f(st.VbotA, fVA); upVB = dpp_shr1_f(st.VbotB, fVB); }
        const int jB = 2 * (t - lane), jA = jB - 1;
        bool act = true;
        if constexpr (PRED) act = (jA >= 1) && (jA <= tk.n);
        if (act) {
            float rsA[R], rsB[R];
            expand_scores<R, S>(scA, pm, rsA);
            expand_scores<R, S>(scB, pm, rsB);
            bool okB = true;
            if constexpr (PRED) okB = jB <= tk.n;
            if constexpr (HAS_OUT) {
                // the strip below needs S and V of this strip's last row (lane 63, register R-1);
                // KEEP materialises V in the collapsed recurrence at no extra arithmetic
                dp_step2<R, LH, LV, true, false>(st, rsA, rsB, upA, upB, upVA, upVB, p, nullptr, 0, nullptr, nullptr);
                if (lane == 63) {
                    float* o = tk.bnd_out + 2 * (size_t)(jA - 1);
                    o[0] = st.SbotA; o[1] = st.VbotA;
                    if (okB) { o[2] = st.S[R - 1]; o[3] = st.VbotB; }
                }
            } else {
                float candA, candB;
                if constexpr (RM_LAST || RMX == R - 1) {
                    dp_step2<R, LH, LV, KEEP, false>(st, rsA, rsB, upA, upB, upVA, upVB, p, nullptr, 0, nullptr, nullptr);
                    candA = st.SbotA; candB = st.S[R - 1];
                } else if constexpr (RMX >= 0) {
                    dp_step2<R, LH, LV, KEEP, false, RMX>(st, rsA, rsB, upA, upB, upVA, upVB, p, nullptr, rM, &candA, &candB);
                } else {
                    dp_step2<R, LH, LV, KEEP, false>(st, rsA, rsB, upA, upB, upVA, upVB, p, nullptr, rM, &candA, &candB);
                }
                // leftmost maximum of the last row: remember the step of the last strict improvement and
                // column A's value there (column A is left of column B; decoded after the loop)
                const float cb = okB ? candB : candA;
                const float nb = __builtin_fmaxf(__builtin_fmaxf(best, candA), cb);
                if (nb > best) { bestt = t; bestA = candA; }
                best = nb;
            }
        }
        qq = qn;
#pragma unroll
        for (int c = 0; c < Shape<R, S>::C; ++c) { scA[c] = nA[c]; scB[c] = nB[c]; }
    }
};

// The registers (m - 1) % R can be for a flank of m = k * S rows: x with x + 1 a multiple of gcd(R, S).  `f` runs for the
// one that equals rM, as a compile-time constant; returns false when none does (a flank that is no whole number of runs).
template <int R, int S, int X, class F>
static __device__ __forceinline__ void rm_case(int rM, bool& hit, F& f)
{
    if constexpr ((X + 1) % Shape<R, S>::G == 0) {
        if (!hit && rM == X) { f(std::integral_constant<int, X>{}); hit = true; }
    }
}
template <int R, int S, class F, int... X>
static __device__ __forceinline__ bool rm_dispatch(int rM, F& f, std::integer_sequence<int, X...>)
{
    bool hit = false;
    (rm_case<R, S, X>(rM, hit, f), ...);
    return hit;
}

// RMSW: the steady-state loop (every lane busy for the 64 steps of a chunk -- all but the first and last chunks of a piece)
// is compiled once per possible register of the last flank row and chosen by a wave-uniform branch outside the loop, so
// that EVERY flank length runs the 152-instruction step of STRique's own 870 rows (round 3 knew that register at compile
// time for m = 870 and m % R == 0 only; any other flank read it through pick_row, ~40 instructions per step more).
template <int R, int S, bool LH, bool LV, int MODE, bool RM_LAST, bool PK, bool STAGE = true, int RMC = -1, bool RMSW = false>
static __device__ __forceinline__ void forward_one(const AlignTask& tk, AlignResult* res, const AlignParams& p,
                                                   float* lds, int lds_base, const char* ldsb, int lane)
{
    constexpr bool HAS_IN = (MODE & 1) != 0, HAS_OUT = (MODE & 2) != 0;
    if constexpr (STAGE) stage_table<PK>(tk, lds, lane);
    LaneConst lc; uint64_t pm[Shape<R, S>::NMASK];
    load_lane_consts<R, S, PK>(tk, lane, lds_base, lc, pm);
    const int lM = (tk.m - 1) / R, rM = (tk.m - 1) % R;
    Forward<R, S, LH, LV, MODE, RM_LAST, PK, RMC> f{tk, p, ldsb, lc, pm, lane, rM};
    init_lane<R, PK>(tk, lane, f.st);
    f.best = tk.col0[tk.m] * (PK ? STRQ_PK_SCALE : 1.0f); f.bestA = 0.0f; f.bestt = -1;
    __builtin_amdgcn_s_waitcnt(0);   // LDS table written by this wave is visible to it

    const int nsteps = (tk.n + 1) / 2 + 63;
    int rM_u = 0;
    if constexpr (RMSW) rM_u = __builtin_amdgcn_readfirstlane(rM);      // in an SGPR: the switch over it is a scalar branch around the loops, not an exec mask inside them
    (void)rM_u;
    int qcur = load_chunk<PK>(tk, 0, lane);
    f.prime(qcur);
    Bnd4 bnext{0.0f, STRQ_NINF, 0.0f, STRQ_NINF};
    if constexpr (HAS_IN) f.bcur = load_bnd(tk, 0, lane); else f.bcur = bnext;
    for (int t0 = 0; t0 < nsteps; t0 += 64) {
        const int qnext = load_chunk<PK>(tk, t0 / 64 + 1, lane);   // prefetch next 128 columns
        if constexpr (HAS_IN) bnext = load_bnd(tk, t0 / 64 + 1, lane);
        // every lane busy with two valid columns for all 64 steps?
        const bool full = (t0 >= 63) && (2 * (t0 + 64) <= tk.n);
        const bool ckpt_here = ((t0 + 64) % STRQ_CKPT_STEPS) == 0 && (t0 + 64) < nsteps;
        const int send = nsteps - t0 < 64 ? nsteps - t0 : 64;
        if (full) {
#ifdef STRQ_DP_PAD
            // placement experiment (profiles/r03_dead_ends.md 9): the steady-state loop STRQ_DP_PAD dwords behind a 64-byte boundary
            asm volatile(".p2align 6\n\t.rept " STRQ_STR(STRQ_DP_PAD) "\n\ts_nop 0\n\t.endr");
#endif
            if constexpr (RMSW && !HAS_OUT && !RM_LAST) {
                auto chunk = [&](auto rmx) {
                    constexpr int X = decltype(rmx)::value;
                    for (int s = 0; s < 63; ++s) f.template step<false, false, X>(t0 + s + 1, s, qcur, s + 1);
                    if (ckpt_here) f.template step<false, true, X>(t0 + 64, 63, qnext, 0);
                    else f.template step<false, false, X>(t0 + 64, 63, qnext, 0);
                };
                if (!rm_dispatch<R, S>(rM_u, chunk, std::make_integer_sequence<int, R>{})) chunk(std::integral_constant<int, -1>{});
            } else {
                for (int s = 0; s < 63; ++s) f.template step<false, false>(t0 + s + 1, s, qcur, s + 1);
                if (ckpt_here) f.template step<false, true>(t0 + 64, 63, qnext, 0);
                else f.template step<false, false>(t0 + 64, 63, qnext, 0);
            }
        } else {
            for (int s = 0; s < send; ++s) {
                const int qsrc = s == 63 ? qnext : qcur, snext = (s + 1) & 63;
                if (ckpt_here && s == 63) f.template step<true, true>(t0 + s + 1, s, qsrc, snext);
                else f.template step<true, false>(t0 + s + 1, s, qsrc, snext);
            }
        }
        if (ckpt_here)
            save_ckpt<R>(tk.ckpt + (size_t)((t0 + 64) / STRQ_CKPT_STEPS - 1) * (STRQ_CKPT_FIELDS(R) * 64), lane, f.st);
        qcur = qnext;
        if constexpr (HAS_IN) f.bcur = bnext;
    }
    if constexpr (!HAS_OUT) {
        int bestj = 0;      // no improvement over column 0
        if (f.bestt >= 0) bestj = 2 * (f.bestt - lane) - (f.bestA == f.best ? 1 : 0);
        const float b = __shfl(f.best, lM, 64);
        const int bj = __shfl(bestj, lM, 64);
        res->best = b * (PK ? 1.0f / STRQ_PK_SCALE : 1.0f); res->j_end = bj;   // every lane stores the same value
    }
}

template <int R, int S, bool LH, bool LV, int MODE, bool PK>
__global__ void __launch_bounds__(512)
align_forward_kernel(const AlignTask* __restrict__ tasks, AlignResult* __restrict__ results, int n_tasks,
                     int* __restrict__ queue, AlignParams p, int lds_floats_per_wave)
{
    extern __shared__ float lds_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* lds = lds_all + (size_t)wave * lds_floats_per_wave;
    const int lds_base = wave * lds_floats_per_wave * 4;
    const char* ldsb = reinterpret_cast<const char*>(lds_all);
    for (;;) {
        const int ti = next_task(queue, lane);
        if (ti >= n_tasks) break;
        const AlignTask& tk = tasks[ti];
        if constexpr ((MODE & 2) != 0) forward_one<R, S, LH, LV, MODE, true, PK>(tk, results + ti, p, lds, lds_base, ldsb, lane);
        else {
            if ((tk.m - 1) % R == R - 1) forward_one<R, S, LH, LV, MODE, true, PK>(tk, results + ti, p, lds, lds_base, ldsb, lane);
            else forward_one<R, S, LH, LV, MODE, false, PK>(tk, results + ti, p, lds, lds_base, ldsb, lane);
        }
    }
}

// Column segments: a workgroup of SEG waves shares ONE score table in LDS and each wave runs the forward
// pass of one piece of the read (AlignTask::col_off; the pieces of alignment g are tasks[g * SEG + w]).
// More waves per CU at the same LDS footprint is what the issue-latency-bound DP needs; WPE (waves per
// SIMD) caps the registers the compiler may use so that all of them are resident.
// LISTED: the (normally empty) second round over the alignments the combine kernel listed -- a kernel of its
// own so that profiles show the first round alone.
// KNOWN: the launch holds only flanks whose last row sits in the lane's last register (m % R == 0) or, at 14 rows per lane,
// STRique's own 870 rows: the forward passes round 3 compiled for exactly those (the benchmarked instance).  Otherwise the
// pass whose steady-state loop is switched on that register (RMSW).  Two kernels instead of one body with all of them: together
// they spill (10 ... 42 VGPRs at the 128 the four waves per SIMD allow) and the 870-row loop ran 17 % slower (gpurun_out/r4c).
template <int R, int S, bool PK, int SEG, int WPE, bool LISTED = false, bool KNOWN = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(64 * SEG, 64 * SEG), amdgpu_waves_per_eu(WPE, WPE)))
align_forward_seg_kernel(const AlignTask* __restrict__ tasks, AlignResult* __restrict__ results, int n_groups,
                         int* __restrict__ queue, AlignParams p, const int* __restrict__ group_list,
                         const int* __restrict__ n_list)
{
    extern __shared__ float lds_all[];
    __shared__ int next_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* ldsb = reinterpret_cast<const char*>(lds_all);
    const int lds_base = 0;                    // table offsets are relative to lds_all
    // a second launch over the alignments the combine kernel listed (device-side count): see launch_align_segments
    if constexpr (LISTED) n_groups = __builtin_amdgcn_readfirstlane(*n_list);
    for (;;) {
        __syncthreads();                       // every wave is done with the previous table
        if (threadIdx.x == 0) next_group = atomicAdd(queue, 1);
        __syncthreads();
        int gi = __builtin_amdgcn_readfirstlane(next_group);
        if (gi >= n_groups) break;
        if constexpr (LISTED) gi = __builtin_amdgcn_readfirstlane(group_list[gi]);
        {
            const AlignTask& t0 = tasks[(size_t)gi * SEG];      // all pieces share the table of the alignment
            if constexpr (PK) {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(t0.table3);
                uint32_t* dst = reinterpret_cast<uint32_t*>(lds_all);
                const int nd = (((2 * t0.tsize + 3) & ~3) + t0.tsize + 3) / 4;
                for (int i = threadIdx.x; i < nd; i += 64 * SEG) dst[i] = src[i];
            } else {
                for (int i = threadIdx.x; i < t0.tsize; i += 64 * SEG) lds_all[i] = t0.table[i];
            }
        }
        __syncthreads();
        const int ti = gi * SEG + wave;
        const AlignTask& tk = tasks[ti];
        if (tk.n <= 0) continue;               // unused piece of a short read (wave-uniform)
        // the last flank row sits in register (m - 1) % R of its lane: STRique's own flanks (145 k-mer classes, 870 rows) at 14 rows
        // per lane in register 1 of lane 62
        if constexpr (KNOWN) {
            constexpr int RMC870 = R == 14 ? (870 - 1) % 14 : -1;
            if ((tk.m - 1) % R == R - 1) forward_one<R, S, true, true, 0, true, PK, false>(tk, results + ti, p, lds_all, lds_base, ldsb, lane);
            else if (RMC870 >= 0 && (tk.m - 1) % R == RMC870) forward_one<R, S, true, true, 0, false, PK, false, RMC870>(tk, results + ti, p, lds_all, lds_base, ldsb, lane);
            else forward_one<R, S, true, true, 0, false, PK, false>(tk, results + ti, p, lds_all, lds_base, ldsb, lane);      // not reached: the host launches this kernel for such flanks only
        } else {
            forward_one<R, S, true, true, 0, false, PK, false, -1, true>(tk, results + ti, p, lds_all, lds_base, ldsb, lane);
        }
    }
}

// Best piece of every alignment.  `list` / `n_list`: only the listed alignments (second round).
// `min_score`: the pieces were cut with a shorter overlap than the worst-case span bound; that is exact
// as long as the best score reaches min_score[a] (every path scoring that much spans less than the overlap
// used) -- alignments that do not are appended to `redo` and run again with the worst-case overlap.
__global__ void align_combine_kernel(const AlignTask* __restrict__ tasks, const AlignResult* __restrict__ seg, int n_align,
                                     int segs, AlignResult* __restrict__ out, int32_t* __restrict__ pick, int pick_base,
                                     const int* __restrict__ list, const int* __restrict__ n_list,
                                     const float* __restrict__ min_score, int* __restrict__ redo, int* __restrict__ redo_count,
                                     unsigned int* __restrict__ redo_total)
{
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= (n_list ? *n_list : n_align)) return;
    if (list) a = list[a];
    int best_k = 0;
    float best = seg[(size_t)a * segs].best;
    for (int k = 1; k < segs; ++k) {
        const size_t t = (size_t)a * segs + k;
        if (tasks[t].n <= 0) continue;
        if (seg[t].best > best) { best = seg[t].best; best_k = k; }      // strict: the leftmost piece wins ties
    }
    const size_t t = (size_t)a * segs + best_k;
    AlignResult r = seg[t];
    r.j_end += tasks[t].col_off;
    out[a] = r;
    pick[a] = pick_base + (int32_t)t;
    if (min_score && !(best >= min_score[a])) {
        redo[atomicAdd(redo_count, 1)] = a;
        if (redo_total) atomicAdd(redo_total, 1u);          // running total of second-round alignments (strq_last_second_round)
    }
}

// ------------------------------------------------------------------------------------------
// trace pass: re-run the blocks of steps the optimal path crosses, keep 4 bits per cell,
// walk back, and emit one record per flank row.
// ------------------------------------------------------------------------------------------
template <int R, int S, bool PK>
__global__ void __launch_bounds__(512)      // at most 8 waves per CU (one workgroup): the full 256-VGPR budget, no spills
align_trace_kernel(const AlignTask* __restrict__ tasks, AlignResult* __restrict__ results, int n_tasks,
                   int* __restrict__ queue, AlignParams p, int lds_floats_per_wave,
                   uint64_t* __restrict__ scratch_all, const int32_t* __restrict__ pick)
{
    extern __shared__ float lds_all[];
    constexpr int W = STRQ_TRACE_WORDS(R);
    constexpr size_t STEP_WORDS = 2 * W * 64;   // [col A/B][word][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* lds = lds_all + (size_t)wave * lds_floats_per_wave;
    const int lds_base = wave * lds_floats_per_wave * 4;
    const char* ldsb = reinterpret_cast<const char*>(lds_all);
    uint64_t* scratch = scratch_all + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * (STRQ_CKPT_STEPS * STEP_WORDS);

    for (;;) {
        const int ti = next_task(queue, lane);
        if (ti >= n_tasks) break;
        const AlignTask* cur = &tasks[pick ? __builtin_amdgcn_readfirstlane(pick[ti]) : ti];   // the strip (and column segment) that holds the end of the path
        const AlignTask* staged = nullptr;
        int32_t* rec = cur->rec;
        const int col_off = __builtin_amdgcn_readfirstlane(cur->col_off);      // read column of the segment's column 0
        LaneConst lc; uint64_t pm[Shape<R, S>::NMASK];

        int ci = cur->m_total, cj = __builtin_amdgcn_readfirstlane(results[ti].j_end) - col_off, state = 0;     // walker position (wave-uniform), segment columns
        while (ci > 0 && cj > 0) {
            while (ci <= cur->row0) cur = uniform_ptr(cur->up);       // the strip that holds row ci
            if (cur != staged) {
                __builtin_amdgcn_s_waitcnt(0);
                stage_table<PK>(*cur, lds, lane);
                load_lane_consts<R, S, PK>(*cur, lane, lds_base, lc, pm);
                __builtin_amdgcn_s_waitcnt(0);
                staged = cur;
            }
            const AlignTask& tk = *cur;
            const bool has_in = tk.bnd_in != nullptr;
            uint64_t pmu[Shape<R, S>::NMASK];       // lane masks, re-pinned to SGPRs inside this (data dependent) loop
#pragma unroll
            for (int x = 0; x < Shape<R, S>::NMASK; ++x) {
                const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)pm[x]), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(pm[x] >> 32));
                pmu[x] = ((uint64_t)hi << 32) | lo;
            }
            // step at which the walker's current cell was computed
            const int tcur = (ci - tk.row0 - 1) / R + (cj + 1) / 2;
            const int blk = (tcur - 1) / STRQ_CKPT_STEPS;
            const int tb = blk * STRQ_CKPT_STEPS;          // state after step tb is the restart point
            Lane<R> st;
            if (blk == 0) init_lane<R, PK>(tk, lane, st);
            else load_ckpt<R>(tk.ckpt + (size_t)(blk - 1) * (STRQ_CKPT_FIELDS(R) * 64), lane, st);
            // packed levels of the two columns this lane finished at step tb
            int qq = 0;
            {
                const int jB = 2 * (tb - lane), jA = jB - 1;
                int a = 0, b = 0;
                if (jA >= 1 && jA <= tk.n) a = tk.levels[jA - 1];
                if (jB >= 1 && jB <= tk.n) b = tk.levels[jB - 1];
                qq = (a * (PK ? 1 : 4)) | ((b * (PK ? 1 : 4)) << 16);
            }
            int qcur = load_chunk<PK>(tk, tb / 64, lane);
            Bnd4 bcur{0.0f, STRQ_NINF, 0.0f, STRQ_NINF}, bnext = bcur;
            if (has_in) bcur = load_bnd(tk, tb / 64, lane);
            for (int t0 = tb; t0 < tcur; t0 += 64) {
                const int qnext = load_chunk<PK>(tk, t0 / 64 + 1, lane);
                if (has_in) bnext = load_bnd(tk, t0 / 64 + 1, lane);
                const int send = tcur - t0 < 64 ? tcur - t0 : 64;
                for (int s = 0; s < send; ++s) {
                    const int t = t0 + s + 1;
                    qq = dpp_shr1_i(qq, __builtin_amdgcn_readlane(qcur, s));
                    const float fA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bcur.sA), s));
                    const float fB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bcur.sB), s));
                    const float fVA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bcur.vA), s));
                    const float fVB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bcur.vB), s));
                    const float upA = dpp_shr1_f(st.SbotA, fA);
                    const float upB = dpp_shr1_f(st.S[R - 1], fB);
                    const float upVA = dpp_shr1_f(st.VbotA, fVA);
                    const float upVB = dpp_shr1_f(st.VbotB, fVB);
                    const int jB = 2 * (t - lane), jA = jB - 1;
                    TraceWords w; w.a[0] = w.a[1] = w.b[0] = w.b[1] = 0;
                    if (jA >= 1 && jA <= tk.n) {
                        float scA[Shape<R, S>::C], scB[Shape<R, S>::C], rsA[R], rsB[R];
                        fetch_scores<R, S, PK>(ldsb, lc, qq & 0xffff, scA);
                        fetch_scores<R, S, PK>(ldsb, lc, (int)((unsigned)qq >> 16), scB);
                        expand_scores<R, S>(scA, pmu, rsA);
                        expand_scores<R, S>(scB, pmu, rsB);
                        dp_step2<R, false, false, true, true>(st, rsA, rsB, upA, upB, upVA, upVB, p, &w, 0, nullptr, nullptr);
                    }
                    uint64_t* dst = scratch + (size_t)(t - tb - 1) * STEP_WORDS;
#pragma unroll
                    for (int x = 0; x < W; ++x) {
                        dst[(0 * W + x) * 64 + lane] = w.a[x];
                        dst[(1 * W + x) * 64 + lane] = w.b[x];
                    }
                }
                qcur = qnext; bcur = bnext;
            }
            __threadfence();   // the walker below reads words written by other lanes of this wave
            // walk back while the current cell lies inside this strip and this block.  The walk is a serial
            // chase through the trace words, so they are fetched 64 at a time: the words of the walker's lane
            // position for the next 32 steps x 2 columns sit one per lane and every hop is a v_readlane.
            int pf_l = -1, pf_hi = 0, pf_lo = 0, pf_word = -1; uint64_t pf = 0;
            while (ci > tk.row0 && cj > 0) {
                const int il = ci - tk.row0;
                const int l = (il - 1) / R, r = (il - 1) % R;
                const int t = l + (cj + 1) / 2;
                if (t <= tb) break;
                const int colsel = (cj & 1) ? 0 : 1;
                if (l != pf_l || r / 16 != pf_word || t > pf_hi || t <= pf_lo) {
                    pf_l = l; pf_word = r / 16; pf_hi = t; pf_lo = t - 32 > tb ? t - 32 : tb;
                    const int st = t - (lane >> 1), cs = lane & 1;
                    pf = st > pf_lo ? scratch[(size_t)(st - tb - 1) * STEP_WORDS + (cs * W + pf_word) * 64 + l] : 0;
                }
                const int src = ((pf_hi - t) << 1) | colsel;
                const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pf, src);
                const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pf >> 32), src);
                const uint64_t w = ((uint64_t)whi << 32) | wlo;
                const uint32_t code = (uint32_t)(w >> trace_shift<R>(r)) & 15u;
                if (state == 0) {
                    if (code & 1u) { rec[ci - 1] = (cj + col_off) << 1; --ci; --cj; }      // the diagonal won
                    else state = (code & 2u) ? 1 : 2;                                      // H over V
                } else if (state == 1) {
                    --cj; if (!(code & 4u)) state = 0;
                } else {
                    rec[ci - 1] = ((cj + col_off) << 1) | 1;
                    --ci; if (!(code & 8u)) state = 0;
                }
            }
        }
        // column 0 is not free: whatever is left of the flank is a vertical run before a[0] (in a later
        // column segment: before the segment's first sample -- a path the full DP has as well)
        for (int i = ci - lane; i > 0; i -= 64) rec[i - 1] = (col_off << 1) | 1;
        results[ti].j0 = cj + col_off;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <int R, int S>
static int launch_shape(hipStream_t stream, const AlignTask* tasks, AlignResult* results, int n_tasks,
                        int* queue, const AlignParams& p, int lds_floats_per_wave, int waves_per_block,
                        int n_blocks, uint64_t* scratch, int phase, int mode, int packed, const int32_t* pick)
{
    const size_t lds_bytes = (size_t)lds_floats_per_wave * 4 * waves_per_block;
    const dim3 grid(n_blocks), block(64 * waves_per_block);
    const bool lh = p.open_h == p.ext_h, lv = p.open_v == p.ext_v;
#define STRQ_FWD1(LH_, LV_, MODE_)                                                                      \
    do {                                                                                                \
        (void)hipFuncSetAttribute((const void*)align_forward_kernel<R, S, LH_, LV_, MODE_, false>,      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);          \
        hipLaunchKernelGGL((align_forward_kernel<R, S, LH_, LV_, MODE_, false>), grid, block, lds_bytes, stream,\
                           tasks, results, n_tasks, queue, p, lds_floats_per_wave);                     \
    } while (0)
#define STRQ_FWD(LH_, LV_)                                                                              \
    do {                                                                                                \
        if (mode == 0) STRQ_FWD1(LH_, LV_, 0); else if (mode == 1) STRQ_FWD1(LH_, LV_, 1);              \
        else if (mode == 2) STRQ_FWD1(LH_, LV_, 2); else STRQ_FWD1(LH_, LV_, 3);                        \
    } while (0)
    if (packed && !(lh && lv && mode == 0)) return 2;      // packed tables: collapsed single-strip kernels only
    if (phase == 0 && packed) {
        // gap parameters in table units (2^-20): exact, a power-of-two scaling
        AlignParams ps = p;
        ps.open_h *= STRQ_PK_SCALE; ps.ext_h *= STRQ_PK_SCALE; ps.open_v *= STRQ_PK_SCALE; ps.ext_v *= STRQ_PK_SCALE;
        (void)hipFuncSetAttribute((const void*)align_forward_kernel<R, S, true, true, 0, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL((align_forward_kernel<R, S, true, true, 0, true>), grid, block, lds_bytes, stream,
                           tasks, results, n_tasks, queue, ps, lds_floats_per_wave);
    } else if (phase == 0) {
        if (lh && lv) STRQ_FWD(true, true);
        else if (lh) STRQ_FWD(true, false);
        else if (lv) STRQ_FWD(false, true);
        else STRQ_FWD(false, false);
    } else if (packed) {
        AlignParams ps = p;
        ps.open_h *= STRQ_PK_SCALE; ps.ext_h *= STRQ_PK_SCALE; ps.open_v *= STRQ_PK_SCALE; ps.ext_v *= STRQ_PK_SCALE;
        (void)hipFuncSetAttribute((const void*)align_trace_kernel<R, S, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL((align_trace_kernel<R, S, true>), grid, block, lds_bytes, stream, tasks, results,
                           n_tasks, queue, ps, lds_floats_per_wave, scratch, pick);
    } else {
        (void)hipFuncSetAttribute((const void*)align_trace_kernel<R, S, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL((align_trace_kernel<R, S, false>), grid, block, lds_bytes, stream, tasks, results,
                           n_tasks, queue, p, lds_floats_per_wave, scratch, pick);
    }
#undef STRQ_FWD
#undef STRQ_FWD1
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Shapes for `samples` other than 6: R = 2 * samples rows per lane (a lane holds exactly two k-mer classes: no
// class selects), 128 classes per strip.  Only the general affine kernels are instantiated for them -- the
// collapsed recurrence, column segments and 24-bit tables are optimisations of STRique's own configuration.
template <int R, int S>
static int launch_shape_general(hipStream_t stream, const AlignTask* tasks, AlignResult* results, int n_tasks,
                                int* queue, const AlignParams& p, int lds_floats_per_wave, int waves_per_block,
                                int n_blocks, uint64_t* scratch, int phase, int mode, int packed, const int32_t* pick)
{
    if (packed) return 2;
    const size_t lds_bytes = (size_t)lds_floats_per_wave * 4 * waves_per_block;
    const dim3 grid(n_blocks), block(64 * waves_per_block);
#define STRQ_FWDG(MODE_)                                                                                \
    do {                                                                                                \
        (void)hipFuncSetAttribute((const void*)align_forward_kernel<R, S, false, false, MODE_, false>,  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);          \
        hipLaunchKernelGGL((align_forward_kernel<R, S, false, false, MODE_, false>), grid, block, lds_bytes, stream,\
                           tasks, results, n_tasks, queue, p, lds_floats_per_wave);                     \
    } while (0)
    if (phase == 0) {
        if (mode == 0) STRQ_FWDG(0); else if (mode == 1) STRQ_FWDG(1); else if (mode == 2) STRQ_FWDG(2); else STRQ_FWDG(3);
    } else {
        (void)hipFuncSetAttribute((const void*)align_trace_kernel<R, S, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL((align_trace_kernel<R, S, false>), grid, block, lds_bytes, stream, tasks, results,
                           n_tasks, queue, p, lds_floats_per_wave, scratch, pick);
    }
#undef STRQ_FWDG
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

#define STRQ_SHAPES(X) X(6, 6) X(7, 6) X(8, 6) X(12, 6) X(14, 6) X(15, 6)
#define STRQ_GENERAL_SHAPES(X) X(2, 1) X(4, 2) X(6, 3) X(8, 4) X(10, 5) X(14, 7) X(16, 8) X(18, 9) X(20, 10) X(22, 11) X(26, 13)

// The run length the kernels work with: 6 when `samples` is a multiple of 6 (STRique's value), else the largest
// divisor of `samples` among the compiled run lengths -- a run of `samples` equal values is so many runs of that length.
int align_effective_samples(int samples)
{
    if (samples < 1) return 0;
    if (samples % 6 == 0) return 6;
    for (int d : {13, 11, 10, 9, 8, 7, 5, 4, 3, 2}) if (samples % d == 0) return d;
    return 1;
}

// Rows per lane R and number of strips for a flank of m rows.
int align_plan(int m, int samples, int* rows_per_lane, int* n_strips)
{
    if (samples < 1 || m < 1) return 0;
    if (samples != 6) {
        if (align_effective_samples(samples) != samples) return 0;       // callers pass the effective run length
        const int r = 2 * samples, rows = 64 * r;
        if (m > STRQ_MAX_STRIPS * rows) return 0;
        *rows_per_lane = r; *n_strips = (m + rows - 1) / rows;
        return r;
    }
    // Measured on MI355X (50 kb reads, 870-row flanks): two strips at two waves per SIMD take as long
    // as one strip at one wave per SIMD -- the per-step overhead is amortised over half the rows --
    // so one strip is preferred whenever a single-strip shape fits.  STRQ_STRIPS=2 forces two.
    const char* e = strq::opt("STRQ_STRIPS");
    const bool force_two = e && e[0] == '2';
    // 14 rows per lane: STRique's 870-row flanks keep 63 of 64 lanes busy (15 rows: 58).  STRQ_NO_R14 leaves the shape
    // out (A/B runs).
    const bool no14 = strq::opt("STRQ_NO_R14") != nullptr;          // read on every call, like STRQ_STRIPS
    const int single[] = {6, 7, 8, 12, 14, 15};
    const int two[] = {6, 7, 8, 12};
    // Flanks of 129 ... 149 classes fit 14 and 15 rows per lane; 14 keeps more lanes busy, and since round 4 the register the
    // last flank row sits in is a compile-time constant of the steady-state loop for every flank length (forward_one, RMSW),
    // so nothing speaks for 15 any more (round 3 chose it when only 15 | m had the register: 191 against 157 instructions).
    if (!force_two)
        for (int r : single) if (64 * r >= m && !(r == 14 && no14)) { *rows_per_lane = r; *n_strips = 1; return r; }
    for (int r : two) if (64 * r < m && 128 * r >= m) { *rows_per_lane = r; *n_strips = 2; return r; }
    for (int r : single) if (64 * r >= m && !(r == 14 && no14)) { *rows_per_lane = r; *n_strips = 1; return r; }
    // longer flanks: strips of 64 x 12 rows (128 k-mer classes each, so that every strip's score table fits the LDS
    // of the kernel that builds it), top to bottom with the strip's last row handed on through HBM
    if (m <= STRQ_MAX_STRIPS * 64 * 12) { *rows_per_lane = 12; *n_strips = (m + 64 * 12 - 1) / (64 * 12); return 12; }
    return 0;
}

size_t align_trace_scratch_words_per_wave(int R)
{
    return (size_t)STRQ_CKPT_STEPS * 2 * STRQ_TRACE_WORDS(R) * 64;
}

int launch_align(hipStream_t stream, int R, int S, const AlignTask* tasks, AlignResult* results,
                 int n_tasks, int* queue, const AlignParams& p, int lds_floats_per_wave,
                 int waves_per_block, int n_blocks, uint64_t* scratch, int phase, int mode, int packed,
                 const int32_t* pick)
{
#define STRQ_CASE(R_, S_)                                                                               \
    if (R == R_ && S == S_)                                                                             \
        return launch_shape<R_, S_>(stream, tasks, results, n_tasks, queue, p, lds_floats_per_wave,     \
                                    waves_per_block, n_blocks, scratch, phase, mode, packed, pick);
    STRQ_SHAPES(STRQ_CASE)
#undef STRQ_CASE
#define STRQ_CASE(R_, S_)                                                                               \
    if (R == R_ && S == S_)                                                                             \
        return launch_shape_general<R_, S_>(stream, tasks, results, n_tasks, queue, p, lds_floats_per_wave, \
                                            waves_per_block, n_blocks, scratch, phase, mode, packed, pick);
    STRQ_GENERAL_SHAPES(STRQ_CASE)
#undef STRQ_CASE
    return 2;
}

int align_segment_overlap(const AlignParams& p, int m)
{
    // horizontal steps cost at least c_h each, a diagonal step gains at most dist_offset, vertical steps
    // cost: a path that scores >= 0 has h * c_h <= m * dist_offset horizontal steps, i.e. spans at most
    // m + h columns.  1 % + 64 columns cover the float32 rounding of the running sums (<= 2^-11 relative
    // per addition on values below 2^14 * dist_offset / 16).
    if (!(p.dist_min >= 0.0f) || !(p.open_h < 0.0f) || !(p.ext_h < 0.0f) || !(p.dist_offset > 0.0f)) return 0;
    if (!(p.open_v <= 0.0f) || !(p.ext_v <= 0.0f)) return 0;
    const double c_h = -(double)(p.open_h > p.ext_h ? p.open_h : p.ext_h);
    // a cell scores max(dist_offset - d^1.2, dist_min): at most the larger of the two
    const double gain = (double)(p.dist_min > p.dist_offset ? p.dist_min : p.dist_offset);
    const double h = (double)m * gain / c_h;
    const double L = (double)m + h * 1.01 + 64.0;
    if (!(L < 1e8)) return 0;
    return (int)L + 1;
}

template <int R, int S, bool PK, int SEG, int WPE, bool KNOWN>
static int launch_seg2(hipStream_t stream, const AlignTask* tasks, AlignResult* seg_results, int n_groups, int* queue,
                       const AlignParams& p, int lds_dwords, int n_blocks, const int* group_list, const int* n_list)
{
    const size_t lds_bytes = (size_t)lds_dwords * 4;
    if (group_list && n_list) {
        (void)hipFuncSetAttribute((const void*)align_forward_seg_kernel<R, S, PK, SEG, WPE, true, KNOWN>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL((align_forward_seg_kernel<R, S, PK, SEG, WPE, true, KNOWN>), dim3(n_blocks), dim3(64 * SEG), lds_bytes, stream,
                           tasks, seg_results, n_groups, queue, p, group_list, n_list);
    } else {
        (void)hipFuncSetAttribute((const void*)align_forward_seg_kernel<R, S, PK, SEG, WPE, false, KNOWN>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL((align_forward_seg_kernel<R, S, PK, SEG, WPE, false, KNOWN>), dim3(n_blocks), dim3(64 * SEG), lds_bytes, stream,
                           tasks, seg_results, n_groups, queue, p, group_list, n_list);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// known_last_row: every flank of the launch has m % R == 0 or (R == 14) m == 870 -- the round-3 kernel body; compiled for 14 rows
// per lane only (STRique's flank length), every other shape runs the switched loop, which covers m % R == 0 as one of its cases
template <int R, int S, bool PK, int SEG, int WPE>
static int launch_seg1(hipStream_t stream, const AlignTask* tasks, AlignResult* seg_results, int n_groups, int* queue,
                       const AlignParams& p, int lds_dwords, int n_blocks, const int* group_list, const int* n_list, bool known_last_row)
{
    if constexpr (R == 14) {
        if (known_last_row) return launch_seg2<R, S, PK, SEG, WPE, true>(stream, tasks, seg_results, n_groups, queue, p, lds_dwords, n_blocks, group_list, n_list);
    }
    return launch_seg2<R, S, PK, SEG, WPE, false>(stream, tasks, seg_results, n_groups, queue, p, lds_dwords, n_blocks, group_list, n_list);
}

// compiled (segments, waves per SIMD) pairs
#define STRQ_SEG_CONFIGS(X, R_, S_, PK_) X(R_, S_, PK_, 1, 2) X(R_, S_, PK_, 2, 3) X(R_, S_, PK_, 2, 4) X(R_, S_, PK_, 3, 3) X(R_, S_, PK_, 4, 4) X(R_, S_, PK_, 3, 4) X(R_, S_, PK_, 4, 3)

// the WPE template argument a (waves per table, tables per CU) launch runs on
int align_segments_wpe(int segs, int tables_per_cu)
{
    const int waves = tables_per_cu * segs;
    int wpe = (waves + 3) / 4; if (wpe < 2) wpe = 2;
    if (segs == 1) wpe = 2;
    // WPE only caps the registers of a wave (amdgpu_waves_per_eu); the smallest compiled value for several waves per
    // table is 3, and a launch with fewer resident waves (large tables: one or two per CU) runs on it unchanged
    else if (wpe < 3) wpe = 3;
    if (wpe > 4) wpe = 4;
    return wpe;
}

int launch_align_segments(hipStream_t stream, int R, int S, const AlignTask* tasks, AlignResult* seg_results,
                          int n_groups, int segs, int* queue, const AlignParams& p, int lds_dwords,
                          int tables_per_cu, int n_cu, int packed, const int* group_list, const int* n_list, bool known_last_row)
{
    if (!(p.open_h == p.ext_h && p.open_v == p.ext_v)) return 2;
    AlignParams ps = p;
    if (packed) { ps.open_h *= STRQ_PK_SCALE; ps.ext_h *= STRQ_PK_SCALE; ps.open_v *= STRQ_PK_SCALE; ps.ext_v *= STRQ_PK_SCALE; }
    const int wpe = align_segments_wpe(segs, tables_per_cu);
    const int n_blocks = tables_per_cu * n_cu;
#define STRQ_SEGCASE(R_, S_, PK_, SEG_, WPE_)                                                             \
    if (R == R_ && S == S_ && (packed != 0) == PK_ && segs == SEG_ && wpe == WPE_)                        \
        return launch_seg1<R_, S_, PK_, SEG_, WPE_>(stream, tasks, seg_results, n_groups, queue, ps, lds_dwords, n_blocks, group_list, n_list, known_last_row);
#define STRQ_SEGSHAPE(R_, S_) STRQ_SEG_CONFIGS(STRQ_SEGCASE, R_, S_, true) STRQ_SEG_CONFIGS(STRQ_SEGCASE, R_, S_, false)
    STRQ_SHAPES(STRQ_SEGSHAPE)
#undef STRQ_SEGSHAPE
#undef STRQ_SEGCASE
    return 2;
}

int launch_align_combine(hipStream_t stream, const AlignTask* tasks, const AlignResult* seg_results, int n_align,
                         int segs, AlignResult* results, int32_t* pick, int pick_base, const int* list, const int* n_list,
                         const float* min_score, int* redo, int* redo_count, unsigned int* redo_total)
{
    if (n_align <= 0) return 0;
    hipLaunchKernelGGL(align_combine_kernel, dim3((n_align + 255) / 256), dim3(256), 0, stream, tasks, seg_results, n_align, segs,
                       results, pick, pick_base, list, n_list, min_score, redo, redo_count, redo_total);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// The coarse screen's second look cuts an alignment's windows into groups of four pieces (one workgroup each): the best of an
// alignment's groups [first[a], first[a + 1]) -- the leftmost on ties, groups are in column order -- goes to the alignment's slot.
__global__ void align_scatter_kernel(const AlignResult* __restrict__ src, const int32_t* __restrict__ pick_src, const int32_t* __restrict__ first,
                                     const int32_t* __restrict__ pos, int n, AlignResult* __restrict__ dst, int32_t* __restrict__ pick_dst)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    int best = first[a];
    for (int g = first[a] + 1; g < first[a + 1]; ++g) if (src[g].best > src[best].best) best = g;
    dst[pos[a]] = src[best]; pick_dst[pos[a]] = pick_src[best];
}

int launch_align_scatter(hipStream_t stream, const AlignResult* src, const int32_t* pick_src, const int32_t* first, const int32_t* pos, int n,
                         AlignResult* dst, int32_t* pick_dst)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(align_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, pick_src, first, pos, n, dst, pick_dst);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int align_overlap_for_score(const AlignParams& p, int m, float score)
{
    // align_segment_overlap with the bound B = score instead of 0; never more than the worst case
    const int worst = align_segment_overlap(p, m);
    if (worst <= 0 || !(score > 0.0f)) return worst;
    const double c_h = -(double)(p.open_h > p.ext_h ? p.open_h : p.ext_h);
    const double gain = (double)(p.dist_min > p.dist_offset ? p.dist_min : p.dist_offset);
    double h = ((double)m * gain - (double)score + 1.0) / c_h;
    if (h < 0) h = 0;
    const double L = (double)m + h * 1.01 + 66.0;
    return L + 1 < (double)worst ? (int)L + 1 : worst;
}

float align_segment_min_score(const AlignParams& p, int m, int overlap_used)
{
    // inverse of align_segment_overlap for paths that score at least B: span <= m + (m * dist_offset - B) / c_h (+ 1 % + 64)
    const double c_h = -(double)(p.open_h > p.ext_h ? p.open_h : p.ext_h);
    const double h_ok = ((double)overlap_used - (double)m - 65.0) / 1.01;
    const double gain = (double)(p.dist_min > p.dist_offset ? p.dist_min : p.dist_offset);
    const double b = (double)m * gain - h_ok * c_h;
    return (float)(b + 1.0 + 1e-6 * (b < 0 ? -b : b));       // rounded up
}

}  // namespace strq
