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
// value vector of the previous/current time step lives in this wave's LDS slice and every lane
// gathers its predecessors from it.  The silent states form a DAG (delete chains ~50 long); instead
// of walking it serially, all silent states are relaxed together until nothing changes.  Max-plus on
// a DAG has a unique fixed point and every relaxation evaluates the same `v[k] + A` sums, so the
// fixed point is bit-identical to the serial topological pass, including the argmax of the final
// sweep; a typical time step needs 2-3 sweeps because long delete chains are improbable.
//
// Repeat counting does not need a traceback: the number of visits of the counted states
// (the two `dummy` states of repeatHMM, STRique.py:374-378) is carried along the best path.
// Back-pointers are written only when the caller wants the state path (modification pass).
#include <hip/hip_runtime.h>
#include <stdint.h>
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

template <int EPL, int SPL, bool BP>
__global__ void __launch_bounds__(1024)
viterbi_kernel(const VitModel* __restrict__ mp, const VitTask* __restrict__ tasks, VitResult* __restrict__ results,
               int n_tasks, int* __restrict__ queue)
{
    extern __shared__ double lds_d[];
    const VitModel& M = *mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int n = M.n_states, ne = M.n_emit, ns = M.n_silent;
    const int NP = (n + 2) & ~1;                    // value cells per buffer incl. the -inf cell v[n]
    const int nrows = M.n_edge_rows;
    double* e_lp = lds_d;                                                   // nrows*64
    double* vbase = e_lp + (size_t)nrows * 64 + (size_t)wave * 2 * NP;       // two value buffers per wave
    int* e_src = reinterpret_cast<int*>(e_lp + (size_t)nrows * 64 + (size_t)nw * 2 * NP);
    int* cbase = e_src + (size_t)nrows * 64 + (size_t)wave * 2 * NP;
    for (int i = threadIdx.x; i < nrows * 64; i += blockDim.x) { e_lp[i] = M.edge_logp[i]; e_src[i] = M.edge_src[i]; }
    __syncthreads();

    // per-lane emission parameters and count increments
    int ekind[EPL]; double ea[EPL], eb[EPL], ec[EPL]; int einc[EPL], sinc[SPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int e = s * 64 + lane;
        ekind[s] = (s < M.epl) ? M.emis_kind[e] : 0;
        ea[s] = (s < M.epl) ? M.emis_a[e] : 0.0; eb[s] = (s < M.epl) ? M.emis_b[e] : 0.0; ec[s] = (s < M.epl) ? M.emis_c[e] : 0.0;
        einc[s] = (e < ne) ? M.count_inc[e] : 0;
    }
#pragma unroll
    for (int s = 0; s < SPL; ++s) { const int q = s * 64 + lane; sinc[s] = (q < ns) ? M.count_inc[ne + q] : 0; }
    const double NEGINF = -__builtin_inf();

    for (;;) {
        const int ti = vit_next_task(queue, lane);
        if (ti >= n_tasks) break;
        const VitTask& tk = tasks[ti];
        const int64_t T = tk.T;
        double* vcur = vbase; double* vnxt = vbase + NP;
        int* ccur = cbase; int* cnxt = cbase + NP;
        for (int i = lane; i < NP; i += 64) { vcur[i] = NEGINF; vnxt[i] = NEGINF; ccur[i] = 0; cnxt[i] = 0; }
        VIT_FENCE();
        if (lane == 0) vcur[M.start] = 0.0;
        VIT_FENCE();

        // relax all silent states of buffer `vb` (values) / `cb` (carried counts) to their fixed
        // point; `pin`: keep start at 0 (t = 0).  Counts ride along: a state's count is final one
        // sweep after its predecessor's, exactly like its value.
        auto relax_silent = [&](double* vb, int* cb, bool pin, int64_t trow) {
            double oldv[SPL]; int oldc[SPL]; int arg[SPL];
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int q = s * 64 + lane;
                oldv[s] = (s < M.spl && q < ns) ? vb[ne + q] : NEGINF;
                oldc[s] = (s < M.spl && q < ns) ? cb[ne + q] : 0;
                arg[s] = n;
            }
            for (;;) {
                bool changed = false;
                double newv[SPL]; int newc[SPL];
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    newv[s] = NEGINF; newc[s] = 0;
                    if (s >= M.spl) continue;
                    const int q = s * 64 + lane, l = ne + q;
                    double best = NEGINF; int a = n;
                    const int base = M.s_base[s], deg = M.s_deg[s];
                    for (int j = 0; j < deg; ++j) {
                        const int src = e_src[(base + j) * 64 + lane];
                        const double c = vb[src] + e_lp[(base + j) * 64 + lane];
                        if (c > best) { best = c; a = src; }
                    }
                    if (pin && l == M.start) { best = 0.0; a = n; }
                    if (q >= ns) { best = NEGINF; a = n; }
                    newv[s] = best; arg[s] = a; newc[s] = cb[a] + sinc[s];
                    if (q < ns && (!(best == oldv[s]) || newc[s] != oldc[s])) changed = true;
                }
                if (!__any(changed)) break;
                VIT_FENCE();
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    const int q = s * 64 + lane;
                    if (s < M.spl && q < ns) { vb[ne + q] = newv[s]; cb[ne + q] = newc[s]; }
                    oldv[s] = newv[s]; oldc[s] = newc[s];
                }
                VIT_FENCE();
            }
            if (BP) {
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    const int q = s * 64 + lane;
                    if (s < M.spl && q < ns) tk.bp[(size_t)trow * n + ne + q] = (uint16_t)arg[s];
                }
            }
        };

        relax_silent(vcur, ccur, true, 0);

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
            for (int s0 = 0; s0 < send; ++s0) {
                const double x = readlane_f64(xchunk, s0);
                const int64_t t = t0 + s0;
#pragma unroll
                for (int s = 0; s < EPL; ++s) {
                    if (s >= M.epl) continue;
                    const int e = s * 64 + lane;
                    double best = NEGINF; int a = n;
                    const int base = M.e_base[s], deg = M.e_deg[s];
                    for (int j = 0; j < deg; ++j) {
                        const int src = e_src[(base + j) * 64 + lane];
                        const double c = vcur[src] + e_lp[(base + j) * 64 + lane];
                        if (c > best) { best = c; a = src; }
                    }
                    double em;
                    if (ekind[s] == 1) { const double d = x - ea[s]; em = ec[s] - (d * d) * eb[s]; }
                    else if (ekind[s] == 2) em = (x >= ea[s] && x <= eb[s]) ? ec[s] : NEGINF;
                    else em = NEGINF;
                    if (e < ne) {
                        vnxt[e] = best + em;
                        cnxt[e] = ccur[a] + einc[s];
                        if (BP) tk.bp[(size_t)(t + 1) * n + e] = (uint16_t)a;
                    }
                }
#pragma unroll
                for (int s = 0; s < SPL; ++s) { const int q = s * 64 + lane; if (s < M.spl && q < ns) { vnxt[ne + q] = NEGINF; cnxt[ne + q] = 0; } }
                VIT_FENCE();
                relax_silent(vnxt, cnxt, false, t + 1);
                double* tv = vcur; vcur = vnxt; vnxt = tv;
                int* tc = ccur; ccur = cnxt; cnxt = tc;
            }
        }
        const double lp = vcur[M.end];
        const int cnt = ccur[M.end];
        VitResult r; r.logp = lp; r.counted = cnt; r.status = (lp > NEGINF) ? 0 : 1; r.pad_ = 0;
        results[ti] = r;     // every lane stores the same value
        VIT_FENCE();
    }
}

// one thread per task: follow the back-pointers from (T, end) and write the emitting state of
// every observation (first version: latency-bound, used by the modification pass and the parity API)
__global__ void vit_traceback_kernel(const VitModel* __restrict__ mp, const VitTask* __restrict__ tasks,
                                     const VitResult* __restrict__ results, int32_t* const* __restrict__ paths, int n_tasks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tasks) return;
    const VitModel& M = *mp;
    const VitTask& tk = tasks[i];
    if (results[i].status != 0 || !tk.bp || !paths[i]) return;
    int64_t t = tk.T; int l = M.end;
    const int n = M.n_states, ne = M.n_emit;
    int guard = 0;
    while (!(t == 0 && l == M.start)) {
        const int prev = tk.bp[(size_t)t * n + l];
        if (prev >= n) break;
        if (l < ne) { paths[i][t - 1] = l; --t; guard = 0; }
        else if (++guard > n) break;
        l = prev;
    }
}

int launch_viterbi(hipStream_t stream, const VitModel& mh, const VitModel* model_dev, const VitTask* tasks,
                   VitResult* results, int n_tasks, int* queue, int n_cu, int want_bp)
{
    const int NP = (mh.n_states + 2) & ~1;
    // one block per CU: the edge lists are shared, every wave adds two value/count buffers
    int nw = 16;
    while (nw > 1 && ((size_t)mh.n_edge_rows * 64 + (size_t)nw * 2 * NP) * 12 > 160 * 1024) nw >>= 1;
    const size_t lds = ((size_t)mh.n_edge_rows * 64 + (size_t)nw * 2 * NP) * (8 + 4);
    if (lds > 160 * 1024) return 3;
    const dim3 grid(n_cu), block(64 * nw);
#define VIT_LAUNCH(E_, S_)                                                                                  \
    do {                                                                                                    \
        if (want_bp) {                                                                                      \
            (void)hipFuncSetAttribute((const void*)viterbi_kernel<E_, S_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((viterbi_kernel<E_, S_, true>), grid, block, lds, stream, model_dev, tasks, results, n_tasks, queue); \
        } else {                                                                                            \
            (void)hipFuncSetAttribute((const void*)viterbi_kernel<E_, S_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((viterbi_kernel<E_, S_, false>), grid, block, lds, stream, model_dev, tasks, results, n_tasks, queue); \
        }                                                                                                   \
    } while (0)
    const int e = mh.epl, s = mh.spl;
    if (e <= 1 && s <= 1) VIT_LAUNCH(1, 1);
    else if (e <= 2 && s <= 2) VIT_LAUNCH(2, 2);
    else if (e <= 4 && s <= 2) VIT_LAUNCH(4, 2);
    else if (e <= 4 && s <= 4) VIT_LAUNCH(4, 4);
    else if (e <= 8 && s <= 4) VIT_LAUNCH(8, 4);
    else return 2;
#undef VIT_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_vit_traceback(hipStream_t stream, const VitModel* model_dev, const VitTask* tasks, const VitResult* results,
                         int32_t* const* paths, int n_tasks)
{
    if (n_tasks <= 0) return 0;
    hipLaunchKernelGGL(vit_traceback_kernel, dim3((n_tasks + 63) / 64), dim3(64), 0, stream, model_dev, tasks, results, paths, n_tasks);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq
