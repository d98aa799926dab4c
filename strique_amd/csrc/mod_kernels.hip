// Base-modification pass of repeatCounter.detect (reference scripts/STRique.py:605-609, 492-500):
//   nrm  = pm.normalize2model(raw, 'minmax')                      raw, unfiltered signal
//   rep  = nrm[prefix_begin:suffix_end][ 'repeat' in state ]      samples decoded into repeat states
//   path = modHMM.viterbi(clip(rep, model_min, model_max))
//   one character per run of non-hub states: '1' if its first state is in the modified branch
// Two streaming kernels around the Viterbi kernels, one wave64 per read: 64 samples per iteration,
// kept elements compacted with a ballot + lane prefix count (coalesced reads and writes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mod_kernels.h"
#include "viterbi_kernels.h"

namespace strq {

static __device__ __forceinline__ int lane_prefix(uint64_t mask)      // set bits below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__global__ void __launch_bounds__(256)
mod_compact_kernel(const ModTask* __restrict__ tasks, int n, int64_t* __restrict__ out_len)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const ModTask tk = tasks[i];
    int64_t k = 0;
    if (tk.T > 0) {
        for (int64_t t0 = 0; t0 < tk.T; t0 += 64) {
            const int64_t t = t0 + lane;
            bool keep = false; double v = 0.0;
            if (t < tk.T) {
                keep = tk.path ? tk.tag[tk.path[t]] == 1 : true;      // no path: the task is already the contiguous repeat stretch
                v = tk.is_f64 ? reinterpret_cast<const double*>(tk.raw)[t] : (double)reinterpret_cast<const int16_t*>(tk.raw)[t];
                v = (v - tk.c1) / tk.h1;
                v = v * tk.h2 + tk.c2;
                v = v < tk.clip_lo ? tk.clip_lo : v;  v = v > tk.clip_hi ? tk.clip_hi : v;     // normalize2model clip
                v = v < tk.mod_lo ? tk.mod_lo : v;    v = v > tk.mod_hi ? tk.mod_hi : v;       // mod_repeats clip
            }
            const uint64_t m = __ballot(keep);
            if (keep) tk.out[k + lane_prefix(m)] = v;
            k += __popcll(m);
        }
    }
    if (lane == 0) out_len[i] = k;
}

__global__ void __launch_bounds__(256)
mod_pattern_kernel(const PatTask* __restrict__ tasks, int n, int64_t* __restrict__ out_len)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const PatTask tk = tasks[i];
    if (*tk.status != 0) { if (lane == 0) { if (tk.out) tk.out[0] = '-'; out_len[i] = tk.out ? 1 : 0; } return; }
    int64_t k = 0;
    uint64_t carry = 1;                 // "previous state was a hub" for the first sample
    for (int64_t t0 = 0; t0 < tk.T; t0 += 64) {
        const int64_t t = t0 + lane;
        int tg = 2;                     // past the end: treated as hub, starts nothing
        if (t < tk.T) tg = tk.tag[tk.path[t]];
        const uint64_t hubs = __ballot(tg == 2);
        const uint64_t prev = (hubs << 1) | carry;             // bit l = sample l-1 was a hub
        const bool start = t < tk.T && tg != 2 && ((prev >> lane) & 1);
        const uint64_t m = __ballot(start);
        if (start) tk.out[k + lane_prefix(m)] = tg == 1 ? '1' : '0';
        k += __popcll(m);
        carry = hubs >> 63;
    }
    if (lane == 0) out_len[i] = k;
}

// one thread per read: hop from hub record to hub record (one hop per repeat unit), last unit first
__global__ void __launch_bounds__(64)
mod_hub_pattern_kernel(const HubTask* __restrict__ tasks, int n, int64_t* __restrict__ out_len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const HubTask tk = tasks[i];
    const VitResult* r = reinterpret_cast<const VitResult*>(tk.result);
    if (r->status != 0) { tk.out[0] = '-'; out_len[i] = 1; return; }
    int64_t cnt = 0;
    for (uint32_t t = r->dbg[0]; t != 0; t = (uint32_t)tk.rec[t]) ++cnt;
    int64_t k = cnt;
    for (uint32_t t = r->dbg[0]; t != 0;) {
        const uint64_t v = tk.rec[t];
        tk.out[--k] = (v >> 32) ? '1' : '0';
        t = (uint32_t)v;
    }
    out_len[i] = cnt;
}

// The pattern kernels write the characters of read k at an offset sized for the worst case (one per time step); the strings
// themselves are ~1000 characters.  Gather them into a dense pool before the read-back: 4 MB instead of ~180 MB per 4096 reads.
__global__ void mod_gather_kernel(const GatherTask* __restrict__ tasks, int n, const char* __restrict__ src, char* __restrict__ dst)
{
    const int k = blockIdx.x;
    if (k >= n) return;
    const GatherTask t = tasks[k];
    for (int64_t i = threadIdx.x; i < t.len; i += blockDim.x) dst[t.dst + i] = src[t.src + i];
}

int launch_mod_gather(hipStream_t s, const GatherTask* tasks, int n, const char* src, char* dst)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mod_gather_kernel, dim3(n), dim3(128), 0, s, tasks, n, src, dst);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_mod_hub_pattern(hipStream_t s, const HubTask* tasks, int n, int64_t* out_len)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mod_hub_pattern_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tasks, n, out_len);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_mod_compact(hipStream_t s, const ModTask* tasks, int n, int64_t* out_len)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mod_compact_kernel, dim3((n + 3) / 4), dim3(256), 0, s, tasks, n, out_len);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_mod_pattern(hipStream_t s, const PatTask* tasks, int n, int64_t* out_len)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mod_pattern_kernel, dim3((n + 3) / 4), dim3(256), 0, s, tasks, n, out_len);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq
