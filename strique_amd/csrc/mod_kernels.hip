// Base-modification pass of repeatCounter.detect (reference scripts/STRique.py:605-609, 492-500):
//   nrm  = pm.normalize2model(raw, 'minmax')                      raw, unfiltered signal
//   rep  = nrm[prefix_begin:suffix_end][ 'repeat' in state ]      samples decoded into repeat states
//   path = modHMM.viterbi(clip(rep, model_min, model_max))
//   one character per run of non-hub states: '1' if its first state is in the modified branch
// Two small streaming kernels around the Viterbi kernels; one thread per read (the per-read work is
// a sequential compaction over the window).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mod_kernels.h"

namespace strq {

__global__ void mod_compact_kernel(const ModTask* __restrict__ tasks, int n, int64_t* __restrict__ out_len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ModTask tk = tasks[i];
    int64_t k = 0;
    if (tk.path && tk.T > 0) {
        for (int64_t t = 0; t < tk.T; ++t) {
            if (tk.tag[tk.path[t]] != 1) continue;
            double v = tk.is_f64 ? reinterpret_cast<const double*>(tk.raw)[t] : (double)reinterpret_cast<const int16_t*>(tk.raw)[t];
            v = (v - tk.c1) / tk.h1;
            v = v * tk.h2 + tk.c2;
            v = v < tk.clip_lo ? tk.clip_lo : v;  v = v > tk.clip_hi ? tk.clip_hi : v;     // normalize2model clip
            v = v < tk.mod_lo ? tk.mod_lo : v;    v = v > tk.mod_hi ? tk.mod_hi : v;       // mod_repeats clip
            tk.out[k++] = v;
        }
    }
    out_len[i] = k;
}

__global__ void mod_pattern_kernel(const PatTask* __restrict__ tasks, int n, int64_t* __restrict__ out_len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PatTask tk = tasks[i];
    int64_t k = 0;
    if (!tk.ok) { if (tk.out) tk.out[k++] = '-'; out_len[i] = k; return; }
    bool prev_hub = true;
    for (int64_t t = 0; t < tk.T; ++t) {
        const int tg = tk.tag[tk.path[t]];
        const bool hub = tg == 2;
        if (!hub && prev_hub) tk.out[k++] = tg == 1 ? '1' : '0';
        prev_hub = hub;
    }
    out_len[i] = k;
}

int launch_mod_compact(hipStream_t s, const ModTask* tasks, int n, int64_t* out_len)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mod_compact_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tasks, n, out_len);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_mod_pattern(hipStream_t s, const PatTask* tasks, int n, int64_t* out_len)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mod_pattern_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tasks, n, out_len);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace strq
