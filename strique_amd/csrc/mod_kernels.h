// Internal: modification-pass helper kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace strq {

struct ModTask {
    const int32_t* path;     // emitting state of every sample of the window (flanked model); null = keep every sample
    const int32_t* tag;      // state tags of that model (1 = repeat section)
    const void* raw;         // raw signal at prefix_begin
    double* out;             // compacted, normalised, clipped samples
    int64_t T;
    int32_t is_f64, pad_;
    double c1, h1, h2, c2, clip_lo, clip_hi, mod_lo, mod_hi;
};
struct PatTask {
    const int32_t* path;     // emitting states of the modification model
    const int32_t* tag;      // 2 = hub (s0/e0), 1 = modified branch
    char* out;
    int64_t T;
    const int32_t* status;   // status of the modification-model Viterbi of this read (device): 0 = a path exists
};

// hub records of the modification-model Viterbi (viterbi_kernels.hip, HUB) -> one character per repeat unit
struct HubTask {
    const uint64_t* rec;     // T + 1 records: low 32 bits = time of the previous e0 emission, high = branch (1 = modified)
    const void* result;      // VitResult of the window (device): status, dbg[0] = time of the last e0 emission
    char* out;
};
struct GatherTask { int64_t src, dst, len; };      // bytes [src, src + len) of the sparse pattern buffer -> [dst, dst + len) of the dense pool
int launch_mod_gather(hipStream_t s, const GatherTask* tasks, int n, const char* src, char* dst);
int launch_mod_hub_pattern(hipStream_t s, const HubTask* tasks, int n, int64_t* out_len);
int launch_mod_compact(hipStream_t s, const ModTask* tasks, int n, int64_t* out_len);
int launch_mod_pattern(hipStream_t s, const PatTask* tasks, int n, int64_t* out_len);

}  // namespace strq
