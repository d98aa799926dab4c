// Internal: score-table build kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "align_kernels.h"

#define STRQ_LUT_MAX_K 158          // classes per flank: k x 256 floats + bookkeeping must fit the 160 KB LDS build buffer
#define STRQ_LUT_LOCAL_HARD 64
#define STRQ_TABLE_SLOT_FLOATS(k) ((size_t)(k) * 256)

namespace strq {

struct LutJob {
    const float* level_val;   // 256 level values of the read
    const float* cls_val;     // k class values of the flank
    float* table;             // slot of STRQ_TABLE_SLOT_FLOATS(k) floats
    uint8_t* table3;          // slot of 3 * STRQ_TABLE_SLOT_FLOATS(k) + 8 bytes: the same entries as 24-bit fixed point (16-bit plane, 8-bit plane)
    int32_t* band_lo;         // k packed row descriptors (see AlignTask::band_lo)
    int32_t k, pad_;
};
struct LutInfo { int32_t total, n_hard, need, packed; };   // total: entries of the ragged table; need: widest row; packed: table3 is exact
struct HardEntry { int32_t job, k, level, index; };

int launch_lut_build(hipStream_t stream, const LutJob* jobs, LutInfo* info, int n_jobs, int max_k,
                     HardEntry* hard, int* hard_count, int hard_cap, const AlignParams& p);
int launch_lut_patch(hipStream_t stream, const LutJob* jobs, const HardEntry* hard, const float* vals, int n);

}  // namespace strq
