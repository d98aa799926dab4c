// Internal: generic align_overlap path (not part of the C ABI), see align_generic.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "align_kernels.h"

namespace strq {

struct GenericHard { int32_t ia, ib; };      // table entry whose pow needs the host libm

struct GenericAlignArgs {
    const uint32_t* code_a;     // n dictionary codes of the read samples
    const uint32_t* code_b;     // m dictionary codes of the flank samples
    const float* table;         // nb x na scores: table[code_b * na + code_a]
    const float* col0;          // m + 1: S[i][0]
    float* bnd_S[2]; float* bnd_V[2];      // n + 1 each: last row of a strip, ping-pong between strips
    uint8_t* trace;             // (m + 1) x (n + 1) trace bytes, row major
    AlignResult* result;
    AlignParams p;
    int32_t n, m, na, nb;
};

int launch_generic_table(hipStream_t st, const float* va, int na, const float* vb, int nb, float* table,
                         GenericHard* hard, int* hard_count, int hard_cap, const AlignParams& p);
int launch_generic_patch(hipStream_t st, float* table, int na, const GenericHard* hard, const float* vals, int n);
int launch_generic_align(hipStream_t st, const GenericAlignArgs& a);

}  // namespace strq
