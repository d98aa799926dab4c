// Internal context of libstrique_hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <cstdio>
#include "align_kernels.h"
#include "viterbi_kernels.h"

namespace strq {

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct HostModel {
    VitModel h;              // host copy (pointers are device pointers)
    const VitModel* dev = nullptr;
    DevBuf blob;
};

}  // namespace strq

struct strq_ctx {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    strq::AlignParams ap{-2.0f, -8.0f, -2.0f, -8.0f, 8.0f, -16.0f};   // src/align_raw.h:51-60
    std::string err;
    float timing[8] = {};
    // workspace
    strq::DevBuf levels, level_val, flank_cls, tables, band_lo, col0, ckpt, rec, tasks, results,
        queue, scratch, lutinfo, hard, misc, vit_x, vit_tasks, vit_bp, vit_path;
    std::vector<strq::HostModel*> models;
    size_t max_ws_bytes = (size_t)48 << 30;   // cap for checkpoint workspace per sub-batch
};

#define STRQ_HIP(ctx, call)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return STRQ_ERR_DEVICE;                                                            \
        }                                                                                      \
    } while (0)
