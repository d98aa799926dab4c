// gfx950 issue rates of the packed 16-bit integer instructions the upper-bound screen is made of (csrc/screen_kernels.hip),
// next to their 32-bit counterparts:  hipcc --offload-arch=gfx950 -O3 tools/ubench_pk16.hip -o /tmp/ubench_pk16 && /tmp/ubench_pk16
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(s) asm volatile(s " %0, %0, %8\n\t" s " %1, %1, %8\n\t" s " %2, %2, %8\n\t" s " %3, %3, %8\n\t" s " %4, %4, %8\n\t" s " %5, %5, %8\n\t" s " %6, %6, %8\n\t" s " %7, %7, %8" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e))
template <int MODE> __global__ void k_rate(unsigned* out, int iters, unsigned e)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) OP8("v_add_u32");
        else if (MODE == 1) OP8("v_pk_add_u16");
        else if (MODE == 2) OP8("v_pk_max_u16");
        else if (MODE == 3) OP8("v_max_u32");
        else if (MODE == 4) OP8("v_add_f32");
        else if (MODE == 5) OP8("v_pk_add_f16");
        else if (MODE == 6) OP8("v_pk_max_i16");
        else if (MODE == 8) {      // v_max3_f32 with three VGPR operands (the committed screen's maximum)
            asm volatile("v_max3_f32 %0, %0, %1, %8\n\tv_max3_f32 %1, %1, %2, %8\n\tv_max3_f32 %2, %2, %3, %8\n\tv_max3_f32 %3, %3, %4, %8\n\t"
                         "v_max3_f32 %4, %4, %5, %8\n\tv_max3_f32 %5, %5, %6, %8\n\tv_max3_f32 %6, %6, %7, %8\n\tv_max3_f32 %7, %7, %0, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));
        } else if (MODE == 9) {    // the committed screen's cell: v_add_u32 then v_max3_f32, two interleaved row chains
            unsigned t0, t1;
            asm volatile("v_add_u32 %8, %2, %10\n\tv_add_u32 %9, %3, %10\n\tv_max3_f32 %0, %8, %4, %0\n\tv_max3_f32 %1, %9, %5, %1\n\t"
                         "v_add_u32 %8, %4, %10\n\tv_add_u32 %9, %5, %10\n\tv_max3_f32 %0, %8, %6, %0\n\tv_max3_f32 %1, %9, %7, %1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1) : "v"(e));
        } else if (MODE == 10) {   // the float32 kernel's cell: three v_add_f32 (two with an SGPR operand) and a v_max3_f32
            unsigned t0, t1, t2;
            asm volatile("v_add_f32 %8, %2, %11\n\tv_add_f32 %9, %12, %3\n\tv_add_f32 %10, %12, %0\n\tv_max3_f32 %0, %8, %9, %10\n\t"
                         "v_add_f32 %8, %4, %11\n\tv_add_f32 %9, %12, %5\n\tv_add_f32 %10, %12, %0\n\tv_max3_f32 %0, %8, %9, %10"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(e), "s"(e));
        }
        else if (MODE == 7) {      // the screen's cell: add, max, max (dependent), four chains
            asm volatile("v_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\t"
                         "v_pk_max_u16 %0, %0, %4\n\tv_pk_max_u16 %1, %1, %5\n\tv_pk_max_u16 %2, %2, %6\n\tv_pk_max_u16 %3, %3, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run(const char* name, int waves_per_simd)
{
    unsigned* d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    const int iters = 100000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_rate<MODE><<<grid, block>>>(d, 1000, 3u); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); k_rate<MODE><<<grid, block>>>(d, iters, 3u); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-22s waves/simd=%d  %.3f ms  -> %.2f cycles(@2.4GHz) per wave-instruction and SIMD\n", name, waves_per_simd, ms, cyc / ((double)iters * 8 * waves_per_simd));
    (void)hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_add_u32", w); run<3>("v_max_u32", w); run<4>("v_add_f32", w);
        run<1>("v_pk_add_u16", w); run<2>("v_pk_max_u16", w); run<6>("v_pk_max_i16", w); run<5>("v_pk_add_f16", w);
        run<7>("pk add + max mix", w);
        run<8>("v_max3_f32 (3 VGPRs)", w); run<9>("screen cell add+max3", w); run<10>("f32 cell 3 add + max3", w);
    }
    return 0;
}
