// Does the issue time of v_max3_f32 (and of the screen's cell: v_add_u32 then v_max3_f32) on gfx950 depend on WHICH registers its
// sources sit in?  Round-4 verdict item 5.  Sources pinned by hand: all three in one class of (register number mod 4), in three
// different classes, one source repeated; at 4 and 6 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_operands.hip -o /tmp/ubench_operands && /tmp/ubench_operands
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOBBER "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"
#define INIT asm volatile("v_mov_b32 v40, 1.0\n v_mov_b32 v41, 2.0\n v_mov_b32 v42, 0.5\n v_mov_b32 v43, 4.0\n v_mov_b32 v44, 1.0\n v_mov_b32 v45, 2.0\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 4.0\n" \
                          "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 2.0\n v_mov_b32 v50, 0.5\n v_mov_b32 v51, 4.0\n v_mov_b32 v52, 1.0\n v_mov_b32 v53, 2.0\n v_mov_b32 v54, 0.5\n v_mov_b32 v55, 4.0\n" \
                          "v_mov_b32 v56, 1.0\n v_mov_b32 v57, 2.0\n v_mov_b32 v58, 0.5\n v_mov_b32 v59, 4.0\n v_mov_b32 v60, 1.0\n v_mov_b32 v61, 2.0\n v_mov_b32 v62, 0.5\n v_mov_b32 v63, 4.0\n" \
                          "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 2.0\n v_mov_b32 v66, 0.5\n v_mov_b32 v67, 4.0\n v_mov_b32 v68, 1.0\n v_mov_b32 v69, 2.0\n v_mov_b32 v70, 0.5\n v_mov_b32 v71, 4.0" ::: CLOBBER)

template <int MODE> __global__ void k_rate(unsigned* out, int iters)
{
    INIT;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0)          // three sources in ONE class (mod 4): v41, v45, v49 ...; destinations v64..v71
            asm volatile("v_max3_f32 v64, v41, v45, v49\n v_max3_f32 v65, v42, v46, v50\n v_max3_f32 v66, v43, v47, v51\n v_max3_f32 v67, v44, v48, v52\n"
                         "v_max3_f32 v68, v45, v49, v53\n v_max3_f32 v69, v46, v50, v54\n v_max3_f32 v70, v47, v51, v55\n v_max3_f32 v71, v48, v52, v56" ::: CLOBBER);
        else if (MODE == 1)     // three sources in three DIFFERENT classes
            asm volatile("v_max3_f32 v64, v41, v42, v43\n v_max3_f32 v65, v42, v43, v44\n v_max3_f32 v66, v43, v44, v45\n v_max3_f32 v67, v44, v45, v46\n"
                         "v_max3_f32 v68, v45, v46, v47\n v_max3_f32 v69, v46, v47, v48\n v_max3_f32 v70, v47, v48, v49\n v_max3_f32 v71, v48, v49, v50" ::: CLOBBER);
        else if (MODE == 2)     // one source repeated (two register reads)
            asm volatile("v_max3_f32 v64, v41, v41, v42\n v_max3_f32 v65, v42, v42, v43\n v_max3_f32 v66, v43, v43, v44\n v_max3_f32 v67, v44, v44, v45\n"
                         "v_max3_f32 v68, v45, v45, v46\n v_max3_f32 v69, v46, v46, v47\n v_max3_f32 v70, v47, v47, v48\n v_max3_f32 v71, v48, v48, v49" ::: CLOBBER);
        else if (MODE == 3)     // the screen's cell, sources of every instruction in different classes: add t = diag + score; max3(t, left, up)
            asm volatile("v_add_u32 v64, v41, v42\n v_max3_f32 v56, v64, v43, v57\n v_add_u32 v65, v42, v43\n v_max3_f32 v57, v65, v44, v58\n"
                         "v_add_u32 v66, v43, v44\n v_max3_f32 v58, v66, v45, v59\n v_add_u32 v67, v44, v45\n v_max3_f32 v59, v67, v46, v56" ::: CLOBBER);
        else if (MODE == 4)     // the same cell with the three max3 sources in ONE class
            asm volatile("v_add_u32 v64, v41, v45\n v_max3_f32 v56, v64, v44, v60\n v_add_u32 v68, v42, v46\n v_max3_f32 v60, v68, v48, v52\n"
                         "v_add_u32 v64, v43, v47\n v_max3_f32 v52, v64, v44, v56\n v_add_u32 v68, v41, v49\n v_max3_f32 v48, v68, v60, v52" ::: CLOBBER);
        else if (MODE == 5)     // v_max_f32 (two sources), different classes
            asm volatile("v_max_f32 v64, v41, v42\n v_max_f32 v65, v42, v43\n v_max_f32 v66, v43, v44\n v_max_f32 v67, v44, v45\n"
                         "v_max_f32 v68, v45, v46\n v_max_f32 v69, v46, v47\n v_max_f32 v70, v47, v48\n v_max_f32 v71, v48, v49" ::: CLOBBER);
        else if (MODE == 6)     // v_add_u32, different classes (the full-rate reference)
            asm volatile("v_add_u32 v64, v41, v42\n v_add_u32 v65, v42, v43\n v_add_u32 v66, v43, v44\n v_add_u32 v67, v44, v45\n"
                         "v_add_u32 v68, v45, v46\n v_add_u32 v69, v46, v47\n v_add_u32 v70, v47, v48\n v_add_u32 v71, v48, v49" ::: CLOBBER);
        else if (MODE == 7)     // v_add_u32, both sources in one class
            asm volatile("v_add_u32 v64, v41, v45\n v_add_u32 v65, v42, v46\n v_add_u32 v66, v43, v47\n v_add_u32 v67, v44, v48\n"
                         "v_add_u32 v68, v45, v49\n v_add_u32 v69, v46, v50\n v_add_u32 v70, v47, v51\n v_add_u32 v71, v48, v52" ::: CLOBBER);
        else if (MODE == 8)     // v_max3_f32 with an inline constant and two registers
            asm volatile("v_max3_f32 v64, v41, v42, 1.0\n v_max3_f32 v65, v42, v43, 1.0\n v_max3_f32 v66, v43, v44, 1.0\n v_max3_f32 v67, v44, v45, 1.0\n"
                         "v_max3_f32 v68, v45, v46, 1.0\n v_max3_f32 v69, v46, v47, 1.0\n v_max3_f32 v70, v47, v48, 1.0\n v_max3_f32 v71, v48, v49, 1.0" ::: CLOBBER);
    }
    unsigned r;
    asm volatile("v_add_u32 %0, v64, v71\n v_add_u32 %0, %0, v56" : "=v"(r) :: CLOBBER);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, int waves_per_simd)
{
    unsigned* d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    const int iters = 100000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_rate<MODE><<<grid, block>>>(d, 1000); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); k_rate<MODE><<<grid, block>>>(d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-58s waves/simd=%d  %.3f ms  -> %.2f cycles(@2.4GHz) per wave-instruction and SIMD\n", name, waves_per_simd, ms, cyc / ((double)iters * 8 * waves_per_simd));
    (void)hipFree(d);
}

int main()
{
    for (int w : {4, 6}) {
        run<6>("v_add_u32, sources in two classes", w);
        run<7>("v_add_u32, both sources in one class", w);
        run<5>("v_max_f32, sources in two classes", w);
        run<0>("v_max3_f32, three sources in ONE class (mod 4)", w);
        run<1>("v_max3_f32, three sources in THREE classes", w);
        run<2>("v_max3_f32, one source repeated", w);
        run<8>("v_max3_f32, two registers + inline constant", w);
        run<3>("screen cell add + max3, sources in different classes", w);
        run<4>("screen cell add + max3, max3 sources in one class", w);
    }
    return 0;
}
