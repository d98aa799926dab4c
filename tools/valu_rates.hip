// Issue cost of the VALU instructions the two hot kernels are made of, measured on the device:
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates tools/valu_rates.hip && ./valu_rates
// Every kernel runs ITER iterations of 32 independent instances of one instruction (inline asm, so the
// compiler cannot fold or reorder them away) in `waves` waves per SIMD on every CU.  Reported per
// instruction: cycles of SIMD time per wave64 instruction = elapsed shader cycles x waves-per-SIMD-slots /
// instructions per SIMD, with the shader clock taken from s_memtime against s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 32768;

enum Op { ADD_F32, MAX3_F32, PK_ADD_F32, ADD_F64, MAX_F64, FMA_F64, MUL_F64, CMP_F64, CMP_F64_CND, CMP_U64, CMP_I32, CNDMASK, ADD_U32, MED3_I32, ADD_F32_DEP, ADD_F64_DEP, MAX_F64_DEP, MOV_DPP, MAX_F32, FMA_F32, MAX3_2REG, CND_SGPR, CMP_F32, CMP_F32_CND, MOV_B32, CELL, CELL_CHAIN, N_OPS };
static const char* op_name[N_OPS] = {"v_add_f32", "v_max3_f32", "v_pk_add_f32", "v_add_f64", "v_max_f64", "v_fma_f64", "v_mul_f64",
    "v_cmp_gt_f64", "v_cmp_gt_f64 + v_cndmask_b32", "v_cmp_lt_u64", "v_cmp_gt_i32", "v_cndmask_b32", "v_add_u32", "v_med3_i32",
    "v_add_f32 (one dependent chain)", "v_add_f64 (one dependent chain)", "v_max_f64 (one dependent chain)", "v_mov_b32_dpp wave_shr:1",
    "v_max_f32", "v_fma_f32", "v_max3_f32 (two distinct registers)", "v_cndmask_b32_e64 (SGPR-pair mask)", "v_cmp_gt_f32", "v_cmp_gt_f32 + v_cndmask_b32", "v_mov_b32",
    "DP cell: 3 v_add_f32 + v_max3_f32, independent cells", "DP cell: 3 v_add_f32 + v_max3_f32, chained like a DP column"};
static const int op_insts[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 4, 4};

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(double* out, uint64_t* clocks, int iters)
{
    float f[32]; double d[16]; int u[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { f[i] = (float)(threadIdx.x + i) * 1e-3f; u[i] = threadIdx.x * 7 + i; }
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = (double)(threadIdx.x + i) * 1e-3;
    const float cf = 1.0000001f; const double cd = 1.0000000001;
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == ADD_F32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
        } else if constexpr (OP == MAX3_F32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(cf), "v"(f[(i + 1) & 31]));
        } else if constexpr (OP == PK_ADD_F32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
        } else if constexpr (OP == ADD_F64) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
        } else if constexpr (OP == MAX_F64) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
        } else if constexpr (OP == FMA_F64) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));
        } else if constexpr (OP == MUL_F64) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
        } else if constexpr (OP == CMP_F64) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(cd) : "vcc");
        } else if constexpr (OP == CMP_F64_CND) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(d[i]), "v"(cd), "v"(u[i + 16]) : "vcc");
        } else if constexpr (OP == CMP_U64) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(d[i]), "v"(cd) : "vcc");
        } else if constexpr (OP == CMP_I32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(u[i]), "v"(u[(i + 1) & 31]) : "vcc");
        } else if constexpr (OP == CNDMASK) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 31]) : "vcc");
        } else if constexpr (OP == ADD_U32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 31]));
        } else if constexpr (OP == MED3_I32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 31]), "v"(u[(i + 2) & 31]));
        } else if constexpr (OP == ADD_F32_DEP) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(cf));
        } else if constexpr (OP == ADD_F64_DEP) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(cd));
        } else if constexpr (OP == MAX_F64_DEP) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[0]) : "v"(cd));
        } else if constexpr (OP == MAX_F32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 31]));
        } else if constexpr (OP == FMA_F32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(cf), "v"(f[(i + 1) & 31]));
        } else if constexpr (OP == MAX3_2REG) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_max3_f32 %0, %0, %0, %1" : "+v"(f[i]) : "v"(cf));
        } else if constexpr (OP == CND_SGPR) {
            const uint64_t mask = __builtin_amdgcn_read_exec() ^ (uint64_t)(0x5555555555555555ull * (unsigned)(iters & 1));
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 31]), "s"(mask));
        } else if constexpr (OP == CMP_F32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(cf) : "vcc");
        } else if constexpr (OP == CMP_F32_CND) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(f[i]), "v"(cf), "v"(u[i + 16]) : "vcc");
        } else if constexpr (OP == MOV_B32) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(u[(i + 5) & 31]));
        } else if constexpr (OP == CELL) {
            // 32 cells: S = max3(diag + sc, left + gh, up + gv), every cell on its own registers
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                float a, b, c;
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(f[i]), "v"(cf));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(f[(i + 1) & 31]), "v"(cf));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(c) : "v"(f[(i + 2) & 31]), "v"(cf));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(a), "v"(b), "v"(c));
            }
        } else if constexpr (OP == CELL_CHAIN) {
            // two interleaved columns of 16 rows: the cell below needs this cell's S (up + gv), as in dp_step2
            float upA = f[31], upB = f[30];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float a, b, c;
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(f[i]), "v"(cf));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(f[i + 16]), "v"(cf));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(c) : "v"(upA), "v"(cf));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(upA) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(f[i + 16]), "v"(cf));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(upA), "v"(cf));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(c) : "v"(upB), "v"(cf));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(upB) : "v"(a), "v"(b), "v"(c));
                f[i] = upA; f[i + 16] = upB;
            }
        } else if constexpr (OP == MOV_DPP) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 5) & 31]));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = wall_clock64();
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += f[i] + u[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += d[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP>
static void run(int n_cu, int waves_per_simd, double* d_out, uint64_t* d_clk)
{
    const int blocks = n_cu * waves_per_simd;          // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, 64);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk, ITER);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t* clk = (uint64_t*)malloc((size_t)blocks * 16);
    CHECK(hipMemcpy(clk, d_clk, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) { cyc += (double)clk[2 * b]; real += (double)clk[2 * b + 1]; }
    cyc /= blocks; real /= blocks;
    const double ghz = cyc / (real * 10.0);             // s_memrealtime ticks at 100 MHz -> 10 ns
    const double insts_per_wave = (double)ITER * 32 * op_insts[OP];
    const double cycles_per_inst = cyc / (insts_per_wave * waves_per_simd);
    printf("| %-34s | %d | %8.3f | %6.2f | %6.3f | %8.1f |\n", op_name[OP], waves_per_simd, ms, cycles_per_inst, ghz,
           insts_per_wave * waves_per_simd * n_cu * 4 / (ms * 1e-3) * 1e-9);
    free(clk);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device: %s, %d CUs, reported clock %.0f MHz\n\n", prop.name, n_cu, prop.clockRate * 1e-3);
    printf("| instruction | waves / SIMD | ms | shader cycles of SIMD time per wave64 instruction | shader clock (GHz) | G wave-instructions / s, chip |\n|---|---|---|---|---|---|\n");
    double* d_out; uint64_t* d_clk;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 16 * 256 * 8)); CHECK(hipMalloc(&d_clk, (size_t)n_cu * 16 * 16));
    for (int w : {1, 2, 4, 8}) {
        run<ADD_F32>(n_cu, w, d_out, d_clk); run<MAX3_F32>(n_cu, w, d_out, d_clk); run<PK_ADD_F32>(n_cu, w, d_out, d_clk);
        run<ADD_U32>(n_cu, w, d_out, d_clk); run<MED3_I32>(n_cu, w, d_out, d_clk); run<CNDMASK>(n_cu, w, d_out, d_clk);
        run<CMP_I32>(n_cu, w, d_out, d_clk); run<MOV_DPP>(n_cu, w, d_out, d_clk);
        run<ADD_F64>(n_cu, w, d_out, d_clk); run<MAX_F64>(n_cu, w, d_out, d_clk); run<MUL_F64>(n_cu, w, d_out, d_clk); run<FMA_F64>(n_cu, w, d_out, d_clk);
        run<CMP_F64>(n_cu, w, d_out, d_clk); run<CMP_F64_CND>(n_cu, w, d_out, d_clk); run<CMP_U64>(n_cu, w, d_out, d_clk);
        run<MAX_F32>(n_cu, w, d_out, d_clk); run<FMA_F32>(n_cu, w, d_out, d_clk); run<MAX3_2REG>(n_cu, w, d_out, d_clk); run<CND_SGPR>(n_cu, w, d_out, d_clk);
        run<CMP_F32>(n_cu, w, d_out, d_clk); run<CMP_F32_CND>(n_cu, w, d_out, d_clk); run<MOV_B32>(n_cu, w, d_out, d_clk);
        run<CELL>(n_cu, w, d_out, d_clk); run<CELL_CHAIN>(n_cu, w, d_out, d_clk);
        run<ADD_F32_DEP>(n_cu, w, d_out, d_clk); run<ADD_F64_DEP>(n_cu, w, d_out, d_clk); run<MAX_F64_DEP>(n_cu, w, d_out, d_clk);
    }
    return 0;
}
