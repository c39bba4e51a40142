// Micro-benchmark: issue rate of DPP-modified VALU ops on gfx950 (does a cross-lane operand cost an issue slot?).
// hipcc --offload-arch=gfx950 -O3 tools/dpp_rate.hip -o tools/dpp_rate_bin
// Eight independent accumulators per lane; every instruction reads a neighbour lane's register through the DPP
// operand path.  Compared with the plain v_add_f32 / v_fmac_f32 stream of tools/valu_rate.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y = a + threadIdx.x;
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 0) {
            asm volatile(
                "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
        } else if (MODE == 1) {
            asm volatile(
                "v_add_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
        } else if (MODE == 2) {
            asm volatile(
                "v_add_f32_dpp %0, %8, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %1, %8, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %2, %8, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %3, %8, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %4, %8, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %5, %8, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %6, %8, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %7, %8, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
        } else if (MODE == 3) {
            asm volatile(
                "v_add_f32_dpp %0, %8, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %1, %8, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %2, %8, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %3, %8, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %4, %8, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %5, %8, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %6, %8, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32_dpp %7, %8, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
        } else if (MODE == 4) {
            asm volatile(
                "v_fmac_f32_dpp %0, %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_fmac_f32_dpp %1, %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_fmac_f32_dpp %2, %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_fmac_f32_dpp %3, %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_fmac_f32_dpp %4, %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_fmac_f32_dpp %5, %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_fmac_f32_dpp %6, %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_fmac_f32_dpp %7, %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(a));
        } else if (MODE == 5) {
            asm volatile(
                "v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
        } else if (MODE == 6) {   // plain v_fmac_f32
            asm volatile(
                "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(a));
        } else {                  // a DPP add whose source was written by the instruction just before (hazard cost)
            asm volatile(
                "v_add_f32 %0, %8, %0\n v_add_f32_dpp %1, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32 %2, %8, %2\n v_add_f32_dpp %3, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32 %4, %8, %4\n v_add_f32_dpp %5, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                "v_add_f32 %6, %8, %6\n v_add_f32_dpp %7, %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE> void run(const char *name, int blocks_per_cu, float *d)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double winstr = (double)blocks * 4 * ITER * 8;
    printf("%-34s waves/SIMD %d : %.3f ms -> %.3f wave-instr per SIMD-cycle (2.4 GHz)\n", name, blocks_per_cu, ms,
           winstr / (ms * 1e-3) / (1024.0 * 2.4e9));
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 3, 8}) {
        run<0>("v_add_f32", w, d);
        run<1>("v_add_f32_dpp row_shr:1", w, d);
        run<2>("v_add_f32_dpp wave_shr:1", w, d);
        run<3>("v_add_f32_dpp wave_shl:1", w, d);
        run<6>("v_fmac_f32", w, d);
        run<4>("v_fmac_f32_dpp wave_shr:1", w, d);
        run<5>("v_mov_b32_dpp wave_shr:1", w, d);
        run<7>("v_add_f32 ; v_add_f32_dpp (RAW)", w, d);
    }
    return 0;
}
