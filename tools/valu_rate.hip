// Micro-benchmark: issue rate of scalar vs packed f32 VALU ops on gfx950 (how many wave64 instructions per
// cycle a SIMD retires).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 0) {        // 8 independent v_add_f32 chains
            x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a;
        } else if (MODE == 1) { // 8 independent v_mul_f32
            x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a;
        } else if (MODE == 2) { // 8 v_fma_f32
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (MODE == 3) { // 4 v_pk_add_f32 (8 adds)
            p0 += pa; p1 += pa; p2 += pa; p3 += pa;
        } else if (MODE == 4) { // 4 v_pk_mul_f32
            p0 *= pa; p1 *= pa; p2 *= pa; p3 *= pa;
        } else {                // 4 v_pk_fma_f32
            p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
            p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char *name, float *d)
{
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double elem_ops = (double)blocks * 256 * ITER * 8;            // scalar element operations
    const double winstr = (double)blocks * 4 * ITER * (MODE < 3 ? 8 : 4); // wave-level instructions
    printf("%-14s %.3f ms  %.1f Gelem-op/s  wave-instr per SIMD-cycle @2.4GHz: %.3f\n", name, ms, elem_ops / ms / 1e6,
           winstr / (ms * 1e-3) / (1024.0 * 2.4e9));
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_f32", d); run<1>("v_mul_f32", d); run<2>("v_fma_f32", d);
    run<3>("v_pk_add_f32", d); run<4>("v_pk_mul_f32", d); run<5>("v_pk_fma_f32", d);
    return 0;
}
