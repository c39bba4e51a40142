// Micro-benchmark: plain against packed f32 VALU ops on gfx950 BY RESIDENT WAVES AND CHAIN COUNT -- the regime of the
// training kernel (3 waves per SIMD, short dependent chains), not the saturated one of tools/valu_rate.hip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/pk_rate.hip -o /tmp/pk_rate
// Prints SIMD cycles (at 2.4 GHz) per wave instruction and per ELEMENT operation.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 2048
// C independent chains of fma; PK: each chain is a float2
template <int C, bool PK>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
    float x[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; p[i] = f2{x[i], x[i] + 0.5f}; }
    const f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 8 / C; ++r)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (PK) p[c] = __builtin_elementwise_fma(p[c], pa, pb);
                else x[c] = __builtin_fmaf(x[c], a, b);
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int C, bool PK> void run(float *d, int waves)
{
    const int blocks = 256 * waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<C, PK>), dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.5f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<C, PK>), dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double winstr_per_simd = (double)waves * ITER * 8;          // per SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("waves/SIMD %d  chains %d  %-6s  %.3f ms  cycles per wave-instr %.2f  per element-op %.2f\n", waves, C, PK ? "packed" : "plain",
           ms, cyc / winstr_per_simd, cyc / winstr_per_simd / (PK ? 2 : 1));
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 3, 4, 8}) {
        run<1, false>(d, w); run<1, true>(d, w);
        run<2, false>(d, w); run<2, true>(d, w);
        run<4, false>(d, w); run<4, true>(d, w);
        run<8, false>(d, w); run<8, true>(d, w);
    }
    return 0;
}
