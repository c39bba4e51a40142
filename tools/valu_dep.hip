// Micro-benchmark: VALU issue rate of DEPENDENT chains vs waves per SIMD on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, float a)
{
    float x[8];
    for (int c = 0; c < 8; ++c) x[c] = threadIdx.x + c;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8 / CHAINS; ++j)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) x[c] += a;     // CHAINS independent chains, 8 adds per iteration
    }
    float s = 0;
    for (int c = 0; c < 8; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS> void run(int blocks_per_cu, float *d)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double winstr = (double)blocks * 4 * ITER * 8;
    printf("chains/wave %d  waves/SIMD %d : %.3f ms  -> %.3f VALU wave-instr per SIMD-cycle (2.4 GHz)\n", CHAINS,
           blocks_per_cu, ms, winstr / (ms * 1e-3) / (1024.0 * 2.4e9));
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 3, 4, 5, 8}) run<1>(w, d);
    for (int w : {1, 2, 4, 8}) run<2>(w, d);
    for (int w : {1, 2, 4, 8}) run<4>(w, d);
    for (int w : {1, 2, 4, 8}) run<8>(w, d);
    return 0;
}
