// Does the written-out division sequence of csrc/photo_train.hip (quot_rcp / refined_rcp + quot_with: the compiler's own
// IEEE expansion without div_scale / div_fmas / div_fixup) give the bits of `/`?  2^32 random pairs per domain, on the GPU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/check_fastdiv_bin tools/check_fastdiv.hip && tools/check_fastdiv_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint64_t &s)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}
// a float whose exponent is uniform in [elo, ehi] (unbiased powers of two), random mantissa, random sign if `sgn`
__device__ __forceinline__ float draw(uint64_t &s, int elo, int ehi, bool sgn)
{
    const uint32_t r = mix(s), m = mix(s);
    const uint32_t e = (uint32_t)(elo + (int)(r % (uint32_t)(ehi - elo + 1)) + 127);
    const uint32_t bits = ((sgn && (r >> 31)) ? 0x80000000u : 0u) | (e << 23) | (m & 0x7FFFFFu);
    return __builtin_bit_cast(float, bits);
}
__device__ __forceinline__ float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float quot_with(float n, float d, float r1)
{
    const float q0 = n * r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r1, q1);
}
// domain 0: SSIM quotient n / d   d in [2^-24, 2^40], n = 0 or |n| in [2^-47, 2^40]
// domain 1: projection    q / z   |z| in [2^-47, 2^20], q = 0 or |q| in [2^-40, 2^30]
// domain 2: depth         1 / sd  sd in [2^-20, 2^20]
__global__ void check(int domain, uint64_t seed, unsigned long long *bad, float *ex)
{
    uint64_t s = seed + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull;
    unsigned long long nb = 0;
    for (int it = 0; it < 4096; ++it) {
        float n, d;
        if (domain == 0) { d = draw(s, -24, 40, false); n = (mix(s) & 1023u) == 0 ? 0.f : draw(s, -47, 40, true); }
        else if (domain == 1) { d = draw(s, -47, 20, true); n = (mix(s) & 1023u) == 0 ? 0.f : draw(s, -40, 30, true); }
        else { d = draw(s, -20, 20, false); n = 1.0f; }
        const float ref = n / d;
        const float got = quot_with(n, d, refined_rcp(d));
        if (__builtin_bit_cast(uint32_t, ref) != __builtin_bit_cast(uint32_t, got)) {
            if (nb == 0) { ex[0] = n; ex[1] = d; ex[2] = ref; ex[3] = got; }
            ++nb;
        }
    }
    if (nb) atomicAdd(bad, nb);
}
int main()
{
    unsigned long long *bad; float *ex;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&ex, 16);
    const char *names[3] = {"SSIM quotient n/d", "projection q/z", "depth 1/sd"};
    int rc = 0;
    for (int dom = 0; dom < 3; ++dom) {
        (void)hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, dom, 0x1234567ull + dom, bad, ex);   // 4096*256*4096 = 2^32 pairs
        unsigned long long h = 0; float hx[4] = {0, 0, 0, 0};
        (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hx, ex, 16, hipMemcpyDeviceToHost);
        printf("%-20s 2^32 random pairs: %llu differ from the IEEE divide", names[dom], h);
        if (h) printf("  (e.g. %.9g / %.9g: / gives %.9g, sequence gives %.9g)", hx[0], hx[1], hx[2], hx[3]);
        printf("\n");
        rc |= h != 0;
    }
    return rc;
}
