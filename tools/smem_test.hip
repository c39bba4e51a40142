// Does s_load_dwordx16 / x8 accept any dword-aligned address on gfx950?  (It does: all offsets below read back correctly.)
//   hipcc --offload-arch=gfx950 -O2 tools/smem_test.hip -o tools/smem_test_bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void k(const int *tab, int off, int *out)
{
    const int *p = tab + off;
    unsigned long long a = (unsigned long long)p;
    i32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a));
    if (threadIdx.x < 16) out[threadIdx.x] = v[threadIdx.x & 15];
    i32x8 w;
    asm volatile("s_load_dwordx8 %0, %1, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(a));
    if (threadIdx.x < 8) out[16 + threadIdx.x] = w[threadIdx.x & 7];
}
int main()
{
    int h[256]; for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    int *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 24 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int off : {0, 1, 3, 5, 13, 17, 31}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, off, o);
        int r[24]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        bool ok = true; for (int i = 0; i < 24; ++i) ok &= r[i] == 1000 + off + i;
        printf("off %d: %s  (%d %d .. %d | %d .. %d)\n", off, ok ? "ok" : "WRONG", r[0], r[1], r[15], r[16], r[23]);
    }
    return 0;
}
