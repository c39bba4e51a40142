// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the photometric kernels use
// (MI355X_MICROARCH.md: "x2 for 16 B/lane streaming reads; other widths uncalibrated: calibrate on a known byte count").
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib_bin
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./tools/fetch_calib_bin
// Each kernel reads the same 1 GiB buffer once (far larger than the 256 MiB Infinity Cache), coalesced, with 4, 8 or
// 16 bytes per lane; rows = a 640-float row read by 10 consecutive waves as the training kernel reads its rows.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ __launch_bounds__(256) void stream_read(const T *__restrict__ p, size_t n, float *out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (; i < n; i += stride) {
        T v = p[i];
        acc += reinterpret_cast<const float *>(&v)[0];
    }
    if (acc == 123.456f) out[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *buf; float *out;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream_read<float>, dim3(4096), dim3(256), 0, 0, (const float *)buf, bytes / 4, out);
        hipLaunchKernelGGL(stream_read<float2>, dim3(4096), dim3(256), 0, 0, (const float2 *)buf, bytes / 8, out);
        hipLaunchKernelGGL(stream_read<float4>, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, out);
    }
    (void)hipDeviceSynchronize();
    printf("read %zu bytes per kernel\n", bytes);
    return 0;
}
