// thinconv_nhwc.hip -- weight gradient of the decoder's thin 3x3 convolutions (16 output channels on the biggest maps), gfx950 MFMA.
//
// model_layer/depth_decoder.py:96-106: the last decoder stages convolve 16- and 32-channel maps of 96x320 and 192x640 pixels into 16
// channels.  Their weight gradient is a reduction over 0.4-1.5 M pixels of a 16 x (9 * Cin) outer product: 190 MB of input for 6.8
// GFLOP on the 16 -> 16 map of scale 0 -- 30 us of HBM time, 43 us of matrix time -- and MIOpen's split-K implicit GEMM takes 176 us
// (tools/convbench.py).  Here a wave walks a run of pixel quads; per quad ONE coalesced 256-byte load of the output gradient (16
// channels x 4 pixels = the A operand of v_mfma_f32_16x16x4_f32 as it lies in memory) and one per tap and 16 input channels of the
// padded input (the B operand, likewise), 9 * Cin / 16 MFMAs into as many 16 x 16 accumulator tiles that stay in registers for the
// whole run.  A block's four waves are added through LDS, the blocks by a finishing pass in a fixed order: no atomics.
#include "nhwc_common.hpp"

namespace mdx {
namespace nhwc {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int TW_WAVES = 4;                    // waves per block

// x [B][h+2][w+2][CIN] (the reflection-padded input), gy [B][h][w][16]; part [blocks][9 * CIN / 16][16][16]
template <int CIN>
__global__ __launch_bounds__(64 * TW_WAVES) void thin_wgrad16_kernel(const float *__restrict__ x, const float *__restrict__ gy, int B, int h,
                                                                      int w, int quads_per_wave, float *__restrict__ part)
{
    constexpr int CH = CIN / 16, NT = 9 * CH;
    __shared__ float lds[TW_WAVES][256];        // one accumulator tile at a time: a small LDS footprint keeps 6 waves per SIMD resident
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Hp = h + 2, Wp = w + 2, wq = w / 4;
    const long long nquads = (long long)B * h * wq;
    const long long q0 = ((long long)blockIdx.x * TW_WAVES + wave) * quads_per_wave;
    const long long q1 = q0 + quads_per_wave < nquads ? q0 + quads_per_wave : nquads;
    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int k = lane >> 4, j = lane & 15;                       // pixel of the quad, channel
    if (q0 < q1) {
        // position of the first quad, once (64-bit divisions); from then on pointers only: a quad further is 64 floats of gy and
        // 4 pixels of x, a row further two more (padded) pixels, an image further two more (padded) rows
        int qx = (int)(q0 % wq);
        const long long row0 = q0 / wq;
        int i = (int)(row0 % h);
        const int b0 = (int)(row0 / h);
        const float *pg = gy + (size_t)q0 * 64 + lane;
        const float *px = x + (((size_t)b0 * Hp + i) * Wp + (size_t)qx * 4 + k) * CIN + j;
        auto load_b = [&](const float *xb, float (&bv)[NT]) {
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                    for (int c = 0; c < CH; ++c) bv[(ty * 3 + tx) * CH + c] = xb[((size_t)ty * Wp + tx) * CIN + c * 16];
        };
        auto advance = [&]() {
            pg += 64;
            px += 4 * CIN;
            if (++qx == wq) {
                qx = 0;
                px += 2 * CIN;
                if (++i == h) { i = 0; px += (size_t)2 * Wp * CIN; }
            }
        };
        // software pipeline: the next quad's ten loads are issued before this quad's MFMAs
        float a0 = *pg, b0v[NT];
        load_b(px, b0v);
        for (long long q = q0; q < q1; ++q) {
            float a1 = 0.f, b1v[NT];
            advance();
            if (q + 1 < q1) {
                a1 = *pg;
                load_b(px, b1v);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) b1v[t] = 0.f;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0v[t], acc[t], 0, 0, 0);
            a0 = a1;
#pragma unroll
            for (int t = 0; t < NT; ++t) b0v[t] = b1v[t];
        }
    }
    // D[i][j]: lane holds rows 4 * (lane / 16) + v, column lane % 16
    float *dst = part + (size_t)blockIdx.x * NT * 256;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int v = 0; v < 4; ++v) lds[wave][(4 * k + v) * 16 + j] = acc[t][v];
        __syncthreads();
        {
            const int r = threadIdx.x;                 // 256 threads, 256 elements of the tile
            float s = lds[0][r];
#pragma unroll
            for (int u = 1; u < TW_WAVES; ++u) s += lds[u][r];
            dst[t * 256 + r] = s;
        }
        __syncthreads();
    }
}

// gw[co][ci][ty][tx] (the weight's strides) = sum over the blocks of part[.][(ty * 3 + tx) * CH + ci / 16][co][ci % 16]
constexpr int TF_S = 64;
__global__ __launch_bounds__(16 * TF_S) void thin_wgrad16_finish_kernel(const float *__restrict__ part, int nblk, int CIN, long wso, long wsc,
                                                                        long wsy, long wsx, float *__restrict__ gw)
{
    __shared__ float lds[TF_S / 4][16];
    const int n = 9 * CIN * 16;                                   // elements of the gradient
    const int cl = threadIdx.x % 16, sl = threadIdx.x / 16, e = blockIdx.x * 16 + cl;
    float s = 0.f;
    for (int i = sl; i < nblk; i += TF_S) s += (e < n) ? part[(size_t)i * n + e] : 0.f;
    s += __shfl_down(s, 32, 64);
    s += __shfl_down(s, 16, 64);
    if ((threadIdx.x & 63) < 16) lds[threadIdx.x >> 6][cl] = s;
    __syncthreads();
    if (threadIdx.x < 16 && e < n) {
        float tot = 0.f;
#pragma unroll
        for (int u = 0; u < TF_S / 4; ++u) tot += lds[u][cl];
        const int CH = CIN / 16;
        const int t = e >> 8, r = e & 255, co = r >> 4, cj = r & 15;
        const int tap = t / CH, ci = (t % CH) * 16 + cj;
        gw[co * wso + ci * wsc + (tap / 3) * wsy + (tap % 3) * wsx] = tot;
    }
}

struct ThinGeom { int blocks, quads_per_wave; };
static inline ThinGeom thin_geom(int B, int h, int w)
{
    const long long nquads = (long long)B * h * (w / 4);
    ThinGeom g;
    long long waves = 6144;                                        // six per SIMD ...
    if (waves * 48 > nquads) waves = (nquads + 47) / 48;           // ... of at least 48 quads each (shorter runs: the partials cost more than they hide)
    if (waves < 1) waves = 1;
    g.quads_per_wave = (int)((nquads + waves - 1) / waves);
    g.blocks = (int)((nquads + (long long)g.quads_per_wave * TW_WAVES - 1) / ((long long)g.quads_per_wave * TW_WAVES));
    return g;
}

}  // namespace nhwc
}  // namespace mdx

using namespace mdx;
using namespace mdx::nhwc;

static int thin_args_ok(int B, int Cin, int Cout, int h, int w)
{
    if (B <= 0 || h <= 0 || w <= 0 || w % 4 || Cout != 16 || (Cin != 16 && Cin != 32)) return MDX_ERR_BAD_SHAPE;
    if ((long long)B * (h + 2) * (w + 2) * Cin >= (1ll << 40)) return MDX_ERR_BAD_SHAPE;
    return MDX_OK;
}

MDX_EXPORT size_t mdx_thin_conv3x3_wgrad_workspace_bytes(int B, int Cin, int Cout, int h, int w)
{
    if (thin_args_ok(B, Cin, Cout, h, w)) return 0;
    return (size_t)thin_geom(B, h, w).blocks * 9 * Cin * 16 * sizeof(float);
}

// x [B][h+2][w+2][Cin], gy [B][h][w][16], float32 channels-last; gweight: element (co, ci, ky, kx) at
// gweight[co * s_o + ci * s_c + ky * s_y + kx * s_x].  Cout = 16, Cin 16 or 32, w a multiple of 4.
MDX_EXPORT int mdx_thin_conv3x3_wgrad(const float *x, const float *gy, float *gweight, int64_t s_o, int64_t s_c, int64_t s_y, int64_t s_x,
                                      int B, int Cin, int Cout, int h, int w, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !gy || !gweight || !workspace) return MDX_ERR_NULL_POINTER;
    const int bad = thin_args_ok(B, Cin, Cout, h, w);
    if (bad) return bad;
    if (workspace_bytes < mdx_thin_conv3x3_wgrad_workspace_bytes(B, Cin, Cout, h, w)) return MDX_ERR_WORKSPACE;
    const ThinGeom g = thin_geom(B, h, w);
    float *part = (float *)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 16)
        hipLaunchKernelGGL((thin_wgrad16_kernel<16>), dim3(g.blocks), dim3(64 * TW_WAVES), 0, st, x, gy, B, h, w, g.quads_per_wave, part);
    else
        hipLaunchKernelGGL((thin_wgrad16_kernel<32>), dim3(g.blocks), dim3(64 * TW_WAVES), 0, st, x, gy, B, h, w, g.quads_per_wave, part);
    const int n = 9 * Cin * 16;
    hipLaunchKernelGGL(thin_wgrad16_finish_kernel, dim3((n + 15) / 16), dim3(16 * TF_S), 0, st, part, g.blocks, Cin, (long)s_o, (long)s_c,
                       (long)s_y, (long)s_x, gweight);
    return check_launch();
}
