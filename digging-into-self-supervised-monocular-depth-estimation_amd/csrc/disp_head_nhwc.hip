// disp_head_nhwc.hip -- the disparity heads of the decoder: sigmoid(Conv3x3(C -> 1)(x)) (model_layer/depth_decoder.py:73-74,
// 108-110), forward and backward, on channels-last maps, gfx950.
//
// A convolution with ONE output channel is not a matrix-core problem: 2 * 9 * C flops per 4 * C bytes of input, 2.7 us of MFMA
// time against 16 us of HBM time on the 16 x 194 x 642 map of scale 0 -- and MIOpen's implicit-GEMM kernels take 137 us forward,
// 77 us for the data gradient and 167 us for the weight gradient there (tools/convbench.py), 760 us per step over the four
// scales, plus a bias add, a sigmoid, its backward, a bias reduction and a zero-fill, each a launch.  Here:
//   forward   ONE launch: a thread owns one 16-byte channel vector of one output COLUMN and walks down the rows with a rolling
//             window, so every input row is loaded once per column triple (L1 serves the two neighbours); the C / N lanes of a
//             pixel combine their partial dot products by shuffles; bias and sigmoid on the way out.  x [B][h+2][w+2][C] is the
//             reflection-padded map the decoder glue wrote; out [B][h][w] float32.
//   backward  ONE launch + a finishing pass: a thread owns one channel vector of one INPUT column and walks down the rows; the
//             3x3 window of  g * (1 - y) * y  it keeps in registers gives both the data gradient of its element (9 FMAs per
//             channel with the weights) and its contribution to the weight gradient (9 FMAs per channel with x, which is loaded
//             ONCE); the block's weight-gradient partials are combined by shuffles + LDS and summed over the blocks by the
//             finishing pass in a fixed order (no atomics).  One read of x, one write of gx: 2 x the map.
#include "nhwc_common.hpp"

namespace mdx {
namespace nhwc {

constexpr int DH_NB = 256;
constexpr int DH_N = 4;          // channels per thread, float32 AND bfloat16 (8 bfloat16 per thread doubled the accumulators: 192 registers,
                                 // two waves per SIMD, the backward pass twice as slow as the float32 one)
enum { MDX_F32 = 0, MDX_BF16 = 1 };   // the dtype codes of include/mdx.h

template <typename T, int N>
__device__ __forceinline__ void load_weights(const float *__restrict__ wt, long wsc, long wsy, long wsx, int c0, float (&wr)[9][N])
{
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < N; ++n) wr[t][n] = wt[(long)(c0 + n) * wsc + (t / 3) * wsy + (t % 3) * wsx];
}

// grid (ceil(w / CPB), ceil(h / R), B), CPB = 256 / LP output columns per block, LP = C / N lanes per pixel (a power of two <= 64)
template <typename T>
__global__ __launch_bounds__(DH_NB) void disp_head_fwd_kernel(const T *__restrict__ x, const float *__restrict__ wt, long wsc, long wsy,
                                                              long wsx, const float *__restrict__ bias, int Hp, int Wp, int C, int LP,
                                                              int R, float *__restrict__ out)
{
    constexpr int N = DH_N;
    const int h = Hp - 2, w = Wp - 2;
    const int l = threadIdx.x % LP, cb = threadIdx.x / LP, CPB = DH_NB / LP;
    const int j = blockIdx.x * CPB + cb;
    const bool live = j < w;
    const int jc = live ? j : w - 1;
    const int i0 = blockIdx.y * R, i1 = min(i0 + R, h), b = blockIdx.z;
    float wr[9][N];
    load_weights<T, N>(wt, wsc, wsy, wsx, l * N, wr);
    const float bv = bias ? bias[0] : 0.f;
    const T *px = x + ((size_t)b * Hp * Wp + jc) * C + (size_t)l * N;
    const size_t rs = (size_t)Wp * C;
    float r1 = 0.f, r2 = 0.f;
    // the next row's loads are issued before this row's arithmetic: a thread walks its rows one after the other, and without the
    // prefetch every row costs a full memory latency
    Vec<T, N> n0 = load_vec<T, N>(px + (size_t)i0 * rs), n1 = load_vec<T, N>(px + (size_t)i0 * rs + C),
              n2 = load_vec<T, N>(px + (size_t)i0 * rs + 2 * C);
    for (int p = i0; p < i1 + 2; ++p) {          // input row p feeds output rows p, p-1, p-2 through kernel rows 0, 1, 2
        const Vec<T, N> v0 = n0, v1 = n1, v2 = n2;
        if (p + 1 < i1 + 2) {
            const T *row = px + (size_t)(p + 1) * rs;
            n0 = load_vec<T, N>(row); n1 = load_vec<T, N>(row + C); n2 = load_vec<T, N>(row + 2 * C);
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float a = to_float(v0.v[n]), bq = to_float(v1.v[n]), c = to_float(v2.v[n]);
            s0 = __builtin_fmaf(wr[0][n], a, s0); s0 = __builtin_fmaf(wr[1][n], bq, s0); s0 = __builtin_fmaf(wr[2][n], c, s0);
            s1 = __builtin_fmaf(wr[3][n], a, s1); s1 = __builtin_fmaf(wr[4][n], bq, s1); s1 = __builtin_fmaf(wr[5][n], c, s1);
            s2 = __builtin_fmaf(wr[6][n], a, s2); s2 = __builtin_fmaf(wr[7][n], bq, s2); s2 = __builtin_fmaf(wr[8][n], c, s2);
        }
        float done = r2 + s2;
        r2 = r1 + s1;
        r1 = s0;
        if (p >= i0 + 2) {                       // output row p - 2 is complete
            for (int m = LP >> 1; m; m >>= 1) done += __shfl_xor(done, m, 64);
            if (l == 0 && live) {
                const float v = done + bv;
                out[((size_t)b * h + (p - 2)) * w + j] = 1.0f / (1.0f + expf(-v));
            }
        }
    }
}

// grid (ceil(Wp / CPB), ceil(Hp / R), B); part [blocks][9 * C + 1]: the block's weight-gradient partials [ky][kx][c] and its bias one
template <typename T, int WAVES>      // WAVES: the occupancy the register allocator is held to (float32: 4 -> 124 registers, no spill)
__global__ __launch_bounds__(DH_NB) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void disp_head_bwd_kernel(const T *__restrict__ x, const float *__restrict__ wt, long wsc, long wsy,
                                                              long wsx, const float *__restrict__ g, const float *__restrict__ y, int Hp,
                                                              int Wp, int C, int LP, int R, T *__restrict__ gx, float *__restrict__ part)
{
    constexpr int N = DH_N;
    extern __shared__ float lds[];               // [4 waves][9 * C] + [4]
    const int h = Hp - 2, w = Wp - 2;
    const int l = threadIdx.x % LP, cb = threadIdx.x / LP, CPB = DH_NB / LP;
    const int q = blockIdx.x * CPB + cb;
    const bool live = q < Wp;
    const int qc = live ? q : Wp - 1;
    const int p0 = blockIdx.y * R, p1 = min(p0 + R, Hp), b = blockIdx.z;
    float wr[9][N];
    load_weights<T, N>(wt, wsc, wsy, wsx, l * N, wr);
    // d loss / d (convolution output) = sigmoid_backward's  g * (1 - y) * y; loaded raw (0 outside the map) so that the next row's
    // loads can be issued before this row's arithmetic
    auto raw = [&](int i, int jj, float &gv, float &yv) {
        gv = 0.f; yv = 0.f;
        if (i >= 0 && i < h && jj >= 0 && jj < w) {
            const size_t o = ((size_t)b * h + i) * w + jj;
            gv = g[o]; yv = y[o];
        }
    };
    auto gpre = [&](float gv, float yv) -> float { return gv * (1.0f - yv) * yv; };
    float G[3][3];                               // G[ky][kx] = gpre(p - ky, q - kx): the outputs input element (p, q) feeds
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        float gv, yv;
        raw(p0 - 1, q - kx, gv, yv); G[1][kx] = gpre(gv, yv);
        raw(p0 - 2, q - kx, gv, yv); G[2][kx] = gpre(gv, yv);
    }
    float gw[9][N], gb = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < N; ++n) gw[t][n] = 0.f;
    const size_t col = (size_t)qc * C + (size_t)l * N;
    const size_t rs = (size_t)Wp * C;
    size_t off = ((size_t)b * Hp + p0) * rs + col;
    Vec<T, N> xn = load_vec<T, N>(x + off);
    float gn[3], yn[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) raw(p0, q - kx, gn[kx], yn[kx]);
    for (int p = p0; p < p1; ++p, off += rs) {
        const Vec<T, N> xv = xn;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) G[0][kx] = gpre(gn[kx], yn[kx]);
        if (p + 1 < p1) {
            xn = load_vec<T, N>(x + off + rs);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) raw(p + 1, q - kx, gn[kx], yn[kx]);
        }
        Vec<T, N> o;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float xf = to_float(xv.v[n]);
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                a = __builtin_fmaf(G[t / 3][t % 3], wr[t][n], a);
                gw[t][n] = __builtin_fmaf(G[t / 3][t % 3], xf, gw[t][n]);
            }
            o.v[n] = from_float<T>(a);
        }
        if (live) store_vec<T, N>(gx + off, o);
        if (l == 0) gb += G[0][0];               // every output pixel is (p, q) of exactly one thread column
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) { G[2][kx] = G[1][kx]; G[1][kx] = G[0][kx]; }
    }
    // threads beyond the map's width hold zeros (their window lies outside the output): no masking needed below
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float v = gw[t][n];
            for (int m = LP; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);      // over the pixels of the wave with this channel vector
            if (lane < LP) lds[(size_t)wave * 9 * C + t * C + l * N + n] = v;
        }
    for (int m = 1; m < 64; m <<= 1) gb += __shfl_xor(gb, m, 64);
    if (lane == 0) lds[(size_t)(DH_NB / 64) * 9 * C + wave] = gb;
    __syncthreads();
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float *dst = part + blk * (9 * C + 1);
    for (int e = threadIdx.x; e < 9 * C; e += DH_NB) {
        float s = lds[e];
#pragma unroll
        for (int k = 1; k < DH_NB / 64; ++k) s += lds[(size_t)k * 9 * C + e];
        dst[e] = s;
    }
    if (threadIdx.x == 0) {
        float s = lds[(size_t)(DH_NB / 64) * 9 * C];
#pragma unroll
        for (int k = 1; k < DH_NB / 64; ++k) s += lds[(size_t)(DH_NB / 64) * 9 * C + k];
        dst[9 * C] = s;
    }
}

// column sums of part [n][cols] in a fixed order -> the weight gradient (with the weight's own strides) and the bias gradient
constexpr int DF_C = 16, DF_S = 64, DF_U = 8;
__global__ __launch_bounds__(DF_C *DF_S) void disp_head_finish_kernel(const float *__restrict__ part, int n, int C, long wsc, long wsy,
                                                                     long wsx, float *__restrict__ gw, float *__restrict__ gb)
{
    __shared__ float lds[DF_S / 4][DF_C];
    const int cols = 9 * C + 1;
    const int cl = threadIdx.x % DF_C, sl = threadIdx.x / DF_C, c = blockIdx.x * DF_C + cl;
    float s = 0.f;
    for (int i0 = sl; i0 < n; i0 += DF_S * DF_U) {
        float v[DF_U];
#pragma unroll
        for (int u = 0; u < DF_U; ++u) {
            const int i = i0 + u * DF_S;
            v[u] = (c < cols && i < n) ? part[(size_t)i * cols + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < DF_U; ++u) s += v[u];
    }
    s += __shfl_down(s, 32, 64);
    s += __shfl_down(s, 16, 64);
    if ((threadIdx.x & 63) < DF_C) lds[threadIdx.x >> 6][cl] = s;
    __syncthreads();
    if (threadIdx.x < DF_C && c < cols) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < DF_S / 4; ++k) tot += lds[k][cl];
        if (c == 9 * C) {
            if (gb) gb[0] = tot;
        } else {
            const int t = c / C, ch = c - t * C;
            gw[(long)ch * wsc + (t / 3) * wsy + (t % 3) * wsx] = tot;
        }
    }
}

struct HeadGeom { int LP, CPB, R; dim3 grid; };
static inline bool head_geom(int B, int rows, int cols, int C, int dtype, HeadGeom *g)
{
    const int N = DH_N;
    (void)dtype;
    if (C % N) return false;
    g->LP = C / N;
    if (g->LP < 1 || g->LP > 64 || (g->LP & (g->LP - 1))) return false;
    g->CPB = DH_NB / g->LP;
    const int gx = (cols + g->CPB - 1) / g->CPB;
    g->R = 32;
    while (g->R > 8 && (long long)gx * ((rows + g->R - 1) / g->R) * B < 1024) g->R >>= 1;
    const int gy = (rows + g->R - 1) / g->R;
    if (gy > 65535 || B > 65535) return false;
    g->grid = dim3(gx, gy, B);
    return true;
}

}  // namespace nhwc
}  // namespace mdx

using namespace mdx;
using namespace mdx::nhwc;

static int head_args_ok(int B, int C, int h, int w, int dtype)
{
    if (dtype != MDX_F32 && dtype != MDX_BF16) return MDX_ERR_BAD_SHAPE;
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || C > 1024) return MDX_ERR_BAD_SHAPE;
    if ((long long)B * (h + 2) * (w + 2) >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
    return MDX_OK;
}

MDX_EXPORT size_t mdx_disp_head_nhwc_workspace_bytes(int B, int C, int h, int w, int dtype)
{
    HeadGeom g;
    if (head_args_ok(B, C, h, w, dtype) || !head_geom(B, h + 2, w + 2, C, dtype, &g)) return 0;
    return (size_t)g.grid.x * g.grid.y * g.grid.z * (9 * (size_t)C + 1) * sizeof(float);
}

MDX_EXPORT int mdx_disp_head_nhwc_fwd(const void *x, const float *weight, int64_t w_stride_c, int64_t w_stride_ky, int64_t w_stride_kx,
                                      const float *bias, float *disp, int B, int C, int h, int w, int dtype, void *stream)
{
    if (!x || !weight || !disp) return MDX_ERR_NULL_POINTER;
    const int bad = head_args_ok(B, C, h, w, dtype);
    if (bad) return bad;
    if (!aligned(x, 16)) return MDX_ERR_MISALIGNED;
    HeadGeom g;
    if (!head_geom(B, h, w, C, dtype, &g)) return MDX_ERR_BAD_SHAPE;
    if (dtype == MDX_F32)
        hipLaunchKernelGGL((disp_head_fwd_kernel<float>), g.grid, dim3(DH_NB), 0, (hipStream_t)stream, (const float *)x, weight,
                           (long)w_stride_c, (long)w_stride_ky, (long)w_stride_kx, bias, h + 2, w + 2, C, g.LP, g.R, disp);
    else
        hipLaunchKernelGGL((disp_head_fwd_kernel<bf16>), g.grid, dim3(DH_NB), 0, (hipStream_t)stream, (const bf16 *)x, weight,
                           (long)w_stride_c, (long)w_stride_ky, (long)w_stride_kx, bias, h + 2, w + 2, C, g.LP, g.R, disp);
    return check_launch();
}

MDX_EXPORT int mdx_disp_head_nhwc_bwd(const void *x, const float *weight, int64_t w_stride_c, int64_t w_stride_ky, int64_t w_stride_kx,
                                      const float *gdisp, const float *disp, void *gx, float *gweight, float *gbias, int B, int C,
                                      int h, int w, int dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !weight || !gdisp || !disp || !gx || !gweight || !workspace) return MDX_ERR_NULL_POINTER;
    const int bad = head_args_ok(B, C, h, w, dtype);
    if (bad) return bad;
    if (!aligned(x, 16) || !aligned(gx, 16)) return MDX_ERR_MISALIGNED;
    HeadGeom g;
    if (!head_geom(B, h + 2, w + 2, C, dtype, &g)) return MDX_ERR_BAD_SHAPE;
    if (workspace_bytes < mdx_disp_head_nhwc_workspace_bytes(B, C, h, w, dtype)) return MDX_ERR_WORKSPACE;
    float *part = (float *)workspace;
    const size_t shmem = ((size_t)(DH_NB / 64) * 9 * C + DH_NB / 64) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MDX_F32)
        hipLaunchKernelGGL((disp_head_bwd_kernel<float, 4>), g.grid, dim3(DH_NB), shmem, st, (const float *)x, weight, (long)w_stride_c,
                           (long)w_stride_ky, (long)w_stride_kx, gdisp, disp, h + 2, w + 2, C, g.LP, g.R, (float *)gx, part);
    else
        hipLaunchKernelGGL((disp_head_bwd_kernel<bf16, 4>), g.grid, dim3(DH_NB), shmem, st, (const bf16 *)x, weight, (long)w_stride_c,
                           (long)w_stride_ky, (long)w_stride_kx, gdisp, disp, h + 2, w + 2, C, g.LP, g.R, (bf16 *)gx, part);
    const int nblk = (int)(g.grid.x * g.grid.y * g.grid.z), cols = 9 * C + 1;
    hipLaunchKernelGGL(disp_head_finish_kernel, dim3((cols + DF_C - 1) / DF_C), dim3(DF_C * DF_S), 0, st, part, nblk, C,
                       (long)w_stride_c, (long)w_stride_ky, (long)w_stride_kx, gweight, gbias);
    return check_launch();
}
