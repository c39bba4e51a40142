// pose_head_nhwc.hip -- what sits between the pose decoder's convolutions (model_layer/pose_decoder.py:24-53), channels-last, gfx950.
//
// The pose decoder is four small convolutions on a [2B][256][H/32][W/32] map, each followed by a bias add and a ReLU, then the
// spatial mean and the 0.01 scale: ATen runs conv + add_ + clamp_min forward and threshold_backward + a 28 us bias reduction +
// two convolution gradients backward -- a dozen launches of 5-30 us each on the pose network's chain, the longer of the step's
// two (LABNOTES round 5: a sleep on the pose stream costs the step half of its length, on the depth stream a sixth).  Here the
// convolution runs WITHOUT its bias and
//   bias_act    y = act(x + b):  one launch; backward one launch (dx = dy * (y > 0), per-block column sums) + a finishing pass
//   mean_bias   out[m][c] = scale * (mean over the pixels of x[m][.][c] + b[c]): one launch; backward one launch for dx and db.
// Sums in a fixed order, no atomics.
#include "nhwc_common.hpp"

namespace mdx {
namespace nhwc {

enum { PH_F32 = 0, PH_BF16 = 1 };
constexpr int BA_BLOCKS = 256, BA_ITERS = 4;

template <typename T>
__global__ __launch_bounds__(NB) void bias_act_nhwc_fwd_kernel(const T *__restrict__ x, const float *__restrict__ bias, int M, int C,
                                                               int CVB, int PL, int RB, int relu, T *__restrict__ y)
{
    constexpr int N = VecN<T>::N;
    const Pos p = position<N>(M, C, CVB, PL, RB);
    if (!p.active) return;
    float b[N];
#pragma unroll
    for (int j = 0; j < N; ++j) b[j] = bias[p.cv * N + j];
    const T *px = x + (size_t)p.cv * N;
    T *py = y + (size_t)p.cv * N;
    for (int r = p.r0 + p.pl; r < p.r1; r += PL) {
        const Vec<T, N> v = load_vec<T, N>(px + (size_t)r * C);
        Vec<T, N> w;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float f = to_float(v.v[j]) + b[j];
            w.v[j] = from_float<T>((relu && f < 0.f) ? 0.f : f);
        }
        store_vec<T, N>(py + (size_t)r * C, w);
    }
}

// part [blocks along the rows][C]: the block's column sums of dz
template <typename T>
__global__ __launch_bounds__(NB) void bias_act_nhwc_bwd_kernel(const T *__restrict__ dy, const T *__restrict__ y, int M, int C, int CVB,
                                                               int PL, int RB, int relu, T *__restrict__ dx, float *__restrict__ part)
{
    constexpr int N = VecN<T>::N;
    __shared__ float lds[NB * N];
    const Pos p = position<N>(M, C, CVB, PL, RB);
    float a[N];
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] = 0.f;
    if (p.active) {
        const T *__restrict__ pd = dy + (size_t)p.cv * N, *__restrict__ py = y + (size_t)p.cv * N;
        T *__restrict__ ox = dx + (size_t)p.cv * N;
        auto one = [&](const Vec<T, N> &d, const Vec<T, N> &v, size_t off) {
            Vec<T, N> w;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float dz = (relu && !(to_float(v.v[j]) > 0.f)) ? 0.f : to_float(d.v[j]);
                a[j] += dz;
                w.v[j] = from_float<T>(dz);
            }
            store_vec<T, N>(ox + off, w);
        };
        int r = p.r0 + p.pl;
        for (; r + 3 * PL < p.r1; r += 4 * PL) {          // every load of four rows before the first store
            const size_t o0 = (size_t)r * C, o1 = (size_t)(r + PL) * C, o2 = (size_t)(r + 2 * PL) * C, o3 = (size_t)(r + 3 * PL) * C;
            const Vec<T, N> d0 = load_vec<T, N>(pd + o0), d1 = load_vec<T, N>(pd + o1), d2 = load_vec<T, N>(pd + o2), d3 = load_vec<T, N>(pd + o3);
            const Vec<T, N> v0 = load_vec<T, N>(py + o0), v1 = load_vec<T, N>(py + o1), v2 = load_vec<T, N>(py + o2), v3 = load_vec<T, N>(py + o3);
            one(d0, v0, o0); one(d1, v1, o1); one(d2, v2, o2); one(d3, v3, o3);
        }
        for (; r < p.r1; r += PL) one(load_vec<T, N>(pd + (size_t)r * C), load_vec<T, N>(py + (size_t)r * C), (size_t)r * C);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) lds[threadIdx.x * N + j] = a[j];
    __syncthreads();
    const int width = CVB * N;
    for (int e = threadIdx.x; e < width; e += NB) {
        const int c = blockIdx.y * width + e;
        if (c >= C) break;
        float s = 0.f;
        for (int l = 0; l < PL; ++l) s += lds[l * width + e];
        part[(size_t)blockIdx.x * C + c] = s;
    }
}

// out[c] = scale * sum over the n rows of part[.][c], fixed order
constexpr int PF_C = 16, PF_S = 64;
__global__ __launch_bounds__(PF_C *PF_S) void colsum_scale_kernel(const float *__restrict__ part, int n, int C, float scale,
                                                                  float *__restrict__ out)
{
    __shared__ float lds[PF_S / 4][PF_C];
    const int cl = threadIdx.x % PF_C, sl = threadIdx.x / PF_C, c = blockIdx.x * PF_C + cl;
    float s = 0.f;
    for (int i = sl; i < n; i += PF_S) s += (c < C) ? part[(size_t)i * C + c] : 0.f;
    s += __shfl_down(s, 32, 64);
    s += __shfl_down(s, 16, 64);
    if ((threadIdx.x & 63) < PF_C) lds[threadIdx.x >> 6][cl] = s;
    __syncthreads();
    if (threadIdx.x < PF_C && c < C) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < PF_S / 4; ++k) tot += lds[k][cl];
        out[c] = tot * scale;
    }
}

// x [M][HW][C] -> out [M][C] float32: one thread per (m, c), four running sums (pixels p, p+4, ...) added in order at the end
template <typename T>
__global__ __launch_bounds__(NB) void mean_bias_nhwc_fwd_kernel(const T *__restrict__ x, const float *__restrict__ bias, int M, int HW,
                                                                int C, float inv_hw, float scale, float *__restrict__ out)
{
    const int e = blockIdx.x * NB + threadIdx.x;
    if (e >= M * C) return;
    const int m = e / C, c = e - m * C;
    const T *px = x + (size_t)m * HW * C + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 3 < HW; p += 4) {
        s0 += to_float(px[(size_t)p * C]); s1 += to_float(px[(size_t)(p + 1) * C]);
        s2 += to_float(px[(size_t)(p + 2) * C]); s3 += to_float(px[(size_t)(p + 3) * C]);
    }
    for (; p < HW; ++p) s0 += to_float(px[(size_t)p * C]);
    const float mean = ((s0 + s1) + (s2 + s3)) * inv_hw;
    out[e] = scale * (mean + (bias ? bias[c] : 0.f));
}

// g [M][C] float32 -> dx [M][HW][C] = g * scale / HW; db [C] = scale * sum over m of g (threads with m == 0, fixed order)
template <typename T>
__global__ __launch_bounds__(NB) void mean_bias_nhwc_bwd_kernel(const float *__restrict__ g, int M, int HW, int C, float inv_hw, float scale,
                                                                T *__restrict__ dx, float *__restrict__ dbias)
{
    const int e = blockIdx.x * NB + threadIdx.x;
    if (e >= M * C) return;
    const int m = e / C, c = e - m * C;
    const T v = from_float<T>((g[e] * scale) * inv_hw);
    T *px = dx + (size_t)m * HW * C + c;
    for (int p = 0; p < HW; ++p) px[(size_t)p * C] = v;
    if (m == 0 && dbias) {
        float s = 0.f;
        for (int i = 0; i < M; ++i) s += g[(size_t)i * C + c];
        dbias[c] = s * scale;
    }
}

static inline int ph_vec(int dtype) { return dtype == PH_F32 ? 4 : 8; }
static int ph_args_ok(long long M, int C, int dtype, bool vectors)
{
    if (dtype != PH_F32 && dtype != PH_BF16) return MDX_ERR_BAD_SHAPE;
    if (M <= 0 || C <= 0 || M >= (1ll << 31) || M * C >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
    if (vectors && C % ph_vec(dtype)) return MDX_ERR_BAD_SHAPE;
    return MDX_OK;
}

}  // namespace nhwc
}  // namespace mdx

using namespace mdx;
using namespace mdx::nhwc;

MDX_EXPORT size_t mdx_bias_act_nhwc_workspace_bytes(int B, int C, int H, int W, int dtype)
{
    const long long M = (long long)B * H * W;
    if (ph_args_ok(M, C, dtype, true)) return 0;
    return (size_t)make_rows(M, C, ph_vec(dtype), BA_BLOCKS, BA_ITERS).nblk * C * sizeof(float);
}

MDX_EXPORT int mdx_bias_act_nhwc_fwd(const void *x, const float *bias, void *y, int B, int C, int H, int W, int relu, int dtype,
                                     void *stream)
{
    if (!x || !bias || !y) return MDX_ERR_NULL_POINTER;
    const long long M = (long long)B * H * W;
    const int bad = ph_args_ok(M, C, dtype, true);
    if (bad) return bad;
    if (!aligned(x, 16) || !aligned(y, 16)) return MDX_ERR_MISALIGNED;
    const Rows g = make_rows(M, C, ph_vec(dtype), 4096, BA_ITERS);
    const dim3 grid(g.nblk, g.t.ny), block(NB);
    if (dtype == PH_F32)
        hipLaunchKernelGGL((bias_act_nhwc_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float *)x, bias, (int)M, C,
                           g.t.CVB, g.t.PL, g.RB, relu, (float *)y);
    else
        hipLaunchKernelGGL((bias_act_nhwc_fwd_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16 *)x, bias, (int)M, C,
                           g.t.CVB, g.t.PL, g.RB, relu, (bf16 *)y);
    return check_launch();
}

MDX_EXPORT int mdx_bias_act_nhwc_bwd(const void *dy, const void *y, void *dx, float *dbias, int B, int C, int H, int W, int relu,
                                     int dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!dy || !y || !dx || !dbias || !workspace) return MDX_ERR_NULL_POINTER;
    const long long M = (long long)B * H * W;
    const int bad = ph_args_ok(M, C, dtype, true);
    if (bad) return bad;
    if (!aligned(dy, 16) || !aligned(y, 16) || !aligned(dx, 16)) return MDX_ERR_MISALIGNED;
    if (workspace_bytes < mdx_bias_act_nhwc_workspace_bytes(B, C, H, W, dtype)) return MDX_ERR_WORKSPACE;
    const Rows g = make_rows(M, C, ph_vec(dtype), BA_BLOCKS, BA_ITERS);
    const dim3 grid(g.nblk, g.t.ny), block(NB);
    float *part = (float *)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PH_F32)
        hipLaunchKernelGGL((bias_act_nhwc_bwd_kernel<float>), grid, block, 0, st, (const float *)dy, (const float *)y, (int)M, C, g.t.CVB,
                           g.t.PL, g.RB, relu, (float *)dx, part);
    else
        hipLaunchKernelGGL((bias_act_nhwc_bwd_kernel<bf16>), grid, block, 0, st, (const bf16 *)dy, (const bf16 *)y, (int)M, C, g.t.CVB,
                           g.t.PL, g.RB, relu, (bf16 *)dx, part);
    hipLaunchKernelGGL(colsum_scale_kernel, dim3((C + PF_C - 1) / PF_C), dim3(PF_C * PF_S), 0, st, part, g.nblk, C, 1.0f, dbias);
    return check_launch();
}

MDX_EXPORT int mdx_mean_bias_nhwc_fwd(const void *x, const float *bias, float *out, int B, int C, int H, int W, float scale, int dtype,
                                      void *stream)
{
    if (!x || !out) return MDX_ERR_NULL_POINTER;
    const long long HW = (long long)H * W;
    const int bad = ph_args_ok((long long)B * HW, C, dtype, false);
    if (bad) return bad;
    const int n = B * C;
    const float inv = 1.0f / (float)HW;
    if (dtype == PH_F32)
        hipLaunchKernelGGL((mean_bias_nhwc_fwd_kernel<float>), dim3((n + NB - 1) / NB), dim3(NB), 0, (hipStream_t)stream, (const float *)x,
                           bias, B, (int)HW, C, inv, scale, out);
    else
        hipLaunchKernelGGL((mean_bias_nhwc_fwd_kernel<bf16>), dim3((n + NB - 1) / NB), dim3(NB), 0, (hipStream_t)stream, (const bf16 *)x,
                           bias, B, (int)HW, C, inv, scale, out);
    return check_launch();
}

MDX_EXPORT int mdx_mean_bias_nhwc_bwd(const float *gout, void *dx, float *dbias, int B, int C, int H, int W, float scale, int dtype,
                                      void *stream)
{
    if (!gout || !dx) return MDX_ERR_NULL_POINTER;
    const long long HW = (long long)H * W;
    const int bad = ph_args_ok((long long)B * HW, C, dtype, false);
    if (bad) return bad;
    const int n = B * C;
    const float inv = 1.0f / (float)HW;
    if (dtype == PH_F32)
        hipLaunchKernelGGL((mean_bias_nhwc_bwd_kernel<float>), dim3((n + NB - 1) / NB), dim3(NB), 0, (hipStream_t)stream, gout, B, (int)HW, C,
                           inv, scale, (float *)dx, dbias);
    else
        hipLaunchKernelGGL((mean_bias_nhwc_bwd_kernel<bf16>), dim3((n + NB - 1) / NB), dim3(NB), 0, (hipStream_t)stream, gout, B, (int)HW, C,
                           inv, scale, (bf16 *)dx, dbias);
    return check_launch();
}

// ---- the networks' input: (frame - 0.45) / 0.225 of up to two frames per image, written channels-last ----------------------------
// depth_encoder.py:89 normalises the input; processor.py:61-75 concatenates the frame pairs of the pose network along the channels
// (and this package the pairs along the batch); the convolution wants channels-last.  Three concatenations, a subtraction, a
// multiplication and a layout copy of a [2B][6][H][W] map -- 130 us at the very start of the pose network's chain -- are one pass:
// out[k * n + b][h][w][3 * g + c] = (src[k][g][b][c][h][w] - mean) * inv_std.
namespace mdx {
namespace nhwc {

struct InputSrc { const float *p[4]; };      // [block k][group g] -> p[2 * k + g], planar [n][3][H][W]

template <typename T>
__global__ __launch_bounds__(NB) void encoder_input_nhwc_kernel(InputSrc src, int groups, int n, int HW, float mean, float inv_std,
                                                                T *__restrict__ out)
{
    // grid (ceil(HW / NB), images): no division, every index in 32 bits
    const int pix = blockIdx.x * NB + threadIdx.x, img = blockIdx.y;
    if (pix >= HW) return;
    const int k = img >= n ? 1 : 0, b = img - k * n;
    const int C = 3 * groups;
    float v[6];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (g >= groups) break;
        const float *__restrict__ s = (k == 0 ? (g == 0 ? src.p[0] : src.p[1]) : (g == 0 ? src.p[2] : src.p[3])) + (size_t)b * 3 * HW + pix;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[3 * g + c] = (s[(size_t)c * HW] - mean) * inv_std;
    }
    T *o = out + ((size_t)img * HW + pix) * C;
#pragma unroll
    for (int c = 0; c < 6; ++c)
        if (c < C) o[c] = from_float<T>(v[c]);
}

}  // namespace nhwc
}  // namespace mdx

// src: HOST array of blocks * groups device pointers, each a planar float32 [n][3][H][W]; out [blocks * n][H][W][3 * groups] in
// dtype (0 float32 / 1 bfloat16).  inv_std is the float32 reciprocal ATen multiplies by for `/ 0.225`.
MDX_EXPORT int mdx_encoder_input_nhwc(const float *const *src, int blocks, int groups, int n, int H, int W, float mean, float inv_std,
                                      void *out, int dtype, void *stream)
{
    if (!src || !out) return MDX_ERR_NULL_POINTER;
    if (blocks < 1 || blocks > 2 || groups < 1 || groups > 2 || n <= 0 || H <= 0 || W <= 0) return MDX_ERR_BAD_SHAPE;
    if (dtype != PH_F32 && dtype != PH_BF16) return MDX_ERR_BAD_SHAPE;
    const long long total = (long long)blocks * n * H * W;
    if (total * 6 >= (1ll << 40)) return MDX_ERR_BAD_SHAPE;
    InputSrc s = {{nullptr, nullptr, nullptr, nullptr}};
    for (int k = 0; k < blocks; ++k)
        for (int g = 0; g < groups; ++g) {
            if (!src[k * groups + g]) return MDX_ERR_NULL_POINTER;
            s.p[2 * k + g] = src[k * groups + g];               // [block][group], two slots per block
        }
    if ((long long)H * W >= (1ll << 31) || blocks * n > 65535) return MDX_ERR_BAD_SHAPE;
    const dim3 grid((unsigned)(((long long)H * W + NB - 1) / NB), (unsigned)(blocks * n));
    if (dtype == PH_F32)
        hipLaunchKernelGGL((encoder_input_nhwc_kernel<float>), grid, dim3(NB), 0, (hipStream_t)stream, s, groups, n, H * W, mean, inv_std,
                           (float *)out);
    else
        hipLaunchKernelGGL((encoder_input_nhwc_kernel<bf16>), grid, dim3(NB), 0, (hipStream_t)stream, s, groups, n, H * W, mean, inv_std,
                           (bf16 *)out);
    return check_launch();
}
