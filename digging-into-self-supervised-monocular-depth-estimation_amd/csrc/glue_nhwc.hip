// glue_nhwc.hip -- the network glue of glue.hip for CHANNELS-LAST maps ([B][H][W][C] memory), gfx950.
//
//   decoder glue   model_layer/depth_decoder.py:44-47,96-106 : (+ bias) -> ELU -> nearest x2 -> concat encoder skip ->
//                  ReflectionPad2d(1) as ONE pass, and its backward as gathers (no atomics), d(bias) included;
//   max-pool 3x3 / stride 2 / pad 1   (ResNet stem): forward keeps the window tap of the first maximum per channel,
//                  backward gathers from the <= 4 windows that contain an input position.
// Same arithmetic, same summation order per element as glue.hip; only the addressing differs: a thread owns one 16-byte
// channel vector (4 float32 / 8 bfloat16) of one pixel (nhwc_common.hpp), a block is CVB channel vectors x XL pixels of
// a row, and every thread works on RPT rows so that several 16-byte accesses are in flight per lane.
#include "nhwc_common.hpp"

namespace mdx {
namespace nhwc {

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }   // ATen: exp(x) - 1

constexpr int RPT = 4;           // rows per thread

// thread -> (pixel lane, channel vector) inside the block; x = position along the row
struct Lane { int x, cv; bool in_tile; };
__device__ __forceinline__ Lane lane_of(int CVB, int XL)
{
    Lane l;
    const int xl = threadIdx.x / CVB;
    l.x = blockIdx.y * XL + xl;
    l.cv = blockIdx.z * CVB + threadIdx.x % CVB;
    l.in_tile = xl < XL;
    return l;
}

// ---- decoder glue, forward.  rows = B * Hp; grid (ceil(rows / RPT), ceil(Wp / XL), ny) ----------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(NB) void decoder_glue_nhwc_fwd_kernel(const TI *__restrict__ raw, const TI *__restrict__ skip,
                                                                   const float *__restrict__ bias, TO *__restrict__ out, int B,
                                                                   int C1, int C2, int h, int w, int up, int elu, int CVB, int XL)
{
    constexpr int N = VecN<TI>::N;
    const int u = up ? 2 : 1;
    const int H = h * u, W = w * u, Hp = H + 2, Wp = W + 2, C = C1 + C2, rows = B * Hp;
    const Lane l = lane_of(CVB, XL);
    const int c0 = l.cv * N;
    if (!l.in_tile || l.x >= Wp || c0 >= C) return;
    const int x = reflect1(l.x - 1, W);
    const bool from_raw = c0 < C1;
    Vec<TI, N> v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = min(blockIdx.x * RPT + r, rows - 1);
        const int b = row / Hp, y = reflect1(row - b * Hp - 1, H);
        v[r] = from_raw ? load_vec<TI, N>(raw + (((size_t)b * h + y / u) * w + x / u) * C1 + c0)
                        : load_vec<TI, N>(skip + (((size_t)b * H + y) * W + x) * C2 + (c0 - C1));
    }
    float bv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) bv[j] = (from_raw && bias) ? bias[c0 + j] : 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = blockIdx.x * RPT + r;
        if (row >= rows) break;
        Vec<TO, N> o;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float f = to_float(v[r].v[j]);
            if (from_raw) {
                if (bias) f += bv[j];            // the convolution ran without its bias: added here, on the way in
                if (elu) f = elu1(f);
            }
            o.v[j] = from_float<TO>(f);
        }
        store_vec<TO, N>(out + ((size_t)row * Wp + l.x) * C + c0, o);
    }
}

// Sum of the padded-gradient positions that reflect onto unpadded (y, x); gp = this image's padded gradient at the thread's
// channel vector, pixel stride C.  acc already holds the position's own image (y+1, x+1); the up to eight others exist only
// on the rows 1 / H-2 and columns 1 / W-2 -- added in glue.hip's order.
__device__ __forceinline__ bool on_fold_ring(int y, int x, int H, int W) { return y == 1 || y == H - 2 || x == 1 || x == W - 2; }
template <typename T, int N>
__device__ __forceinline__ void fold_rest(float (&acc)[N], const T *__restrict__ gp, int y, int x, int H, int W, int C)
{
    const int Wp = W + 2;
    const int ys[3] = {y + 1, y == 1 ? 0 : -1, y == H - 2 ? H + 1 : -1};
    const int xs[3] = {x + 1, x == 1 ? 0 : -1, x == W - 2 ? W + 1 : -1};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (ys[i] < 0) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (xs[j] >= 0 && (i | j) != 0) {
                const Vec<T, N> g = load_vec<T, N>(gp + ((size_t)ys[i] * Wp + xs[j]) * C);
#pragma unroll
                for (int k = 0; k < N; ++k) acc[k] += to_float(g.v[k]);
            }
    }
}

// ---- decoder glue, backward w.r.t. raw.  rows = B * h; grid (ceil(rows / RPT), ceil(w / XL), ny) ------------------
template <typename TI, typename TO, int U>
__global__ __launch_bounds__(NB) void decoder_glue_nhwc_bwd_raw_kernel(const TO *__restrict__ gout, const TI *__restrict__ raw,
                                                                       const float *__restrict__ bias, TI *__restrict__ graw,
                                                                       float *__restrict__ bias_part, int B, int C1, int C2, int h,
                                                                       int w, int elu, int CVB, int XL)
{
    constexpr int N = VecN<TI>::N;
    __shared__ float lds[NB * N];
    const int H = h * U, W = w * U, Hp = H + 2, Wp = W + 2, C = C1 + C2, rows = B * h;
    const Lane l = lane_of(CVB, XL);
    const int c0 = l.cv * N;
    const bool active = l.in_tile && l.x < w && c0 < C1;
    float bsum[N];
#pragma unroll
    for (int j = 0; j < N; ++j) bsum[j] = 0.f;
    if (active) {
        float bv[N];
#pragma unroll
        for (int j = 0; j < N; ++j) bv[j] = bias ? bias[c0 + j] : 0.f;
        Vec<TO, N> own[RPT][U * U];
        Vec<TI, N> rv[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = min(blockIdx.x * RPT + r, rows - 1);
            const int b = row / h, yy = row - b * h;
            const TO *gp = gout + (size_t)b * Hp * Wp * C + c0;
#pragma unroll
            for (int dy = 0; dy < U; ++dy)
#pragma unroll
                for (int dx = 0; dx < U; ++dx)
                    own[r][dy * U + dx] = load_vec<TO, N>(gp + ((size_t)(U * yy + dy + 1) * Wp + (U * l.x + dx + 1)) * C);
            rv[r] = load_vec<TI, N>(raw + ((size_t)row * w + l.x) * C1 + c0);
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = blockIdx.x * RPT + r;
            if (row >= rows) break;
            const int b = row / h, yy = row - b * h;
            const TO *gp = gout + (size_t)b * Hp * Wp * C + c0;
            float g[N];
#pragma unroll
            for (int j = 0; j < N; ++j) g[j] = 0.f;
#pragma unroll
            for (int dy = 0; dy < U; ++dy)
#pragma unroll
                for (int dx = 0; dx < U; ++dx) {
                    float f[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) f[j] = to_float(own[r][dy * U + dx].v[j]);
                    if (on_fold_ring(U * yy + dy, U * l.x + dx, H, W)) fold_rest<TO, N>(f, gp, U * yy + dy, U * l.x + dx, H, W, C);
#pragma unroll
                    for (int j = 0; j < N; ++j) g[j] += f[j];
                }
            Vec<TI, N> o;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float pre = to_float(rv[r].v[j]) + bv[j];
                const float gr = (elu && !(pre > 0.f)) ? g[j] * __expf(pre) : g[j];
                o.v[j] = from_float<TI>(gr);
                bsum[j] += to_float(o.v[j]);
            }
            store_vec<TI, N>(graw + ((size_t)row * w + l.x) * C1 + c0, o);
        }
    }
    if (bias_part) {
        // d(bias)[c] = sum of graw over (b, y, x): one partial per block and channel, lanes summed in fixed order
#pragma unroll
        for (int j = 0; j < N; ++j) lds[threadIdx.x * N + j] = bsum[j];
        __syncthreads();
        const int width = CVB * N;
        float *dst = bias_part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * C1;
        for (int e = threadIdx.x; e < width; e += NB) {
            const int c = blockIdx.z * width + e;
            if (c >= C1) break;
            float s = 0.f;
            for (int k = 0; k < XL; ++k) s += lds[k * width + e];
            dst[c] = s;
        }
    }
}

// column sums of part [n][C] -> out [C]: a block = 16 channels x 64 sub-sums (1024 threads), eight independent loads per
// thread and round (the partials come from blocks on all eight XCDs: a loop of single loads is a chain of fabric round trips),
// fixed order: sub-sums in float32, combined per wave by shuffles and across the 16 waves through LDS
constexpr int CS_C = 16, CS_S = 64, CS_U = 8;
__global__ __launch_bounds__(CS_C *CS_S) void colsum_finish_kernel(const float *__restrict__ part, int n, int C, float *__restrict__ out)
{
    __shared__ float lds[CS_S / 4][CS_C];
    const int cl = threadIdx.x % CS_C, sl = threadIdx.x / CS_C, c = blockIdx.x * CS_C + cl;
    float s = 0.f;
    for (int i0 = sl; i0 < n; i0 += CS_S * CS_U) {
        float v[CS_U];
#pragma unroll
        for (int u = 0; u < CS_U; ++u) {
            const int i = i0 + u * CS_S;
            v[u] = (c < C && i < n) ? part[(size_t)i * C + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < CS_U; ++u) s += v[u];
    }
    s += __shfl_down(s, 32, 64);
    s += __shfl_down(s, 16, 64);
    if ((threadIdx.x & 63) < CS_C) lds[threadIdx.x >> 6][cl] = s;
    __syncthreads();
    if (threadIdx.x < CS_C && c < C) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < CS_S / 4; ++k) tot += lds[k][cl];
        out[c] = tot;
    }
}

// ---- decoder glue, backward w.r.t. skip.  rows = B * H; grid (ceil(rows / RPT), ceil(W / XL), ny over C2) ----------
template <typename TI, typename TO>
__global__ __launch_bounds__(NB) void decoder_glue_nhwc_bwd_skip_kernel(const TO *__restrict__ gout, TI *__restrict__ gskip, int B,
                                                                        int C1, int C2, int H, int W, int CVB, int XL)
{
    constexpr int N = VecN<TI>::N;
    const int Hp = H + 2, Wp = W + 2, C = C1 + C2, rows = B * H;
    const Lane l = lane_of(CVB, XL);
    const int c0 = l.cv * N;
    if (!l.in_tile || l.x >= W || c0 >= C2) return;
    Vec<TO, N> own[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = min(blockIdx.x * RPT + r, rows - 1);
        const int b = row / H, y = row - b * H;
        own[r] = load_vec<TO, N>(gout + (((size_t)b * Hp + y + 1) * Wp + l.x + 1) * C + C1 + c0);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int row = blockIdx.x * RPT + r;
        if (row >= rows) break;
        const int b = row / H, y = row - b * H;
        float g[N];
#pragma unroll
        for (int j = 0; j < N; ++j) g[j] = to_float(own[r].v[j]);
        if (on_fold_ring(y, l.x, H, W)) fold_rest<TO, N>(g, gout + (size_t)b * Hp * Wp * C + C1 + c0, y, l.x, H, W, C);
        Vec<TI, N> o;
#pragma unroll
        for (int j = 0; j < N; ++j) o.v[j] = from_float<TI>(g[j]);
        store_vec<TI, N>(gskip + ((size_t)row * W + l.x) * C2 + c0, o);
    }
}

// ---- max-pool 3x3 / 2 / 1, forward.  rows = B * Ho; grid (rows, ceil(Wo / XL), ny) --------------------------------
template <typename T>
__global__ __launch_bounds__(NB) void maxpool3s2_nhwc_fwd_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                                 uint8_t *__restrict__ arg, int C, int H, int W, int Ho, int Wo,
                                                                 int CVB, int XL)
{
    constexpr int N = VecN<T>::N;
    const Lane l = lane_of(CVB, XL);
    const int c0 = l.cv * N;
    if (!l.in_tile || l.x >= Wo || c0 >= C) return;
    const int b = blockIdx.x / Ho, yo = blockIdx.x - b * Ho;
    const T *p = in + (size_t)b * H * W * C + c0;
    Vec<T, N> tap[9];
    bool ok[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int y = 2 * yo - 1 + k / 3, x = 2 * l.x - 1 + k % 3;
        ok[k] = y >= 0 && y < H && x >= 0 && x < W;
        tap[k] = load_vec<T, N>(p + ((size_t)(ok[k] ? y : 0) * W + (ok[k] ? x : 0)) * C);
    }
    Vec<T, N> o;
    Vec<uint8_t, N> a;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float best = -INFINITY;
        int bi = -1;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (!ok[k]) continue;
            const float v = to_float(tap[k].v[j]);
            if (bi < 0) bi = k;                               // ATen starts from the first in-bounds tap
            if (v > best || v != v) { best = v; bi = k; }     // strict >: the first maximum wins; NaN propagates
        }
        o.v[j] = from_float<T>(best);
        a.v[j] = (uint8_t)bi;
    }
    const size_t off = ((size_t)blockIdx.x * Wo + l.x) * C + c0;
    store_vec<T, N>(out + off, o);
    store_vec<uint8_t, N>(arg + off, a);
}

// ---- max-pool backward.  rows = B * H (input rows); grid (rows, ceil(W / XL), ny) ---------------------------------
// input pixel (y, x) is tap (y - 2 yo + 1, x - 2 xo + 1) of window (yo, xo): one window per axis for an even coordinate,
// two for an odd one.  Candidates in (yo, xo) ascending order -- glue.hip's order of additions.
template <typename T>
__global__ __launch_bounds__(NB) void maxpool3s2_nhwc_bwd_kernel(const T *__restrict__ gout, const T *__restrict__ gout2,
                                                                 const uint8_t *__restrict__ arg, T *__restrict__ gin, int C, int H,
                                                                 int W, int Ho, int Wo, int CVB, int XL)
{
    constexpr int N = VecN<T>::N;
    const Lane l = lane_of(CVB, XL);
    const int c0 = l.cv * N;
    if (!l.in_tile || l.x >= W || c0 >= C) return;
    const int b = blockIdx.x / H, y = blockIdx.x - b * H, x = l.x;
    const int yo0 = y >> 1, xo0 = x >> 1;                    // even: the only window; odd: the first of two
    const int ny = (y & 1) ? 2 : 1, nx = (x & 1) ? 2 : 1;
    const size_t img = (size_t)b * Ho * Wo * C + c0;
    float g[N];
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = 0.f;
    Vec<T, N> gv[2][2], gw[2][2] = {};
    Vec<uint8_t, N> av[2][2];
    bool ok[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int yo = yo0 + i, xo = xo0 + k;
            ok[i][k] = i < ny && k < nx && yo < Ho && xo < Wo;
            gv[i][k] = Vec<T, N>{};
            av[i][k] = Vec<uint8_t, N>{};
            if (ok[i][k]) {            // an even coordinate lies in ONE window of its axis: 2.25 windows per pixel on average, not 4
                const size_t off = img + ((size_t)yo * Wo + xo) * C;
                gv[i][k] = load_vec<T, N>(gout + off);
                if (gout2) gw[i][k] = load_vec<T, N>(gout2 + off);
                av[i][k] = load_vec<uint8_t, N>(arg + off);
            }
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!ok[i][k]) continue;
            const int tapi = (y - 2 * (yo0 + i) + 1) * 3 + (x - 2 * (xo0 + k) + 1);
#pragma unroll
            for (int j = 0; j < N; ++j)
                if ((int)av[i][k].v[j] == tapi) g[j] += gout2 ? to_float(gv[i][k].v[j]) + to_float(gw[i][k].v[j]) : to_float(gv[i][k].v[j]);
        }
    Vec<T, N> o;
#pragma unroll
    for (int j = 0; j < N; ++j) o.v[j] = from_float<T>(g[j]);
    store_vec<T, N>(gin + ((size_t)blockIdx.x * W + x) * C + c0, o);
}

static inline int vec_elems(int dtype) { return dtype == 0 ? 4 : 8; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace nhwc
}  // namespace mdx

using namespace mdx;
using namespace mdx::nhwc;

enum { MDX_F32 = 0, MDX_BF16 = 1 };

static int glue_shape_ok(int B, int C1, int C2, int h, int w, int upsample, int in_dtype)
{
    const int u = upsample ? 2 : 1, N = vec_elems(in_dtype);
    if (B <= 0 || C1 <= 0 || C2 < 0 || h <= 0 || w <= 0 || h * u < 2 || w * u < 2) return MDX_ERR_BAD_SHAPE;
    if (C1 % N || C2 % N) return MDX_ERR_BAD_SHAPE;                 // a channel vector never straddles raw | skip
    if ((long long)B * (h * u + 2) * (w * u + 2) * (C1 + C2) >= (1ll << 40)) return MDX_ERR_BAD_SHAPE;
    if ((long long)B * (h * u + 2) >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
    return MDX_OK;
}

/* channels-last decoder glue: raw [B][h][w][C1], skip [B][u*h][u*w][C2], out [B][u*h+2][u*w+2][C1+C2] */
MDX_EXPORT int mdx_decoder_glue_nhwc_fwd(const void *raw, const void *skip, const float *bias, void *out, int B, int C1, int C2,
                                         int h, int w, int upsample, int elu, int in_dtype, int out_dtype, void *stream)
{
    if (!raw || !out || (C2 > 0 && !skip)) return MDX_ERR_NULL_POINTER;
    if ((in_dtype != MDX_F32 && in_dtype != MDX_BF16) || (out_dtype != MDX_F32 && out_dtype != MDX_BF16)) return MDX_ERR_BAD_SHAPE;
    const int bad = glue_shape_ok(B, C1, C2, h, w, upsample, in_dtype);
    if (bad) return bad;
    const size_t out_align = (size_t)vec_elems(in_dtype) * (out_dtype == MDX_F32 ? 4 : 2);   // one channel vector of `out`
    if (!aligned(raw, 16) || !aligned(out, out_align) || (skip && !aligned(skip, 16))) return MDX_ERR_MISALIGNED;
    const int u = upsample ? 2 : 1, Hp = h * u + 2, Wp = w * u + 2;
    const Tile t = make_tile(C1 + C2, vec_elems(in_dtype));
    const dim3 grid(ceil_div(B * Hp, RPT), ceil_div(Wp, t.PL), t.ny), block(NB);
    if (grid.y > 65535 || grid.z > 65535) return MDX_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == MDX_F32 && out_dtype == MDX_F32)
        hipLaunchKernelGGL((decoder_glue_nhwc_fwd_kernel<float, float>), grid, block, 0, st, (const float *)raw, (const float *)skip,
                           bias, (float *)out, B, C1, C2, h, w, upsample, elu, t.CVB, t.PL);
    else if (in_dtype == MDX_BF16 && out_dtype == MDX_BF16)
        hipLaunchKernelGGL((decoder_glue_nhwc_fwd_kernel<bf16, bf16>), grid, block, 0, st, (const bf16 *)raw, (const bf16 *)skip,
                           bias, (bf16 *)out, B, C1, C2, h, w, upsample, elu, t.CVB, t.PL);
    else if (in_dtype == MDX_BF16 && out_dtype == MDX_F32)
        hipLaunchKernelGGL((decoder_glue_nhwc_fwd_kernel<bf16, float>), grid, block, 0, st, (const bf16 *)raw, (const bf16 *)skip,
                           bias, (float *)out, B, C1, C2, h, w, upsample, elu, t.CVB, t.PL);
    else
        return MDX_ERR_BAD_SHAPE;
    return check_launch();
}

MDX_EXPORT size_t mdx_decoder_glue_nhwc_workspace_bytes(int B, int C1, int h, int w, int in_dtype)
{
    if (B <= 0 || C1 <= 0 || h <= 0 || w <= 0 || (in_dtype != MDX_F32 && in_dtype != MDX_BF16) || C1 % vec_elems(in_dtype)) return 0;
    const Tile t = make_tile(C1, vec_elems(in_dtype));
    return (size_t)ceil_div(B * h, RPT) * ceil_div(w, t.PL) * C1 * sizeof(float);
}

MDX_EXPORT int mdx_decoder_glue_nhwc_bwd(const void *gout, const void *raw, const float *bias, void *graw, void *gskip,
                                         float *dbias, int B, int C1, int C2, int h, int w, int upsample, int elu, int in_dtype,
                                         int out_dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!gout || !raw || !graw || (C2 > 0 && !gskip)) return MDX_ERR_NULL_POINTER;
    if ((in_dtype != MDX_F32 && in_dtype != MDX_BF16) || (out_dtype != MDX_F32 && out_dtype != MDX_BF16)) return MDX_ERR_BAD_SHAPE;
    const int bad = glue_shape_ok(B, C1, C2, h, w, upsample, in_dtype);
    if (bad) return bad;
    if (dbias && (!workspace || workspace_bytes < mdx_decoder_glue_nhwc_workspace_bytes(B, C1, h, w, in_dtype))) return MDX_ERR_WORKSPACE;
    const size_t out_align = (size_t)vec_elems(in_dtype) * (out_dtype == MDX_F32 ? 4 : 2);
    if (!aligned(gout, out_align) || !aligned(raw, 16) || !aligned(graw, 16) || (gskip && !aligned(gskip, 16))) return MDX_ERR_MISALIGNED;
    float *bias_part = dbias ? (float *)workspace : nullptr;
    const int u = upsample ? 2 : 1, N = vec_elems(in_dtype);
    const Tile t1 = make_tile(C1, N);
    const dim3 block(NB), graw_grid(ceil_div(B * h, RPT), ceil_div(w, t1.PL), t1.ny);
    if (graw_grid.y > 65535) return MDX_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
#define MDX_GLUE_BWD(TI, TO)                                                                                                     \
    do {                                                                                                                         \
        if (upsample)                                                                                                            \
            hipLaunchKernelGGL((decoder_glue_nhwc_bwd_raw_kernel<TI, TO, 2>), graw_grid, block, 0, st, (const TO *)gout,         \
                               (const TI *)raw, bias, (TI *)graw, bias_part, B, C1, C2, h, w, elu, t1.CVB, t1.PL);                \
        else                                                                                                                     \
            hipLaunchKernelGGL((decoder_glue_nhwc_bwd_raw_kernel<TI, TO, 1>), graw_grid, block, 0, st, (const TO *)gout,         \
                               (const TI *)raw, bias, (TI *)graw, bias_part, B, C1, C2, h, w, elu, t1.CVB, t1.PL);                \
        if (C2 > 0) {                                                                                                            \
            const Tile t2 = make_tile(C2, N);                                                                                    \
            const dim3 gskip_grid(ceil_div(B * h * u, RPT), ceil_div(w * u, t2.PL), t2.ny);                                      \
            hipLaunchKernelGGL((decoder_glue_nhwc_bwd_skip_kernel<TI, TO>), gskip_grid, block, 0, st, (const TO *)gout,          \
                               (TI *)gskip, B, C1, C2, h * u, w * u, t2.CVB, t2.PL);                                              \
        }                                                                                                                        \
    } while (0)
    if (in_dtype == MDX_F32 && out_dtype == MDX_F32) MDX_GLUE_BWD(float, float);
    else if (in_dtype == MDX_BF16 && out_dtype == MDX_BF16) MDX_GLUE_BWD(bf16, bf16);
    else if (in_dtype == MDX_BF16 && out_dtype == MDX_F32) MDX_GLUE_BWD(bf16, float);
    else return MDX_ERR_BAD_SHAPE;
#undef MDX_GLUE_BWD
    if (dbias)
        hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(C1, CS_C)), dim3(CS_C * CS_S), 0, st, bias_part,
                           (int)(graw_grid.x * graw_grid.y), C1, dbias);
    return check_launch();
}

/* channels-last max-pool: in [B][H][W][C] -> out, arg [B][Ho][Wo][C] */
MDX_EXPORT int mdx_maxpool3s2_nhwc_fwd(const void *in, void *out, uint8_t *arg, int B, int C, int H, int W, int dtype,
                                       void *stream)
{
    if (!in || !out || !arg) return MDX_ERR_NULL_POINTER;
    if (dtype != MDX_F32 && dtype != MDX_BF16) return MDX_ERR_BAD_SHAPE;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || C % vec_elems(dtype)) return MDX_ERR_BAD_SHAPE;
    if (!aligned(in, 16) || !aligned(out, 16) || !aligned(arg, 8)) return MDX_ERR_MISALIGNED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;   // floor((H + 2 - 3) / 2) + 1
    if ((long long)B * H >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
    const Tile t = make_tile(C, vec_elems(dtype));
    const dim3 grid(B * Ho, ceil_div(Wo, t.PL), t.ny), block(NB);
    if (grid.y > 65535) return MDX_ERR_BAD_SHAPE;
    if (dtype == MDX_F32)
        hipLaunchKernelGGL((maxpool3s2_nhwc_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float *)in, (float *)out,
                           arg, C, H, W, Ho, Wo, t.CVB, t.PL);
    else
        hipLaunchKernelGGL((maxpool3s2_nhwc_fwd_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16 *)in, (bf16 *)out,
                           arg, C, H, W, Ho, Wo, t.CVB, t.PL);
    return check_launch();
}

MDX_EXPORT int mdx_maxpool3s2_nhwc_bwd(const void *gout, const void *gout2, const uint8_t *arg, void *gin, int B, int C, int H,
                                       int W, int dtype, void *stream)
{
    if (!gout || !arg || !gin) return MDX_ERR_NULL_POINTER;
    if (gout2 && !aligned(gout2, 16)) return MDX_ERR_MISALIGNED;
    if (dtype != MDX_F32 && dtype != MDX_BF16) return MDX_ERR_BAD_SHAPE;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || C % vec_elems(dtype)) return MDX_ERR_BAD_SHAPE;
    if (!aligned(gout, 16) || !aligned(gin, 16) || !aligned(arg, 8)) return MDX_ERR_MISALIGNED;
    if ((long long)B * H >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const Tile t = make_tile(C, vec_elems(dtype));
    const dim3 grid(B * H, ceil_div(W, t.PL), t.ny), block(NB);
    if (grid.y > 65535) return MDX_ERR_BAD_SHAPE;
    if (dtype == MDX_F32)
        hipLaunchKernelGGL((maxpool3s2_nhwc_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float *)gout,
                           (const float *)gout2, arg, (float *)gin, C, H, W, Ho, Wo, t.CVB, t.PL);
    else
        hipLaunchKernelGGL((maxpool3s2_nhwc_bwd_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16 *)gout,
                           (const bf16 *)gout2, arg, (bf16 *)gin, C, H, W, Ho, Wo, t.CVB, t.PL);
    return check_launch();
}
