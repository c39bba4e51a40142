// glue.hip -- memory-bound glue of the depth / pose networks around MIOpen's convolutions, gfx950.
//
//   decoder glue   model_layer/depth_decoder.py:44-47,96-106 : ELU -> nearest x2 -> concat encoder skip ->
//                  ReflectionPad2d(1), i.e. everything between two decoder convolutions, as ONE pass:
//                      out[b, c, yo, xo] = c < C1 ? act(raw[b, c, y / u, x / u]) : skip[b, c - C1, y, x]
//                      with (y, x) = reflect(yo - 1, xo - 1), u = 2 (upsample) or 1, act = ELU or identity.
//                  The unfused sequence moves ~23 N + 4 S floats through HBM (N = |raw|, S = |skip|), this ~5 N + 2 S.
//                  Backward is a gather (no atomics): each unpadded position sums the <= 2 x 2 padded positions that
//                  reflect onto it; the raw gradient additionally folds the 2 x 2 upsample block and ELU'.
//   max-pool 3x3 / stride 2 / pad 1   (ResNet stem, model_layer/depth_encoder.py): forward keeps the 0..8 window
//                  index of the first maximum (ATen's strict-> scan), backward gathers from the <= 4 windows that
//                  contain an input position.
// float32 and bfloat16 (networks under autocast) storage, arithmetic in float32.  Thread <-> x: rows coalesce.
#include "mdx_common.hpp"
#include <stdint.h>

namespace mdx {

struct bf16 { uint16_t v; };

__device__ __forceinline__ float to_float(float x) { return x; }
__device__ __forceinline__ float to_float(bf16 x) { return __uint_as_float((uint32_t)x.v << 16); }
template <typename T> __device__ __forceinline__ T from_float(float x);
template <> __device__ __forceinline__ float from_float<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_float<bf16>(float x)
{
    // round to nearest even, NaN stays NaN (torch's float -> bfloat16)
    uint32_t u = __float_as_uint(x);
    bf16 r;
    if ((u & 0x7fffffffu) > 0x7f800000u) { r.v = (uint16_t)((u >> 16) | 0x40u); return r; }
    u += 0x7fffu + ((u >> 16) & 1u);
    r.v = (uint16_t)(u >> 16);
    return r;
}

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }   // ATen: exp(x) - 1

constexpr int GX = 64, GY = 4;   // block = 64 x 4 threads: a wave per output row segment
constexpr int RPT = 4;           // rows per thread (GY apart): four independent loads in flight per lane --
                                 // with one 4-byte access per thread a CU cannot keep enough bytes in flight

// ---- decoder glue, forward.  grid: (ceil(Wp/64), ceil(Hp/4), B*(C1+C2)) ----
template <typename TI, typename TO>
__global__ __launch_bounds__(GX *GY) void decoder_glue_fwd_kernel(const TI *__restrict__ raw, const TI *__restrict__ skip,
                                                                   const float *__restrict__ bias, TO *__restrict__ out,
                                                                   int C1, int C2, int h, int w, int up, int elu)
{
    const int u = up ? 2 : 1;
    const int H = h * u, W = w * u, Hp = H + 2, Wp = W + 2;
    const int xo = blockIdx.x * GX + threadIdx.x, y0 = blockIdx.y * (GY * RPT) + threadIdx.y;
    if (xo >= Wp) return;
    const int bc = blockIdx.z, C = C1 + C2, b = bc / C, c = bc - b * C;
    const int x = reflect1(xo - 1, W);
    float v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yo = y0 + r * GY;
        const int y = reflect1((yo < Hp ? yo : 0) - 1, H);
        v[r] = c < C1 ? to_float(raw[(((size_t)b * C1 + c) * h + y / u) * w + x / u])
                      : to_float(skip[(((size_t)b * C2 + (c - C1)) * H + y) * W + x]);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yo = y0 + r * GY;
        if (yo >= Hp) break;
        float f = v[r];
        if (c < C1) {
            if (bias) f += bias[c];          // the convolution ran without its bias: added here, on the way in
            if (elu) f = elu1(f);
        }
        out[((size_t)bc * Hp + yo) * Wp + xo] = from_float<TO>(f);
    }
}

// Sum of the padded-gradient positions that reflect onto unpadded (y, x); gp = plane [Hp][Wp].  The position's own image
// (y+1, x+1) always exists and comes first in the sum; the up to eight others exist only on the rows 1 / H-2 and columns
// 1 / W-2.  Round 4: the own position is loaded UNCONDITIONALLY by the callers (all rows and sub-pixels of a thread back to back)
// and the others are added under one rare branch -- with every one of the nine positions behind its own `if`, each load sat in
// an exec-masked block followed by s_waitcnt vmcnt(0): 36 memory round trips in a row per thread of the raw-gradient kernel
// (tools/isa_loadwaits.py).  Same terms in the same order: the same bits.
__device__ __forceinline__ bool on_fold_ring(int y, int x, int H, int W) { return y == 1 || y == H - 2 || x == 1 || x == W - 2; }
template <typename T> __device__ __forceinline__ float fold_rest(float acc, const T *__restrict__ gp, int y, int x, int H, int W)
{
    const int Wp = W + 2;
    const int ys[3] = {y + 1, y == 1 ? 0 : -1, y == H - 2 ? H + 1 : -1};
    const int xs[3] = {x + 1, x == 1 ? 0 : -1, x == W - 2 ? W + 1 : -1};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (ys[i] < 0) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (xs[j] >= 0 && (i | j) != 0) acc += to_float(gp[(size_t)ys[i] * Wp + xs[j]]);
    }
    return acc;
}

// ---- decoder glue, backward w.r.t. raw.  grid: (ceil(w/64), ceil(h/4), B*C1) ----
template <typename TI, typename TO, int U>
__global__ __launch_bounds__(GX *GY) void decoder_glue_bwd_raw_kernel(const TO *__restrict__ gout, const TI *__restrict__ raw,
                                                                       const float *__restrict__ bias, TI *__restrict__ graw,
                                                                       float *__restrict__ bias_part, int C1, int C2, int h,
                                                                       int w, int elu)
{
    __shared__ float lds[GX * GY / 64];
    const int H = h * U, W = w * U, Wp = W + 2;
    const int xx_raw = blockIdx.x * GX + threadIdx.x, y0 = blockIdx.y * (GY * RPT) + threadIdx.y;
    const bool in_x = xx_raw < w;
    const int xx = in_x ? xx_raw : 0;
    const int bc = blockIdx.z, b = bc / C1, c = bc - b * C1;
    const TO *gp = gout + ((size_t)b * (C1 + C2) + c) * (size_t)(H + 2) * Wp;
    const float bv = bias ? bias[c] : 0.f;
    float own[RPT][U * U], rv[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yy = (y0 + r * GY < h) ? y0 + r * GY : 0;
#pragma unroll
        for (int dy = 0; dy < U; ++dy)
#pragma unroll
            for (int dx = 0; dx < U; ++dx) own[r][dy * U + dx] = to_float(gp[(size_t)(U * yy + dy + 1) * Wp + (U * xx + dx + 1)]);
        rv[r] = to_float(raw[((size_t)bc * h + yy) * w + xx]);
    }
    float g[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yy = (y0 + r * GY < h) ? y0 + r * GY : 0;
        g[r] = 0.f;
#pragma unroll
        for (int dy = 0; dy < U; ++dy)
#pragma unroll
            for (int dx = 0; dx < U; ++dx) {
                float f = own[r][dy * U + dx];
                if (on_fold_ring(U * yy + dy, U * xx + dx, H, W)) f = fold_rest(f, gp, U * yy + dy, U * xx + dx, H, W);
                g[r] += f;
            }
        rv[r] = elu ? rv[r] + bv : 1.f;
    }
    float bsum = 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yy = y0 + r * GY;
        if (yy >= h || !in_x) continue;
        const float gr = (elu && !(rv[r] > 0.f)) ? g[r] * __expf(rv[r]) : g[r];
        const TI stored = from_float<TI>(gr);
        graw[((size_t)bc * h + yy) * w + xx] = stored;
        bsum += to_float(stored);
    }
    if (bias_part) {
        // d(bias)[c] = sum of graw over (b, y, x): one partial per block, summed in fixed order by the finish kernel
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bsum += __shfl_down(bsum, off, 64);
        const int tid = threadIdx.y * GX + threadIdx.x;
        if ((tid & 63) == 0) lds[tid >> 6] = bsum;
        __syncthreads();
        if (tid == 0)
            bias_part[((size_t)bc * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    }
}

// d(bias)[c] = sum over b and the blocks of plane (b, c): one wave per channel, fixed order
__global__ __launch_bounds__(64) void decoder_glue_bias_finish_kernel(const float *__restrict__ part, int B, int C1,
                                                                       int nblk, float *__restrict__ dbias)
{
    const int c = blockIdx.x;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *p = part + ((size_t)b * C1 + c) * nblk;
        for (int i = threadIdx.x; i < nblk; i += 64) acc += p[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (threadIdx.x == 0) dbias[c] = acc;
}

// ---- decoder glue, backward w.r.t. skip.  grid: (ceil(W/64), ceil(H/4), B*C2) ----
template <typename TI, typename TO>
__global__ __launch_bounds__(GX *GY) void decoder_glue_bwd_skip_kernel(const TO *__restrict__ gout, TI *__restrict__ gskip,
                                                                        int C1, int C2, int H, int W)
{
    const int x = blockIdx.x * GX + threadIdx.x, y0 = blockIdx.y * (GY * RPT) + threadIdx.y;
    if (x >= W) return;
    const int bc = blockIdx.z, b = bc / C2, c = bc - b * C2;
    const TO *gp = gout + ((size_t)b * (C1 + C2) + C1 + c) * (size_t)(H + 2) * (W + 2);
    float g[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) g[r] = to_float(gp[(size_t)(((y0 + r * GY < H) ? y0 + r * GY : 0) + 1) * (W + 2) + x + 1]);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int y = y0 + r * GY;
        if (y >= H) break;
        if (on_fold_ring(y, x, H, W)) g[r] = fold_rest(g[r], gp, y, x, H, W);
        gskip[((size_t)bc * H + y) * W + x] = from_float<TI>(g[r]);
    }
}

// ---- max-pool 3x3 / 2 / 1.  grid: (ceil(Wo/64), ceil(Ho/4), B*C) ----
template <typename T>
__global__ __launch_bounds__(GX *GY) void maxpool3s2_fwd_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                                 uint8_t *__restrict__ arg, int H, int W, int Ho, int Wo)
{
    const int xo = blockIdx.x * GX + threadIdx.x, yo = blockIdx.y * GY + threadIdx.y;
    if (xo >= Wo || yo >= Ho) return;
    const T *p = in + (size_t)blockIdx.z * H * W;
    float best = -INFINITY;
    int bi = -1;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int y = 2 * yo - 1 + k / 3, x = 2 * xo - 1 + k % 3;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        const float v = to_float(p[(size_t)y * W + x]);
        if (bi < 0) bi = k;                               // ATen starts from the first in-bounds tap
        if (v > best || v != v) { best = v; bi = k; }     // strict >: the first maximum wins; NaN propagates
    }
    const size_t o = ((size_t)blockIdx.z * Ho + yo) * Wo + xo;
    out[o] = from_float<T>(best);
    arg[o] = (uint8_t)bi;
}

// W % 4 == 0 variant: one thread per TWO adjacent outputs (xo = 2j, 2j+1).  Their windows span input columns
// 4j-1 .. 4j+3: per row one aligned 4-element load plus one scalar, instead of six scalar loads.
// grid: (ceil(Wo/2/64), ceil(Ho/4), B*C)
template <typename T> struct __attribute__((aligned(sizeof(T) * 4))) Quad { T v[4]; };
template <typename T> struct __attribute__((aligned(sizeof(T) * 2))) Pair { T v[2]; };

template <typename T>
__global__ __launch_bounds__(GX *GY) void maxpool3s2_fwd2_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                                  uint8_t *__restrict__ arg, int H, int W, int Ho, int Wo)
{
    const int j = blockIdx.x * GX + threadIdx.x, yo = blockIdx.y * GY + threadIdx.y;
    if (2 * j >= Wo || yo >= Ho) return;
    const T *p = in + (size_t)blockIdx.z * H * W;
    float row[3][5];
    bool rok[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int y = 2 * yo - 1 + ky;
        rok[ky] = y >= 0 && y < H;
        const size_t base = (size_t)(rok[ky] ? y : 0) * W + 4 * j;
        const Quad<T> q = *reinterpret_cast<const Quad<T> *>(p + base);
        row[ky][0] = j > 0 ? to_float(p[base - 1]) : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) row[ky][1 + k] = to_float(q.v[k]);
    }
    float best[2] = {-INFINITY, -INFINITY};
    int bi[2] = {-1, -1};
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int ky = k / 3, kx = k % 3;
            if (!rok[ky] || (o == 0 && kx == 0 && j == 0)) continue;       // outside the image
            const float v = row[ky][2 * o + kx];
            if (bi[o] < 0) bi[o] = k;
            if (v > best[o] || v != v) { best[o] = v; bi[o] = k; }
        }
    const size_t oo = ((size_t)blockIdx.z * Ho + yo) * Wo + 2 * j;
    Pair<T> pv;
    pv.v[0] = from_float<T>(best[0]);
    pv.v[1] = from_float<T>(best[1]);
    *reinterpret_cast<Pair<T> *>(out + oo) = pv;
    *reinterpret_cast<uint16_t *>(arg + oo) = (uint16_t)(bi[0] | (bi[1] << 8));
}

// One thread per 2x2 input block (rows 2*yo, 2*yo+1; columns 2*xo, 2*xo+1): only the four windows (yo..yo+1,
// xo..xo+1) can have selected one of its pixels, so four index bytes and four gradients serve four outputs.
// grid: (ceil(ceil(W/2)/64), ceil(ceil(H/2)/4), B*C)
template <typename T>
__global__ __launch_bounds__(GX *GY) void maxpool3s2_bwd_kernel(const T *__restrict__ gout, const uint8_t *__restrict__ arg,
                                                                 T *__restrict__ gin, int H, int W, int Ho, int Wo)
{
    const int xo = blockIdx.x * GX + threadIdx.x, yo = blockIdx.y * GY + threadIdx.y;
    if (2 * xo >= W || 2 * yo >= H) return;
    const T *gp = gout + (size_t)blockIdx.z * Ho * Wo;
    const uint8_t *ap = arg + (size_t)blockIdx.z * Ho * Wo;
    float gw[2][2];
    int aw[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool in = (yo + j < Ho) && (xo + i < Wo);
            const size_t o = (size_t)(in ? yo + j : yo) * Wo + (in ? xo + i : xo);
            aw[j][i] = in ? (int)ap[o] : -1;
            gw[j][i] = to_float(gp[o]);
        }
    T *op = gin + (size_t)blockIdx.z * H * W;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int y = 2 * yo + a;
        if (y >= H) break;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int x = 2 * xo + b;
            if (x >= W) break;
            // pixel (a, b) of the block is tap (a+1, b+1) of window (yo, xo), tap (a+1, 0) of (yo, xo+1) when b == 1,
            // tap (0, b+1) of (yo+1, xo) when a == 1 and tap (0, 0) of (yo+1, xo+1) when a == b == 1
            float g = aw[0][0] == (a + 1) * 3 + (b + 1) ? gw[0][0] : 0.f;
            if (b == 1 && aw[0][1] == (a + 1) * 3) g += gw[0][1];
            if (a == 1 && aw[1][0] == (b + 1)) g += gw[1][0];
            if (a == 1 && b == 1 && aw[1][1] == 0) g += gw[1][1];
            op[(size_t)y * W + x] = from_float<T>(g);
        }
    }
}

static inline dim3 grid3(int nx, int ny, int nz) { return dim3((nx + GX - 1) / GX, (ny + GY - 1) / GY, nz); }
static inline dim3 grid3r(int nx, int ny, int nz) { return dim3((nx + GX - 1) / GX, (ny + GY * RPT - 1) / (GY * RPT), nz); }

}  // namespace mdx

using namespace mdx;

// dtype codes of the glue entry points
enum { MDX_F32 = 0, MDX_BF16 = 1 };

MDX_EXPORT int mdx_decoder_glue_fwd(const void *raw, const void *skip, const float *bias, void *out, int B, int C1, int C2,
                                    int h, int w, int upsample, int elu, int in_dtype, int out_dtype, void *stream)
{
    if (!raw || !out || (C2 > 0 && !skip)) return MDX_ERR_NULL_POINTER;
    const int u = upsample ? 2 : 1;
    if (B <= 0 || C1 <= 0 || C2 < 0 || h <= 0 || w <= 0 || h * u < 2 || w * u < 2 || (long long)B * (C1 + C2) > 65535)
        return MDX_ERR_BAD_SHAPE;
    const dim3 grid = grid3r(w * u + 2, h * u + 2, B * (C1 + C2)), block(GX, GY);
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == MDX_F32 && out_dtype == MDX_F32)
        hipLaunchKernelGGL((decoder_glue_fwd_kernel<float, float>), grid, block, 0, st, (const float *)raw,
                           (const float *)skip, bias, (float *)out, C1, C2, h, w, upsample, elu);
    else if (in_dtype == MDX_BF16 && out_dtype == MDX_BF16)
        hipLaunchKernelGGL((decoder_glue_fwd_kernel<bf16, bf16>), grid, block, 0, st, (const bf16 *)raw,
                           (const bf16 *)skip, bias, (bf16 *)out, C1, C2, h, w, upsample, elu);
    else if (in_dtype == MDX_BF16 && out_dtype == MDX_F32)
        hipLaunchKernelGGL((decoder_glue_fwd_kernel<bf16, float>), grid, block, 0, st, (const bf16 *)raw,
                           (const bf16 *)skip, bias, (float *)out, C1, C2, h, w, upsample, elu);
    else
        return MDX_ERR_BAD_SHAPE;
    return check_launch();
}

MDX_EXPORT size_t mdx_decoder_glue_workspace_bytes(int B, int C1, int h, int w)
{
    if (B <= 0 || C1 <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)B * C1 * ((w + GX - 1) / GX) * ((h + GY * RPT - 1) / (GY * RPT)) * sizeof(float);
}

MDX_EXPORT int mdx_decoder_glue_bwd(const void *gout, const void *raw, const float *bias, void *graw, void *gskip,
                                    float *dbias, int B, int C1, int C2, int h, int w, int upsample, int elu,
                                    int in_dtype, int out_dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!gout || !raw || !graw || (C2 > 0 && !gskip)) return MDX_ERR_NULL_POINTER;
    if (dbias && (!workspace || workspace_bytes < mdx_decoder_glue_workspace_bytes(B, C1, h, w))) return MDX_ERR_WORKSPACE;
    float *bias_part = dbias ? (float *)workspace : nullptr;
    const int u = upsample ? 2 : 1;
    if (B <= 0 || C1 <= 0 || C2 < 0 || h <= 0 || w <= 0 || h * u < 2 || w * u < 2 || (long long)B * (C1 + C2) > 65535)
        return MDX_ERR_BAD_SHAPE;
    const dim3 block(GX, GY), graw_grid = grid3r(w, h, B * C1), gskip_grid = grid3r(w * u, h * u, B * (C2 > 0 ? C2 : 1));
    hipStream_t st = (hipStream_t)stream;
#define MDX_GLUE_BWD(TI, TO)                                                                                          \
    do {                                                                                                              \
        if (upsample)                                                                                                 \
            hipLaunchKernelGGL((decoder_glue_bwd_raw_kernel<TI, TO, 2>), graw_grid, block, 0, st, (const TO *)gout,   \
                               (const TI *)raw, bias, (TI *)graw, bias_part, C1, C2, h, w, elu);                       \
        else                                                                                                          \
            hipLaunchKernelGGL((decoder_glue_bwd_raw_kernel<TI, TO, 1>), graw_grid, block, 0, st, (const TO *)gout,   \
                               (const TI *)raw, bias, (TI *)graw, bias_part, C1, C2, h, w, elu);                       \
        if (C2 > 0)                                                                                                   \
            hipLaunchKernelGGL((decoder_glue_bwd_skip_kernel<TI, TO>), gskip_grid, block, 0, st, (const TO *)gout,    \
                               (TI *)gskip, C1, C2, h * u, w * u);                                                     \
    } while (0)
    if (in_dtype == MDX_F32 && out_dtype == MDX_F32) MDX_GLUE_BWD(float, float);
    else if (in_dtype == MDX_BF16 && out_dtype == MDX_BF16) MDX_GLUE_BWD(bf16, bf16);
    else if (in_dtype == MDX_BF16 && out_dtype == MDX_F32) MDX_GLUE_BWD(bf16, float);
    else return MDX_ERR_BAD_SHAPE;
#undef MDX_GLUE_BWD
    if (dbias)
        hipLaunchKernelGGL(decoder_glue_bias_finish_kernel, dim3(C1), dim3(64), 0, st, bias_part, B, C1,
                           (int)(graw_grid.x * graw_grid.y), dbias);
    return check_launch();
}

MDX_EXPORT int mdx_maxpool3s2_fwd(const void *in, void *out, uint8_t *arg, int BC, int H, int W, int dtype, void *stream)
{
    if (!in || !out || !arg) return MDX_ERR_NULL_POINTER;
    if (BC <= 0 || BC > 65535 || H <= 0 || W <= 0) return MDX_ERR_BAD_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;   // floor((H + 2 - 3) / 2) + 1
    const dim3 grid = grid3(Wo, Ho, BC), block(GX, GY);
    if (dtype != MDX_F32 && dtype != MDX_BF16) return MDX_ERR_BAD_SHAPE;
    if (W % 4 == 0) {      // rows 16-byte (8-byte for bf16) aligned, Wo even: two outputs per thread
        const dim3 grid2 = grid3(Wo / 2, Ho, BC);
        if (dtype == MDX_F32)
            hipLaunchKernelGGL((maxpool3s2_fwd2_kernel<float>), grid2, block, 0, (hipStream_t)stream, (const float *)in,
                               (float *)out, arg, H, W, Ho, Wo);
        else
            hipLaunchKernelGGL((maxpool3s2_fwd2_kernel<bf16>), grid2, block, 0, (hipStream_t)stream, (const bf16 *)in,
                               (bf16 *)out, arg, H, W, Ho, Wo);
        return check_launch();
    }
    if (dtype == MDX_F32)
        hipLaunchKernelGGL((maxpool3s2_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float *)in,
                           (float *)out, arg, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL((maxpool3s2_fwd_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16 *)in,
                           (bf16 *)out, arg, H, W, Ho, Wo);
    return check_launch();
}

MDX_EXPORT int mdx_maxpool3s2_bwd(const void *gout, const uint8_t *arg, void *gin, int BC, int H, int W, int dtype,
                                  void *stream)
{
    if (!gout || !arg || !gin) return MDX_ERR_NULL_POINTER;
    if (BC <= 0 || BC > 65535 || H <= 0 || W <= 0) return MDX_ERR_BAD_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const dim3 grid = grid3((W + 1) / 2, (H + 1) / 2, BC), block(GX, GY);
    if (dtype == MDX_F32)
        hipLaunchKernelGGL((maxpool3s2_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float *)gout, arg,
                           (float *)gin, H, W, Ho, Wo);
    else if (dtype == MDX_BF16)
        hipLaunchKernelGGL((maxpool3s2_bwd_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16 *)gout, arg,
                           (bf16 *)gin, H, W, Ho, Wo);
    else
        return MDX_ERR_BAD_SHAPE;
    return check_launch();
}
