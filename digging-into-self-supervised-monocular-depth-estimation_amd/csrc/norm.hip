// norm.hip -- training-mode BatchNorm2d fused with its residual add and ReLU, gfx950.
//
// The ResNet encoders (model_layer/depth_encoder.py; torchvision BasicBlock / Bottleneck layout) run
//     y = relu(bn(x))            and            y = relu(bn(x) + identity)
// 60 times per training step.  Unfused that is MIOpen's batch-norm (1-3 kernels) + an add + a clamp per layer
// forward, and a threshold + MIOpen's batch-norm backward (1-3 kernels) per layer backward: ~5 N / ~8 N floats of
// HBM traffic for an N-element map, and 4-8 launches of a few microseconds each for the small maps.  Here:
//     forward   stats pass (read x)  +  apply pass (read x [, identity], write y)              = 3 N (4 N)
//     backward  stats pass (read dy, y, x)  +  apply pass (read dy, y, x, write dx [, d_identity]) = 7 N (8 N)
// two launches each way (one for small maps, see below).  NCHW: channel c of image b is one contiguous plane of HW elements; a block owns a span of
// one plane (<= 8192 elements), reads it with 16-byte (float32) / 8-byte (bfloat16) loads, four per thread in flight.
// Per-channel sums: float32 inside a block (<= 8192 terms), combined across blocks in float64 (E[x^2] - mean^2 is
// formed in float64).  Statistics are those of torch.nn.functional.batch_norm: biased variance to normalise,
// unbiased for running_var, running = (1 - momentum) * running + momentum * batch.
// ReLU mask in the backward comes from y > 0 (y is alive anyway: the next convolution saved it).
#include "mdx_common.hpp"
#include <stdint.h>

namespace mdx {

struct bf16n { uint16_t v; };
__device__ __forceinline__ float ld(const float *p, size_t i) { return p[i]; }
__device__ __forceinline__ float ld(const bf16n *p, size_t i) { return __uint_as_float((uint32_t)p[i].v << 16); }
__device__ __forceinline__ void st(float *p, size_t i, float x) { p[i] = x; }
__device__ __forceinline__ void st(bf16n *p, size_t i, float x)
{
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) { p[i].v = (uint16_t)((u >> 16) | 0x40u); return; }
    u += 0x7fffu + ((u >> 16) & 1u);
    p[i].v = (uint16_t)(u >> 16);
}

constexpr int NB = 256;          // threads per block
constexpr int VEC = 4;           // elements per thread and access
constexpr int SPAN = 8192;       // elements of one plane a block owns (8 accesses per thread)

template <typename T> struct Vec4 { T v[VEC]; };
template <> struct __attribute__((aligned(16))) Vec4<float> { float v[VEC]; };
template <> struct __attribute__((aligned(8))) Vec4<bf16n> { bf16n v[VEC]; };

// the span [lo, hi) of plane (b, c) this block owns; grid = (C, B * K), K spans per plane
struct SpanId { int c, b, lo, hi; size_t plane; };
__device__ __forceinline__ SpanId span_id(int C, int HW, int K)
{
    SpanId s;
    s.c = blockIdx.x;
    s.b = blockIdx.y / K;
    const int k = blockIdx.y - s.b * K;
    const int per = ((HW + K - 1) / K + VEC - 1) / VEC * VEC;
    s.lo = k * per;
    s.hi = min(HW, s.lo + per);
    s.plane = ((size_t)s.b * C + s.c) * (size_t)HW;
    return s;
}

__device__ __forceinline__ float block_sum(float v, float *lds)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    __syncthreads();
    return v;
}

// iterate the block's span: body(i, n) for a full vector at element offset i (n = VEC) or a scalar tail (n = 1)
#define MDX_SPAN_LOOP(s, vec_ok, BODY_VEC, BODY_ONE)                                        \
    if (vec_ok) {                                                                           \
        _Pragma("unroll 4")                                                                 \
        for (int i = s.lo + VEC * threadIdx.x; i + VEC <= s.hi; i += VEC * NB) { BODY_VEC } \
        const int tail = s.lo + (s.hi - s.lo) / VEC * VEC;                                  \
        for (int i = tail + threadIdx.x; i < s.hi; i += NB) { BODY_ONE }                    \
    } else {                                                                                \
        for (int i = s.lo + threadIdx.x; i < s.hi; i += NB) { BODY_ONE }                    \
    }

// ---- forward, pass 1: per-span (sum, sum of squares) ----
template <typename T>
__global__ __launch_bounds__(NB) void bn_fwd_stats_kernel(const T *__restrict__ x, int C, int HW, int K,
                                                          float *__restrict__ part)
{
    __shared__ float lds[NB / 64];
    const SpanId s = span_id(C, HW, K);
    const T *p = x + s.plane;
    const bool vec_ok = (HW % VEC) == 0;
    float a = 0.f, q = 0.f;
    MDX_SPAN_LOOP(s, vec_ok,
                  const Vec4<T> v = *reinterpret_cast<const Vec4<T> *>(p + i);
                  _Pragma("unroll") for (int j = 0; j < VEC; ++j) { const float f = ld(v.v, j); a += f; q = __builtin_fmaf(f, f, q); },
                  const float f = ld(p, i); a += f; q = __builtin_fmaf(f, f, q);)
    a = block_sum(a, lds);
    q = block_sum(q, lds);
    if (threadIdx.x == 0) {
        part[((size_t)s.c * gridDim.y + blockIdx.y) * 2 + 0] = a;
        part[((size_t)s.c * gridDim.y + blockIdx.y) * 2 + 1] = q;
    }
}

// channel statistics from the partials (every block of the channel repeats this tiny sum in the same order)
__device__ __forceinline__ void channel_stats(const float *__restrict__ part, int c, int nspan, double M, float eps,
                                              float &mean, float &invstd, double &var_out)
{
    double a = 0.0, q = 0.0;
    for (int i = 0; i < nspan; ++i) {
        a += (double)part[((size_t)c * nspan + i) * 2 + 0];
        q += (double)part[((size_t)c * nspan + i) * 2 + 1];
    }
    const double m = a / M;
    double var = q / M - m * m;
    var = var > 0.0 ? var : 0.0;
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    var_out = var;
}

// ---- forward, pass 2: y = act(x * scale + shift [+ res]); block (c, 0) also updates the running statistics ----
template <typename T>
__global__ __launch_bounds__(NB) void bn_fwd_apply_kernel(const T *__restrict__ x, const T *__restrict__ res,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const float *__restrict__ part, int C, int HW, int K, int B,
                                                          float eps, float momentum, int relu, T *__restrict__ y,
                                                          float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                          float *__restrict__ run_mean, float *__restrict__ run_var)
{
    const SpanId s = span_id(C, HW, K);
    const double M = (double)B * HW;
    float mean, invstd;
    double var;
    channel_stats(part, s.c, gridDim.y, M, eps, mean, invstd, var);
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        save_mean[s.c] = mean;
        save_invstd[s.c] = invstd;
        if (run_mean) {
            const double unbiased = M > 1.0 ? var * (M / (M - 1.0)) : var;
            run_mean[s.c] = (1.0f - momentum) * run_mean[s.c] + momentum * mean;
            run_var[s.c] = (1.0f - momentum) * run_var[s.c] + momentum * (float)unbiased;
        }
    }
    const float scale = gamma[s.c] * invstd, shift = beta[s.c] - mean * scale;
    const T *p = x + s.plane;
    const T *r = res ? res + s.plane : nullptr;
    T *o = y + s.plane;
    const bool vec_ok = (HW % VEC) == 0;
    MDX_SPAN_LOOP(s, vec_ok,
                  const Vec4<T> v = *reinterpret_cast<const Vec4<T> *>(p + i);
                  Vec4<T> rv = {}; if (r) rv = *reinterpret_cast<const Vec4<T> *>(r + i);
                  Vec4<T> w;
                  _Pragma("unroll") for (int j = 0; j < VEC; ++j) {
                      float f = __builtin_fmaf(ld(v.v, j), scale, shift);
                      if (r) f += ld(rv.v, j);
                      st(w.v, j, (relu && f < 0.f) ? 0.f : f);
                  }
                  *reinterpret_cast<Vec4<T> *>(o + i) = w;,
                  float f = __builtin_fmaf(ld(p, i), scale, shift); if (r) f += ld(r, i);
                  st(o, i, (relu && f < 0.f) ? 0.f : f);)
}

// ---- backward, pass 1: per-span (sum dz, sum dz * xhat), dz = dy * (y > 0) ----
template <typename T>
__global__ __launch_bounds__(NB) void bn_bwd_stats_kernel(const T *__restrict__ dy, const T *__restrict__ y,
                                                          const T *__restrict__ x, const float *__restrict__ save_mean,
                                                          const float *__restrict__ save_invstd, int C, int HW, int K,
                                                          int relu, float *__restrict__ part)
{
    __shared__ float lds[NB / 64];
    const SpanId s = span_id(C, HW, K);
    const float mean = save_mean[s.c], invstd = save_invstd[s.c];
    const T *pd = dy + s.plane, *py = y + s.plane, *px = x + s.plane;
    const bool vec_ok = (HW % VEC) == 0;
    float a = 0.f, q = 0.f;
    MDX_SPAN_LOOP(s, vec_ok,
                  const Vec4<T> vd = *reinterpret_cast<const Vec4<T> *>(pd + i);
                  const Vec4<T> vy = *reinterpret_cast<const Vec4<T> *>(py + i);
                  const Vec4<T> vx = *reinterpret_cast<const Vec4<T> *>(px + i);
                  _Pragma("unroll") for (int j = 0; j < VEC; ++j) {
                      const float dz = (relu && !(ld(vy.v, j) > 0.f)) ? 0.f : ld(vd.v, j);
                      a += dz; q = __builtin_fmaf(dz, (ld(vx.v, j) - mean) * invstd, q);
                  },
                  const float dz = (relu && !(ld(py, i) > 0.f)) ? 0.f : ld(pd, i);
                  a += dz; q = __builtin_fmaf(dz, (ld(px, i) - mean) * invstd, q);)
    a = block_sum(a, lds);
    q = block_sum(q, lds);
    if (threadIdx.x == 0) {
        part[((size_t)s.c * gridDim.y + blockIdx.y) * 2 + 0] = a;
        part[((size_t)s.c * gridDim.y + blockIdx.y) * 2 + 1] = q;
    }
}

// ---- backward, pass 2: dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)); d_res = dz ----
template <typename T>
__global__ __launch_bounds__(NB) void bn_bwd_apply_kernel(const T *__restrict__ dy, const T *__restrict__ y,
                                                          const T *__restrict__ x, const float *__restrict__ gamma,
                                                          const float *__restrict__ save_mean,
                                                          const float *__restrict__ save_invstd,
                                                          const float *__restrict__ part, int C, int HW, int K, int B,
                                                          int relu, int accum, T *__restrict__ dx,
                                                          T *__restrict__ dres, float *__restrict__ dgamma,
                                                          float *__restrict__ dbeta)
{
    const SpanId s = span_id(C, HW, K);
    double a = 0.0, q = 0.0;
    for (int i = 0; i < (int)gridDim.y; ++i) {
        a += (double)part[((size_t)s.c * gridDim.y + i) * 2 + 0];
        q += (double)part[((size_t)s.c * gridDim.y + i) * 2 + 1];
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        dbeta[s.c] = (accum ? dbeta[s.c] : 0.f) + (float)a;
        dgamma[s.c] = (accum ? dgamma[s.c] : 0.f) + (float)q;
    }
    const double M = (double)B * HW;
    const float mean = save_mean[s.c], invstd = save_invstd[s.c];
    const float k0 = gamma[s.c] * invstd, mdz = (float)(a / M), mdzx = (float)(q / M);
    const T *pd = dy + s.plane, *py = y + s.plane, *px = x + s.plane;
    T *ox = dx + s.plane;
    T *orr = dres ? dres + s.plane : nullptr;
    const bool vec_ok = (HW % VEC) == 0;
    MDX_SPAN_LOOP(s, vec_ok,
                  const Vec4<T> vd = *reinterpret_cast<const Vec4<T> *>(pd + i);
                  const Vec4<T> vy = *reinterpret_cast<const Vec4<T> *>(py + i);
                  const Vec4<T> vx = *reinterpret_cast<const Vec4<T> *>(px + i);
                  Vec4<T> wx; Vec4<T> wr;
                  _Pragma("unroll") for (int j = 0; j < VEC; ++j) {
                      const float dz = (relu && !(ld(vy.v, j) > 0.f)) ? 0.f : ld(vd.v, j);
                      const float xh = (ld(vx.v, j) - mean) * invstd;
                      st(wx.v, j, k0 * (dz - mdz - xh * mdzx));
                      st(wr.v, j, dz);
                  }
                  *reinterpret_cast<Vec4<T> *>(ox + i) = wx;
                  if (orr) *reinterpret_cast<Vec4<T> *>(orr + i) = wr;,
                  const float dz = (relu && !(ld(py, i) > 0.f)) ? 0.f : ld(pd, i);
                  const float xh = (ld(px, i) - mean) * invstd;
                  st(ox, i, k0 * (dz - mdz - xh * mdzx));
                  if (orr) st(orr, i, dz);)
}

// -----------------------------------------------------------------------------------------------------------
// Small maps (B*H*W <= SMALL_MAX elements per channel, H*W % 4 == 0): ONE launch each way.  One 1024-thread block
// owns a channel, keeps its <= 24 elements per thread in registers between the reduction and the apply step, so the
// activations cross HBM once (forward: read x [, identity], write y; backward: read dy, y, x, write dx [, d_identity]).
// -----------------------------------------------------------------------------------------------------------
constexpr int SB = 1024;                     // threads per block (largest form)
constexpr int SB_TINY = 256;                 // ... and for channels of at most 2 vectors per thread of this size (2048 elements).  Round 4:
                                             // the 512-channel maps of a 192x640 batch (1440 elements per channel) fill 360 of a 1024-thread
                                             // block's lanes and its reductions run over sixteen waves: forward 11.9 -> 7.2 us, backward
                                             // 11.1 -> 7.3 us with 256 threads.  (The 256-channel maps, 5760 elements, are SLOWER that
                                             // way -- 8.0 -> 9.0 / 9.8 -> 11.4 us with six vectors per thread -- and keep the large block.)
constexpr int SV = 6;                        // 4-element vectors per thread
constexpr int SMALL_MAX = SB * SV * VEC;     // 24576 elements per channel

template <int NTH> __device__ __forceinline__ float block_sum_small(float v, float *lds)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NTH / 64; ++k) t += lds[k];
    __syncthreads();
    return t;
}

// element offset (inside the [B,C,H,W] tensor) of vector j of this thread, or -1 beyond the channel's data
template <int NTH> __device__ __forceinline__ long long small_offset(int j, int c, int C, int HW, int M)
{
    const int e = (j * NTH + threadIdx.x) * VEC;      // linear index over (b, hw) of channel c
    if (e >= M) return -1;
    const int b = e / HW, off = e - b * HW;
    return ((long long)b * C + c) * HW + off;
}

template <typename T, int NTH>
__global__ __launch_bounds__(NTH) void bn_small_fwd_kernel(const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          int C, int HW, int B, float eps, float momentum, int relu,
                                                          T *__restrict__ y_all, float *__restrict__ save_mean_all,
                                                          float *__restrict__ save_invstd_all, float *__restrict__ run_mean,
                                                          float *__restrict__ run_var, const T *__restrict__ x_all,
                                                          const T *__restrict__ res_all, int G)
{
    __shared__ float lds[NTH / 64];
    const int c = blockIdx.x, M = B * HW;
    // G consecutive sub-batches of B images, one after the other: own statistics each, running statistics updated in
    // order (what G calls of the module do) -- from one launch
#pragma unroll 1
    for (int grp = 0; grp < G; ++grp) {
    const size_t goff = (size_t)grp * B * C * HW;
    const T *x = x_all + goff;
    const T *res = res_all ? res_all + goff : nullptr;
    T *y = y_all + goff;
    float *save_mean = save_mean_all + (size_t)grp * C, *save_invstd = save_invstd_all + (size_t)grp * C;
    long long o[SV];
    Vec4<T> v[SV];
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < SV; ++j) {
        o[j] = small_offset<NTH>(j, c, C, HW, M);
        if (o[j] >= 0) v[j] = *reinterpret_cast<const Vec4<T> *>(x + o[j]);
    }
#pragma unroll
    for (int j = 0; j < SV; ++j)
        if (o[j] >= 0)
#pragma unroll
            for (int k = 0; k < VEC; ++k) a += ld(v[j].v, k);
    const float mean = block_sum_small<NTH>(a, lds) / (float)M;
    float q = 0.f;                                   // two-pass variance: the data is in registers
#pragma unroll
    for (int j = 0; j < SV; ++j)
        if (o[j] >= 0)
#pragma unroll
            for (int k = 0; k < VEC; ++k) { const float d = ld(v[j].v, k) - mean; q = __builtin_fmaf(d, d, q); }
    const float var = block_sum_small<NTH>(q, lds) / (float)M;
    const float invstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (run_mean) {
            const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unbiased;
        }
    }
    const float scale = gamma[c] * invstd, shift = beta[c] - mean * scale;
#pragma unroll
    for (int j = 0; j < SV; ++j) {
        if (o[j] < 0) continue;
        Vec4<T> rv = {};
        if (res) rv = *reinterpret_cast<const Vec4<T> *>(res + o[j]);
        Vec4<T> w;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float f = __builtin_fmaf(ld(v[j].v, k), scale, shift);
            if (res) f += ld(rv.v, k);
            st(w.v, k, (relu && f < 0.f) ? 0.f : f);
        }
        *reinterpret_cast<Vec4<T> *>(y + o[j]) = w;
    }
    }   // grp
}

template <typename T, int NTH>
__global__ __launch_bounds__(NTH) void bn_small_bwd_kernel(const T *__restrict__ dy_all, const T *__restrict__ y_all,
                                                          const T *__restrict__ x_all, const float *__restrict__ gamma,
                                                          const float *__restrict__ save_mean,
                                                          const float *__restrict__ save_invstd, int C, int HW, int B,
                                                          int relu, int accum, T *__restrict__ dx_all,
                                                          T *__restrict__ dres_all, float *__restrict__ dgamma,
                                                          float *__restrict__ dbeta, int G)
{
    __shared__ float lds[NTH / 64];
    const int c = blockIdx.x, M = B * HW;
    float tot_a = 0.f, tot_q = 0.f;
#pragma unroll 1
    for (int grp = 0; grp < G; ++grp) {
    const size_t goff = (size_t)grp * B * C * HW;
    const T *dy = dy_all + goff, *y = y_all + goff, *x = x_all + goff;
    T *dx = dx_all + goff;
    T *dres = dres_all ? dres_all + goff : nullptr;
    const float mean = save_mean[(size_t)grp * C + c], invstd = save_invstd[(size_t)grp * C + c];
    long long o[SV];
    float dz[SV][VEC], xh[SV][VEC];
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < SV; ++j) {
        o[j] = small_offset<NTH>(j, c, C, HW, M);
        if (o[j] < 0) continue;
        const Vec4<T> vd = *reinterpret_cast<const Vec4<T> *>(dy + o[j]);
        const Vec4<T> vy = *reinterpret_cast<const Vec4<T> *>(y + o[j]);
        const Vec4<T> vx = *reinterpret_cast<const Vec4<T> *>(x + o[j]);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            dz[j][k] = (relu && !(ld(vy.v, k) > 0.f)) ? 0.f : ld(vd.v, k);
            xh[j][k] = (ld(vx.v, k) - mean) * invstd;
            a += dz[j][k];
            q = __builtin_fmaf(dz[j][k], xh[j][k], q);
        }
    }
    a = block_sum_small<NTH>(a, lds);
    q = block_sum_small<NTH>(q, lds);
    tot_a += a;
    tot_q += q;
    const float k0 = gamma[c] * invstd, mdz = a / (float)M, mdzx = q / (float)M;
#pragma unroll
    for (int j = 0; j < SV; ++j) {
        if (o[j] < 0) continue;
        Vec4<T> wx, wr;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            st(wx.v, k, k0 * (dz[j][k] - mdz - xh[j][k] * mdzx));
            st(wr.v, k, dz[j][k]);
        }
        *reinterpret_cast<Vec4<T> *>(dx + o[j]) = wx;
        if (dres) *reinterpret_cast<Vec4<T> *>(dres + o[j]) = wr;
    }
    }   // grp
    if (threadIdx.x == 0) {
        dbeta[c] = (accum ? dbeta[c] : 0.f) + tot_a;
        dgamma[c] = (accum ? dgamma[c] : 0.f) + tot_q;
    }
}

// the small-map kernels in their 256- or 1024-thread form by the channel's element count
#define MDX_BN_SMALL(kind, T, ...)                                                                                   \
    do {                                                                                                             \
        if ((long long)B * HW <= (long long)SB_TINY * 2 * VEC)                                                       \
            hipLaunchKernelGGL((bn_small_##kind##_kernel<T, SB_TINY>), dim3(C), dim3(SB_TINY), 0, st, __VA_ARGS__);  \
        else                                                                                                         \
            hipLaunchKernelGGL((bn_small_##kind##_kernel<T, SB>), dim3(C), dim3(SB), 0, st, __VA_ARGS__);            \
    } while (0)
static inline bool small_map(int B, int HW) { return (HW % VEC) == 0 && (long long)B * HW <= SMALL_MAX; }

static inline int spans_per_plane(int HW) { return (HW + SPAN - 1) / SPAN; }

}  // namespace mdx

using namespace mdx;

MDX_EXPORT size_t mdx_bn_workspace_bytes(int B, int C, int H, int W)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)C * B * spans_per_plane(H * W) * 2 * sizeof(float);
}

// dtype: 0 float32, 1 bfloat16 (x, res, y); gamma/beta/statistics float32
MDX_EXPORT int mdx_bn_act_fwd(const void *x, const void *res, const float *gamma, const float *beta, float *run_mean,
                              float *run_var, void *y, float *save_mean, float *save_invstd, int B, int C, int H, int W,
                              int groups, float eps, float momentum, int relu, int dtype, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace) return MDX_ERR_NULL_POINTER;
    if ((run_mean == nullptr) != (run_var == nullptr)) return MDX_ERR_NULL_POINTER;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || (long long)H * W > (1ll << 30)) return MDX_ERR_BAD_SHAPE;
    if (dtype != 0 && dtype != 1) return MDX_ERR_BAD_SHAPE;
    const int HW = H * W, K = spans_per_plane(HW);
    if ((long long)B * K > 65535) return MDX_ERR_BAD_SHAPE;
    if (workspace_bytes < mdx_bn_workspace_bytes(B, C, H, W)) return MDX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (small_map(B, HW)) {
        if (dtype == 0)
            MDX_BN_SMALL(fwd, float, gamma, beta, C, HW, B, eps, momentum,
                               relu, (float *)y, save_mean, save_invstd, run_mean, run_var, (const float *)x,
                               (const float *)res, groups);
        else
            MDX_BN_SMALL(fwd, bf16n, gamma, beta, C, HW, B, eps, momentum,
                               relu, (bf16n *)y, save_mean, save_invstd, run_mean, run_var, (const bf16n *)x,
                               (const bf16n *)res, groups);
        return check_launch();
    }
    const dim3 grid(C, B * K), block(NB);
    float *part = (float *)workspace;
    const size_t gelems = (size_t)B * C * HW, esize = dtype == 0 ? 4 : 2;
    for (int g = 0; g < groups; ++g) {   // sub-batches in order (running statistics): stream order does that
        const char *xg = (const char *)x + g * gelems * esize;
        const char *rg = res ? (const char *)res + g * gelems * esize : nullptr;
        char *yg = (char *)y + g * gelems * esize;
        float *sm = save_mean + (size_t)g * C, *si = save_invstd + (size_t)g * C;
        if (dtype == 0) {
            hipLaunchKernelGGL((bn_fwd_stats_kernel<float>), grid, block, 0, st, (const float *)xg, C, HW, K, part);
            hipLaunchKernelGGL((bn_fwd_apply_kernel<float>), grid, block, 0, st, (const float *)xg, (const float *)rg, gamma,
                               beta, part, C, HW, K, B, eps, momentum, relu, (float *)yg, sm, si, run_mean, run_var);
        } else {
            hipLaunchKernelGGL((bn_fwd_stats_kernel<bf16n>), grid, block, 0, st, (const bf16n *)xg, C, HW, K, part);
            hipLaunchKernelGGL((bn_fwd_apply_kernel<bf16n>), grid, block, 0, st, (const bf16n *)xg, (const bf16n *)rg, gamma,
                               beta, part, C, HW, K, B, eps, momentum, relu, (bf16n *)yg, sm, si, run_mean, run_var);
        }
    }
    return check_launch();
}

MDX_EXPORT int mdx_bn_act_bwd(const void *dy, const void *y, const void *x, const float *gamma, const float *save_mean,
                              const float *save_invstd, void *dx, void *dres, float *dgamma, float *dbeta, int B, int C,
                              int H, int W, int groups, int relu, int dtype, void *workspace, size_t workspace_bytes,
                              void *stream)
{
    if (!dy || !y || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace)
        return MDX_ERR_NULL_POINTER;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || (long long)H * W > (1ll << 30)) return MDX_ERR_BAD_SHAPE;
    if (dtype != 0 && dtype != 1) return MDX_ERR_BAD_SHAPE;
    const int HW = H * W, K = spans_per_plane(HW);
    if ((long long)B * K > 65535) return MDX_ERR_BAD_SHAPE;
    if (workspace_bytes < mdx_bn_workspace_bytes(B, C, H, W)) return MDX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (small_map(B, HW)) {
        if (dtype == 0)
            MDX_BN_SMALL(bwd, float, (const float *)dy, (const float *)y,
                               (const float *)x, gamma, save_mean, save_invstd, C, HW, B, relu, 0, (float *)dx,
                               (float *)dres, dgamma, dbeta, groups);
        else
            MDX_BN_SMALL(bwd, bf16n, (const bf16n *)dy, (const bf16n *)y,
                               (const bf16n *)x, gamma, save_mean, save_invstd, C, HW, B, relu, 0, (bf16n *)dx,
                               (bf16n *)dres, dgamma, dbeta, groups);
        return check_launch();
    }
    const dim3 grid(C, B * K), block(NB);
    float *part = (float *)workspace;
    const size_t gelems = (size_t)B * C * HW, esize = dtype == 0 ? 4 : 2;
    for (int g = 0; g < groups; ++g) {
        const char *dyg = (const char *)dy + g * gelems * esize, *yg = (const char *)y + g * gelems * esize;
        const char *xg = (const char *)x + g * gelems * esize;
        char *dxg = (char *)dx + g * gelems * esize;
        char *drg = dres ? (char *)dres + g * gelems * esize : nullptr;
        const float *sm = save_mean + (size_t)g * C, *si = save_invstd + (size_t)g * C;
        if (dtype == 0) {
            hipLaunchKernelGGL((bn_bwd_stats_kernel<float>), grid, block, 0, st, (const float *)dyg, (const float *)yg,
                               (const float *)xg, sm, si, C, HW, K, relu, part);
            hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), grid, block, 0, st, (const float *)dyg, (const float *)yg,
                               (const float *)xg, gamma, sm, si, part, C, HW, K, B, relu, g > 0, (float *)dxg,
                               (float *)drg, dgamma, dbeta);
        } else {
            hipLaunchKernelGGL((bn_bwd_stats_kernel<bf16n>), grid, block, 0, st, (const bf16n *)dyg, (const bf16n *)yg,
                               (const bf16n *)xg, sm, si, C, HW, K, relu, part);
            hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16n>), grid, block, 0, st, (const bf16n *)dyg, (const bf16n *)yg,
                               (const bf16n *)xg, gamma, sm, si, part, C, HW, K, B, relu, g > 0, (bf16n *)dxg,
                               (bf16n *)drg, dgamma, dbeta);
        }
    }
    return check_launch();
}
