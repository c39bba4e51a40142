// norm_nhwc.hip -- training-mode BatchNorm2d + residual add + ReLU on CHANNELS-LAST maps, gfx950.
//
// Same operator as norm.hip (model_layer/depth_encoder.py: y = relu(bn(x) [+ identity]), reference
// model_layer/depth_encoder.py:27,95 through torchvision's blocks), for tensors whose memory is [B][H][W][C]: what
// MIOpen's implicit-GEMM convolutions (`igemm_*_nhwc`) read and write without a layout transpose on either side.
// The map is a matrix [M = B*H*W rows][C]; the per-channel statistics are COLUMN sums:
//     forward    stats   : every block sums its rows -> one (sum, sum of squares) partial per block and channel
//                finalize: partials -> mean, 1/std (float64), running statistics            (one tiny launch)
//                apply   : y = act(x * scale + shift [+ res])
//     backward   stats   : (sum dz, sum dz * xhat) partials, dz = (dy [+ dy2]) * (y > 0)
//                finalize: -> d(beta), d(gamma), the two means the apply pass needs
//                apply   : dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)); d_res = dz
// A thread owns one 16-byte channel vector and walks rows (nhwc_common.hpp): four independent 16-byte loads in flight
// per input, sums in float32 inside a block (sequential per thread, fixed order across the block), float64 across
// blocks -- no atomics, the same bits every run.  dy2: a second upstream gradient (a block's output feeds the next
// block's first convolution AND its identity path); adding it here saves autograd's accumulation pass over the map.
// Statistics are torch.nn.functional.batch_norm's: biased variance to normalise, unbiased for running_var.
#include "nhwc_common.hpp"

namespace mdx {
namespace nhwc {

// Blocks along the rows.  The statistics passes want FEW blocks: every block leaves one partial per channel and the finalize
// pass that adds them is a chain of fabric round trips (tools/ab_netbench.sh over the maps of BASELINE configs[1] and [3], sum of
// the forward / backward times: 1024 blocks 447 / 789 us, 512: 377 / 702, 256: 352 / 663, 128: 350 / 694 in float32; the
// bfloat16 backward, three 2-byte streams per row, is the one that still gains from more blocks: 256: 1463 us, 384: 1407, 512:
// 1424).  The apply passes stream and take many.
constexpr int STATS_FWD_BLOCKS = 256;
constexpr int STATS_BWD_BLOCKS_F32 = 256, STATS_BWD_BLOCKS_BF16 = 384;
constexpr int STATS_MAX_BLOCKS = 384;      // the largest of the three: sizes the workspace and the finalize pass's unrolled loads
constexpr int APPLY_MAX_BLOCKS = 4096;
constexpr int STATS_ITERS = 8, APPLY_ITERS = 4;          // row sweeps a block is given at least (small maps: fewer blocks)
static inline int stats_bwd_blocks(int dtype) { return dtype == 0 ? STATS_BWD_BLOCKS_F32 : STATS_BWD_BLOCKS_BF16; }

// ---- forward, pass 1 ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NB) void bn_nhwc_fwd_stats_kernel(const T *__restrict__ x, int M, int C, int CVB, int PL, int RB,
                                                               float *__restrict__ part)
{
    constexpr int N = VecN<T>::N;
    __shared__ float lds[2 * NB * N];
    const Pos p = position<N>(M, C, CVB, PL, RB);
    float a[N], q[N];
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] = q[j] = 0.f;
    if (p.active) {
        const T *px = x + (size_t)p.g * M * C + (size_t)p.cv * N;
        auto acc = [&](const Vec<T, N> &v) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float f = to_float(v.v[j]);
                a[j] += f;
                q[j] = __builtin_fmaf(f, f, q[j]);
            }
        };
        int r = p.r0 + p.pl;
        for (; r + 3 * PL < p.r1; r += 4 * PL) {
            const Vec<T, N> v0 = load_vec<T, N>(px + (size_t)r * C), v1 = load_vec<T, N>(px + (size_t)(r + PL) * C);
            const Vec<T, N> v2 = load_vec<T, N>(px + (size_t)(r + 2 * PL) * C), v3 = load_vec<T, N>(px + (size_t)(r + 3 * PL) * C);
            acc(v0); acc(v1); acc(v2); acc(v3);
        }
        for (; r < p.r1; r += PL) acc(load_vec<T, N>(px + (size_t)r * C));
    }
    block_partials<N>(a, q, lds, CVB, PL, C, part + ((size_t)p.g * gridDim.x + blockIdx.x) * 2 * C);
}

// ---- partials -> per-channel totals: a block = 16 channels x 64 sub-sums (1024 threads), float64 -------------------
// Every thread has at most FU = STATS_MAX_BLOCKS / FS partials to add: all 2 FU loads are issued before the first add -- the
// partials were written a moment ago by blocks on all eight XCDs, each load is a trip through the fabric, and a loop with
// one load per iteration made this pass (4 blocks for a 64-channel map) the longest of the three.
constexpr int FC = 16, FS = 64, FU = STATS_MAX_BLOCKS / FS;
__device__ __forceinline__ void channel_totals(const float *__restrict__ part_g, int nblk, int C, int c, double (*lds)[FS / 4][FC],
                                               double &ta, double &tq)
{
    const int cl = threadIdx.x % FC, sl = threadIdx.x / FC;
    float va[FU], vq[FU];
#pragma unroll
    for (int i = 0; i < FU; ++i) {
        const int b = sl + i * FS;
        const bool ok = c < C && b < nblk;
        va[i] = ok ? part_g[(size_t)b * 2 * C + c] : 0.f;
        vq[i] = ok ? part_g[(size_t)b * 2 * C + C + c] : 0.f;
    }
    double a = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < FU; ++i) {
        a += (double)va[i];
        q += (double)vq[i];
    }
    // the 4 sub-sums a wave holds per channel (lanes cl, cl+16, cl+32, cl+48), then the 16 waves through LDS: fixed order
    a += __shfl_down(a, 32, 64);
    q += __shfl_down(q, 32, 64);
    a += __shfl_down(a, 16, 64);
    q += __shfl_down(q, 16, 64);
    if ((threadIdx.x & 63) < FC) {
        lds[0][threadIdx.x >> 6][cl] = a;
        lds[1][threadIdx.x >> 6][cl] = q;
    }
    __syncthreads();
    ta = tq = 0.0;
    if (threadIdx.x < FC)
#pragma unroll
        for (int s = 0; s < FS / 4; ++s) {
            ta += lds[0][s][cl];
            tq += lds[1][s][cl];
        }
    __syncthreads();
}

// forward: groups in order (each updates the running statistics like one call of the module)
__global__ __launch_bounds__(FC *FS) void bn_nhwc_fwd_finalize_kernel(const float *__restrict__ part, int nblk, int C, int G, double M,
                                                                       float eps, float momentum, float *__restrict__ save_mean,
                                                                       float *__restrict__ save_invstd, float *__restrict__ run_mean,
                                                                       float *__restrict__ run_var)
{
    __shared__ double lds[2][FS / 4][FC];
    const int c = blockIdx.x * FC + threadIdx.x % FC;
    const bool writer = threadIdx.x < FC && c < C;
    float rm = 0.f, rv = 0.f;
    if (writer && run_mean) { rm = run_mean[c]; rv = run_var[c]; }
    for (int g = 0; g < G; ++g) {
        double a, q;
        channel_totals(part + (size_t)g * nblk * 2 * C, nblk, C, c, lds, a, q);
        if (writer) {
            const double m = a / M;
            double var = q / M - m * m;
            var = var > 0.0 ? var : 0.0;
            const float mean = (float)m;
            save_mean[(size_t)g * C + c] = mean;
            save_invstd[(size_t)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
            const double unbiased = M > 1.0 ? var * (M / (M - 1.0)) : var;
            rm = (1.0f - momentum) * rm + momentum * mean;
            rv = (1.0f - momentum) * rv + momentum * (float)unbiased;
        }
    }
    if (writer && run_mean) { run_mean[c] = rm; run_var[c] = rv; }
}

// backward: totals[g][2][C] = the MEANS over the M rows of (dz, dz * xhat) per group -- what the apply pass subtracts: formed here,
// once per channel, not by every thread of the apply pass (two float64 divisions per channel of its vector in front of four rows of
// work); d(beta), d(gamma) = the sums over the groups
__global__ __launch_bounds__(FC *FS) void bn_nhwc_bwd_finalize_kernel(const float *__restrict__ part, int nblk, int C, int G, double M,
                                                                       float *__restrict__ totals, float *__restrict__ dgamma,
                                                                       float *__restrict__ dbeta)
{
    __shared__ double lds[2][FS / 4][FC];
    const int c = blockIdx.x * FC + threadIdx.x % FC;
    const bool writer = threadIdx.x < FC && c < C;
    float sa = 0.f, sq = 0.f;
    for (int g = 0; g < G; ++g) {
        double a, q;
        channel_totals(part + (size_t)g * nblk * 2 * C, nblk, C, c, lds, a, q);
        if (writer) {
            totals[((size_t)g * 2 + 0) * C + c] = (float)((double)(float)a / M);
            totals[((size_t)g * 2 + 1) * C + c] = (float)((double)(float)q / M);
            sa += (float)a;
            sq += (float)q;
        }
    }
    if (writer) { dbeta[c] = sa; dgamma[c] = sq; }
}

// ---- forward, pass 2 ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NB) void bn_nhwc_fwd_apply_kernel(const T *__restrict__ x, const T *__restrict__ res,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta,
                                                               const float *__restrict__ save_mean,
                                                               const float *__restrict__ save_invstd, int M, int C, int CVB, int PL,
                                                               int RB, int relu, T *__restrict__ y)
{
    constexpr int N = VecN<T>::N;
    const Pos p = position<N>(M, C, CVB, PL, RB);
    if (!p.active) return;
    float scale[N], shift[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int c = p.cv * N + j;
        scale[j] = gamma[c] * save_invstd[(size_t)p.g * C + c];
        shift[j] = beta[c] - save_mean[(size_t)p.g * C + c] * scale[j];
    }
    const size_t base = (size_t)p.g * M * C + (size_t)p.cv * N;
    const T *px = x + base;
    const T *pr = res ? res + base : nullptr;
    T *py = y + base;
    auto apply = [&](const Vec<T, N> &v, const Vec<T, N> &rv, size_t off) {
        Vec<T, N> w;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float f = __builtin_fmaf(to_float(v.v[j]), scale[j], shift[j]);
            if (pr) f += to_float(rv.v[j]);
            w.v[j] = from_float<T>((relu && f < 0.f) ? 0.f : f);
        }
        store_vec<T, N>(py + off, w);
    };
    int r = p.r0 + p.pl;
    for (; r + 3 * PL < p.r1; r += 4 * PL) {
        const size_t o0 = (size_t)r * C, o1 = (size_t)(r + PL) * C, o2 = (size_t)(r + 2 * PL) * C, o3 = (size_t)(r + 3 * PL) * C;
        const Vec<T, N> v0 = load_vec<T, N>(px + o0), v1 = load_vec<T, N>(px + o1), v2 = load_vec<T, N>(px + o2), v3 = load_vec<T, N>(px + o3);
        Vec<T, N> r0 = {}, r1 = {}, r2 = {}, r3 = {};
        if (pr) { r0 = load_vec<T, N>(pr + o0); r1 = load_vec<T, N>(pr + o1); r2 = load_vec<T, N>(pr + o2); r3 = load_vec<T, N>(pr + o3); }
        apply(v0, r0, o0); apply(v1, r1, o1); apply(v2, r2, o2); apply(v3, r3, o3);
    }
    for (; r < p.r1; r += PL) {
        const size_t o = (size_t)r * C;
        Vec<T, N> rv = {};
        if (pr) rv = load_vec<T, N>(pr + o);
        apply(load_vec<T, N>(px + o), rv, o);
    }
}

// ---- backward, pass 1 ---------------------------------------------------------------------------------------------
// MASKX: the ReLU mask is re-derived from x (y == nullptr) -- a separate instantiation, so that the form that reads y carries neither
// its registers nor its branches (as ONE kernel the bfloat16 apply pass went from 31 to 56 us per call)
template <typename T, bool MASKX>
__global__ __launch_bounds__(NB) void bn_nhwc_bwd_stats_kernel(const T *__restrict__ dy, const T *__restrict__ dy2,
                                                               const T *__restrict__ y, const T *__restrict__ x,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta,
                                                               const float *__restrict__ save_mean,
                                                               const float *__restrict__ save_invstd, int M, int C, int CVB, int PL,
                                                               int RB, int relu, float *__restrict__ part)
{
    constexpr int N = VecN<T>::N;
    __shared__ float lds[2 * NB * N];
    const Pos p = position<N>(M, C, CVB, PL, RB);
    float a[N], q[N];
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] = q[j] = 0.f;
    if (p.active) {
        float mean[N], invstd[N], scale[N], shift[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            mean[j] = save_mean[(size_t)p.g * C + p.cv * N + j];
            invstd[j] = save_invstd[(size_t)p.g * C + p.cv * N + j];
            scale[j] = shift[j] = 0.f;
            if (MASKX) {               // the forward pass's own expressions (bn_nhwc_fwd_apply_kernel): y is re-derived, not read
                scale[j] = gamma[p.cv * N + j] * invstd[j];
                shift[j] = beta[p.cv * N + j] - mean[j] * scale[j];
            }
        }
        const size_t base = (size_t)p.g * M * C + (size_t)p.cv * N;
        const T *pd = dy + base, *pd2 = dy2 ? dy2 + base : nullptr, *py = MASKX ? nullptr : y + base, *px = x + base;
        auto acc = [&](const Vec<T, N> &vd, const Vec<T, N> &vd2, const Vec<T, N> &vy, const Vec<T, N> &vx) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float d = to_float(vd.v[j]);
                if (pd2) d += to_float(vd2.v[j]);
                const float yv = !MASKX ? to_float(vy.v[j]) : to_float(from_float<T>(__builtin_fmaf(to_float(vx.v[j]), scale[j], shift[j])));
                const float dz = (relu && !(yv > 0.f)) ? 0.f : d;
                a[j] += dz;
                q[j] = __builtin_fmaf(dz, (to_float(vx.v[j]) - mean[j]) * invstd[j], q[j]);
            }
        };
        int r = p.r0 + p.pl;
        for (; r + PL < p.r1; r += 2 * PL) {
            const size_t o0 = (size_t)r * C, o1 = (size_t)(r + PL) * C;
            const Vec<T, N> d0 = load_vec<T, N>(pd + o0), d1 = load_vec<T, N>(pd + o1);
            Vec<T, N> y0 = {}, y1 = {};
            if (!MASKX) { y0 = load_vec<T, N>(py + o0); y1 = load_vec<T, N>(py + o1); }
            const Vec<T, N> x0 = load_vec<T, N>(px + o0), x1 = load_vec<T, N>(px + o1);
            Vec<T, N> e0 = {}, e1 = {};
            if (pd2) { e0 = load_vec<T, N>(pd2 + o0); e1 = load_vec<T, N>(pd2 + o1); }
            acc(d0, e0, y0, x0);
            acc(d1, e1, y1, x1);
        }
        for (; r < p.r1; r += PL) {
            const size_t o = (size_t)r * C;
            Vec<T, N> e = {}, yv = {};
            if (pd2) e = load_vec<T, N>(pd2 + o);
            if (!MASKX) yv = load_vec<T, N>(py + o);
            acc(load_vec<T, N>(pd + o), e, yv, load_vec<T, N>(px + o));
        }
    }
    block_partials<N>(a, q, lds, CVB, PL, C, part + ((size_t)p.g * gridDim.x + blockIdx.x) * 2 * C);
}

// ---- backward, pass 2 ---------------------------------------------------------------------------------------------
template <typename T, bool MASKX>
__global__ __launch_bounds__(NB) void bn_nhwc_bwd_apply_kernel(const T *__restrict__ dy, const T *__restrict__ dy2,
                                                               const T *__restrict__ y, const T *__restrict__ x,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta,
                                                               const float *__restrict__ save_mean,
                                                               const float *__restrict__ save_invstd,
                                                               const float *__restrict__ totals, int M, int C, int CVB, int PL,
                                                               int RB, int relu, T *__restrict__ dx, T *__restrict__ dres)
{
    constexpr int N = VecN<T>::N;
    const Pos p = position<N>(M, C, CVB, PL, RB);
    if (!p.active) return;
    float mean[N], invstd[N], k0[N], mdz[N], mdzx[N], shift[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int c = p.cv * N + j;
        mean[j] = save_mean[(size_t)p.g * C + c];
        invstd[j] = save_invstd[(size_t)p.g * C + c];
        k0[j] = gamma[c] * invstd[j];                      // = the forward pass's scale
        shift[j] = MASKX ? beta[c] - mean[j] * k0[j] : 0.f;
        mdz[j] = totals[((size_t)p.g * 2 + 0) * C + c];
        mdzx[j] = totals[((size_t)p.g * 2 + 1) * C + c];
    }
    const size_t base = (size_t)p.g * M * C + (size_t)p.cv * N;
    const T *pd = dy + base, *pd2 = dy2 ? dy2 + base : nullptr, *py = MASKX ? nullptr : y + base, *px = x + base;
    T *ox = dx + base;
    T *orr = dres ? dres + base : nullptr;
    auto apply = [&](const Vec<T, N> &vd, const Vec<T, N> &vd2, const Vec<T, N> &vy, const Vec<T, N> &vx, size_t off) {
        Vec<T, N> wx, wr;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float d = to_float(vd.v[j]);
            if (pd2) d += to_float(vd2.v[j]);
            const float yv = !MASKX ? to_float(vy.v[j]) : to_float(from_float<T>(__builtin_fmaf(to_float(vx.v[j]), k0[j], shift[j])));
            const float dz = (relu && !(yv > 0.f)) ? 0.f : d;
            const float xh = (to_float(vx.v[j]) - mean[j]) * invstd[j];
            wx.v[j] = from_float<T>(k0[j] * (dz - mdz[j] - xh * mdzx[j]));
            wr.v[j] = from_float<T>(dz);
        }
        store_vec<T, N>(ox + off, wx);
        if (orr) store_vec<T, N>(orr + off, wr);
    };
    int r = p.r0 + p.pl;
    for (; r + PL < p.r1; r += 2 * PL) {
        const size_t o0 = (size_t)r * C, o1 = (size_t)(r + PL) * C;
        const Vec<T, N> d0 = load_vec<T, N>(pd + o0), d1 = load_vec<T, N>(pd + o1);
        Vec<T, N> y0 = {}, y1 = {};
        if (!MASKX) { y0 = load_vec<T, N>(py + o0); y1 = load_vec<T, N>(py + o1); }
        const Vec<T, N> x0 = load_vec<T, N>(px + o0), x1 = load_vec<T, N>(px + o1);
        Vec<T, N> e0 = {}, e1 = {};
        if (pd2) { e0 = load_vec<T, N>(pd2 + o0); e1 = load_vec<T, N>(pd2 + o1); }
        apply(d0, e0, y0, x0, o0);
        apply(d1, e1, y1, x1, o1);
    }
    for (; r < p.r1; r += PL) {
        const size_t o = (size_t)r * C;
        Vec<T, N> e = {}, yv = {};
        if (pd2) e = load_vec<T, N>(pd2 + o);
        if (!MASKX) yv = load_vec<T, N>(py + o);
        apply(load_vec<T, N>(pd + o), e, yv, load_vec<T, N>(px + o), o);
    }
}

static inline int vec_elems(int dtype) { return dtype == 0 ? 4 : 8; }

}  // namespace nhwc
}  // namespace mdx

using namespace mdx;
using namespace mdx::nhwc;

// workspace: block partials [groups][STATS blocks][2][C] + backward totals [groups][2][C], float32
MDX_EXPORT size_t mdx_bn_nhwc_workspace_bytes(int B, int C, int H, int W, int groups, int dtype)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || (dtype != 0 && dtype != 1)) return 0;
    const Rows g = make_rows((long long)B * H * W, C, vec_elems(dtype), STATS_MAX_BLOCKS, STATS_ITERS);
    return ((size_t)groups * g.nblk * 2 * C + (size_t)groups * 2 * C) * sizeof(float);
}

static int nhwc_args_ok(int B, int C, int H, int W, int groups, int dtype)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || groups > 65535) return MDX_ERR_BAD_SHAPE;
    if (dtype != 0 && dtype != 1) return MDX_ERR_BAD_SHAPE;
    if (C % vec_elems(dtype)) return MDX_ERR_BAD_SHAPE;                       // a thread owns a whole 16-byte channel vector
    if ((long long)B * H * W >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
    return MDX_OK;
}

// x, res, y: [groups * B][H][W][C] (channels-last memory), dtype 0 float32 / 1 bfloat16; statistics and parameters float32.
// B = images PER GROUP; save_mean / save_invstd: [groups][C].
MDX_EXPORT int mdx_bn_act_nhwc_fwd(const void *x, const void *res, const float *gamma, const float *beta, float *run_mean,
                                   float *run_var, void *y, float *save_mean, float *save_invstd, int B, int C, int H, int W,
                                   int groups, float eps, float momentum, int relu, int dtype, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace) return MDX_ERR_NULL_POINTER;
    if ((run_mean == nullptr) != (run_var == nullptr)) return MDX_ERR_NULL_POINTER;
    const int bad = nhwc_args_ok(B, C, H, W, groups, dtype);
    if (bad) return bad;
    if (!aligned(x, 16) || !aligned(y, 16) || (res && !aligned(res, 16))) return MDX_ERR_MISALIGNED;
    if (workspace_bytes < mdx_bn_nhwc_workspace_bytes(B, C, H, W, groups, dtype)) return MDX_ERR_WORKSPACE;
    const int M = B * H * W, N = vec_elems(dtype);
    const Rows gs = make_rows(M, C, N, STATS_FWD_BLOCKS, STATS_ITERS), ga = make_rows(M, C, N, APPLY_MAX_BLOCKS, APPLY_ITERS);
    float *part = (float *)workspace;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid_s(gs.nblk, gs.t.ny, groups), grid_a(ga.nblk, ga.t.ny, groups), block(NB);
    if (dtype == 0)
        hipLaunchKernelGGL((bn_nhwc_fwd_stats_kernel<float>), grid_s, block, 0, st, (const float *)x, M, C, gs.t.CVB, gs.t.PL, gs.RB, part);
    else
        hipLaunchKernelGGL((bn_nhwc_fwd_stats_kernel<bf16>), grid_s, block, 0, st, (const bf16 *)x, M, C, gs.t.CVB, gs.t.PL, gs.RB, part);
    hipLaunchKernelGGL(bn_nhwc_fwd_finalize_kernel, dim3((C + FC - 1) / FC), dim3(FC * FS), 0, st, part, gs.nblk, C, groups, (double)M,
                       eps, momentum, save_mean, save_invstd, run_mean, run_var);
    if (dtype == 0)
        hipLaunchKernelGGL((bn_nhwc_fwd_apply_kernel<float>), grid_a, block, 0, st, (const float *)x, (const float *)res, gamma, beta,
                           save_mean, save_invstd, M, C, ga.t.CVB, ga.t.PL, ga.RB, relu, (float *)y);
    else
        hipLaunchKernelGGL((bn_nhwc_fwd_apply_kernel<bf16>), grid_a, block, 0, st, (const bf16 *)x, (const bf16 *)res, gamma, beta,
                           save_mean, save_invstd, M, C, ga.t.CVB, ga.t.PL, ga.RB, relu, (bf16 *)y);
    return check_launch();
}

// dy2 (optional): a second upstream gradient of y, added on the way in.  dres (optional): gradient of the residual input.
// y NULL (relu, no residual, beta given): the ReLU mask is re-derived from x with the forward pass's own expressions instead of
// being read -- one map less in each of the two passes.
MDX_EXPORT int mdx_bn_act_nhwc_bwd(const void *dy, const void *dy2, const void *y, const void *x, const float *gamma,
                                   const float *beta, const float *save_mean, const float *save_invstd, void *dx, void *dres, float *dgamma,
                                   float *dbeta, int B, int C, int H, int W, int groups, int relu, int dtype, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace)
        return MDX_ERR_NULL_POINTER;
    if (!y && (!relu || !beta || dres)) return MDX_ERR_NULL_POINTER;    // y may be left out only for relu without a residual, and then the mask needs beta
    const int bad = nhwc_args_ok(B, C, H, W, groups, dtype);
    if (bad) return bad;
    if (!aligned(dy, 16) || (y && !aligned(y, 16)) || !aligned(x, 16) || !aligned(dx, 16) || (dy2 && !aligned(dy2, 16)) ||
        (dres && !aligned(dres, 16)))
        return MDX_ERR_MISALIGNED;
    if (workspace_bytes < mdx_bn_nhwc_workspace_bytes(B, C, H, W, groups, dtype)) return MDX_ERR_WORKSPACE;
    const int M = B * H * W, N = vec_elems(dtype);
    const Rows gs = make_rows(M, C, N, stats_bwd_blocks(dtype), STATS_ITERS), ga = make_rows(M, C, N, APPLY_MAX_BLOCKS, APPLY_ITERS);
    float *part = (float *)workspace;
    float *totals = part + (size_t)groups * gs.nblk * 2 * C;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid_s(gs.nblk, gs.t.ny, groups), grid_a(ga.nblk, ga.t.ny, groups), block(NB);
    const bool maskx = !y;              // (checked above: only with relu, beta and no residual)
#define MDX_BN_BWD(KERNEL, T, MX, ...) hipLaunchKernelGGL((KERNEL<T, MX>), __VA_ARGS__)
    if (dtype == 0 && maskx)
        MDX_BN_BWD(bn_nhwc_bwd_stats_kernel, float, true, grid_s, block, 0, st, (const float *)dy, (const float *)dy2, (const float *)y,
                   (const float *)x, gamma, beta, save_mean, save_invstd, M, C, gs.t.CVB, gs.t.PL, gs.RB, relu, part);
    else if (dtype == 0)
        MDX_BN_BWD(bn_nhwc_bwd_stats_kernel, float, false, grid_s, block, 0, st, (const float *)dy, (const float *)dy2, (const float *)y,
                   (const float *)x, gamma, beta, save_mean, save_invstd, M, C, gs.t.CVB, gs.t.PL, gs.RB, relu, part);
    else if (maskx)
        MDX_BN_BWD(bn_nhwc_bwd_stats_kernel, bf16, true, grid_s, block, 0, st, (const bf16 *)dy, (const bf16 *)dy2, (const bf16 *)y,
                   (const bf16 *)x, gamma, beta, save_mean, save_invstd, M, C, gs.t.CVB, gs.t.PL, gs.RB, relu, part);
    else
        MDX_BN_BWD(bn_nhwc_bwd_stats_kernel, bf16, false, grid_s, block, 0, st, (const bf16 *)dy, (const bf16 *)dy2, (const bf16 *)y,
                   (const bf16 *)x, gamma, beta, save_mean, save_invstd, M, C, gs.t.CVB, gs.t.PL, gs.RB, relu, part);
    hipLaunchKernelGGL(bn_nhwc_bwd_finalize_kernel, dim3((C + FC - 1) / FC), dim3(FC * FS), 0, st, part, gs.nblk, C, groups, (double)M, totals,
                       dgamma, dbeta);
    if (dtype == 0 && maskx)
        MDX_BN_BWD(bn_nhwc_bwd_apply_kernel, float, true, grid_a, block, 0, st, (const float *)dy, (const float *)dy2, (const float *)y,
                   (const float *)x, gamma, beta, save_mean, save_invstd, totals, M, C, ga.t.CVB, ga.t.PL, ga.RB, relu, (float *)dx, (float *)dres);
    else if (dtype == 0)
        MDX_BN_BWD(bn_nhwc_bwd_apply_kernel, float, false, grid_a, block, 0, st, (const float *)dy, (const float *)dy2, (const float *)y,
                   (const float *)x, gamma, beta, save_mean, save_invstd, totals, M, C, ga.t.CVB, ga.t.PL, ga.RB, relu, (float *)dx, (float *)dres);
    else if (maskx)
        MDX_BN_BWD(bn_nhwc_bwd_apply_kernel, bf16, true, grid_a, block, 0, st, (const bf16 *)dy, (const bf16 *)dy2, (const bf16 *)y,
                   (const bf16 *)x, gamma, beta, save_mean, save_invstd, totals, M, C, ga.t.CVB, ga.t.PL, ga.RB, relu, (bf16 *)dx, (bf16 *)dres);
    else
        MDX_BN_BWD(bn_nhwc_bwd_apply_kernel, bf16, false, grid_a, block, 0, st, (const bf16 *)dy, (const bf16 *)dy2, (const bf16 *)y,
                   (const bf16 *)x, gamma, beta, save_mean, save_invstd, totals, M, C, ga.t.CVB, ga.t.PL, ga.RB, relu, (bf16 *)dx, (bf16 *)dres);
#undef MDX_BN_BWD
    return check_launch();
}
