// monitor.hip -- the train-time depth monitor (reference: model_loss/model_metric.py:70-105, called every training
// step at model_train.py:69) for gfx950.
//
//   pred_up = clamp(bilinear(pred, gt size, align_corners=False), 1e-3, 80)
//   mask    = (gt > 0) inside the Garg crop window
//   pred_m *= median(gt_m) / median(pred_m);  clamp;  the seven numbers of compute_depth_error
//
// The reference compacts the masked pixels with boolean indexing (a device -> host synchronisation per step) and
// torch.median sorts them; a sync-free restatement in torch ops costs ~100 small kernels and two 0.5 M-element sorts
// (2.0 ms per call at batch 12, tools/metric_probe.py).  Here: EXACT lower medians (torch.median's rule) by radix
// selection on the float bits -- all values are positive, so the uint32 bit pattern orders them -- in two 16-bit
// histogram passes, then one pass of masked sums.  Integer atomics only (deterministic); double partial sums added in a
// fixed order.  Bilinear taps are mdx_device.hpp's (ATen's CPU arithmetic).
#include "mdx_common.hpp"
#include "mdx_device.hpp"

namespace mdx {

constexpr int MON_BINS = 65536;
constexpr int MON_PIX = 8;            // window pixels per thread of the streaming passes

// Depth values cluster: neighbouring bins are hot together, and atomics on one 64-byte line serialise (the first
// version of the level-0 pass took 254 us for 0.36 M atomics).  Logical bin -> slot 4096 entries away from its neighbours.
MDX_DEV unsigned mon_slot(unsigned bin) { return ((bin & 15u) << 12) | (bin >> 4); }

struct MonArgs {
    const float *pred, *gt;
    int B, h, w, gh, gw, r0, r1, c0, c1;
    float lo, hi;
    unsigned *hist;        // [2][MON_BINS]: gt, pred
    unsigned *sel;         // [8]: n, rank K, hi16(gt), rank in bin, hi16(pred), rank in bin, median bits gt, median bits pred
    double *part;          // [blocks][7]
    float *out;            // [8]
    int premul;
};

// the masked pixel of window index i: ground truth and the clamped, upsampled prediction
MDX_DEV bool mon_pixel(const MonArgs &a, unsigned i, float &g, float &p)
{
    const unsigned ww = (unsigned)(a.c1 - a.c0), wh = (unsigned)(a.r1 - a.r0);
    const unsigned row = i / ww, x = i - row * ww;
    const unsigned b = row / wh, y = row - b * wh;
    const int gy = a.r0 + (int)y, gx = a.c0 + (int)x;
    g = a.gt[((size_t)b * a.gh + gy) * a.gw + gx];
    if (!(g > 0.f)) return false;
    const float v = upsample_at(a.pred + (size_t)b * a.h * a.w, a.h, a.w, a.gh, a.gw, gy, gx, a.premul != 0);
    p = fminf(fmaxf(v, a.lo), a.hi);
    return true;
}

// pass 1 / pass 3: histogram of the high (LEVEL 0) or low (LEVEL 1, inside the selected high bin) 16 bits
template <int LEVEL>
__global__ __launch_bounds__(NT) void mon_hist_kernel(MonArgs a)
{
    const unsigned n = (unsigned)a.B * (unsigned)(a.r1 - a.r0) * (unsigned)(a.c1 - a.c0);
    const unsigned hg = LEVEL ? a.sel[2] : 0, hp = LEVEL ? a.sel[4] : 0;
#pragma unroll
    for (int k = 0; k < MON_PIX; ++k) {
        const unsigned i = ((unsigned)blockIdx.x * MON_PIX + k) * NT + threadIdx.x;
        if (i >= n) continue;
        float g, p;
        if (!mon_pixel(a, i, g, p)) continue;
        const unsigned kg = __float_as_uint(g), kp = __float_as_uint(p);
        if (LEVEL == 0) {
            atomicAdd(a.hist + mon_slot(kg >> 16), 1u);
            atomicAdd(a.hist + MON_BINS + mon_slot(kp >> 16), 1u);
        } else {
            if ((kg >> 16) == hg) atomicAdd(a.hist + mon_slot(kg & 0xffffu), 1u);
            if ((kp >> 16) == hp) atomicAdd(a.hist + MON_BINS + mon_slot(kp & 0xffffu), 1u);
        }
    }
}

// pass 2 / pass 4: one block per array walks the histogram's prefix sums to the bin that holds the wanted rank, then
// clears the histogram for the next pass / call.  LEVEL 0 also fixes n and the rank K = (n - 1) / 2.
template <int LEVEL>
__global__ __launch_bounds__(1024) void mon_select_kernel(MonArgs a)
{
    __shared__ unsigned s_sum[1024];
    const int which = blockIdx.x;                   // 0: gt, 1: pred
    unsigned *h = a.hist + which * MON_BINS;
    constexpr int PER = MON_BINS / 1024;
    unsigned loc[PER], tot = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { loc[k] = h[mon_slot(threadIdx.x * PER + k)]; tot += loc[k]; }
    s_sum[threadIdx.x] = tot;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {      // inclusive scan (Hillis-Steele)
        const unsigned v = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0u;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned n = LEVEL ? a.sel[0] : s_sum[1023];
    unsigned rank;
    if (LEVEL == 0) rank = n ? (n - 1) / 2 : 0;
    else rank = a.sel[3 + 2 * which];
    const unsigned before = threadIdx.x ? s_sum[threadIdx.x - 1] : 0u;
    if (n && rank >= before && rank < s_sum[threadIdx.x]) {      // exactly one thread
        unsigned acc = before;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (rank < acc + loc[k]) {
                if (LEVEL == 0) {
                    a.sel[2 + 2 * which] = threadIdx.x * PER + k;      // high 16 bits
                    a.sel[3 + 2 * which] = rank - acc;                 // rank inside that bin
                } else {
                    a.sel[6 + which] = (a.sel[2 + 2 * which] << 16) | (unsigned)(threadIdx.x * PER + k);
                }
                break;
            }
            acc += loc[k];
        }
    }
    if (LEVEL == 0 && which == 0 && threadIdx.x == 0) { a.sel[0] = n; a.sel[1] = rank; }
    if (n == 0 && threadIdx.x == 0) { a.sel[2 + 2 * which] = 0; a.sel[3 + 2 * which] = 0; a.sel[6 + which] = 0; }
#pragma unroll
    for (int k = 0; k < PER; ++k) h[mon_slot(threadIdx.x * PER + k)] = 0u;
}

// pass 5: masked sums with the median ratio; one row of seven doubles per block:
//   [0..2] counts of thresh < 1.25, 1.25^2, 1.25^3; [3] sum d^2; [4] sum (log g - log p)^2; [5] sum |d|/g; [6] sum d^2/g
__global__ __launch_bounds__(NT) void mon_metric_kernel(MonArgs a)
{
    __shared__ double s_red[7][NT / 64];
    const unsigned n = (unsigned)a.B * (unsigned)(a.r1 - a.r0) * (unsigned)(a.c1 - a.c0);
    const float ratio = __uint_as_float(a.sel[6]) / __uint_as_float(a.sel[7]);     // median(gt) / median(pred)
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < MON_PIX; ++k) {
        const unsigned i = ((unsigned)blockIdx.x * MON_PIX + k) * NT + threadIdx.x;
        if (i >= n) continue;
        float g, p;
        if (!mon_pixel(a, i, g, p)) continue;
        p = p * ratio;
        p = fminf(fmaxf(p, a.lo), a.hi);
        const float t = fmaxf(g / p, p / g);
        acc[0] += t < 1.25f ? 1.0 : 0.0;
        acc[1] += t < 1.5625f ? 1.0 : 0.0;
        acc[2] += t < 1.953125f ? 1.0 : 0.0;
        const float d = g - p;
        acc[3] += (double)(d * d);
        const float l = logf(g) - logf(p);
        acc[4] += (double)(l * l);
        acc[5] += (double)(fabsf(d) / g);
        acc[6] += (double)((d * d) / g);
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const double v = wave_sum(acc[q]);
        if ((threadIdx.x & 63) == 0) s_red[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double t = 0.0;
        for (int k = 0; k < NT / 64; ++k) t += s_red[threadIdx.x][k];
        a.part[(size_t)blockIdx.x * 7 + threadIdx.x] = t;
    }
}

// pass 6: fixed-order sums over the blocks -> (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, n)
__global__ __launch_bounds__(NT) void mon_finish_kernel(MonArgs a, int nblk)
{
    __shared__ double s_red[7][NT / 64];
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += NT)
#pragma unroll
        for (int q = 0; q < 7; ++q) acc[q] += a.part[(size_t)i * 7 + q];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const double v = wave_sum(acc[q]);
        if ((threadIdx.x & 63) == 0) s_red[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[7];
        for (int q = 0; q < 7; ++q) {
            t[q] = 0.0;
            for (int k = 0; k < NT / 64; ++k) t[q] += s_red[q][k];
        }
        const double n = (double)a.sel[0];
        const float nanv = __uint_as_float(0x7fc00000u);
        if (a.sel[0] == 0) {
            for (int q = 0; q < 7; ++q) a.out[q] = nanv;       // torch's mean / median of an empty selection
        } else {
            a.out[0] = (float)(t[5] / n);          // abs_rel
            a.out[1] = (float)(t[6] / n);          // sq_rel
            a.out[2] = (float)sqrt(t[3] / n);      // rmse
            a.out[3] = (float)sqrt(t[4] / n);      // rmse_log
            a.out[4] = (float)(t[0] / n);
            a.out[5] = (float)(t[1] / n);
            a.out[6] = (float)(t[2] / n);
        }
        a.out[7] = (float)n;
    }
}

static unsigned mon_blocks(int B, int r0, int r1, int c0, int c1)
{
    const size_t n = (size_t)B * (r1 - r0) * (c1 - c0);
    return (unsigned)((n + (size_t)NT * MON_PIX - 1) / ((size_t)NT * MON_PIX));
}

}  // namespace mdx

using namespace mdx;

// workspace: [2*65536] u32 histograms | [8] u32 selection state | [blocks][7] double partials
MDX_EXPORT size_t mdx_depth_monitor_workspace_bytes(int B, int r0, int r1, int c0, int c1)
{
    if (B <= 0 || r1 <= r0 || c1 <= c0) return 0;
    return (size_t)2 * MON_BINS * 4 + 64 + (size_t)mon_blocks(B, r0, r1, c0, c1) * 7 * sizeof(double);
}

MDX_EXPORT int mdx_depth_monitor(const float *pred, int B, int h, int w, const float *gt, int gh, int gw, int r0, int r1,
                                 int c0, int c1, float min_depth, float max_depth, float *out, void *workspace,
                                 size_t workspace_bytes, void *stream)
{
    if (!pred || !gt || !out) return MDX_ERR_NULL_POINTER;
    if (B <= 0 || h <= 0 || w <= 0 || gh <= 0 || gw <= 0 || r0 < 0 || c0 < 0 || r1 > gh || c1 > gw || r1 <= r0 || c1 <= c0)
        return MDX_ERR_BAD_SHAPE;
    if ((long long)B * gh * gw >= (1ll << 31) || !(min_depth > 0.f) || !(max_depth > min_depth)) return MDX_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < mdx_depth_monitor_workspace_bytes(B, r0, r1, c0, c1)) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 8)) return MDX_ERR_MISALIGNED;
    hipStream_t st = (hipStream_t)stream;
    MonArgs a = {};
    a.pred = pred; a.gt = gt; a.B = B; a.h = h; a.w = w; a.gh = gh; a.gw = gw;
    a.r0 = r0; a.r1 = r1; a.c0 = c0; a.c1 = c1; a.lo = min_depth; a.hi = max_depth;
    a.premul = (gh + gw <= 128) ? 1 : 0;           // ATen's small-output bilinear kernel (see mdx_desc_init)
    a.hist = (unsigned *)workspace;
    a.sel = a.hist + 2 * MON_BINS;
    a.part = (double *)((char *)workspace + (size_t)2 * MON_BINS * 4 + 64);
    a.out = out;
    const unsigned nblk = mon_blocks(B, r0, r1, c0, c1);
    if (hipMemsetAsync(workspace, 0, (size_t)2 * MON_BINS * 4 + 64, st) != hipSuccess) return MDX_ERR_LAUNCH;
    hipLaunchKernelGGL(mon_hist_kernel<0>, dim3(nblk), dim3(NT), 0, st, a);
    hipLaunchKernelGGL(mon_select_kernel<0>, dim3(2), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(mon_hist_kernel<1>, dim3(nblk), dim3(NT), 0, st, a);
    hipLaunchKernelGGL(mon_select_kernel<1>, dim3(2), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(mon_metric_kernel, dim3(nblk), dim3(NT), 0, st, a);
    hipLaunchKernelGGL(mon_finish_kernel, dim3(1), dim3(NT), 0, st, a, (int)nblk);
    return check_launch();
}
