// monitor.hip -- the train-time depth monitor (reference: model_loss/model_metric.py:70-105, called every training
// step at model_train.py:69) for gfx950.
//
//   pred_up = clamp(bilinear(pred, gt size, align_corners=False), 1e-3, 80)
//   mask    = (gt > 0) inside the Garg crop window
//   pred_m *= median(gt_m) / median(pred_m);  clamp;  the seven numbers of compute_depth_error
//
// The reference compacts the masked pixels with boolean indexing (a device -> host synchronisation per step) and
// torch.median sorts them; a sync-free restatement in torch ops costs ~100 small kernels and two 0.5 M-element sorts
// (2.0 ms per call at batch 12, tools/metric_probe.py).  Here (round 3): the masked pixels are COMPACTED on the device --
// a counting pass, then every block finds its offset by summing the counts before it and writes its (gt, pred) pairs in
// pixel order: no atomics, a deterministic layout, nothing leaves the device -- and the rest works on the ~6 % of the
// window that carries lidar: EXACT lower medians (torch.median's rule) by radix selection on the float bits (all values
// are positive, so the uint32 bit pattern orders them) in three passes of 11 / 11 / 10 bits over the compact arrays with
// histograms in LDS, then one pass of sums; double partial sums added in a fixed order.  Bilinear taps are
// mdx_device.hpp's (ATen's CPU arithmetic).  Round 2's form histogrammed the window with global atomics (two 16-bit
// passes, 0.36 M atomics on clustered bins: 97 us of its 155 us) and re-derived every masked pixel three times.
#include "mdx_common.hpp"
#include "mdx_device.hpp"

namespace mdx {

constexpr int MON_PIX = 8;            // window pixels per thread of the streaming passes
constexpr int MON_BLK = NT * MON_PIX;  // window pixels per block
constexpr int MON_MB = 64;            // blocks of the sums pass

struct MonArgs {
    const float *pred, *gt;
    int B, h, w, gh, gw, r0, r1, c0, c1;
    float lo, hi;
    unsigned *count;       // [blocks]: masked pixels of each block of the window
    unsigned *sel;         // [16]: n, -, -, -, -, -, median bits gt, median bits pred | per array: n, prefix, rank, blocks done
    unsigned *ghist;       // [2][2048]: the radix passes' global histograms (zero between passes)
    double *part;          // [MON_MB][7]
    float *cg, *cp;        // [n]: ground truth / clamped upsampled prediction of the masked pixels, in pixel order
    float *out;            // [8]
    int premul;
    unsigned nblk;
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

MDX_DEV bool mon_valid(const MonArgs &a, unsigned i)
{
    const unsigned ww = (unsigned)(a.c1 - a.c0), wh = (unsigned)(a.r1 - a.r0);
    const unsigned row = i / ww, x = i - row * ww;
    const unsigned b = row / wh, y = row - b * wh;
    return a.gt[((size_t)b * a.gh + a.r0 + (int)y) * a.gw + a.c0 + (int)x] > 0.f;
}

// pass 1: masked pixels per block of MON_BLK window pixels
__global__ __launch_bounds__(NT) void mon_count_kernel(MonArgs a)
{
    __shared__ unsigned s_cnt[NT / 64];
    const unsigned n = (unsigned)a.B * (unsigned)(a.r1 - a.r0) * (unsigned)(a.c1 - a.c0);
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < MON_PIX; ++k) {
        const unsigned i = ((unsigned)blockIdx.x * MON_PIX + k) * NT + threadIdx.x;
        c += (i < n && mon_valid(a, i)) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < NT / 64; ++w) t += s_cnt[w];
        a.count[blockIdx.x] = t;
    }
}

// pass 2: the block's offset = the counts before it; its pairs go out in pixel order (k-major, then thread)
__global__ __launch_bounds__(NT) void mon_compact_kernel(MonArgs a)
{
    __shared__ unsigned s_cnt[NT / 64];
    __shared__ unsigned s_base;
    const unsigned n = (unsigned)a.B * (unsigned)(a.r1 - a.r0) * (unsigned)(a.c1 - a.c0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0) {                           // the radix passes start from empty histograms and counters
        for (int k = threadIdx.x; k < 2 * 2048; k += NT) a.ghist[k] = 0u;
        if (threadIdx.x < 16) a.sel[threadIdx.x] = 0u;
    }
    unsigned before = 0;
    for (unsigned j = threadIdx.x; j < blockIdx.x; j += NT) before += a.count[j];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
    if (lane == 0) s_cnt[wave] = before;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < NT / 64; ++w) t += s_cnt[w];
        s_base = t;
    }
    __syncthreads();
    unsigned base = s_base;
#pragma unroll 1
    for (int k = 0; k < MON_PIX; ++k) {
        const unsigned i = ((unsigned)blockIdx.x * MON_PIX + k) * NT + threadIdx.x;
        float g = 0.f, p = 0.f;
        const bool ok = i < n && mon_pixel(a, i, g, p);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        const unsigned pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        __syncthreads();                                 // the previous round's s_cnt has been read
        if (lane == 0) s_cnt[wave] = (unsigned)__builtin_popcountll(m);
        __syncthreads();
        unsigned o = base + pre, tot = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) {
            o += w < wave ? s_cnt[w] : 0u;
            tot += s_cnt[w];
        }
        if (ok) { a.cg[o] = g; a.cp[o] = p; }
        base += tot;
    }
}

// passes 3-5: the lower median (rank (n - 1) / 2) of the compact arrays by radix selection on the float bits, 11 + 11 + 10
// bits.  Per pass MON_RB blocks per array histogram their slice in LDS (eight loads in flight per thread: one block per
// array with one load at a time was a chain of ~180 memory round trips, 94 us), add their non-empty bins to the global
// histogram, and the block that finishes LAST (a counter; nobody waits for anybody) scans it, narrows (prefix, rank)
// and clears histogram and counter for the next pass.  Integer atomics only: deterministic.
constexpr int MON_RB = 16;
template <int PASS>
__global__ __launch_bounds__(1024) void mon_radix_kernel(MonArgs a)
{
    constexpr int SH = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0), WD = PASS == 2 ? 10 : 11, BINS = 1 << WD;
    __shared__ unsigned s_hist[2048];
    __shared__ unsigned s_sum[1024];
    __shared__ unsigned s_last;
    const int which = blockIdx.y;                   // 0: gt, 1: pred
    const float *v = which ? a.cp : a.cg;
    unsigned *ghist = a.ghist + which * 2048;
    unsigned *state = a.sel + 8 + 4 * which;        // n, prefix, rank, blocks done
    unsigned n, prefix = 0, rank = 0;
    if (PASS == 0) {                                // n = all counts (every block sums them: a few KB from L2)
        unsigned cnt = 0;
        for (unsigned j = threadIdx.x; j < a.nblk; j += 1024) cnt += a.count[j];
        s_sum[threadIdx.x] = cnt;
        __syncthreads();
        for (int off = 512; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
            __syncthreads();
        }
        n = s_sum[0];
        rank = n ? (n - 1) / 2 : 0;
        __syncthreads();
    } else {
        n = state[0]; prefix = state[1]; rank = state[2];
    }
    s_hist[threadIdx.x] = 0; s_hist[threadIdx.x + 1024] = 0;
    __syncthreads();
    for (unsigned i0 = blockIdx.x * 8192u; i0 < n; i0 += MON_RB * 8192u) {
        unsigned key[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned i = i0 + u * 1024u + threadIdx.x;
            key[u] = i < n ? __float_as_uint(v[i]) : 0xffffffffu;          // (no depth has these bits: clamped to [lo, hi])
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (key[u] != 0xffffffffu && (PASS == 0 || (key[u] >> (SH + WD)) == prefix))
                atomicAdd(&s_hist[(key[u] >> SH) & (unsigned)(BINS - 1)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < BINS; k += 1024)
        if (s_hist[k]) atomicAdd(&ghist[k], s_hist[k]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&state[3], 1u) == MON_RB - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const unsigned h0 = __hip_atomic_load(&ghist[2 * threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned h1 = __hip_atomic_load(&ghist[2 * threadIdx.x + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_sum[threadIdx.x] = h0 + h1;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {       // inclusive scan (Hillis-Steele)
        const unsigned t = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0u;
        __syncthreads();
        s_sum[threadIdx.x] += t;
        __syncthreads();
    }
    const unsigned before = threadIdx.x ? s_sum[threadIdx.x - 1] : 0u;
    if (n && rank >= before && rank < s_sum[threadIdx.x]) {      // exactly one thread
        const unsigned bin = rank < before + h0 ? 2 * threadIdx.x : 2 * threadIdx.x + 1;
        const unsigned np = (prefix << WD) | bin;
        state[1] = np;
        state[2] = rank - (bin == 2 * threadIdx.x ? before : before + h0);
        if (PASS == 2) a.sel[6 + which] = np;
    }
    if (threadIdx.x == 0) {
        state[0] = n;
        state[3] = 0;
        if (PASS == 2 && n == 0) a.sel[6 + which] = 0;
        if (PASS == 2 && which == 0) a.sel[0] = n;
    }
    ghist[2 * threadIdx.x] = 0; ghist[2 * threadIdx.x + 1] = 0;      // for the next pass / call
}

// pass 4: sums with the median ratio over the compact arrays; one row of seven doubles per block:
//   [0..2] counts of thresh < 1.25, 1.25^2, 1.25^3; [3] sum d^2; [4] sum (log g - log p)^2; [5] sum |d|/g; [6] sum d^2/g
__global__ __launch_bounds__(NT) void mon_metric_kernel(MonArgs a)
{
    __shared__ double s_red[7][NT / 64];
    const unsigned n = a.sel[0];
    const float ratio = __uint_as_float(a.sel[6]) / __uint_as_float(a.sel[7]);     // median(gt) / median(pred)
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (unsigned i = blockIdx.x * NT + threadIdx.x; i < n; i += MON_MB * NT) {
        const float g = a.cg[i];
        float p = a.cp[i] * ratio;
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

// workspace: [blocks] u32 counts | [16] u32 selection state | [2][2048] u32 histograms | [MON_MB][7] double partials |
//            [n] + [n] float compact arrays
static size_t mon_off_sel(unsigned nblk) { return (((size_t)nblk * 4 + 63) / 64) * 64; }
constexpr size_t MON_HDR = 64 + 2 * 2048 * 4 + (size_t)MON_MB * 7 * sizeof(double);
MDX_EXPORT size_t mdx_depth_monitor_workspace_bytes(int B, int r0, int r1, int c0, int c1)
{
    if (B <= 0 || r1 <= r0 || c1 <= c0) return 0;
    const size_t n = (size_t)B * (r1 - r0) * (c1 - c0);
    return mon_off_sel(mon_blocks(B, r0, r1, c0, c1)) + MON_HDR + 2 * n * sizeof(float);
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
    const unsigned nblk = mon_blocks(B, r0, r1, c0, c1);
    const size_t n = (size_t)B * (r1 - r0) * (c1 - c0);
    MonArgs a = {};
    a.pred = pred; a.gt = gt; a.B = B; a.h = h; a.w = w; a.gh = gh; a.gw = gw;
    a.r0 = r0; a.r1 = r1; a.c0 = c0; a.c1 = c1; a.lo = min_depth; a.hi = max_depth;
    a.premul = (gh + gw <= 128) ? 1 : 0;           // ATen's small-output bilinear kernel (see mdx_desc_init)
    char *ws = (char *)workspace;
    a.count = (unsigned *)ws;
    a.sel = (unsigned *)(ws + mon_off_sel(nblk));
    a.ghist = (unsigned *)(ws + mon_off_sel(nblk) + 64);
    a.part = (double *)(ws + mon_off_sel(nblk) + 64 + 2 * 2048 * 4);
    a.cg = (float *)(ws + mon_off_sel(nblk) + MON_HDR);
    a.cp = a.cg + n;
    a.out = out;
    a.nblk = nblk;
    hipLaunchKernelGGL(mon_count_kernel, dim3(nblk), dim3(NT), 0, st, a);
    hipLaunchKernelGGL(mon_compact_kernel, dim3(nblk), dim3(NT), 0, st, a);
    hipLaunchKernelGGL(mon_radix_kernel<0>, dim3(MON_RB, 2), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(mon_radix_kernel<1>, dim3(MON_RB, 2), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(mon_radix_kernel<2>, dim3(MON_RB, 2), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(mon_metric_kernel, dim3(MON_MB), dim3(NT), 0, st, a);
    hipLaunchKernelGGL(mon_finish_kernel, dim3(1), dim3(NT), 0, st, a, (int)MON_MB);
    return check_launch();
}
