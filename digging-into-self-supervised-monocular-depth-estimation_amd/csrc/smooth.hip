// smooth.hip -- edge-aware disparity smoothness (SmoothLoss / EdgeAwareSmooth) for gfx950, forward + gradient of every
// scale of a step in TWO launches (round 3: four dependent ones, 41.6 us of launch latency for 39 MB).
//
// Replaces model_loss/model_loss.py:77-88 and 112-115 (called at model_tool/processor.py:208):
//   m = mean_HW(disp) + 1e-7;  dn = disp / m
//   loss = mean_x |dn[x]-dn[x+1]| * exp(-mean_c |I[x]-I[x+1]|)  +  the same along y
// The per-image mean used to force a pass of its own in front of everything else.  It does not have to:
//   |dn_i - dn_j| = |d_i - d_j| / |m|   and   sign(dn_i - dn_j) = sign(m) * sign(d_i - d_j)
// -- the main pass works on the disparity AS GIVEN and produces, per block, four partial sums (sum d, sum_x, sum_y,
// dot(G', d)) and the raw gradient map G' (SURVEY appendix A.3 with m = 1); the second launch reduces the partials of an
// image in a fixed order (every block for itself: at most 120 x 4 doubles, L2 hits), forms m and finishes:
//   loss = sum_b (Sx_b / |m_b|) / Nx + (Sy_b / |m_b|) / Ny,   gdisp = G' / |m_b| - sign(m_b) dot_b / (m_b^2 * h*w)   (in place)
// (m > 0 for the sigmoid disparities of the networks; the signs make the identity hold for any m != 0).
// The reduction order is not pinned by the reference (tolerance 1e-4 rel); sums run in double, deterministic.
#include "mdx_common.hpp"
#include "mdx_device.hpp"

namespace mdx {

constexpr int APPLY_PIX = 8 * NT;      // pixels of one image per block of the finishing pass

MDX_DEV float sgn(float a, float b) { return (a > b) ? 1.f : ((a < b) ? -1.f : 0.f); }

// the block's four partial sums -> part[(b * nbx + bx) * 4 ..]
MDX_DEV void store_partials(double sd, double sx, double sy, double dot, double *__restrict__ part, int b, int nbx, int bx)
{
    __shared__ double s_red[4][NT / 64];
    sd = wave_sum(sd); sx = wave_sum(sx); sy = wave_sum(sy); dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        s_red[0][wv] = sd; s_red[1][wv] = sx; s_red[2][wv] = sy; s_red[3][wv] = dot;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int k = 0; k < NT / 64; ++k) t += s_red[threadIdx.x][k];
        part[((size_t)b * nbx + bx) * 4 + threadIdx.x] = t;
    }
}

// Both forms of the main pass are BRANCH-FREE around their loads: a neighbour that does not exist (image border) is read at
// the pixel's own address instead -- the difference is then exactly 0, so the edge contributes 0 to the sums and sgn() = 0 to
// the gradient, no mask needed -- and a thread beyond the image repeats the last pixel with its results zeroed.  Round 4's
// first version guarded every neighbour load with its own `if`: the compiler put each one into an exec-masked block with an
// s_waitcnt vmcnt(0) behind it -- sixteen memory round trips one after the other per wave, 14.6 us of a wave's life for 436
// vector instructions (SQ_WAVE_CYCLES / SQ_WAVES, profiles/r04_bench_kernel_pmc.txt), 23.7 us per launch.

// main pass, one pixel per thread (any width / alignment)
MDX_DEV void main_body(const float *__restrict__ disp, const float *__restrict__ color, int B, int h, int w,
                       float *__restrict__ G, double *__restrict__ part, int bx, int nbx, int b)
{
    const size_t hw = (size_t)h * w;
    const float *d = disp + (size_t)b * hw;
    const float *c0 = color + (size_t)b * 3 * hw;
    const double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
    const size_t i0 = (size_t)bx * NT + threadIdx.x;
    const bool live = i0 < hw;
    const size_t i = live ? i0 : hw - 1;
    const int y = (int)((unsigned)i / (unsigned)w), x = (int)((unsigned)i % (unsigned)w);
    const float inv_nx = 1.0f / (float)Nx, inv_ny = 1.0f / (float)Ny;
    const size_t ir = x + 1 < w ? i + 1 : i, il = x > 0 ? i - 1 : i, id = y + 1 < h ? i + w : i, iu = y > 0 ? i - w : i;
    const float n0 = d[i], nr = d[ir], nl = d[il], nd = d[id], nu = d[iu];
    float gr = 0.f, gl = 0.f, gd = 0.f, gu = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *p = c0 + (size_t)c * hw;
        const float cc = p[i];
        gr += fabsf(cc - p[ir]); gl += fabsf(p[il] - cc); gd += fabsf(cc - p[id]); gu += fabsf(p[iu] - cc);
    }
    const float wr = __expf(-gr * (1.0f / 3.0f)), wl = __expf(-gl * (1.0f / 3.0f));      // hardware exponential: 1e-4 tolerance,
    const float wd = __expf(-gd * (1.0f / 3.0f)), wu = __expf(-gu * (1.0f / 3.0f));      // no pinned order
    float gacc = sgn(n0, nr) * wr * inv_nx;
    gacc -= sgn(nl, n0) * wl * inv_nx;
    gacc += sgn(n0, nd) * wd * inv_ny;
    gacc -= sgn(nu, n0) * wu * inv_ny;
    if (live && G) G[(size_t)b * hw + i] = gacc;
    const double sd = live ? (double)n0 : 0.0, sx = live ? (double)(fabsf(n0 - nr) * wr) : 0.0;
    const double sy = live ? (double)(fabsf(n0 - nd) * wd) : 0.0, dot = live ? (double)gacc * (double)n0 : 0.0;
    store_partials(sd, sx, sy, dot, part, b, nbx, bx);
}

// The same pass with FOUR consecutive pixels of a row per thread (w % 4 == 0): the rows y-1, y, y+1 of the disparity and
// of the three colour planes come in as 16-byte loads plus the two columns either side (12 vector + 8 scalar loads for
// four pixels instead of 80 scalar ones -- the one-pixel form is bound by load instructions: 30 us for 29 MB at scale 0),
// and a horizontal edge weight is formed once for the two pixels it joins.  All twenty loads are issued back to back.
MDX_DEV void main_body4(const float *__restrict__ disp, const float *__restrict__ color, int B, int h, int w,
                        float *__restrict__ G, double *__restrict__ part, int bx, int nbx, int b)
{
    const size_t hw = (size_t)h * w;
    const float *d = disp + (size_t)b * hw;
    const float *c0 = color + (size_t)b * 3 * hw;
    const double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
    const unsigned wq = (unsigned)w / 4, nq = (unsigned)h * wq;
    const unsigned q0 = (unsigned)bx * NT + threadIdx.x;          // quad index inside the image
    const bool live = q0 < nq;
    const unsigned q = live ? q0 : nq - 1;
    const int y = (int)(q / wq), x = (int)(q - (unsigned)y * wq) * 4;
    const float inv_nx = 1.0f / (float)Nx, inv_ny = 1.0f / (float)Ny;
    const size_t i = (size_t)y * w + x;
    const size_t iu = y > 0 ? i - w : i, id = y + 1 < h ? i + w : i, il = x > 0 ? i - 1 : i, ir = x + 4 < w ? i + 4 : i + 3;
    // disparity: the row, its neighbours, the columns either side
    const float4 dc = *reinterpret_cast<const float4 *>(d + i);
    const float4 du = *reinterpret_cast<const float4 *>(d + iu);
    const float4 dd = *reinterpret_cast<const float4 *>(d + id);
    const float n[6] = {d[il], dc.x, dc.y, dc.z, dc.w, d[ir]};
    const float nu[4] = {du.x, du.y, du.z, du.w};
    const float nd[4] = {dd.x, dd.y, dd.z, dd.w};
    // edge weights exp(-mean_c |I_a - I_b|): five horizontal (x-1|x ... x+3|x+4), four up, four down
    float4 cc[3], cu[3], cd[3];
    float cl[3], cr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *p = c0 + (size_t)c * hw;
        cc[c] = *reinterpret_cast<const float4 *>(p + i);
        cu[c] = *reinterpret_cast<const float4 *>(p + iu);
        cd[c] = *reinterpret_cast<const float4 *>(p + id);
        cl[c] = p[il];
        cr[c] = p[ir];
    }
    float gh[5] = {0, 0, 0, 0, 0}, gu[4] = {0, 0, 0, 0}, gd[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float row[6] = {cl[c], cc[c].x, cc[c].y, cc[c].z, cc[c].w, cr[c]};
#pragma unroll
        for (int k = 0; k < 5; ++k) gh[k] += fabsf(row[k] - row[k + 1]);
        gu[0] += fabsf(cu[c].x - cc[c].x); gu[1] += fabsf(cu[c].y - cc[c].y); gu[2] += fabsf(cu[c].z - cc[c].z); gu[3] += fabsf(cu[c].w - cc[c].w);
        gd[0] += fabsf(cc[c].x - cd[c].x); gd[1] += fabsf(cc[c].y - cd[c].y); gd[2] += fabsf(cc[c].z - cd[c].z); gd[3] += fabsf(cc[c].w - cd[c].w);
    }
    float wh[5], wu[4], wd[4];
#pragma unroll
    for (int k = 0; k < 5; ++k) wh[k] = __expf(-gh[k] * (1.0f / 3.0f));
#pragma unroll
    for (int k = 0; k < 4; ++k) { wu[k] = __expf(-gu[k] * (1.0f / 3.0f)); wd[k] = __expf(-gd[k] * (1.0f / 3.0f)); }
    float4 gout;
    float *go = &gout.x;
    double sd = 0.0, sx = 0.0, sy = 0.0, dot = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float n0 = n[k + 1], n1 = n[k + 2];
        sx += (double)(fabsf(n0 - n1) * wh[k + 1]);                 // edge to the right neighbour
        float gacc = sgn(n0, n1) * wh[k + 1] * inv_nx;
        gacc -= sgn(n[k], n0) * wh[k] * inv_nx;                    // edge to the left neighbour
        sy += (double)(fabsf(n0 - nd[k]) * wd[k]);
        gacc += sgn(n0, nd[k]) * wd[k] * inv_ny;
        gacc -= sgn(nu[k], n0) * wu[k] * inv_ny;
        go[k] = gacc;
        sd += (double)n0;
        dot += (double)gacc * (double)n0;
    }
    if (live && G) *reinterpret_cast<float4 *>(G + (size_t)b * hw + i) = gout;
    if (!live) sd = sx = sy = dot = 0.0;
    store_partials(sd, sx, sy, dot, part, b, nbx, bx);
}

// ---- all scales of a step: main pass + finishing pass (mdx_smooth_loss_multi; mdx_smooth_loss is the one-scale case) ----
struct SmoothJobs {
    int nscales, B, normalize;
    int h[MDX_MAX_SCALES], w[MDX_MAX_SCALES];
    int nblk[MDX_MAX_SCALES];         // main-pass blocks per image (vec: blocks of NT quads)
    int ngb[MDX_MAX_SCALES];          // finishing-pass blocks per image (APPLY_PIX pixels each)
    int vec[MDX_MAX_SCALES];          // 1: four pixels per thread (w % 4 == 0, 16-byte aligned planes)
    int first_main[MDX_MAX_SCALES + 1], first_apply[MDX_MAX_SCALES + 1];   // block ranges
    const float *disp[MDX_MAX_SCALES], *color[MDX_MAX_SCALES];
    float *gdisp[MDX_MAX_SCALES];
    double *part[MDX_MAX_SCALES];     // [B][nblk][4]: sum d, sum_x, sum_y, dot(G', d)
    float *loss;
};

template <typename T> MDX_DEV T spick(const T (&v)[MDX_MAX_SCALES], int s)
{
    return s == 0 ? v[0] : (s == 1 ? v[1] : (s == 2 ? v[2] : v[3]));
}
MDX_DEV int job_scale(const int (&first)[MDX_MAX_SCALES + 1], int blk)
{
    return blk >= first[3] ? 3 : (blk >= first[2] ? 2 : (blk >= first[1] ? 1 : 0));
}
MDX_DEV int job_first(const int (&first)[MDX_MAX_SCALES + 1], int s)
{
    return s == 0 ? first[0] : (s == 1 ? first[1] : (s == 2 ? first[2] : first[3]));
}

__global__ __launch_bounds__(NT) void smooth_multi_main_kernel(SmoothJobs j)
{
    const int s = job_scale(j.first_main, blockIdx.x), rel = blockIdx.x - job_first(j.first_main, s);
    const int nb = spick(j.nblk, s);
    if (spick(j.vec, s))
        main_body4(spick(j.disp, s), spick(j.color, s), j.B, spick(j.h, s), spick(j.w, s), spick(j.gdisp, s), spick(j.part, s),
                   rel % nb, nb, rel / nb);
    else
        main_body(spick(j.disp, s), spick(j.color, s), j.B, spick(j.h, s), spick(j.w, s), spick(j.gdisp, s), spick(j.part, s),
                  rel % nb, nb, rel / nb);
}

// totals of one image's block partials, by ONE WAVE (all 64 lanes call it), in a fixed order: lane-strided sums, then the
// shuffle tree; every lane returns the totals
MDX_DEV void image_totals(const double *__restrict__ part_b, int nblk, double (&t)[4])
{
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = (int)(threadIdx.x & 63); k < nblk; k += 64)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] += part_b[(size_t)k * 4 + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = __shfl(wave_sum(a[c]), 0, 64);
}

// Finishing pass.  Blocks [0, first_apply[MAX]): gdisp = G' / m - dot / (m^2 * h*w) in place, APPLY_PIX pixels of one image
// per block (every wave re-reduces the image's partials for itself -- no LDS, no barrier -- which is why a block takes 8
// pixels per thread: with one pixel per thread the re-reduction was most of the pass, 14.9 us).  The last `nscales` blocks: the loss of
// one scale each -- wave k takes images k, k+4, ..., the four wave totals are added in a fixed order.
__global__ __launch_bounds__(NT) void smooth_multi_finish_kernel(SmoothJobs j)
{
    const int napply = j.first_apply[MDX_MAX_SCALES];
    if ((int)blockIdx.x < napply) {
        const int s = job_scale(j.first_apply, blockIdx.x), rel = blockIdx.x - job_first(j.first_apply, s);
        const int hw = spick(j.h, s) * spick(j.w, s), ngb = spick(j.ngb, s), nblk = spick(j.nblk, s);
        const int b = rel / ngb, chunk = rel - b * ngb;
        double t[4];
        image_totals(spick(j.part, s) + (size_t)b * nblk * 4, nblk, t);
        const double m = (double)((float)(t[0] / (double)hw) + 1e-7f);      // mean + 1e-7 as the reference forms it (float32)
        const float inv_am = (float)(1.0 / fabs(m)), shift = (float)((m < 0.0 ? -t[3] : t[3]) / (m * m * (double)hw));
        float *G = spick(j.gdisp, s) + (size_t)b * hw;
        const unsigned base = (unsigned)chunk * APPLY_PIX;
        if (spick(j.vec, s)) {              // hw % 4 == 0, 16-byte aligned planes: two float4 per thread
#pragma unroll
            for (int r = 0; r < APPLY_PIX / (4 * NT); ++r) {
                const unsigned i = base + ((unsigned)r * NT + threadIdx.x) * 4u;
                if (i < (unsigned)hw) {
                    float4 g = *reinterpret_cast<float4 *>(G + i);
                    g.x = g.x * inv_am - shift; g.y = g.y * inv_am - shift; g.z = g.z * inv_am - shift; g.w = g.w * inv_am - shift;
                    *reinterpret_cast<float4 *>(G + i) = g;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < APPLY_PIX / NT; ++r) {
                const unsigned i = base + (unsigned)r * NT + threadIdx.x;
                if (i < (unsigned)hw) G[i] = G[i] * inv_am - shift;
            }
        }
        return;
    }
    __shared__ double s_red[NT / 64];
    const int s = (int)blockIdx.x - napply;
    const int h = spick(j.h, s), w = spick(j.w, s), nblk = spick(j.nblk, s);
    const double Nx = (double)j.B * h * (w - 1), Ny = (double)j.B * (h - 1) * w;
    double acc = 0.0;
    for (int b = (int)(threadIdx.x >> 6); b < j.B; b += NT / 64) {
        double t[4];
        image_totals(spick(j.part, s) + (size_t)b * nblk * 4, nblk, t);
        const double m = j.normalize ? (double)((float)(t[0] / (double)(h * w)) + 1e-7f) : 1.0;
        acc += (t[1] / fabs(m)) / Nx + (t[2] / fabs(m)) / Ny;
    }
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < NT / 64; ++k) tot += s_red[k];
        j.loss[s] = (float)tot;
    }
}

static size_t smooth_nblk(int h, int w) { return ((size_t)h * w + NT - 1) / NT; }
static size_t smooth_napply(int h, int w) { return ((size_t)h * w + APPLY_PIX - 1) / APPLY_PIX; }

}  // namespace mdx

using namespace mdx;

// workspace of one scale: [B][blocks][4] double block partials (sized for the one-pixel-per-thread form)
MDX_EXPORT size_t mdx_smooth_workspace_bytes(int B, int h, int w)
{
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)B * smooth_nblk(h, w) * 4 * sizeof(double);
}

MDX_EXPORT size_t mdx_smooth_multi_workspace_bytes(int nscales, int B, const int32_t *h, const int32_t *w)
{
    if (nscales < 1 || nscales > MDX_MAX_SCALES || !h || !w) return 0;
    size_t tot = 0;
    for (int s = 0; s < nscales; ++s) {
        const size_t one = mdx_smooth_workspace_bytes(B, h[s], w[s]);
        if (!one) return 0;
        tot += (one + 15) & ~(size_t)15;
    }
    return tot;
}

MDX_EXPORT int mdx_smooth_loss_multi(int nscales, int B, const int32_t *h, const int32_t *w, const float *const *disp,
                                     const float *const *color, int normalize, float *loss, float *const *gdisp,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    if (!h || !w || !disp || !color || !loss) return MDX_ERR_NULL_POINTER;
    if (nscales < 1 || nscales > MDX_MAX_SCALES || B <= 0) return MDX_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < mdx_smooth_multi_workspace_bytes(nscales, B, h, w)) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 8)) return MDX_ERR_MISALIGNED;
    SmoothJobs j = {};
    j.nscales = nscales; j.B = B; j.normalize = normalize; j.loss = loss;
    char *ws = (char *)workspace;
    int nm = 0, na = 0;
    for (int s = 0; s < MDX_MAX_SCALES; ++s) {
        j.first_main[s] = nm; j.first_apply[s] = na;
        const int ss = s < nscales ? s : 0;
        if (!disp[ss] || !color[ss]) return MDX_ERR_NULL_POINTER;
        if (h[ss] < 2 || w[ss] < 2 || (long long)B * h[ss] * w[ss] >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
        j.h[s] = h[ss]; j.w[s] = w[ss]; j.disp[s] = disp[ss]; j.color[s] = color[ss];
        j.gdisp[s] = gdisp ? gdisp[ss] : nullptr;
        j.vec[s] = (w[ss] % 4 == 0) && aligned(disp[ss], 16) && aligned(color[ss], 16) && (!gdisp || aligned(gdisp[ss], 16));
        j.nblk[s] = j.vec[s] ? (int)(((size_t)h[ss] * (w[ss] / 4) + NT - 1) / NT) : (int)smooth_nblk(h[ss], w[ss]);
        j.ngb[s] = (int)smooth_napply(h[ss], w[ss]);
        if (s >= nscales) { j.part[s] = j.part[0]; continue; }
        if ((j.gdisp[s] != nullptr) != (j.gdisp[0] != nullptr)) return MDX_ERR_NULL_POINTER;   // all or none
        j.part[s] = (double *)ws;
        ws += (mdx_smooth_workspace_bytes(B, h[s], w[s]) + 15) & ~(size_t)15;
        nm += j.nblk[s] * B;
        if (j.gdisp[s] && normalize) na += j.ngb[s] * B;          // normalize = 0: m = 1, the raw map IS the gradient
    }
    j.first_main[MDX_MAX_SCALES] = nm; j.first_apply[MDX_MAX_SCALES] = na;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smooth_multi_main_kernel, dim3(nm), dim3(NT), 0, st, j);
    hipLaunchKernelGGL(smooth_multi_finish_kernel, dim3(na + nscales), dim3(NT), 0, st, j);
    return check_launch();
}

MDX_EXPORT int mdx_smooth_loss(int B, int h, int w, const float *disp, const float *color, int normalize, float *loss,
                               float *gdisp, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!disp || !color || !loss) return MDX_ERR_NULL_POINTER;
    if (B <= 0 || h < 2 || w < 2) return MDX_ERR_BAD_SHAPE;
    const int32_t hh = h, ww = w;
    if (!workspace || workspace_bytes < mdx_smooth_workspace_bytes(B, h, w)) return MDX_ERR_WORKSPACE;
    float *const g1[1] = {gdisp};
    const float *const d1[1] = {disp};
    const float *const c1[1] = {color};
    // (the multi entry asks for each scale's share rounded up to 16 bytes; the kernels touch the unrounded size only)
    return mdx_smooth_loss_multi(1, B, &hh, &ww, d1, c1, normalize, loss, gdisp ? g1 : nullptr, workspace,
                                 ((mdx_smooth_workspace_bytes(B, h, w) + 15) & ~(size_t)15), stream);
}
