// smooth.hip -- edge-aware disparity smoothness (SmoothLoss / EdgeAwareSmooth) for gfx950.
//
// Replaces model_loss/model_loss.py:77-88 and 112-115 (called at model_tool/processor.py:208):
//   m = mean_HW(disp) + 1e-7;  dn = disp / m
//   loss = mean_x |dn[x]-dn[x+1]| * exp(-mean_c |I[x]-I[x+1]|)  +  the same along y
// Two phases: (1) per-image mean (one block per image, wave64 shuffles), (2) one pass over the
// pixels producing the loss partials and -- for training -- the gradient map G and the per-image
// dot(G, disp) partials needed by d(disp/m)/d(disp); (3) a finishing pass.  Closed form: SURVEY A.3.
// The reduction order is not pinned by the reference (tolerance 1e-4 rel); sums run in double.
#include "mdx_common.hpp"
#include "mdx_device.hpp"

namespace mdx {

constexpr int MEAN_CHUNK = NT * 4;   // pixels per block of the partial-mean pass

// pass 1: per-block partial sums of disp (double); block (bx of nbx, image by)
MDX_DEV void partial_sum_body(const float *__restrict__ disp, int hw, double *__restrict__ psum, int bx, int nbx, int by)
{
    __shared__ double s_red[NT / 64];
    const float *d = disp + (size_t)by * hw;
    double acc = 0.0;
    const int base = bx * MEAN_CHUNK;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + k * NT + threadIdx.x;
        if (i < hw) acc += (double)d[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < NT / 64; ++k) t += s_red[k];
        psum[(size_t)by * nbx + bx] = t;
    }
}

__global__ __launch_bounds__(NT) void smooth_partial_sum_kernel(const float *__restrict__ disp, int hw,
                                                                double *__restrict__ psum)
{
    partial_sum_body(disp, hw, psum, blockIdx.x, gridDim.x, blockIdx.y);
}

// every block of the main pass re-reduces the (few) partial sums of its image: den = mean + 1e-7
MDX_DEV float block_den(const double *__restrict__ psum, int nchunk, int hw, int normalize, double *s_red)
{
    if (!normalize) return 1.0f;   // EdgeAwareSmooth on the disparity as given: divide by exactly 1
    double acc = 0.0;
    for (int i = threadIdx.x; i < nchunk; i += NT) acc += psum[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < NT / 64; ++k) t += s_red[k];
    __syncthreads();
    const float mean = (float)(t / (double)hw);
    return mean + 1e-7f;
}

MDX_DEV float edge_weight(const float *__restrict__ c0, size_t hw, size_t i, size_t j)
{
    float a0 = fabsf(c0[i] - c0[j]);
    float a1 = fabsf(c0[hw + i] - c0[hw + j]);
    float a2 = fabsf(c0[2 * hw + i] - c0[2 * hw + j]);
    const float g = ((a0 + a1) + a2) * (1.0f / 3.0f);
    return __expf(-g);      // hardware exponential: the smoothness term carries a 1e-4 tolerance, no pinned order
}

// partials layout per block: [0] sum_x, [1] sum_y, [2] dot(G, disp)   (block -> one image row band)
MDX_DEV void main_body(const float *__restrict__ disp, const float *__restrict__ color, const double *__restrict__ psum,
                       int nchunk, int normalize, float *__restrict__ den, int B, int h, int w, float *__restrict__ G,
                       double *__restrict__ part, int bx, int nbx, int b)
{
    __shared__ double s_red[3][NT / 64];
    const size_t hw = (size_t)h * w;
    const float *d = disp + (size_t)b * hw;
    const float *c0 = color + (size_t)b * 3 * hw;
    const float m = block_den(psum + (size_t)b * nchunk, nchunk, (int)hw, normalize, &s_red[0][0]);
    if (bx == 0 && threadIdx.x == 0) den[b] = m;
    const double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
    double sx = 0.0, sy = 0.0, dot = 0.0;
    const size_t i = (size_t)bx * NT + threadIdx.x;
    if (i < hw) {
        const int y = (int)((unsigned)i / (unsigned)w), x = (int)((unsigned)i % (unsigned)w);
        const float inv_m = 1.0f / m, inv_nx = 1.0f / (float)Nx, inv_ny = 1.0f / (float)Ny;
        const float n0 = d[i] * inv_m;
        float gacc = 0.f;
        if (x + 1 < w) {
            const float n1 = d[i + 1] * inv_m;
            const float wg = edge_weight(c0, hw, i, i + 1);
            sx = (double)(fabsf(n0 - n1) * wg);
            gacc += ((n0 > n1) ? 1.f : ((n0 < n1) ? -1.f : 0.f)) * wg * inv_nx;
        }
        if (x > 0) {
            const float nm = d[i - 1] * inv_m;
            const float wg = edge_weight(c0, hw, i - 1, i);
            gacc -= ((nm > n0) ? 1.f : ((nm < n0) ? -1.f : 0.f)) * wg * inv_nx;
        }
        if (y + 1 < h) {
            const float n1 = d[i + w] * inv_m;
            const float wg = edge_weight(c0, hw, i, i + w);
            sy = (double)(fabsf(n0 - n1) * wg);
            gacc += ((n0 > n1) ? 1.f : ((n0 < n1) ? -1.f : 0.f)) * wg * inv_ny;
        }
        if (y > 0) {
            const float nm = d[i - w] * inv_m;
            const float wg = edge_weight(c0, hw, i - w, i);
            gacc -= ((nm > n0) ? 1.f : ((nm < n0) ? -1.f : 0.f)) * wg * inv_ny;
        }
        if (G) G[(size_t)b * hw + i] = gacc;
        dot = (double)gacc * (double)d[i];
    }
    sx = wave_sum(sx); sy = wave_sum(sy); dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) {
        s_red[0][threadIdx.x >> 6] = sx; s_red[1][threadIdx.x >> 6] = sy; s_red[2][threadIdx.x >> 6] = dot;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int k = 0; k < NT / 64; ++k) t += s_red[threadIdx.x][k];
        part[((size_t)b * nbx + bx) * 3 + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(NT) void smooth_main_kernel(const float *__restrict__ disp,
                                                         const float *__restrict__ color,
                                                         const double *__restrict__ psum, int nchunk, int normalize,
                                                         float *__restrict__ den, int B, int h, int w,
                                                         float *__restrict__ G, double *__restrict__ part)
{
    main_body(disp, color, psum, nchunk, normalize, den, B, h, w, G, part, blockIdx.x, gridDim.x, blockIdx.y);
}

// one block: loss = sum_x/Nx + sum_y/Ny (parallel strided sums + LDS tree); per-image dot -> dots[b]
// The same pass with FOUR consecutive pixels of a row per thread (w % 4 == 0): the rows y-1, y, y+1 of the disparity and
// of the three colour planes come in as 16-byte loads plus the two columns either side (12 vector + 8 scalar loads for
// four pixels instead of 80 scalar ones -- the one-pixel form is bound by load instructions: 30 us for 29 MB at scale 0),
// and a horizontal edge weight is formed once for the two pixels it joins.
MDX_DEV void main_body4(const float *__restrict__ disp, const float *__restrict__ color, const double *__restrict__ psum,
                        int nchunk, int normalize, float *__restrict__ den, int B, int h, int w, float *__restrict__ G,
                        double *__restrict__ part, int bx, int nbx, int b)
{
    __shared__ double s_red[3][NT / 64];
    const size_t hw = (size_t)h * w;
    const float *d = disp + (size_t)b * hw;
    const float *c0 = color + (size_t)b * 3 * hw;
    const float m = block_den(psum + (size_t)b * nchunk, nchunk, (int)hw, normalize, &s_red[0][0]);
    if (bx == 0 && threadIdx.x == 0) den[b] = m;
    const double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
    double sx = 0.0, sy = 0.0, dot = 0.0;
    const unsigned q = (unsigned)bx * NT + threadIdx.x;          // quad index inside the image
    const unsigned wq = (unsigned)w / 4;
    if (q < (unsigned)h * wq) {
        const int y = (int)(q / wq), x = (int)(q - (unsigned)y * wq) * 4;
        const float inv_m = 1.0f / m, inv_nx = 1.0f / (float)Nx, inv_ny = 1.0f / (float)Ny;
        const size_t i = (size_t)y * w + x;
        const bool up = y > 0, dn = y + 1 < h, lf = x > 0, rt = x + 4 < w;
        // disparity: the row, its neighbours, the columns either side
        const float4 dc = *reinterpret_cast<const float4 *>(d + i);
        const float4 du = up ? *reinterpret_cast<const float4 *>(d + i - w) : dc;
        const float4 dd = dn ? *reinterpret_cast<const float4 *>(d + i + w) : dc;
        const float dl = lf ? d[i - 1] : 0.f, dr = rt ? d[i + 4] : 0.f;
        const float n[6] = {dl * inv_m, dc.x * inv_m, dc.y * inv_m, dc.z * inv_m, dc.w * inv_m, dr * inv_m};
        const float nu[4] = {du.x * inv_m, du.y * inv_m, du.z * inv_m, du.w * inv_m};
        const float nd[4] = {dd.x * inv_m, dd.y * inv_m, dd.z * inv_m, dd.w * inv_m};
        // edge weights exp(-mean_c |I_a - I_b|): five horizontal (x-1|x ... x+3|x+4), four up, four down
        float gh[5] = {0, 0, 0, 0, 0}, gu[4] = {0, 0, 0, 0}, gd[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *p = c0 + (size_t)c * hw + i;
            const float4 cc = *reinterpret_cast<const float4 *>(p);
            const float4 cu = up ? *reinterpret_cast<const float4 *>(p - w) : cc;
            const float4 cd = dn ? *reinterpret_cast<const float4 *>(p + w) : cc;
            const float cl = lf ? p[-1] : cc.x, cr = rt ? p[4] : cc.w;
            const float row[6] = {cl, cc.x, cc.y, cc.z, cc.w, cr};
#pragma unroll
            for (int k = 0; k < 5; ++k) gh[k] += fabsf(row[k] - row[k + 1]);
            gu[0] += fabsf(cu.x - cc.x); gu[1] += fabsf(cu.y - cc.y); gu[2] += fabsf(cu.z - cc.z); gu[3] += fabsf(cu.w - cc.w);
            gd[0] += fabsf(cc.x - cd.x); gd[1] += fabsf(cc.y - cd.y); gd[2] += fabsf(cc.z - cd.z); gd[3] += fabsf(cc.w - cd.w);
        }
        float wh[5], wu[4], wd[4];
#pragma unroll
        for (int k = 0; k < 5; ++k) wh[k] = __expf(-gh[k] * (1.0f / 3.0f));
#pragma unroll
        for (int k = 0; k < 4; ++k) { wu[k] = __expf(-gu[k] * (1.0f / 3.0f)); wd[k] = __expf(-gd[k] * (1.0f / 3.0f)); }
        float4 gout;
        float *go = &gout.x;
        const float dv[4] = {dc.x, dc.y, dc.z, dc.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float n0 = n[k + 1];
            float gacc = 0.f;
            if (k < 3 || rt) {                                   // edge to the right neighbour
                const float n1 = n[k + 2];
                sx += (double)(fabsf(n0 - n1) * wh[k + 1]);
                gacc += ((n0 > n1) ? 1.f : ((n0 < n1) ? -1.f : 0.f)) * wh[k + 1] * inv_nx;
            }
            if (k > 0 || lf) {                                   // edge to the left neighbour
                const float nm = n[k];
                gacc -= ((nm > n0) ? 1.f : ((nm < n0) ? -1.f : 0.f)) * wh[k] * inv_nx;
            }
            if (dn) {
                sy += (double)(fabsf(n0 - nd[k]) * wd[k]);
                gacc += ((n0 > nd[k]) ? 1.f : ((n0 < nd[k]) ? -1.f : 0.f)) * wd[k] * inv_ny;
            }
            if (up) gacc -= ((nu[k] > n0) ? 1.f : ((nu[k] < n0) ? -1.f : 0.f)) * wu[k] * inv_ny;
            go[k] = gacc;
            dot += (double)gacc * (double)dv[k];
        }
        if (G) *reinterpret_cast<float4 *>(G + (size_t)b * hw + i) = gout;
    }
    sx = wave_sum(sx); sy = wave_sum(sy); dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) {
        s_red[0][threadIdx.x >> 6] = sx; s_red[1][threadIdx.x >> 6] = sy; s_red[2][threadIdx.x >> 6] = dot;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int k = 0; k < NT / 64; ++k) t += s_red[threadIdx.x][k];
        part[((size_t)b * nbx + bx) * 3 + threadIdx.x] = t;
    }
}

MDX_DEV void finish_body(const double *__restrict__ part, int B, int nblk, int h, int w, float *__restrict__ loss,
                         double *__restrict__ dots)
{
    __shared__ double s_x[NT / 64], s_y[NT / 64];
    double sx = 0.0, sy = 0.0;
    for (int i = threadIdx.x; i < B * nblk; i += NT) { sx += part[(size_t)i * 3]; sy += part[(size_t)i * 3 + 1]; }
    sx = wave_sum(sx); sy = wave_sum(sy);
    if ((threadIdx.x & 63) == 0) { s_x[threadIdx.x >> 6] = sx; s_y[threadIdx.x >> 6] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tx = 0.0, ty = 0.0;
        for (int k = 0; k < NT / 64; ++k) { tx += s_x[k]; ty += s_y[k]; }
        const double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
        loss[0] = (float)(tx / Nx + ty / Ny);
    }
    // one wave per image (round-robin) for dot(G, disp)
    for (int b = threadIdx.x >> 6; b < B; b += NT / 64) {
        double t = 0.0;
        for (int k = threadIdx.x & 63; k < nblk; k += 64) t += part[((size_t)b * nblk + k) * 3 + 2];
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) dots[b] = t;
    }
}

__global__ __launch_bounds__(NT) void smooth_finish_kernel(const double *__restrict__ part, int B, int nblk,
                                                           int h, int w, float *__restrict__ loss,
                                                           double *__restrict__ dots)
{
    finish_body(part, B, nblk, h, w, loss, dots);
}

// gdisp = G/m - dot/(m^2 * h*w)      (in place over G)
__global__ __launch_bounds__(NT) void smooth_grad_kernel(float *__restrict__ G, const float *__restrict__ den,
                                                         const double *__restrict__ dots, int hw, size_t n,
                                                         int normalize)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n || !normalize) return;
    const int b = (int)(i / hw);
    const double m = (double)den[b];
    G[i] = (float)((double)G[i] / m - dots[b] / (m * m * (double)hw));
}

// ---- all scales of a step in one launch per pass (mdx_smooth_loss_multi) ----
struct SmoothJobs {
    int nscales, B, normalize;
    int h[MDX_MAX_SCALES], w[MDX_MAX_SCALES], nblk[MDX_MAX_SCALES], nchunk[MDX_MAX_SCALES];
    int vec[MDX_MAX_SCALES];          // 1: four pixels per thread (w % 4 == 0, 16-byte aligned planes); nblk counts those blocks
    int first_main[MDX_MAX_SCALES + 1], first_sum[MDX_MAX_SCALES + 1], first_grad[MDX_MAX_SCALES + 1];   // block ranges
    const float *disp[MDX_MAX_SCALES], *color[MDX_MAX_SCALES];
    float *gdisp[MDX_MAX_SCALES], *den[MDX_MAX_SCALES];
    double *dots[MDX_MAX_SCALES], *part[MDX_MAX_SCALES], *psum[MDX_MAX_SCALES];
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

__global__ __launch_bounds__(NT) void smooth_multi_sum_kernel(SmoothJobs j)
{
    const int s = job_scale(j.first_sum, blockIdx.x), rel = blockIdx.x - job_first(j.first_sum, s);
    const int nc = spick(j.nchunk, s);
    partial_sum_body(spick(j.disp, s), spick(j.h, s) * spick(j.w, s), spick(j.psum, s), rel % nc, nc, rel / nc);
}

__global__ __launch_bounds__(NT) void smooth_multi_main_kernel(SmoothJobs j)
{
    const int s = job_scale(j.first_main, blockIdx.x), rel = blockIdx.x - job_first(j.first_main, s);
    const int nb = spick(j.nblk, s);
    if (spick(j.vec, s))
        main_body4(spick(j.disp, s), spick(j.color, s), spick(j.psum, s), spick(j.nchunk, s), j.normalize, spick(j.den, s), j.B,
                   spick(j.h, s), spick(j.w, s), spick(j.gdisp, s), spick(j.part, s), rel % nb, nb, rel / nb);
    else
        main_body(spick(j.disp, s), spick(j.color, s), spick(j.psum, s), spick(j.nchunk, s), j.normalize, spick(j.den, s), j.B,
                  spick(j.h, s), spick(j.w, s), spick(j.gdisp, s), spick(j.part, s), rel % nb, nb, rel / nb);
}

__global__ __launch_bounds__(NT) void smooth_multi_finish_kernel(SmoothJobs j)
{
    const int s = blockIdx.x;
    finish_body(spick(j.part, s), j.B, spick(j.nblk, s), spick(j.h, s), spick(j.w, s), j.loss + s, spick(j.dots, s));
}

__global__ __launch_bounds__(NT) void smooth_multi_grad_kernel(SmoothJobs j)
{
    if (!j.normalize) return;
    const int s = job_scale(j.first_grad, blockIdx.x), rel = blockIdx.x - job_first(j.first_grad, s);
    const int hw = spick(j.h, s) * spick(j.w, s);
    const unsigned i = (unsigned)rel * NT + threadIdx.x;
    if (i >= (unsigned)j.B * (unsigned)hw) return;
    float *G = spick(j.gdisp, s);
    const int b = (int)(i / (unsigned)hw);
    const double m = (double)spick(j.den, s)[b];
    G[i] = (float)((double)G[i] / m - spick(j.dots, s)[b] / (m * m * (double)hw));
}

static size_t smooth_nblk(int h, int w) { return ((size_t)h * w + NT - 1) / NT; }
static size_t smooth_nchunk(int h, int w) { return ((size_t)h * w + MEAN_CHUNK - 1) / MEAN_CHUNK; }

}  // namespace mdx

using namespace mdx;

// workspace: [B] float den (padded to 8) | [B] double dots | [B*nblk*3] double partials | [B*nchunk] double psum
MDX_EXPORT size_t mdx_smooth_workspace_bytes(int B, int h, int w)
{
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    const size_t den = ((size_t)B * sizeof(float) + 7) & ~(size_t)7;
    return den + (size_t)B * sizeof(double) + (size_t)B * smooth_nblk(h, w) * 3 * sizeof(double) +
           (size_t)B * smooth_nchunk(h, w) * sizeof(double);
}

MDX_EXPORT int mdx_smooth_loss(int B, int h, int w, const float *disp, const float *color, int normalize, float *loss,
                               float *gdisp, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!disp || !color || !loss) return MDX_ERR_NULL_POINTER;
    if (B <= 0 || h < 2 || w < 2) return MDX_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < mdx_smooth_workspace_bytes(B, h, w)) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 8)) return MDX_ERR_MISALIGNED;
    hipStream_t st = (hipStream_t)stream;
    const size_t den_bytes = ((size_t)B * sizeof(float) + 7) & ~(size_t)7;
    float *den = (float *)workspace;
    double *dots = (double *)((char *)workspace + den_bytes);
    double *part = dots + B;
    const int hw = h * w;
    const int nblk = (int)smooth_nblk(h, w);
    const int nchunk = (int)smooth_nchunk(h, w);
    double *psum = part + (size_t)B * nblk * 3;
    if (normalize)
        hipLaunchKernelGGL(smooth_partial_sum_kernel, dim3(nchunk, B), dim3(NT), 0, st, disp, hw, psum);
    hipLaunchKernelGGL(smooth_main_kernel, dim3(nblk, B), dim3(NT), 0, st, disp, color, (const double *)psum,
                       nchunk, normalize, den, B, h, w, gdisp, part);
    hipLaunchKernelGGL(smooth_finish_kernel, dim3(1), dim3(NT), 0, st, (const double *)part, B, nblk, h, w,
                       loss, dots);
    if (gdisp) {
        const size_t n = (size_t)B * hw;
        hipLaunchKernelGGL(smooth_grad_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, gdisp,
                           (const float *)den, (const double *)dots, hw, n, normalize);
    }
    return check_launch();
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
    int nm = 0, ns = 0, ng = 0;
    bool grads = false;
    for (int s = 0; s < MDX_MAX_SCALES; ++s) {
        j.first_main[s] = nm; j.first_sum[s] = ns; j.first_grad[s] = ng;
        const int ss = s < nscales ? s : 0;
        if (!disp[ss] || !color[ss]) return MDX_ERR_NULL_POINTER;
        if (h[ss] < 2 || w[ss] < 2 || (long long)B * h[ss] * w[ss] >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
        j.h[s] = h[ss]; j.w[s] = w[ss]; j.disp[s] = disp[ss]; j.color[s] = color[ss];
        j.gdisp[s] = gdisp ? gdisp[ss] : nullptr;
        j.vec[s] = (w[ss] % 4 == 0) && aligned(disp[ss], 16) && aligned(color[ss], 16) && (!gdisp || aligned(gdisp[ss], 16));
        j.nblk[s] = j.vec[s] ? (int)(((size_t)h[ss] * (w[ss] / 4) + NT - 1) / NT) : (int)smooth_nblk(h[ss], w[ss]);
        j.nchunk[s] = (int)smooth_nchunk(h[ss], w[ss]);
        if (s >= nscales) { j.den[s] = j.den[0]; j.dots[s] = j.dots[0]; j.part[s] = j.part[0]; j.psum[s] = j.psum[0]; continue; }
        const size_t den_bytes = ((size_t)B * sizeof(float) + 7) & ~(size_t)7;
        j.den[s] = (float *)ws;
        j.dots[s] = (double *)(ws + den_bytes);
        j.part[s] = j.dots[s] + B;
        j.psum[s] = j.part[s] + (size_t)B * j.nblk[s] * 3;
        ws += (mdx_smooth_workspace_bytes(B, h[s], w[s]) + 15) & ~(size_t)15;
        nm += j.nblk[s] * B; ns += j.nchunk[s] * B;
        ng += (int)(((size_t)B * h[s] * w[s] + NT - 1) / NT);
        grads = grads || j.gdisp[s] != nullptr;
        if ((j.gdisp[s] != nullptr) != (j.gdisp[0] != nullptr)) return MDX_ERR_NULL_POINTER;   // all or none
    }
    j.first_main[MDX_MAX_SCALES] = nm; j.first_sum[MDX_MAX_SCALES] = ns; j.first_grad[MDX_MAX_SCALES] = ng;
    hipStream_t st = (hipStream_t)stream;
    if (normalize) hipLaunchKernelGGL(smooth_multi_sum_kernel, dim3(ns), dim3(NT), 0, st, j);
    hipLaunchKernelGGL(smooth_multi_main_kernel, dim3(nm), dim3(NT), 0, st, j);
    hipLaunchKernelGGL(smooth_multi_finish_kernel, dim3(nscales), dim3(NT), 0, st, j);
    if (grads && normalize) hipLaunchKernelGGL(smooth_multi_grad_kernel, dim3(ng), dim3(NT), 0, st, j);
    return check_launch();
}
