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

__global__ __launch_bounds__(NT) void smooth_mean_kernel(const float *__restrict__ disp, int hw,
                                                         float *__restrict__ den, int normalize)
{
    __shared__ double s_red[NT / 64];
    if (!normalize) {   // EdgeAwareSmooth on the disparity as given: divide by exactly 1
        if (threadIdx.x == 0) den[blockIdx.x] = 1.0f;
        return;
    }
    const float *d = disp + (size_t)blockIdx.x * hw;
    double acc = 0.0;
    for (int i = threadIdx.x; i < hw; i += NT) acc += (double)d[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < NT / 64; ++k) t += s_red[k];
        const float mean = (float)(t / (double)hw);
        den[blockIdx.x] = mean + 1e-7f;
    }
}

MDX_DEV float edge_weight(const float *__restrict__ c0, size_t hw, size_t i, size_t j)
{
    float a0 = fabsf(c0[i] - c0[j]);
    float a1 = fabsf(c0[hw + i] - c0[hw + j]);
    float a2 = fabsf(c0[2 * hw + i] - c0[2 * hw + j]);
    float g = ((a0 + a1) + a2) / 3.0f;
    return expf(-g);
}

// partials layout per block: [0] sum_x, [1] sum_y, [2] dot(G, disp)   (block -> one image row band)
__global__ __launch_bounds__(NT) void smooth_main_kernel(const float *__restrict__ disp,
                                                         const float *__restrict__ color,
                                                         const float *__restrict__ den, int B, int h, int w,
                                                         float *__restrict__ G, double *__restrict__ part)
{
    __shared__ double s_red[3][NT / 64];
    const int b = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const float *d = disp + (size_t)b * hw;
    const float *c0 = color + (size_t)b * 3 * hw;
    const float m = den[b];
    const double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
    double sx = 0.0, sy = 0.0, dot = 0.0;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i < hw) {
        const int y = (int)(i / w), x = (int)(i % w);
        const float n0 = d[i] / m;
        float gacc = 0.f;
        if (x + 1 < w) {
            const float n1 = d[i + 1] / m;
            const float wg = edge_weight(c0, hw, i, i + 1);
            sx = (double)(fabsf(n0 - n1) * wg);
            gacc += ((n0 > n1) ? 1.f : ((n0 < n1) ? -1.f : 0.f)) * wg / (float)Nx;
        }
        if (x > 0) {
            const float nm = d[i - 1] / m;
            const float wg = edge_weight(c0, hw, i - 1, i);
            gacc -= ((nm > n0) ? 1.f : ((nm < n0) ? -1.f : 0.f)) * wg / (float)Nx;
        }
        if (y + 1 < h) {
            const float n1 = d[i + w] / m;
            const float wg = edge_weight(c0, hw, i, i + w);
            sy = (double)(fabsf(n0 - n1) * wg);
            gacc += ((n0 > n1) ? 1.f : ((n0 < n1) ? -1.f : 0.f)) * wg / (float)Ny;
        }
        if (y > 0) {
            const float nm = d[i - w] / m;
            const float wg = edge_weight(c0, hw, i - w, i);
            gacc -= ((nm > n0) ? 1.f : ((nm < n0) ? -1.f : 0.f)) * wg / (float)Ny;
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
        part[((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = t;
    }
}

// one block: loss = sum_x/Nx + sum_y/Ny; per-image dot -> dots[b]
__global__ __launch_bounds__(NT) void smooth_finish_kernel(const double *__restrict__ part, int B, int nblk,
                                                           int h, int w, float *__restrict__ loss,
                                                           double *__restrict__ dots)
{
    __shared__ double s_x[NT], s_y[NT];
    double sx = 0.0, sy = 0.0;
    for (int i = threadIdx.x; i < B * nblk; i += NT) { sx += part[(size_t)i * 3]; sy += part[(size_t)i * 3 + 1]; }
    s_x[threadIdx.x] = sx; s_y[threadIdx.x] = sy;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tx = 0.0, ty = 0.0;
        for (int k = 0; k < NT; ++k) { tx += s_x[k]; ty += s_y[k]; }
        const double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
        loss[0] = (float)(tx / Nx + ty / Ny);
    }
    if (threadIdx.x < B) {
        double t = 0.0;
        for (int k = 0; k < nblk; ++k) t += part[((size_t)threadIdx.x * nblk + k) * 3 + 2];
        dots[threadIdx.x] = t;
    }
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

static size_t smooth_nblk(int h, int w) { return ((size_t)h * w + NT - 1) / NT; }

}  // namespace mdx

using namespace mdx;

// workspace: [B] float den (padded to 8) | [B] double dots | [B*nblk*3] double partials
MDX_EXPORT size_t mdx_smooth_workspace_bytes(int B, int h, int w)
{
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    const size_t den = ((size_t)B * sizeof(float) + 7) & ~(size_t)7;
    return den + (size_t)B * sizeof(double) + (size_t)B * smooth_nblk(h, w) * 3 * sizeof(double);
}

MDX_EXPORT int mdx_smooth_loss(int B, int h, int w, const float *disp, const float *color, int normalize, float *loss,
                               float *gdisp, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!disp || !color || !loss) return MDX_ERR_NULL_POINTER;
    if (B <= 0 || B > NT || h < 2 || w < 2) return MDX_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < mdx_smooth_workspace_bytes(B, h, w)) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 8)) return MDX_ERR_MISALIGNED;
    hipStream_t st = (hipStream_t)stream;
    const size_t den_bytes = ((size_t)B * sizeof(float) + 7) & ~(size_t)7;
    float *den = (float *)workspace;
    double *dots = (double *)((char *)workspace + den_bytes);
    double *part = dots + B;
    const int hw = h * w;
    const int nblk = (int)smooth_nblk(h, w);
    hipLaunchKernelGGL(smooth_mean_kernel, dim3(B), dim3(NT), 0, st, disp, hw, den, normalize);
    hipLaunchKernelGGL(smooth_main_kernel, dim3(nblk, B), dim3(NT), 0, st, disp, color, (const float *)den,
                       B, h, w, gdisp, part);
    hipLaunchKernelGGL(smooth_finish_kernel, dim3(1), dim3(NT), 0, st, (const double *)part, B, nblk, h, w,
                       loss, dots);
    if (gdisp) {
        const size_t n = (size_t)B * hw;
        hipLaunchKernelGGL(smooth_grad_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, gdisp,
                           (const float *)den, (const double *)dots, hw, n, normalize);
    }
    return check_launch();
}
