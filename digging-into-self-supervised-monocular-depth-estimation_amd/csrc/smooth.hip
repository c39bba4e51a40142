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

// pass 1: per-block partial sums of disp (double), grid (nchunk, B)
__global__ __launch_bounds__(NT) void smooth_partial_sum_kernel(const float *__restrict__ disp, int hw,
                                                                double *__restrict__ psum)
{
    __shared__ double s_red[NT / 64];
    const float *d = disp + (size_t)blockIdx.y * hw;
    double acc = 0.0;
    const int base = blockIdx.x * MEAN_CHUNK;
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
        psum[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
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
    float g = ((a0 + a1) + a2) / 3.0f;
    return expf(-g);
}

// partials layout per block: [0] sum_x, [1] sum_y, [2] dot(G, disp)   (block -> one image row band)
__global__ __launch_bounds__(NT) void smooth_main_kernel(const float *__restrict__ disp,
                                                         const float *__restrict__ color,
                                                         const double *__restrict__ psum, int nchunk, int normalize,
                                                         float *__restrict__ den, int B, int h, int w,
                                                         float *__restrict__ G, double *__restrict__ part)
{
    __shared__ double s_red[3][NT / 64];
    const int b = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const float *d = disp + (size_t)b * hw;
    const float *c0 = color + (size_t)b * 3 * hw;
    const float m = block_den(psum + (size_t)b * nchunk, nchunk, (int)hw, normalize, &s_red[0][0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) den[b] = m;
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

// one block: loss = sum_x/Nx + sum_y/Ny (parallel strided sums + LDS tree); per-image dot -> dots[b]
__global__ __launch_bounds__(NT) void smooth_finish_kernel(const double *__restrict__ part, int B, int nblk,
                                                           int h, int w, float *__restrict__ loss,
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
