// loss_total.hip -- the scalar tail of the training loss and its gradient fan-out (processor.py:208-217), gfx950.
//
// After the per-scale kernels the reference finishes the loss with ~5 scalar torch ops per scale
//     scale_loss = to_optimise.mean() + disp_smoothness * smooth_loss / 2**scale;  total += scale_loss;  total /= len(scales)
// and autograd walks the same ops back: ~20 launches forward, ~30 backward plus three whole-map passes per scale (the photometric
// gradient times its scalar, the smoothness gradient times its scalar, their sum) -- all of it on the one stretch of a step where
// neither network can run.  Here: ONE single-thread launch forward, ONE launch backward that writes every scale's disparity
// gradient  gd_photo * c1 + gd_smooth * c2[s]  and the projection gradient.  Every product and sum is rounded where ATen rounds it
// (a division by a Python scalar is a multiplication by its float32 reciprocal on the GPU: BinaryDivTrueKernel.cu), so the loss and
// the disparity gradients are the op-by-op path's bit for bit.
#include "mdx_common.hpp"

namespace mdx {

struct TotalCoef {
    float r_pow[MDX_MAX_SCALES];      // 1 / 2**scale
    float r_pix, r_nsc, lambda;
};

struct TotalJobs {
    const float *photo[MDX_MAX_SCALES];
    const float *smooth[MDX_MAX_SCALES];
    float *out[MDX_MAX_SCALES];
    long long count[MDX_MAX_SCALES];
    int block0[MDX_MAX_SCALES];       // first block of every scale
};

__global__ void loss_total_fwd_kernel(int nsc, const float *__restrict__ sums, const float *__restrict__ smooth, TotalCoef c,
                                      float *__restrict__ total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < MDX_MAX_SCALES; ++k) {
        if (k < nsc) {
            const float mean = sums[k] * c.r_pix;                       // to_optimise.mean()
            const float sm = (c.lambda * smooth[k]) * c.r_pow[k];       // disp_smoothness * smooth_loss / 2**scale
            const float l = mean + sm;
            t = (k == 0) ? l : t + l;                                   // 0 + l is l
        }
    }
    total[0] = t * c.r_nsc;
}

__global__ __launch_bounds__(NT) void loss_total_bwd_kernel(TotalJobs J, int nsc, int pblock, const float *__restrict__ g_total, TotalCoef c,
                                                            const float *__restrict__ gP, int nP, int per_scale_P,
                                                            float *__restrict__ gP_out)
{
    const float g1 = g_total[0] * c.r_nsc;        // through total / len(scales)
    const float c1 = g1 * c.r_pix;                // through sum / (B*H*W)
    const int blk = blockIdx.x;
    if (blk >= pblock) {                          // the projection gradient: [nscales][nP] -> per scale, or summed over the scales
        const int i = (blk - pblock) * NT + threadIdx.x;
        if (per_scale_P) {
            if (i < nsc * nP) gP_out[i] = gP[i] * c1;
        } else if (i < nP) {
            float acc = gP[i] * c1;
            for (int s = 1; s < nsc; ++s) acc = acc + gP[(size_t)s * nP + i] * c1;
            gP_out[i] = acc;
        }
        return;
    }
    int s = 0;
#pragma unroll
    for (int k = 1; k < MDX_MAX_SCALES; ++k)
        if (k < nsc && blk >= J.block0[k]) s = k;
    const float *__restrict__ a = J.photo[0];
    const float *__restrict__ b = J.smooth[0];
    float *__restrict__ o = J.out[0];
    long long n = J.count[0];
    int first = J.block0[0];
    float rp = c.r_pow[0];
#pragma unroll
    for (int k = 1; k < MDX_MAX_SCALES; ++k)
        if (s == k) { a = J.photo[k]; b = J.smooth[k]; o = J.out[k]; n = J.count[k]; first = J.block0[k]; rp = c.r_pow[k]; }
    const float c2 = (g1 * rp) * c.lambda;        // through (lambda * smooth) / 2**scale
    const long long i = ((long long)(blk - first) * NT + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 x = *reinterpret_cast<const float4 *>(a + i);
        const float4 y = *reinterpret_cast<const float4 *>(b + i);
        float4 r;
        r.x = x.x * c1 + y.x * c2; r.y = x.y * c1 + y.y * c2; r.z = x.z * c1 + y.z * c2; r.w = x.w * c1 + y.w * c2;
        *reinterpret_cast<float4 *>(o + i) = r;
    } else {
        for (long long j = i; j < n; ++j) o[j] = a[j] * c1 + b[j] * c2;
    }
}

static int fill_coef(int nsc, const int32_t *scale, long long pixels, double disp_smoothness, TotalCoef *c)
{
    if (nsc < 1 || nsc > MDX_MAX_SCALES || pixels <= 0) return MDX_ERR_BAD_SHAPE;
    if (!scale) return MDX_ERR_NULL_POINTER;
    for (int k = 0; k < nsc; ++k) {
        if (scale[k] < 0 || scale[k] > 30) return MDX_ERR_BAD_SHAPE;
        c->r_pow[k] = 1.0f / (float)(1 << scale[k]);
    }
    for (int k = nsc; k < MDX_MAX_SCALES; ++k) c->r_pow[k] = 0.f;
    c->r_pix = 1.0f / (float)(double)pixels;      // float(B*H*W) -> opmath float -> reciprocal
    c->r_nsc = 1.0f / (float)nsc;
    c->lambda = (float)disp_smoothness;
    return MDX_OK;
}

}  // namespace mdx

using namespace mdx;

MDX_EXPORT int mdx_loss_total_fwd(int nscales, const float *sums, const float *smooth, const int32_t *scale, int64_t pixels,
                                  double disp_smoothness, float *total, void *stream)
{
    TotalCoef c;
    int rc = fill_coef(nscales, scale, pixels, disp_smoothness, &c);
    if (rc) return rc;
    if (!sums || !smooth || !total) return MDX_ERR_NULL_POINTER;
    hipLaunchKernelGGL(loss_total_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nscales, sums, smooth, c, total);
    return check_launch();
}

MDX_EXPORT int mdx_loss_total_bwd(int nscales, const float *g_total, const int32_t *scale, int64_t pixels, double disp_smoothness,
                                  const float *const *gd_photo, const float *const *gd_smooth, const int64_t *count,
                                  float *const *gdisp, const float *gP, int nP, int per_scale_P, float *gP_out, void *stream)
{
    TotalCoef c;
    int rc = fill_coef(nscales, scale, pixels, disp_smoothness, &c);
    if (rc) return rc;
    if (!g_total || !gd_photo || !gd_smooth || !count || !gdisp) return MDX_ERR_NULL_POINTER;
    if ((gP == nullptr) != (gP_out == nullptr) || (gP && nP <= 0)) return MDX_ERR_BAD_SHAPE;
    TotalJobs J;
    int blocks = 0;
    for (int k = 0; k < MDX_MAX_SCALES; ++k) {
        J.block0[k] = blocks;
        if (k < nscales) {
            if (!gd_photo[k] || !gd_smooth[k] || !gdisp[k]) return MDX_ERR_NULL_POINTER;
            if (count[k] <= 0 || count[k] >= (1ll << 40)) return MDX_ERR_BAD_SHAPE;
            if (!aligned(gd_photo[k], 16) || !aligned(gd_smooth[k], 16) || !aligned(gdisp[k], 16)) return MDX_ERR_MISALIGNED;
            J.photo[k] = gd_photo[k]; J.smooth[k] = gd_smooth[k]; J.out[k] = gdisp[k]; J.count[k] = count[k];
            blocks += (int)((count[k] + 4 * NT - 1) / (4 * NT));
        } else {
            J.photo[k] = nullptr; J.smooth[k] = nullptr; J.out[k] = nullptr; J.count[k] = 0;
        }
    }
    const int pblock = blocks;                    // first block of the projection part
    if (gP) blocks += ((per_scale_P ? nscales * nP : nP) + NT - 1) / NT;
    hipLaunchKernelGGL(loss_total_bwd_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, J, nscales, pblock, g_total, c, gP, nP,
                       per_scale_P ? 1 : 0, gP_out);
    return check_launch();
}
