// ops.hip -- fine-grained gfx950 kernels behind the reference's model_layer / model_loss API.
//
// One kernel per reference function (forward and closed-form backward), each using the same pinned
// device math as the fused kernels (mdx_device.hpp):
//   interpolate        model_layer/warp.py:18-20        disparity2depth   model_layer/warp.py:29-39
//   Depth2PointCloud   model_layer/warp.py:237-246      PointCloud2Pixel  model_layer/warp.py:259-269
//   grid_sample        model_layer/warp.py:12-14        SSIM              model_loss/model_loss.py:28-41
//   ReprojectionLoss   model_loss/model_loss.py:97-103  min / auto-mask   model_tool/processor.py:194-204
// These exist for drop-in use and per-stage parity tests; the training step runs photometric.hip.
// Thread <-> x, consecutive lanes on consecutive pixels of a row: all planar accesses coalesce.
#include "mdx_common.hpp"
#include "mdx_device.hpp"

namespace mdx {

static inline dim3 grid1d(size_t n) { return dim3((unsigned)((n + NT - 1) / NT)); }

__global__ __launch_bounds__(NT) void interpolate_fwd_kernel(const float *__restrict__ x, int BC, int h, int w,
                                                             float *__restrict__ out, int H, int W, bool premul)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)BC * H * W) return;
    const int ox = (int)(i % W), oy = (int)((i / W) % H);
    const size_t bc = i / ((size_t)W * H);
    out[i] = upsample_at(x + bc * (size_t)h * w, h, w, H, W, oy, ox, premul);
}

__global__ __launch_bounds__(NT) void disp2depth_fwd_kernel(const float *__restrict__ disp, size_t n, float a,
                                                            float b, float *__restrict__ sd, float *__restrict__ depth)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const float s = scaled_disp(disp[i], a, b);
    if (sd) sd[i] = s;
    if (depth) depth[i] = 1.0f / s;
}

__global__ __launch_bounds__(NT) void disp2depth_bwd_kernel(const float *__restrict__ disp, const float *__restrict__ gsd,
                                                            const float *__restrict__ gdepth, size_t n, float a,
                                                            float b, float *__restrict__ gdisp)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const float s = scaled_disp(disp[i], a, b);
    const float dep = 1.0f / s;
    float g = 0.f;
    if (gsd) g += gsd[i] * b;
    if (gdepth) g += gdepth[i] * (-b * dep * dep);
    gdisp[i] = g;
}

__global__ __launch_bounds__(NT) void backproject_fwd_kernel(const float *__restrict__ depth, const float *__restrict__ invK,
                                                             int B, int H, int W, float *__restrict__ cam)
{
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const int b = (int)(i / HW);
    const size_t p = i % HW;
    float r[3];
    pixel_ray(invK + b * 16, (float)(p % W), (float)(p / W), r);
    const float d = depth[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) cam[((size_t)b * 4 + k) * HW + p] = d * r[k];
    cam[((size_t)b * 4 + 3) * HW + p] = 1.0f;
}

__global__ __launch_bounds__(NT) void backproject_bwd_kernel(const float *__restrict__ gcam, const float *__restrict__ invK,
                                                             int B, int H, int W, float *__restrict__ gdepth)
{
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const int b = (int)(i / HW);
    const size_t p = i % HW;
    float r[3];
    pixel_ray(invK + b * 16, (float)(p % W), (float)(p / W), r);
    float g = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) g += gcam[((size_t)b * 4 + k) * HW + p] * r[k];
    gdepth[i] = g;
}

__global__ __launch_bounds__(NT) void project_fwd_kernel(const float *__restrict__ cam, const float *__restrict__ P, int B,
                                                         int H, int W, float eps, float *__restrict__ grid, bool fw, bool fh)
{
    Norm2 nd;
    nd.w = make_normdiv(W - 1, fw);
    nd.h = make_normdiv(H - 1, fh);
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const int b = (int)(i / HW);
    const size_t p = i % HW;
    const float *c = cam + (size_t)b * 4 * HW + p;
    const Proj pr = project_point(P + b * 12, c[0], c[HW], c[2 * HW], c[3 * HW], nd, eps);
    grid[i * 2] = pr.gx;
    grid[i * 2 + 1] = pr.gy;
}

// ggrid -> gcam, and per-block partials of gP ([nblk][12], blocks never straddle images)
__global__ __launch_bounds__(NT) void project_bwd_kernel(const float *__restrict__ cam, const float *__restrict__ P,
                                                         const float *__restrict__ ggrid, int B, int H, int W,
                                                         float eps, float *__restrict__ gcam, float *__restrict__ partP)
{
    __shared__ float s_red[NT / 64][12];
    Norm2 nd;
    nd.w = make_normdiv(W - 1, false);
    nd.h = make_normdiv(H - 1, false);
    const size_t HW = (size_t)H * W;
    const int b = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * NT + threadIdx.x;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    if (p < HW) {
        const float *c = cam + (size_t)b * 4 * HW + p;
        const float *Pb = P + b * 12;
        const float X[4] = {c[0], c[HW], c[2 * HW], c[3 * HW]};
        const Proj pr = project_point(Pb, X[0], X[1], X[2], X[3], nd, eps);
        const float gu = ggrid[((size_t)b * HW + p) * 2] * (2.0f / (float)(W - 1));
        const float gv = ggrid[((size_t)b * HW + p) * 2 + 1] * (2.0f / (float)(H - 1));
        const float iz = 1.0f / pr.z;
        const float gq[3] = {gu * iz, gv * iz, -(gu * pr.u + gv * pr.v) * iz};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            gcam[((size_t)b * 4 + j) * HW + p] = gq[0] * Pb[j] + gq[1] * Pb[4 + j] + gq[2] * Pb[8 + j];
        gcam[((size_t)b * 4 + 3) * HW + p] = gq[0] * Pb[3] + gq[1] * Pb[7] + gq[2] * Pb[11];
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ii * 4 + j] = gq[ii] * X[j];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float t = 0.f;
        for (int k = 0; k < NT / 64; ++k) t += s_red[k][threadIdx.x];
        partP[((size_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = t;
    }
}

__global__ void project_finish_kernel(const float *__restrict__ partP, int B, int nblk, float *__restrict__ gP)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 12) return;
    const int b = i / 12, k = i % 12;
    double acc = 0.0;
    for (int t = 0; t < nblk; ++t) acc += (double)partP[((size_t)b * nblk + t) * 12 + k];
    gP[i] = (float)acc;
}

__global__ __launch_bounds__(NT) void grid_sample_fwd_kernel(const float *__restrict__ img, const float *__restrict__ grid,
                                                             int B, int C, int Hi, int Wi, int Ho, int Wo,
                                                             float *__restrict__ out)
{
    const size_t HWo = (size_t)Ho * Wo, HWi = (size_t)Hi * Wi;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * HWo) return;
    const int b = (int)(i / HWo);
    const size_t p = i % HWo;
    const Tap t = make_tap(grid[i * 2], grid[i * 2 + 1], Hi, Wi);
    for (int c = 0; c < C; ++c)
        out[((size_t)b * C + c) * HWo + p] = sample(load_corners(img + ((size_t)b * C + c) * HWi, Hi, Wi, t), t);
}

__global__ __launch_bounds__(NT) void grid_sample_bwd_kernel(const float *__restrict__ img, const float *__restrict__ grid,
                                                             const float *__restrict__ gout, int B, int C, int Hi,
                                                             int Wi, int Ho, int Wo, float *__restrict__ ggrid,
                                                             float *__restrict__ gimg)
{
    const size_t HWo = (size_t)Ho * Wo, HWi = (size_t)Hi * Wi;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * HWo) return;
    const int b = (int)(i / HWo);
    const size_t p = i % HWo;
    const Tap t = make_tap(grid[i * 2], grid[i * 2 + 1], Hi, Wi);
    const float x1 = (float)(t.x0 + 1), y1 = (float)(t.y0 + 1), xf0 = (float)t.x0, yf0 = (float)t.y0;
    const bool xe = t.x0 + 1 < Wi, ys = t.y0 + 1 < Hi;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float go = gout[((size_t)b * C + c) * HWo + p];
        const Corners cn = load_corners(img + ((size_t)b * C + c) * HWi, Hi, Wi, t);
        gix += go * (-cn.nw * (y1 - t.iy) + cn.ne * (y1 - t.iy) - cn.sw * (t.iy - yf0) + cn.se * (t.iy - yf0));
        giy += go * (-cn.nw * (x1 - t.ix) - cn.ne * (t.ix - xf0) + cn.sw * (x1 - t.ix) + cn.se * (t.ix - xf0));
        if (gimg) {
            float *gi = gimg + ((size_t)b * C + c) * HWi + (size_t)t.y0 * Wi + t.x0;
            atomicAdd(gi, go * t.nw);
            if (xe) atomicAdd(gi + 1, go * t.ne);
            if (ys) atomicAdd(gi + Wi, go * t.sw);
            if (xe && ys) atomicAdd(gi + Wi + 1, go * t.se);
        }
    }
    ggrid[i * 2] = t.inx ? gix * ((float)(Wi - 1) / 2.0f) : 0.f;
    ggrid[i * 2 + 1] = t.iny ? giy * ((float)(Hi - 1) / 2.0f) : 0.f;
}

MDX_DEV void load9(const float *__restrict__ img, int H, int W, int py, int px, float v[9])
{
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = img[(size_t)reflect(py + k / 3 - 1, H) * W + reflect(px + k % 3 - 1, W)];
}

__global__ __launch_bounds__(NT) void ssim_fwd_kernel(const float *__restrict__ x, const float *__restrict__ y, int BC,
                                                      int H, int W, float *__restrict__ out)
{
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)BC * HW) return;
    const size_t bc = i / HW, p = i % HW;
    float x9[9], y9[9];
    load9(x + bc * HW, H, W, (int)(p / W), (int)(p % W), x9);
    load9(y + bc * HW, H, W, (int)(p / W), (int)(p % W), y9);
    out[i] = clamp01(ssim_raw(pred_stats(x9, y9), target_stats(y9)));
}

__global__ __launch_bounds__(NT) void reprojection_fwd_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                              int B, int H, int W, float *__restrict__ out)
{
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const size_t b = i / HW, p = i % HW;
    float ss[3], ad[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float x9[9], y9[9];
        load9(pred + (b * 3 + c) * HW, H, W, (int)(p / W), (int)(p % W), x9);
        load9(target + (b * 3 + c) * HW, H, W, (int)(p / W), (int)(p % W), y9);
        ss[c] = clamp01(ssim_raw(pred_stats(x9, y9), target_stats(y9)));
        ad[c] = fabsf(y9[4] - x9[4]);
    }
    out[i] = reprojection_combine(ss, ad);
}

// gather form of the SSIM+L1 backward: every pixel q sums the contributions of the (up to 25)
// padded taps that reflect onto it.  swap = true computes the gradient wrt the TARGET instead.
// SSIM_ONLY: the backward of SSIM.forward alone (model_loss.py:28-41): the upstream gradient is per channel
// ([BC,H,W], `B` counts channel planes), no channel mean, no L1 term.
template <bool SSIM_ONLY>
__global__ __launch_bounds__(NT) void reprojection_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                              const float *__restrict__ gout, int B, int H, int W,
                                                              float *__restrict__ gres, bool swap)
{
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * (SSIM_ONLY ? 1 : 3) * HW) return;
    const size_t bc = i / HW, p = i % HW, b = SSIM_ONLY ? bc : bc / 3;
    const int qy = (int)(p / W), qx = (int)(p % W);
    const float *xs = (swap ? target : pred) + bc * HW, *ys = (swap ? pred : target) + bc * HW;
    const float *go = gout + b * HW;
    const float xq = xs[p], yq = ys[p];
    float A = 0.f, Bq = 0.f, Cq = 0.f;
    for (int dy = -1; dy <= 1; ++dy) {
        const int cy = qy + dy;
        if (cy < 0 || cy >= H) continue;
        const float wy = 1.f + ((dy == -1 && qy == 1) ? 1.f : 0.f) + ((dy == 1 && qy == H - 2) ? 1.f : 0.f);
        for (int dx = -1; dx <= 1; ++dx) {
            const int cx = qx + dx;
            if (cx < 0 || cx >= W) continue;
            const float wx = 1.f + ((dx == -1 && qx == 1) ? 1.f : 0.f) + ((dx == 1 && qx == W - 2) ? 1.f : 0.f);
            float x9[9], y9[9];
            load9(xs, H, W, cy, cx, x9);
            load9(ys, H, W, cy, cx, y9);
            const SsimGrad sg = ssim_grad(pred_stats(x9, y9), target_stats(y9),
                                          (SSIM_ONLY ? 1.0f : 0.85f / 3.0f) * go[(size_t)cy * W + cx]);
            const float wgt = wy * wx;
            A += wgt * sg.alpha; Bq += wgt * sg.beta; Cq += wgt * sg.gamma;
        }
    }
    float g = (A + 2.0f * xq * Bq + yq * Cq) * (1.0f / 9.0f);
    // L1: d|y-x|/dx = -sign(y-x); wrt y (swap) the roles exchange and the sign is the same expression
    if (!SSIM_ONLY) {
        const float sgn = (yq > xq) ? 1.f : ((yq < xq) ? -1.f : 0.f);
        g -= 0.05f * sgn * go[p];
    }
    gres[i] = g;
}

__global__ __launch_bounds__(NT) void min_automask_kernel(const float *__restrict__ ident, const float *__restrict__ noise,
                                                          const float *__restrict__ reproj, int B, int S, int H, int W,
                                                          int automask, float *__restrict__ combined,
                                                          float *__restrict__ to_opt, uint8_t *__restrict__ idx)
{
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const size_t b = i / HW, p = i % HW;
    const int C = automask ? 2 * S : S;
    float best = 0.f;
    int bi = 0;
    for (int c = 0; c < C; ++c) {
        float v;
        if (automask && c < S) {
            const float t = 1e-5f * noise[(b * S + c) * HW + p];
            v = ident[(b * S + c) * HW + p] + t;
        } else {
            v = reproj[(b * S + (automask ? c - S : c)) * HW + p];
        }
        if (combined) combined[(b * C + c) * HW + p] = v;
        if (c == 0 || v < best) { best = v; bi = c; }
    }
    to_opt[i] = best;
    idx[i] = (uint8_t)bi;
}

}  // namespace mdx

using namespace mdx;

#define MDX_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)

MDX_EXPORT int mdx_interpolate_bilinear_fwd(const float *x, int BC, int h, int w, float *out, int H, int W,
                                            void *stream)
{
    MDX_REQUIRE(x && out, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(BC > 0 && h > 0 && w > 0 && H > 0 && W > 0, MDX_ERR_BAD_SHAPE);
    const size_t n = (size_t)BC * H * W;
    hipLaunchKernelGGL(interpolate_fwd_kernel, grid1d(n), dim3(NT), 0, (hipStream_t)stream, x, BC, h, w, out, H, W,
                       (H + W <= 128));
    return check_launch();
}

static void d2d_consts(double min_depth, double max_depth, float *a, float *b)
{
    const double min_disp = 1.0 / max_depth, max_disp = 1.0 / min_depth;
    *a = (float)min_disp;
    *b = (float)(max_disp - min_disp);
}

MDX_EXPORT int mdx_disparity2depth_fwd(const float *disp, size_t n, double min_depth, double max_depth,
                                       float *sd, float *depth, void *stream)
{
    MDX_REQUIRE(disp && (sd || depth), MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(n > 0 && min_depth > 0 && max_depth > min_depth, MDX_ERR_BAD_SHAPE);
    float a, b;
    d2d_consts(min_depth, max_depth, &a, &b);
    hipLaunchKernelGGL(disp2depth_fwd_kernel, grid1d(n), dim3(NT), 0, (hipStream_t)stream, disp, n, a, b, sd, depth);
    return check_launch();
}

MDX_EXPORT int mdx_disparity2depth_bwd(const float *disp, const float *gsd, const float *gdepth, size_t n,
                                       double min_depth, double max_depth, float *gdisp, void *stream)
{
    MDX_REQUIRE(disp && gdisp && (gsd || gdepth), MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(n > 0 && min_depth > 0 && max_depth > min_depth, MDX_ERR_BAD_SHAPE);
    float a, b;
    d2d_consts(min_depth, max_depth, &a, &b);
    hipLaunchKernelGGL(disp2depth_bwd_kernel, grid1d(n), dim3(NT), 0, (hipStream_t)stream, disp, gsd, gdepth, n, a, b,
                       gdisp);
    return check_launch();
}

MDX_EXPORT int mdx_backproject_fwd(const float *depth, const float *invK, int B, int H, int W, float *cam,
                                   void *stream)
{
    MDX_REQUIRE(depth && invK && cam, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && H > 0 && W > 0, MDX_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(backproject_fwd_kernel, grid1d((size_t)B * H * W), dim3(NT), 0, (hipStream_t)stream, depth,
                       invK, B, H, W, cam);
    return check_launch();
}

MDX_EXPORT int mdx_backproject_bwd(const float *gcam, const float *invK, int B, int H, int W, float *gdepth,
                                   void *stream)
{
    MDX_REQUIRE(gcam && invK && gdepth, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && H > 0 && W > 0, MDX_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(backproject_bwd_kernel, grid1d((size_t)B * H * W), dim3(NT), 0, (hipStream_t)stream, gcam,
                       invK, B, H, W, gdepth);
    return check_launch();
}

MDX_EXPORT size_t mdx_project_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * (((size_t)H * W + NT - 1) / NT) * 12 * sizeof(float);
}

MDX_EXPORT int mdx_project_fwd(const float *cam, const float *P, int B, int H, int W, float eps, float *grid,
                               void *stream)
{
    MDX_REQUIRE(cam && P && grid, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && H > 1 && W > 1, MDX_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(project_fwd_kernel, grid1d((size_t)B * H * W), dim3(NT), 0, (hipStream_t)stream, cam, P, B, H,
                       W, eps, grid, div_verified(W - 1), div_verified(H - 1));
    return check_launch();
}

MDX_EXPORT int mdx_project_bwd(const float *cam, const float *P, const float *ggrid, int B, int H, int W,
                               float eps, float *gcam, float *gP, void *workspace, size_t workspace_bytes,
                               void *stream)
{
    MDX_REQUIRE(cam && P && ggrid && gcam && gP, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && H > 1 && W > 1, MDX_ERR_BAD_SHAPE);
    MDX_REQUIRE(workspace && workspace_bytes >= mdx_project_workspace_bytes(B, H, W), MDX_ERR_WORKSPACE);
    const int nblk = (int)(((size_t)H * W + NT - 1) / NT);
    hipLaunchKernelGGL(project_bwd_kernel, dim3(nblk, B), dim3(NT), 0, (hipStream_t)stream, cam, P, ggrid, B, H, W,
                       eps, gcam, (float *)workspace);
    hipLaunchKernelGGL(project_finish_kernel, dim3((B * 12 + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       (const float *)workspace, B, nblk, gP);
    return check_launch();
}

MDX_EXPORT int mdx_grid_sample_border_fwd(const float *img, const float *grid, int B, int C, int Hi, int Wi,
                                          int Ho, int Wo, float *out, void *stream)
{
    MDX_REQUIRE(img && grid && out, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, MDX_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(grid_sample_fwd_kernel, grid1d((size_t)B * Ho * Wo), dim3(NT), 0, (hipStream_t)stream, img,
                       grid, B, C, Hi, Wi, Ho, Wo, out);
    return check_launch();
}

MDX_EXPORT int mdx_grid_sample_border_bwd(const float *img, const float *grid, const float *gout, int B, int C,
                                          int Hi, int Wi, int Ho, int Wo, float *ggrid, float *gimg,
                                          void *stream)
{
    MDX_REQUIRE(img && grid && gout && ggrid, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, MDX_ERR_BAD_SHAPE);
    if (gimg && hipMemsetAsync(gimg, 0, sizeof(float) * (size_t)B * C * Hi * Wi, (hipStream_t)stream) != hipSuccess)
        return MDX_ERR_LAUNCH;
    hipLaunchKernelGGL(grid_sample_bwd_kernel, grid1d((size_t)B * Ho * Wo), dim3(NT), 0, (hipStream_t)stream, img,
                       grid, gout, B, C, Hi, Wi, Ho, Wo, ggrid, gimg);
    return check_launch();
}

MDX_EXPORT int mdx_ssim_fwd(const float *x, const float *y, int BC, int H, int W, float *out, void *stream)
{
    MDX_REQUIRE(x && y && out, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(BC > 0 && H >= 2 && W >= 2, MDX_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid1d((size_t)BC * H * W), dim3(NT), 0, (hipStream_t)stream, x, y, BC, H, W,
                       out);
    return check_launch();
}

MDX_EXPORT int mdx_reprojection_loss_fwd(const float *pred, const float *target, int B, int H, int W,
                                         float *out, void *stream)
{
    MDX_REQUIRE(pred && target && out, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && H >= 2 && W >= 2, MDX_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(reprojection_fwd_kernel, grid1d((size_t)B * H * W), dim3(NT), 0, (hipStream_t)stream, pred,
                       target, B, H, W, out);
    return check_launch();
}

MDX_EXPORT int mdx_reprojection_loss_bwd(const float *pred, const float *target, const float *gout, int B,
                                         int H, int W, float *gpred, float *gtarget, void *stream)
{
    MDX_REQUIRE(pred && target && gout && (gpred || gtarget), MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && H >= 4 && W >= 4, MDX_ERR_BAD_SHAPE);
    const size_t n = (size_t)B * 3 * H * W;
    if (gpred)
        hipLaunchKernelGGL(reprojection_bwd_kernel<false>, grid1d(n), dim3(NT), 0, (hipStream_t)stream, pred, target, gout,
                           B, H, W, gpred, false);
    if (gtarget)
        hipLaunchKernelGGL(reprojection_bwd_kernel<false>, grid1d(n), dim3(NT), 0, (hipStream_t)stream, pred, target, gout,
                           B, H, W, gtarget, true);
    return check_launch();
}

MDX_EXPORT int mdx_ssim_bwd(const float *x, const float *y, const float *gout, int BC, int H, int W, float *gx, float *gy,
                            void *stream)
{
    MDX_REQUIRE(x && y && gout && (gx || gy), MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(BC > 0 && H >= 4 && W >= 4, MDX_ERR_BAD_SHAPE);
    const size_t n = (size_t)BC * H * W;
    if (gx)
        hipLaunchKernelGGL(reprojection_bwd_kernel<true>, grid1d(n), dim3(NT), 0, (hipStream_t)stream, x, y, gout, BC, H, W,
                           gx, false);
    if (gy)
        hipLaunchKernelGGL(reprojection_bwd_kernel<true>, grid1d(n), dim3(NT), 0, (hipStream_t)stream, x, y, gout, BC, H, W,
                           gy, true);
    return check_launch();
}

MDX_EXPORT int mdx_min_automask_fwd(const float *ident, const float *noise, const float *reproj, int B, int S,
                                    int H, int W, int automask, float *combined, float *to_opt, uint8_t *idx,
                                    void *stream)
{
    MDX_REQUIRE(reproj && to_opt && idx, MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(!automask || (ident && noise), MDX_ERR_NULL_POINTER);
    MDX_REQUIRE(B > 0 && S >= 1 && S <= MDX_MAX_SRC && H > 0 && W > 0, MDX_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(min_automask_kernel, grid1d((size_t)B * H * W), dim3(NT), 0, (hipStream_t)stream, ident, noise,
                       reproj, B, S, H, W, automask, combined, to_opt, idx);
    return check_launch();
}
