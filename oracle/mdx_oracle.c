/*
 * mdx_oracle.c -- CPU restatement of the reference's photometric hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link or call this file:
 * it is the checker used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  Every function below is checked against golden vectors produced by
 * running the reference's own Python modules on CPU (tests/golden/make_golden.py, which imports
 * /root/reference/model_layer/warp.py, model_loss/model_loss.py, model_tool/processor.py).
 * Per-pixel tensors and arg-min indices are reproduced bit-for-bit; scalar reductions and
 * gradients to <= 1e-4 relative (tests/test_oracle_vs_golden.py).
 *
 * The arithmetic that actually executes for the reference lives in PyTorch ATen (torch 2.10.0 CPU,
 * MKL sgemm, AVX-512 kernels).  The operation ORDER restated here (where an FMA is fused, where it
 * is not, sequential 9-tap sums, true IEEE divides) is the order that reproduces those kernels
 * bit-for-bit.  Compile with -ffp-contract=off: every fused multiply-add is an explicit fmaf().
 *
 * All tensors are contiguous float32 NCHW.  Sizes: B images, full resolution H x W, disparity at
 * h x w (h = H >> scale), S source frames.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * A1  interpolate(disp, H, W, "bilinear", align_corners=False)
 * reference: model_layer/warp.py:18-20, called at model_tool/processor.py:142
 * ATen upsample_bilinear2d: src = max(scale*(dst+0.5)-0.5, 0); i1 = min(i0+1, in-1).
 * ATen picks one of two CPU kernels by OUTPUT size (UpSampleKernel.cpp,
 * _use_vectorized_kernel_cond_2d): when H + W <= 128 the pre-multiplied-weights kernel
 *     w_ij = ly_i*lx_j;  out = fma(w11,v11, fma(w10,v10, fma(w00,v00, w01*v01)))
 * otherwise (every real training size, e.g. 192x640) the generic separable kernel
 *     out = fma(ly0, fma(lx0,v00, lx1*v01), ly1*fma(lx0,v10, lx1*v11)).
 * Both restated; bit-exact for the power-of-two ratios the path produces (h = H >> scale).
 * ---------------------------------------------------------------------------------------- */
static inline void orc_src_index(float scale, int dst, int in_size, int *i0, int *i1, float *l0,
                                 float *l1)
{
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    int a = (int)src;
    if (a > in_size - 1) a = in_size - 1;
    *i0 = a;
    *i1 = a + ((a < in_size - 1) ? 1 : 0);
    *l1 = src - (float)a;
    *l0 = 1.f - *l1;
}

ORC_API void orc_upsample_bilinear(const float *in, int BC, int h, int w, float *out, int H, int W)
{
    float sy = (float)h / (float)H, sx = (float)w / (float)W;
    if (h == H && w == W) { memcpy(out, in, sizeof(float) * (size_t)BC * H * W); return; }
#pragma omp parallel for
    for (int bc = 0; bc < BC; ++bc) {
        const float *p = in + (size_t)bc * h * w;
        float *o = out + (size_t)bc * H * W;
        for (int y = 0; y < H; ++y) {
            int y0, y1; float ly0, ly1;
            orc_src_index(sy, y, h, &y0, &y1, &ly0, &ly1);
            for (int x = 0; x < W; ++x) {
                int x0, x1; float lx0, lx1;
                orc_src_index(sx, x, w, &x0, &x1, &lx0, &lx1);
                if (H + W <= 128) {
                    float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
                    float acc = w01 * p[y0 * w + x1];
                    acc = fmaf(w00, p[y0 * w + x0], acc);
                    acc = fmaf(w10, p[y1 * w + x0], acc);
                    o[y * W + x] = fmaf(w11, p[y1 * w + x1], acc);
                } else {
                    float top = fmaf(lx0, p[y0 * w + x0], lx1 * p[y0 * w + x1]);
                    float bot = fmaf(lx0, p[y1 * w + x0], lx1 * p[y1 * w + x1]);
                    o[y * W + x] = fmaf(ly0, top, ly1 * bot);
                }
            }
        }
    }
}

/* transpose of A1 (autograd of upsample_bilinear2d): scatter-add with the same weights */
ORC_API void orc_upsample_bilinear_bwd(const float *gout, int BC, int H, int W, float *gin, int h,
                                       int w)
{
    float sy = (float)h / (float)H, sx = (float)w / (float)W;
    if (h == H && w == W) { memcpy(gin, gout, sizeof(float) * (size_t)BC * H * W); return; }
#pragma omp parallel for
    for (int bc = 0; bc < BC; ++bc) {
        const float *g = gout + (size_t)bc * H * W;
        float *o = gin + (size_t)bc * h * w;
        double *acc = (double *)calloc((size_t)h * w, sizeof(double));
        for (int y = 0; y < H; ++y) {
            int y0, y1; float ly0, ly1;
            orc_src_index(sy, y, h, &y0, &y1, &ly0, &ly1);
            for (int x = 0; x < W; ++x) {
                int x0, x1; float lx0, lx1;
                orc_src_index(sx, x, w, &x0, &x1, &lx0, &lx1);
                double v = g[y * W + x];
                acc[y0 * w + x0] += v * ly0 * lx0;
                acc[y0 * w + x1] += v * ly0 * lx1;
                acc[y1 * w + x0] += v * ly1 * lx0;
                acc[y1 * w + x1] += v * ly1 * lx1;
            }
        }
        for (int i = 0; i < h * w; ++i) o[i] = (float)acc[i];
        free(acc);
    }
}

/* ------------------------------------------------------------------------------------------
 * A2  disparity2depth(disp, min_depth, max_depth) -> (scaled_disp, depth)
 * reference: model_layer/warp.py:29-39.  Constants are formed in Python double and rounded to
 * f32 when they meet the tensor: sd = f32(min_disp) + f32(max_disp-min_disp)*disp (mul, add --
 * two ATen ops, no FMA); depth = 1/sd (IEEE divide).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_disp2depth_consts(double min_depth, double max_depth, float *a, float *b)
{
    double min_disp = 1.0 / max_depth, max_disp = 1.0 / min_depth;
    *a = (float)min_disp;
    *b = (float)(max_disp - min_disp);
}

ORC_API void orc_disparity2depth(const float *disp, size_t n, double min_depth, double max_depth,
                                 float *sd, float *depth)
{
    float a, b;
    orc_disp2depth_consts(min_depth, max_depth, &a, &b);
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) {
        float t = b * disp[i];
        float s = a + t;
        if (sd) sd[i] = s;
        if (depth) depth[i] = 1.0f / s;
    }
}

/* ------------------------------------------------------------------------------------------
 * P = (K @ T)[:, :3, :]   reference: model_layer/warp.py:260.  A 4x4x4 bmm is below ATen's
 * small-matrix threshold (contraction*rows*cols < 400) and runs its naive baddbmm kernel, built
 * without FMA: acc = 0; acc += K[i][k]*T[k][j] for k = 0..3 (separate mul and add roundings).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_compose_projection(const float *K, const float *T, int B, float *P)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) {
                const float *k = K + b * 16 + i * 4, *t = T + b * 16 + j;
                float acc = 0.0f;
                for (int kk = 0; kk < 4; ++kk) {
                    float prod = k[kk] * t[kk * 4];
                    acc = acc + prod;
                }
                P[b * 12 + i * 4 + j] = acc;
            }
}

/* grad_T = K[:3,:]^T @ grad_P   (autograd of the matmul above; K carries no grad) */
ORC_API void orc_compose_projection_bwd(const float *K, const float *gP, int B, float *gT)
{
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                double acc = 0;
                for (int i = 0; i < 3; ++i) acc += (double)K[b * 16 + i * 4 + r] * gP[b * 12 + i * 4 + c];
                gT[b * 16 + r * 4 + c] = (float)acc;
            }
}

/* ------------------------------------------------------------------------------------------
 * A4  Depth2PointCloud.forward   reference: model_layer/warp.py:237-246
 * r_i = fma(a_i2, 1, fma(a_i1, y, a_i0*x));  X_i = depth * r_i;  X_3 = 1
 * ---------------------------------------------------------------------------------------- */
static inline void orc_ray(const float *invK, float x, float y, float r[3])
{
    for (int i = 0; i < 3; ++i) {
        const float *a = invK + i * 4;
        r[i] = fmaf(a[2], 1.0f, fmaf(a[1], y, a[0] * x));
    }
}

ORC_API void orc_backproject(const float *depth, const float *invK, int B, int H, int W, float *cam)
{
    size_t HW = (size_t)H * W;
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float r[3];
                orc_ray(invK + b * 16, (float)x, (float)y, r);
                float d = depth[b * HW + (size_t)y * W + x];
                for (int i = 0; i < 3; ++i) cam[((size_t)b * 4 + i) * HW + (size_t)y * W + x] = d * r[i];
                cam[((size_t)b * 4 + 3) * HW + (size_t)y * W + x] = 1.0f;
            }
}

/* ------------------------------------------------------------------------------------------
 * A5  PointCloud2Pixel.forward   reference: model_layer/warp.py:259-269
 * q_i = fma(P_i3,X3, fma(P_i2,X2, fma(P_i1,X1, P_i0*X0)));  z = q2 + 1e-7f;
 * u = q0/z; v = q1/z; gx = (u/(W-1) - 0.5f)*2f; gy likewise with H-1.  grid layout [B,H,W,2].
 * ---------------------------------------------------------------------------------------- */
static inline void orc_project_point(const float *P, const float X[4], int H, int W, float eps,
                                     float *gx, float *gy, float *u_, float *v_, float *z_)
{
    float q[3];
    for (int i = 0; i < 3; ++i) {
        const float *p = P + i * 4;
        q[i] = fmaf(p[3], X[3], fmaf(p[2], X[2], fmaf(p[1], X[1], p[0] * X[0])));
    }
    float z = q[2] + eps;
    float u = q[0] / z, v = q[1] / z;
    if (u_) { *u_ = u; *v_ = v; *z_ = z; }
    float nx = u / (float)(W - 1), ny = v / (float)(H - 1);
    *gx = (nx - 0.5f) * 2.0f;
    *gy = (ny - 0.5f) * 2.0f;
}

ORC_API void orc_project(const float *cam, const float *P, int B, int H, int W, float *grid)
{
    size_t HW = (size_t)H * W;
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < HW; ++i) {
            float X[4];
            for (int k = 0; k < 4; ++k) X[k] = cam[((size_t)b * 4 + k) * HW + i];
            orc_project_point(P + b * 12, X, H, W, 1e-7f, &grid[((size_t)b * HW + i) * 2],
                              &grid[((size_t)b * HW + i) * 2 + 1], 0, 0, 0);
        }
}

/* ------------------------------------------------------------------------------------------
 * A6  F.grid_sample(img, grid, bilinear, padding_mode="border", align_corners=True)
 * reference: model_layer/warp.py:12-14, called at processor.py:161-162
 * ATen vectorised CPU kernel: ix=(gx+1)*((W-1)/2); clip; floor; weights nw=s*e, ne=s*w, sw=n*e,
 * se=n*w; out = fma(se_v,se, fma(sw_v,sw, fma(ne_v,ne, nw_v*nw))); out-of-range corner reads 0.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int x0, y0;               /* north-west corner */
    float nw, ne, sw, se;     /* bilinear weights */
    float ix, iy;             /* clipped sample position */
    int inx, iny;             /* 1 when the unclipped coordinate is strictly inside (grad passes) */
} orc_tap;

static inline void orc_make_tap(float gx, float gy, int H, int W, orc_tap *t)
{
    float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f);
    float iy = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
    t->inx = (ix > 0.f && ix < (float)(W - 1));
    t->iny = (iy > 0.f && iy < (float)(H - 1));
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    float xw = floorf(ix), yn = floorf(iy);
    float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
    t->nw = s * e; t->ne = s * w; t->sw = n * e; t->se = n * w;
    t->x0 = (int)xw; t->y0 = (int)yn; t->ix = ix; t->iy = iy;
}

static inline float orc_at(const float *img, int H, int W, int y, int x)
{
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.0f;
}

static inline float orc_sample(const float *img, int H, int W, const orc_tap *t)
{
    float nwv = orc_at(img, H, W, t->y0, t->x0), nev = orc_at(img, H, W, t->y0, t->x0 + 1);
    float swv = orc_at(img, H, W, t->y0 + 1, t->x0), sev = orc_at(img, H, W, t->y0 + 1, t->x0 + 1);
    return fmaf(sev, t->se, fmaf(swv, t->sw, fmaf(nev, t->ne, nwv * t->nw)));
}

/* img [B,C,Hi,Wi], grid [B,Ho,Wo,2] -> out [B,C,Ho,Wo] */
ORC_API void orc_grid_sample(const float *img, const float *grid, int B, int C, int Hi, int Wi,
                             int Ho, int Wo, float *out)
{
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                const float *g = grid + (((size_t)b * Ho + y) * Wo + x) * 2;
                orc_tap t;
                orc_make_tap(g[0], g[1], Hi, Wi, &t);
                for (int c = 0; c < C; ++c)
                    out[(((size_t)b * C + c) * Ho + y) * Wo + x] =
                        orc_sample(img + ((size_t)b * C + c) * Hi * Wi, Hi, Wi, &t);
            }
}

/* autograd of A6 wrt grid (ATen grid_sampler_2d_backward, border padding, align_corners=True) */
ORC_API void orc_grid_sample_bwd_grid(const float *img, const float *grid, const float *gout, int B,
                                      int C, int Hi, int Wi, int Ho, int Wo, float *ggrid)
{
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                size_t gi = (((size_t)b * Ho + y) * Wo + x) * 2;
                orc_tap t;
                orc_make_tap(grid[gi], grid[gi + 1], Hi, Wi, &t);
                float x1 = (float)(t.x0 + 1), y1 = (float)(t.y0 + 1), x0 = (float)t.x0, y0 = (float)t.y0;
                double gix = 0, giy = 0;
                for (int c = 0; c < C; ++c) {
                    const float *im = img + ((size_t)b * C + c) * Hi * Wi;
                    double go = gout[(((size_t)b * C + c) * Ho + y) * Wo + x];
                    double nwv = orc_at(im, Hi, Wi, t.y0, t.x0), nev = orc_at(im, Hi, Wi, t.y0, t.x0 + 1);
                    double swv = orc_at(im, Hi, Wi, t.y0 + 1, t.x0), sev = orc_at(im, Hi, Wi, t.y0 + 1, t.x0 + 1);
                    gix += go * (-nwv * (y1 - t.iy) + nev * (y1 - t.iy) - swv * (t.iy - y0) + sev * (t.iy - y0));
                    giy += go * (-nwv * (x1 - t.ix) - nev * (t.ix - x0) + swv * (x1 - t.ix) + sev * (t.ix - x0));
                }
                ggrid[gi] = (float)(t.inx ? gix * ((double)(Wi - 1) / 2.0) : 0.0);
                ggrid[gi + 1] = (float)(t.iny ? giy * ((double)(Hi - 1) / 2.0) : 0.0);
            }
}

/* autograd of A6 wrt the image (scatter-add of the four weights) */
ORC_API void orc_grid_sample_bwd_img(const float *grid, const float *gout, int B, int C, int Hi, int Wi,
                                     int Ho, int Wo, float *gimg)
{
    memset(gimg, 0, sizeof(float) * (size_t)B * C * Hi * Wi);
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                size_t gi = (((size_t)b * Ho + y) * Wo + x) * 2;
                orc_tap t;
                orc_make_tap(grid[gi], grid[gi + 1], Hi, Wi, &t);
                for (int c = 0; c < C; ++c) {
                    float *im = gimg + ((size_t)b * C + c) * Hi * Wi;
                    float go = gout[(((size_t)b * C + c) * Ho + y) * Wo + x];
                    int xs[2] = {t.x0, t.x0 + 1}, ys[2] = {t.y0, t.y0 + 1};
                    float ws[4] = {t.nw, t.ne, t.sw, t.se};
                    for (int k = 0; k < 4; ++k) {
                        int xx = xs[k & 1], yy = ys[k >> 1];
                        if (xx >= 0 && xx < Wi && yy >= 0 && yy < Hi) im[(size_t)yy * Wi + xx] += go * ws[k];
                    }
                }
            }
}

/* ------------------------------------------------------------------------------------------
 * A7  SSIM.forward   reference: model_loss/model_loss.py:28-41
 * A8  ReprojectionLoss.forward   reference: model_loss/model_loss.py:97-103
 * ReflectionPad2d(1); AvgPool2d(3,1) = row-major sequential sum of the 9 taps, then /9;
 * every product/sum is its own f32 op (separate ATen kernels: no FMA); IEEE divides;
 * channel mean ((c0+c1)+c2)/3; 0.85f*ssim + 0.15f*l1.
 * ---------------------------------------------------------------------------------------- */
static inline int orc_reflect(int i, int n)
{
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

typedef struct { float mu_x, mu_y, ex2, ey2, exy; } orc_stats;

static inline void orc_window_stats(const float *x, const float *y, int H, int W, int py, int px,
                                    orc_stats *s)
{
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
    int first = 1;
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = orc_reflect(py + dy, H);
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = orc_reflect(px + dx, W);
            float a = x[(size_t)yy * W + xx], b = y[(size_t)yy * W + xx];
            float aa = a * a, bb = b * b, ab = a * b;
            if (first) { sx = a; sy = b; sxx = aa; syy = bb; sxy = ab; first = 0; }
            else { sx += a; sy += b; sxx += aa; syy += bb; sxy += ab; }
        }
    }
    s->mu_x = sx / 9.0f; s->mu_y = sy / 9.0f;
    s->ex2 = sxx / 9.0f; s->ey2 = syy / 9.0f; s->exy = sxy / 9.0f;
}

#define ORC_C1 ((float)0.0001) /* 0.01 ** 2 (model_loss.py:25) */
#define ORC_C2 ((float)0.0009) /* 0.03 ** 2 (model_loss.py:26) */

static inline float orc_ssim_from_stats(const orc_stats *s, float *raw_out)
{
    float mxx = s->mu_x * s->mu_x, myy = s->mu_y * s->mu_y, mxy = s->mu_x * s->mu_y;
    float sig_x = s->ex2 - mxx, sig_y = s->ey2 - myy, sig_xy = s->exy - mxy;
    float t = 2.0f * s->mu_x;
    t = t * s->mu_y;
    float A1 = t + ORC_C1;
    float A2 = 2.0f * sig_xy + ORC_C2;   /* mul, add: separate ATen ops */
    float n = A1 * A2;
    float B1 = (mxx + myy) + ORC_C1;
    float B2 = (sig_x + sig_y) + ORC_C2;
    float d = B1 * B2;
    float raw = (1.0f - n / d) / 2.0f;
    if (raw_out) *raw_out = raw;
    return fminf(fmaxf(raw, 0.0f), 1.0f);
}

/* SSIM map [B,3,H,W] (model_loss.py:28-41) */
ORC_API void orc_ssim(const float *x, const float *y, int BC, int H, int W, float *out)
{
#pragma omp parallel for
    for (int bc = 0; bc < BC; ++bc) {
        const float *xp = x + (size_t)bc * H * W, *yp = y + (size_t)bc * H * W;
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                orc_stats s;
                orc_window_stats(xp, yp, H, W, py, px, &s);
                out[(size_t)bc * H * W + (size_t)py * W + px] = orc_ssim_from_stats(&s, 0);
            }
    }
}

/* ReprojectionLoss(pred, target) -> [B,1,H,W] */
ORC_API void orc_reprojection_loss(const float *pred, const float *target, int B, int H, int W,
                                   float *out)
{
    size_t HW = (size_t)H * W;
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                float l1 = 0.f, ss = 0.f;
                for (int c = 0; c < 3; ++c) {
                    const float *xp = pred + ((size_t)b * 3 + c) * HW, *yp = target + ((size_t)b * 3 + c) * HW;
                    orc_stats s;
                    orc_window_stats(xp, yp, H, W, py, px, &s);
                    float v = orc_ssim_from_stats(&s, 0);
                    float ad = fabsf(yp[(size_t)py * W + px] - xp[(size_t)py * W + px]);
                    if (c == 0) { l1 = ad; ss = v; } else { l1 += ad; ss += v; }
                }
                l1 = l1 / 3.0f;
                ss = ss / 3.0f;
                float a = 0.85f * ss, bb = 0.15f * l1;
                out[(size_t)b * HW + (size_t)py * W + px] = a + bb;
            }
}

/* autograd of ReprojectionLoss wrt prediction (and optionally target): closed form.
 * gout [B,1,H,W]; gpred/gtarg [B,3,H,W] (either may be NULL). */
ORC_API void orc_reprojection_loss_bwd(const float *pred, const float *target, const float *gout,
                                       int B, int H, int W, float *gpred, float *gtarg)
{
    size_t HW = (size_t)H * W;
    int Hp = H + 2, Wp = W + 2;
#pragma omp parallel for
    for (int bc = 0; bc < B * 3; ++bc) {
        int b = bc / 3;
        const float *xp = pred + (size_t)bc * HW, *yp = target + (size_t)bc * HW;
        const float *go = gout + (size_t)b * HW;
        /* gradient on the padded grids, folded back afterwards */
        double *gxp = (double *)calloc((size_t)Hp * Wp, sizeof(double));
        double *gyp = (double *)calloc((size_t)Hp * Wp, sizeof(double));
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                double g0 = go[(size_t)py * W + px];
                orc_stats s; float raw;
                orc_window_stats(xp, yp, H, W, py, px, &s);
                orc_ssim_from_stats(&s, &raw);
                /* L1 term: 0.15 * mean_c |y - x| */
                double xv = xp[(size_t)py * W + px], yv = yp[(size_t)py * W + px];
                double sg = (yv > xv) - (yv < xv);
                gxp[(size_t)(py + 1) * Wp + px + 1] += -0.15 / 3.0 * sg * g0;
                gyp[(size_t)(py + 1) * Wp + px + 1] += 0.15 / 3.0 * sg * g0;
                if (!(raw >= 0.f && raw <= 1.f)) continue;   /* clamp passes grad on [0,1] */
                double g = 0.85 / 3.0 * g0;
                double mx = s.mu_x, my = s.mu_y;
                double sgx = (double)s.ex2 - mx * mx, sgy = (double)s.ey2 - my * my, sgxy = (double)s.exy - mx * my;
                double A1 = 2 * mx * my + ORC_C1, A2 = 2 * sgxy + ORC_C2;
                double B1 = mx * mx + my * my + ORC_C1, B2 = sgx + sgy + ORC_C2;
                double n = A1 * A2, d = B1 * B2;
                double Ln = -1.0 / (2 * d), Ld = n / (2 * d * d);
                double dA1 = Ln * A2, dA2 = Ln * A1, dB1 = Ld * B2, dB2 = Ld * B1;
                double al_x = g * 2 * (my * (dA1 - dA2) + mx * (dB1 - dB2));
                double al_y = g * 2 * (mx * (dA1 - dA2) + my * (dB1 - dB2));
                double be = g * dB2, ga = g * 2 * dA2;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        int yy = orc_reflect(py + dy, H), xx = orc_reflect(px + dx, W);
                        double a = xp[(size_t)yy * W + xx], bq = yp[(size_t)yy * W + xx];
                        size_t q = (size_t)(py + dy + 1) * Wp + (px + dx + 1);
                        gxp[q] += (al_x + 2 * a * be + bq * ga) / 9.0;
                        gyp[q] += (al_y + 2 * bq * be + a * ga) / 9.0;
                    }
            }
        /* fold the reflection padding back (autograd of ReflectionPad2d(1)) */
        for (int pass = 0; pass < 2; ++pass) {
            double *gp = pass ? gyp : gxp;
            float *dst = pass ? gtarg : gpred;
            if (!dst) continue;
            dst += (size_t)bc * HW;
            for (size_t i = 0; i < HW; ++i) dst[i] = 0.f;
            double *acc = (double *)calloc(HW, sizeof(double));
            for (int qy = -1; qy <= H; ++qy)
                for (int qx = -1; qx <= W; ++qx) {
                    int yy = orc_reflect(qy, H), xx = orc_reflect(qx, W);
                    acc[(size_t)yy * W + xx] += gp[(size_t)(qy + 1) * Wp + qx + 1];
                }
            for (size_t i = 0; i < HW; ++i) dst[i] = (float)acc[i];
            free(acc);
        }
        free(gxp); free(gyp);
    }
}

/* ------------------------------------------------------------------------------------------
 * A9  identity loss + noise, concat, per-pixel min  (reference: processor.py:186-204)
 * channels: [ident(f0)..ident(fS-1), reproj(f0)..reproj(fS-1)] when automasking, else reproj only.
 * ident += 1e-5f * noise (mul, add).  torch.min: first minimal index wins; a NaN wins.
 * ident/reproj/noise: [B,S,H,W];  outputs: combined [B,C,H,W] (may be NULL), to_opt [B,H,W],
 * idx [B,H,W] uint8.  Returns sum(to_opt) in double.
 * ---------------------------------------------------------------------------------------- */
ORC_API double orc_min_automask(const float *ident, const float *noise, const float *reproj, int B,
                                int S, int H, int W, int automask, float *combined, float *to_opt,
                                uint8_t *idx)
{
    size_t HW = (size_t)H * W;
    int C = automask ? 2 * S : S;
    double total = 0;
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < HW; ++i) {
            float best = 0.f; int bi = 0;
            for (int c = 0; c < C; ++c) {
                float v;
                if (automask && c < S) {
                    float t = 1e-5f * noise[((size_t)b * S + c) * HW + i];
                    v = ident[((size_t)b * S + c) * HW + i] + t;
                } else {
                    int f = automask ? c - S : c;
                    v = reproj[((size_t)b * S + f) * HW + i];
                }
                if (combined) combined[((size_t)b * C + c) * HW + i] = v;
                if (c == 0 || v < best || (v != v && best == best)) { best = v; bi = c; }
            }
            to_opt[(size_t)b * HW + i] = best;
            idx[(size_t)b * HW + i] = (uint8_t)bi;
            total += best;
        }
    return total;
}

/* ------------------------------------------------------------------------------------------
 * A10  SmoothLoss / EdgeAwareSmooth   reference: model_loss/model_loss.py:77-88,112-115
 * disp [B,1,h,w], color [B,3,h,w] -> scalar.  Reduction order not pinned (tolerance 1e-4 rel).
 * ---------------------------------------------------------------------------------------- */
ORC_API double orc_smooth_loss(const float *disp, const float *color, int B, int h, int w,
                               float *gdisp /* may be NULL; d(loss)/d(disp) for unit upstream */)
{
    size_t hw = (size_t)h * w;
    double Nx = (double)B * h * (w - 1), Ny = (double)B * (h - 1) * w;
    double tx = 0, ty = 0;
    for (int b = 0; b < B; ++b) {
        const float *d = disp + b * hw;
        const float *c0 = color + (size_t)b * 3 * hw;
        /* mean(2,True).mean(3,True): over H first, then W (float32 in the reference) */
        double m = 0;
        for (size_t i = 0; i < hw; ++i) m += d[i];
        float mf = (float)(m / (double)hw);
        float den = mf + 1e-7f;
        double *G = gdisp ? (double *)calloc(hw, sizeof(double)) : 0;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float n0 = d[y * w + x] / den;
                if (x + 1 < w) {
                    float n1 = d[y * w + x + 1] / den;
                    float gi = 0.f;
                    for (int c = 0; c < 3; ++c) {
                        float a = fabsf(c0[c * hw + y * w + x] - c0[c * hw + y * w + x + 1]);
                        gi = c ? gi + a : a;
                    }
                    gi = gi / 3.0f;
                    double wgt = exp(-(double)gi);
                    tx += fabs((double)n0 - n1) * wgt;
                    if (G) {
                        double sg = ((n0 > n1) - (n0 < n1)) * wgt / Nx;
                        G[y * w + x] += sg; G[y * w + x + 1] -= sg;
                    }
                }
                if (y + 1 < h) {
                    float n1 = d[(y + 1) * w + x] / den;
                    float gi = 0.f;
                    for (int c = 0; c < 3; ++c) {
                        float a = fabsf(c0[c * hw + y * w + x] - c0[c * hw + (y + 1) * w + x]);
                        gi = c ? gi + a : a;
                    }
                    gi = gi / 3.0f;
                    double wgt = exp(-(double)gi);
                    ty += fabs((double)n0 - n1) * wgt;
                    if (G) {
                        double sg = ((n0 > n1) - (n0 < n1)) * wgt / Ny;
                        G[y * w + x] += sg; G[(y + 1) * w + x] -= sg;
                    }
                }
            }
        if (G) {
            double dot = 0;
            for (size_t i = 0; i < hw; ++i) dot += G[i] * d[i];
            for (size_t i = 0; i < hw; ++i)
                gdisp[b * hw + i] = (float)(G[i] / den - dot / ((double)den * den * (double)hw));
            free(G);
        }
    }
    return tx / Nx + ty / Ny;
}

/* ------------------------------------------------------------------------------------------
 * Composite: one scale of compute.image2warping + compute.compute_loss
 * reference: model_tool/processor.py:139-163 (warp) and 166-217 (loss), for one `scale`.
 * Materialises every intermediate exactly as the reference does.
 *
 * disp [B,1,h,w]; target [B,3,H,W]; sources[S] each [B,3,H,W]; invK [B,4,4]; P [S,B,3,4];
 * noise [B,S,H,W] (automask only).
 * Optional outputs (NULL to skip): depth [B,1,H,W]; grid [S,B,H,W,2]; warp [S,B,3,H,W];
 * reproj [B,S,H,W]; ident [B,S,H,W] (before noise); combined [B,C,H,W].
 * Required outputs: to_opt [B,H,W], idx [B,H,W].  Returns sum(to_opt).
 * When C == 1 (no automask, one source; processor.py:201-202) idx is all zero.
 * ---------------------------------------------------------------------------------------- */
ORC_API double orc_photometric_fwd(int B, int H, int W, int h, int w, int S, double min_depth,
                                   double max_depth, int automask, const float *disp,
                                   const float *target, const float *const *sources,
                                   const float *invK, const float *P, const float *noise,
                                   float *depth, float *grid, float *warp, float *reproj,
                                   float *ident, float *combined, float *to_opt, uint8_t *idx)
{
    size_t HW = (size_t)H * W, N = (size_t)B * HW;
    float *up = (float *)malloc(N * sizeof(float)), *dep = depth ? depth : (float *)malloc(N * sizeof(float));
    float *cam = (float *)malloc(N * 4 * sizeof(float));
    float *g = (float *)malloc(N * 2 * sizeof(float)), *wc = (float *)malloc(N * 3 * sizeof(float)), *rl = (float *)malloc(N * sizeof(float));
    float *rp = reproj ? reproj : (float *)malloc(N * S * sizeof(float));
    float *id = ident ? ident : (float *)malloc(N * S * sizeof(float));
    orc_upsample_bilinear(disp, B, h, w, up, H, W);
    orc_disparity2depth(up, N, min_depth, max_depth, 0, dep);
    orc_backproject(dep, invK, B, H, W, cam);
    for (int f = 0; f < S; ++f) {
        orc_project(cam, P + (size_t)f * B * 12, B, H, W, g);
        if (grid) memcpy(grid + (size_t)f * N * 2, g, N * 2 * sizeof(float));
        orc_grid_sample(sources[f], g, B, 3, H, W, H, W, wc);
        if (warp) memcpy(warp + (size_t)f * N * 3, wc, N * 3 * sizeof(float));
        orc_reprojection_loss(wc, target, B, H, W, rl);
        for (int b = 0; b < B; ++b) memcpy(rp + ((size_t)b * S + f) * HW, rl + b * HW, HW * sizeof(float));
        if (automask) {
            orc_reprojection_loss(sources[f], target, B, H, W, rl);
            for (int b = 0; b < B; ++b) memcpy(id + ((size_t)b * S + f) * HW, rl + b * HW, HW * sizeof(float));
        }
    }
    double total = orc_min_automask(id, noise, rp, B, S, H, W, automask, combined, to_opt, idx);
    free(up); if (!depth) free(dep); free(cam); free(g); free(wc); free(rl);
    if (!reproj) free(rp);
    if (!ident) free(id);
    return total;
}

/* Backward of the composite for an upstream gradient g_min on every element of to_opt
 * (g_min = grad_loss / (num_scales * B*H*W) for the reference's mean, processor.py:212-216).
 * Chains the closed-form backward of each stage in autograd order.
 * Outputs: gdisp [B,1,h,w]; gP [S,B,3,4]. */
ORC_API void orc_photometric_bwd(int B, int H, int W, int h, int w, int S, double min_depth,
                                 double max_depth, int automask, const float *disp,
                                 const float *target, const float *const *sources,
                                 const float *invK, const float *P, const uint8_t *idx,
                                 double g_min, float *gdisp, float *gP)
{
    size_t HW = (size_t)H * W, N = (size_t)B * HW;
    float a, bcoef;
    orc_disp2depth_consts(min_depth, max_depth, &a, &bcoef);
    float *up = (float *)malloc(N * sizeof(float)), *dep = (float *)malloc(N * sizeof(float)), *sd = (float *)malloc(N * sizeof(float));
    float *g = (float *)malloc(N * 2 * sizeof(float)), *wc = (float *)malloc(N * 3 * sizeof(float));
    float *grl = (float *)malloc(N * sizeof(float)), *gwc = (float *)malloc(N * 3 * sizeof(float)), *gg = (float *)malloc(N * 2 * sizeof(float));
    float *cam = (float *)malloc(N * 4 * sizeof(float));
    double *gdepth = (double *)calloc(N, sizeof(double));
    float *gup = (float *)malloc(N * sizeof(float));
    orc_upsample_bilinear(disp, B, h, w, up, H, W);
    orc_disparity2depth(up, N, min_depth, max_depth, sd, dep);
    orc_backproject(dep, invK, B, H, W, cam);
    for (int f = 0; f < S; ++f) {
        const float *Pf = P + (size_t)f * B * 12;
        int ch = automask ? S + f : f;
        orc_project(cam, Pf, B, H, W, g);
        orc_grid_sample(sources[f], g, B, 3, H, W, H, W, wc);
        for (size_t i = 0; i < N; ++i) grl[i] = (idx[i] == ch) ? (float)g_min : 0.f;
        orc_reprojection_loss_bwd(wc, target, grl, B, H, W, gwc, 0);
        orc_grid_sample_bwd_grid(sources[f], g, gwc, B, 3, H, W, H, W, gg);
        for (int b = 0; b < B; ++b) {
            double accP[12] = {0};
            const float *Pb = Pf + b * 12;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    size_t i = (size_t)b * HW + (size_t)y * W + x;
                    float X[4], gx, gy, u, v, z, r[3];
                    for (int k = 0; k < 4; ++k) X[k] = cam[((size_t)b * 4 + k) * HW + (size_t)y * W + x];
                    orc_project_point(Pb, X, H, W, 1e-7f, &gx, &gy, &u, &v, &z);
                    double gu = (double)gg[i * 2] * 2.0 / (double)(W - 1);
                    double gv = (double)gg[i * 2 + 1] * 2.0 / (double)(H - 1);
                    double gq[3] = {gu / z, gv / z, -(gu * u + gv * v) / z};
                    orc_ray(invK + b * 16, (float)x, (float)y, r);
                    double gd = 0;
                    for (int j = 0; j < 3; ++j) {
                        double gX = gq[0] * Pb[j] + gq[1] * Pb[4 + j] + gq[2] * Pb[8 + j];
                        gd += gX * r[j];
                    }
                    gdepth[i] += gd;
                    for (int ii = 0; ii < 3; ++ii)
                        for (int j = 0; j < 4; ++j) accP[ii * 4 + j] += gq[ii] * X[j];
                }
            for (int k = 0; k < 12; ++k) gP[((size_t)f * B + b) * 12 + k] = (float)accP[k];
        }
    }
    for (size_t i = 0; i < N; ++i) {
        double dd = dep[i];
        gup[i] = (float)(gdepth[i] * (-(double)bcoef * dd * dd));
    }
    orc_upsample_bilinear_bwd(gup, B, H, W, gdisp, h, w);
    free(up); free(dep); free(sd); free(g); free(wc); free(grl); free(gwc); free(gg); free(cam);
    free(gdepth); free(gup);
}

ORC_API int orc_version(void) { return 1; }
