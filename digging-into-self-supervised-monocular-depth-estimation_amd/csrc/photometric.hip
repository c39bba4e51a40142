// photometric.hip -- fused photometric kernels for gfx950 (MI355X) and their C-ABI entry points.
//
// One launch per scale replaces, for that scale, the whole of
//   compute.image2warping  (model_tool/processor.py:141-162)  and the photometric half of
//   compute.compute_loss   (model_tool/processor.py:172-204,212):
// bilinear disparity upsample -> disparity2depth -> Depth2PointCloud -> PointCloud2Pixel ->
// grid_sample(border) -> SSIM+L1 ReprojectionLoss -> (+identity loss, noise) -> per-pixel min.
//
// Layout: one 256-thread block (4 x wave64) per 64x8 output tile of one image.  A wave owns 64
// consecutive pixels of a row, so every planar read/write is a 256-byte coalesced segment.
//   phase A  every thread warps pixels of the tile + halo and parks the warped colours (and the
//            target) in LDS -- the 3x3 SSIM windows are then LDS reads, never HBM re-reads;
//   phase B  SSIM + L1 per pixel from LDS, min / arg-min in registers, loss partial by wave64
//            shuffles, one double per block to the workspace (deterministic two-pass sum).
// The backward kernel recomputes the warp on a 2-pixel halo, turns the arg-min selection into the
// three SSIM coefficient maps (alpha, beta, gamma) in LDS, gathers them with the reflection-pad
// fold, and chains through grid_sample / projection / depth to d(disp) and d(P).
#include "mdx_common.hpp"
#include "mdx_device.hpp"

namespace mdx {

constexpr int FX = TX + 2, FY = TY + 2;   // tile + 1-pixel halo (SSIM window)
constexpr int BX = TX + 4, BY = TY + 4;   // tile + 2-pixel halo (backward)

struct FwdArgs {
    mdx_desc d;
    const float *disp, *target;
    mdx_sources src;
    const float *invK, *P, *ident, *noise;
    uint8_t *idx;
    float *to_opt, *depth, *warp, *reproj;
    double *partials;
};

// geometry of one pixel: everything that does not depend on the source frame
struct PixelGeom { float depth, X0, X1, X2, r[3]; };

MDX_DEV PixelGeom pixel_geom(const mdx_desc &d, const float *__restrict__ disp_b,
                             const float *__restrict__ invK_b, int px, int py)
{
    PixelGeom g;
    float up = upsample_at(disp_b, d.h, d.w, d.H, d.W, py, px, (d.flags & MDX_FLAG_UPSAMPLE_PREMUL) != 0);
    float sd = scaled_disp(up, d.disp_a, d.disp_b);
    g.depth = 1.0f / sd;
    pixel_ray(invK_b, (float)px, (float)py, g.r);
    g.X0 = g.depth * g.r[0];
    g.X1 = g.depth * g.r[1];
    g.X2 = g.depth * g.r[2];
    return g;
}

// ---------------------------------------------------------------------------------------------
// forward.  IDENT = true: "prediction" is the un-warped source image (identity loss,
// processor.py:187-191) and the per-frame loss maps are the only output.
// ---------------------------------------------------------------------------------------------
template <int S, bool IDENT>
__global__ __launch_bounds__(NT) void photometric_fwd_kernel(FwdArgs a)
{
    __shared__ float s_t[3][FY][FX];
    __shared__ float s_x[S][3][FY][FX];
    __shared__ double s_red[NT / 64];

    const mdx_desc &d = a.d;
    const int H = d.H, W = d.W;
    const size_t HW = (size_t)H * W;
    const int b = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int tid = threadIdx.x;
    const float *tgt_b = a.target + (size_t)b * 3 * HW;
    const float *disp_b = IDENT ? nullptr : a.disp + (size_t)b * d.h * d.w;
    const float *invK_b = IDENT ? nullptr : a.invK + b * 16;

    // ---- phase A: fill the halo tile ----
    for (int i = tid; i < FX * FY; i += NT) {
        const int ly = i / FX, lx = i - ly * FX;
        const int gx = x0 + lx - 1, gy = y0 + ly - 1;
        if (gx > W || gy > H) continue;   // beyond the reflected border: never read
        const int px = reflect(gx, W), py = reflect(gy, H);
        const size_t p = (size_t)py * W + px;
#pragma unroll
        for (int c = 0; c < 3; ++c) s_t[c][ly][lx] = tgt_b[c * HW + p];
        if (IDENT) {
#pragma unroll
            for (int f = 0; f < S; ++f)
#pragma unroll
                for (int c = 0; c < 3; ++c) s_x[f][c][ly][lx] = a.src.img[f][((size_t)b * 3 + c) * HW + p];
        } else {
            const PixelGeom g = pixel_geom(d, disp_b, invK_b, px, py);
            const bool interior = (gx == px) && (gy == py) && lx >= 1 && lx <= TX && ly >= 1 && ly <= TY;
            if (a.depth && interior) a.depth[(size_t)b * HW + p] = g.depth;
#pragma unroll
            for (int f = 0; f < S; ++f) {
                const float *Pf = a.P + ((size_t)f * d.B + b) * 12;
                const Proj pr = project_point(Pf, g.X0, g.X1, g.X2, 1.0f, H, W, 1e-7f);
                const Tap t = make_tap(pr.gx, pr.gy, H, W);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float *img = a.src.img[f] + ((size_t)b * 3 + c) * HW;
                    const float v = sample(load_corners(img, H, W, t), t);
                    s_x[f][c][ly][lx] = v;
                    if (a.warp && interior) a.warp[(((size_t)f * d.B + b) * 3 + c) * HW + p] = v;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase B: SSIM + L1 from LDS, min / arg-min, loss partial ----
    double acc = 0.0;
    const int tx = tid & 63;
    const int px = x0 + tx;
    for (int r = tid >> 6; r < TY; r += NT / 64) {
        const int py = y0 + r;
        if (px >= W || py >= H) continue;
        const size_t p = (size_t)py * W + px;
        float y9[3][9];
        TargetStats ts[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int k = 0; k < 9; ++k) y9[c][k] = s_t[c][r + k / 3][tx + k % 3];
            ts[c] = target_stats(y9[c]);
        }
        float rl[S];
#pragma unroll
        for (int f = 0; f < S; ++f) {
            float ss[3], ad[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float x9[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) x9[k] = s_x[f][c][r + k / 3][tx + k % 3];
                ss[c] = clamp01(ssim_raw(pred_stats(x9, y9[c]), ts[c]));
                ad[c] = fabsf(y9[c][4] - x9[4]);
            }
            rl[f] = reprojection_combine(ss, ad);
            if (a.reproj) a.reproj[((size_t)b * S + f) * HW + p] = rl[f];
        }
        if (IDENT) continue;
        // concat [ident + 1e-5*noise, reproj] and torch.min's first-minimum rule (processor.py:194-204)
        float best = 0.f;
        int bi = 0;
        if (d.flags & MDX_FLAG_AUTOMASK) {
#pragma unroll
            for (int f = 0; f < S; ++f) {
                const size_t q = ((size_t)b * S + f) * HW + p;
                const float t = 1e-5f * a.noise[q];
                const float v = a.ident[q] + t;
                if (f == 0 || v < best) { best = v; bi = f; }
            }
#pragma unroll
            for (int f = 0; f < S; ++f)
                if (rl[f] < best) { best = rl[f]; bi = S + f; }
        } else {
            best = rl[0];
#pragma unroll
            for (int f = 1; f < S; ++f)
                if (rl[f] < best) { best = rl[f]; bi = f; }
        }
        a.idx[(size_t)b * HW + p] = (uint8_t)bi;
        if (a.to_opt) a.to_opt[(size_t)b * HW + p] = best;
        acc += (double)best;
    }
    if (IDENT) return;
    acc = wave_sum(acc);
    if ((tid & 63) == 0) s_red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) t += s_red[k];
        a.partials[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
    }
}

// deterministic second pass: one block sums n doubles in a fixed order
__global__ __launch_bounds__(NT) void sum_partials_kernel(const double *__restrict__ part, int n,
                                                          float *__restrict__ out)
{
    __shared__ double s_red[NT / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += NT) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < NT / 64; ++k) t += s_red[k];
        out[0] = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------
// backward (recompute).  SURVEY Appendix A.1 / A.2.
// ---------------------------------------------------------------------------------------------
struct BwdArgs {
    mdx_desc d;
    const float *disp, *target;
    mdx_sources src;
    const float *invK, *P;
    const uint8_t *idx;
    float g_const;
    const float *g_dev;
    float *gup;        // [B,H,W] d loss / d upsampled disparity
    float *partP;      // [nblocks][S][12]
};

template <int S>
__global__ __launch_bounds__(NT) void photometric_bwd_kernel(BwdArgs a)
{
    __shared__ float s_t[3][BY][BX];
    __shared__ float s_x[S][3][BY][BX];
    __shared__ float s_abg[3][3][FY][FX];   // [channel][alpha,beta,gamma]
    __shared__ int s_sel[FY][FX];           // selected source frame of the window centre, -1 = none
    __shared__ float s_redP[NT / 64][S * 12];

    const mdx_desc &d = a.d;
    const int H = d.H, W = d.W;
    const size_t HW = (size_t)H * W;
    const int b = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int tid = threadIdx.x;
    const float *tgt_b = a.target + (size_t)b * 3 * HW;
    const float *disp_b = a.disp + (size_t)b * d.h * d.w;
    const float *invK_b = a.invK + b * 16;
    const bool automask = (d.flags & MDX_FLAG_AUTOMASK) != 0;

    // ---- phase A: warped colours on the 2-pixel halo ----
    for (int i = tid; i < BX * BY; i += NT) {
        const int ly = i / BX, lx = i - ly * BX;
        const int gx = x0 + lx - 2, gy = y0 + ly - 2;
        if (gx < -1 || gy < -1 || gx > W || gy > H) continue;
        const int px = reflect(gx, W), py = reflect(gy, H);
        const size_t p = (size_t)py * W + px;
#pragma unroll
        for (int c = 0; c < 3; ++c) s_t[c][ly][lx] = tgt_b[c * HW + p];
        const PixelGeom g = pixel_geom(d, disp_b, invK_b, px, py);
#pragma unroll
        for (int f = 0; f < S; ++f) {
            const float *Pf = a.P + ((size_t)f * d.B + b) * 12;
            const Proj pr = project_point(Pf, g.X0, g.X1, g.X2, 1.0f, H, W, 1e-7f);
            const Tap t = make_tap(pr.gx, pr.gy, H, W);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                s_x[f][c][ly][lx] = sample(load_corners(a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, t), t);
        }
    }
    __syncthreads();

    // ---- phase B: SSIM coefficient maps of the selected frame at every window centre ----
    for (int i = tid; i < FX * FY; i += NT) {
        const int ly = i / FX, lx = i - ly * FX;
        const int px = x0 + lx - 1, py = y0 + ly - 1;
        int f = -1;
        if (px >= 0 && px < W && py >= 0 && py < H) {
            const int sel = a.idx[(size_t)b * HW + (size_t)py * W + px];
            f = automask ? sel - S : sel;
            if (f >= S) f = -1;
        }
        s_sel[ly][lx] = f;
        if (f < 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) s_abg[c][k][ly][lx] = 0.f;
            continue;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x9[9], y9[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                x9[k] = s_x[f][c][ly + k / 3][lx + k % 3];
                y9[k] = s_t[c][ly + k / 3][lx + k % 3];
            }
            const SsimGrad sg = ssim_grad(pred_stats(x9, y9), target_stats(y9), 0.85f / 3.0f);
            s_abg[c][0][ly][lx] = sg.alpha;
            s_abg[c][1][ly][lx] = sg.beta;
            s_abg[c][2][ly][lx] = sg.gamma;
        }
    }
    __syncthreads();

    // ---- phase C: gather to d(warped colour), chain to depth / P ----
    const float g_scale = a.g_const * (a.g_dev ? a.g_dev[0] : 1.0f);
    float accP[S][12];
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int k = 0; k < 12; ++k) accP[f][k] = 0.f;
    const int tx = tid & 63;
    const int px = x0 + tx;
    for (int r = tid >> 6; r < TY; r += NT / 64) {
        const int py = y0 + r;
        if (px >= W || py >= H) continue;
        // reflection-pad fold: a centre one step inside the border sees the border tap twice
        float wxs[3] = {1.f + (px == 1 ? 1.f : 0.f), 1.f, 1.f + (px == W - 2 ? 1.f : 0.f)};
        float wys[3] = {1.f + (py == 1 ? 1.f : 0.f), 1.f, 1.f + (py == H - 2 ? 1.f : 0.f)};
        const PixelGeom g = pixel_geom(d, disp_b, invK_b, px, py);
        float gdepth = 0.f;
#pragma unroll
        for (int f = 0; f < S; ++f) {
            float gx_c[3];
            bool any = false;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float A = 0.f, Bq = 0.f, Cq = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int ly = r + k / 3, lx = tx + k % 3;
                    if (s_sel[ly][lx] == f) {
                        const float wgt = wys[k / 3] * wxs[k % 3];
                        A += wgt * s_abg[c][0][ly][lx];
                        Bq += wgt * s_abg[c][1][ly][lx];
                        Cq += wgt * s_abg[c][2][ly][lx];
                        any = true;
                    }
                }
                const float xq = s_x[f][c][r + 2][tx + 2], yq = s_t[c][r + 2][tx + 2];
                float gv = (A + 2.0f * xq * Bq + yq * Cq) * (1.0f / 9.0f);
                if (s_sel[r + 1][tx + 1] == f) {
                    // 0.15 * mean_c |y - x|  ->  -0.05 * sign(y - x)
                    const float sg = (yq > xq) ? 1.f : ((yq < xq) ? -1.f : 0.f);
                    gv -= 0.05f * sg;
                }
                gx_c[c] = gv;
            }
            if (!any) continue;
            const float *Pf = a.P + ((size_t)f * d.B + b) * 12;
            const Proj pr = project_point(Pf, g.X0, g.X1, g.X2, 1.0f, H, W, 1e-7f);
            const Tap t = make_tap(pr.gx, pr.gy, H, W);
            const float x1 = (float)(t.x0 + 1), y1 = (float)(t.y0 + 1), xf0 = (float)t.x0, yf0 = (float)t.y0;
            float gu = 0.f, gv = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const Corners cn = load_corners(a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, t);
                gu += gx_c[c] * (-cn.nw * (y1 - t.iy) + cn.ne * (y1 - t.iy) - cn.sw * (t.iy - yf0) + cn.se * (t.iy - yf0));
                gv += gx_c[c] * (-cn.nw * (x1 - t.ix) - cn.ne * (t.ix - xf0) + cn.sw * (x1 - t.ix) + cn.se * (t.ix - xf0));
            }
            // grid normalisation (2/(W-1)) and grid_sample's un-normalisation ((W-1)/2) cancel
            gu = t.inx ? gu : 0.f;
            gv = t.iny ? gv : 0.f;
            const float iz = 1.0f / pr.z;
            const float gq0 = gu * iz, gq1 = gv * iz, gq2 = -(gu * pr.u + gv * pr.v) * iz;
            const float X[4] = {g.X0, g.X1, g.X2, 1.0f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float gX = gq0 * Pf[j] + gq1 * Pf[4 + j] + gq2 * Pf[8 + j];
                gdepth += gX * g.r[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accP[f][j] += gq0 * X[j];
                accP[f][4 + j] += gq1 * X[j];
                accP[f][8 + j] += gq2 * X[j];
            }
        }
        // depth = 1/(a + b*disp)  ->  d depth / d disp = -b * depth^2
        a.gup[(size_t)b * HW + (size_t)py * W + px] = gdepth * (-d.disp_b * g.depth * g.depth) * g_scale;
    }
    // per-block partial of d(P): wave shuffles, then across the 4 waves through LDS
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float v = wave_sum(accP[f][k]);
            if ((tid & 63) == 0) s_redP[tid >> 6][f * 12 + k] = v;
        }
    __syncthreads();
    if (tid < S * 12) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) t += s_redP[k][tid];
        const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.partP[blk * (S * 12) + tid] = t;
    }
}

// d(P)[f,b,:] = g * sum over the tiles of image b   (fixed order, double accumulation)
__global__ void finish_gP_kernel(const float *__restrict__ partP, int S, int B, int tiles,
                                 float g_const, const float *__restrict__ g_dev, float *__restrict__ gP)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over S*B*12
    if (i >= S * B * 12) return;
    const int k = i % 12, bb = (i / 12) % B, f = i / (12 * B);
    double acc = 0.0;
    for (int t = 0; t < tiles; ++t) acc += (double)partP[((size_t)bb * tiles + t) * (S * 12) + f * 12 + k];
    const float g = g_const * (g_dev ? g_dev[0] : 1.0f);
    gP[i] = (float)(acc * (double)g);
}

// transpose of the bilinear upsample (autograd of warp.py:18-20): gather per low-res pixel
__global__ __launch_bounds__(NT) void upsample_bwd_kernel(const float *__restrict__ gout, int BC, int H, int W,
                                                          float *__restrict__ gin, int h, int w)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)BC * h * w) return;
    const int jx = (int)(i % w), iy = (int)((i / w) % h);
    const size_t bc = i / ((size_t)w * h);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    // output rows whose source index scale*(y+0.5)-0.5 can fall in (iy-1, iy+1), with a margin
    const int ya = max(0, (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1);
    const int yb = min(H - 1, (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1);
    const int xa = max(0, (int)floorf(((float)jx - 0.5f) / sx - 0.5f) - 1);
    const int xb = min(W - 1, (int)ceilf(((float)jx + 1.5f) / sx - 0.5f) + 1);
    const float *g = gout + bc * (size_t)H * W;
    float acc = 0.f;
    for (int y = ya; y <= yb; ++y) {
        const UpTap ty = up_tap(sy, y, h);
        const float wy = (ty.i0 == iy ? ty.l0 : 0.f) + (ty.i1 == iy ? ty.l1 : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int x = xa; x <= xb; ++x) {
            const UpTap tx = up_tap(sx, x, w);
            const float wx = (tx.i0 == jx ? tx.l0 : 0.f) + (tx.i1 == jx ? tx.l1 : 0.f);
            row += wx * g[(size_t)y * W + x];
        }
        acc += wy * row;
    }
    gin[i] = acc;
}

__global__ void compose_projection_kernel(const float *__restrict__ K, const float *__restrict__ T, int B,
                                          float *__restrict__ P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 12) return;
    const int j = i % 4, r = (i / 4) % 3, b = i / 12;
    // ATen's small-matrix bmm kernel: acc = 0; acc += K[r][k]*T[k][j]  (mul and add rounded separately)
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float prod = K[b * 16 + r * 4 + k] * T[b * 16 + k * 4 + j];
        acc = acc + prod;
    }
    P[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <bool IDENT>
static int launch_fwd(const FwdArgs &a, hipStream_t st)
{
    const dim3 grid = tile_grid(&a.d);
    switch (a.d.S) {
    case 1: hipLaunchKernelGGL((photometric_fwd_kernel<1, IDENT>), grid, dim3(NT), 0, st, a); break;
    case 2: hipLaunchKernelGGL((photometric_fwd_kernel<2, IDENT>), grid, dim3(NT), 0, st, a); break;
    case 3: hipLaunchKernelGGL((photometric_fwd_kernel<3, IDENT>), grid, dim3(NT), 0, st, a); break;
    case 4: hipLaunchKernelGGL((photometric_fwd_kernel<4, IDENT>), grid, dim3(NT), 0, st, a); break;
    default: return MDX_ERR_BAD_SHAPE;
    }
    return check_launch();
}

static size_t num_tiles(const mdx_desc *d)
{
    const dim3 g = tile_grid(d);
    return (size_t)g.x * g.y * g.z;
}

// workspace: [tiles] double loss partials | [tiles][S][12] float dP partials | [B*H*W] float gup
static size_t ws_off_partP(const mdx_desc *d) { return num_tiles(d) * sizeof(double); }
static size_t ws_off_gup(const mdx_desc *d) { return ws_off_partP(d) + ((num_tiles(d) * d->S * 12 * sizeof(float) + 7) & ~(size_t)7); }
static size_t ws_total(const mdx_desc *d) { return ws_off_gup(d) + (size_t)d->B * d->H * d->W * sizeof(float); }

}  // namespace mdx

using namespace mdx;

MDX_EXPORT int mdx_version(void) { return MDX_VERSION; }

MDX_EXPORT const char *mdx_status_string(int s)
{
    switch (s) {
    case MDX_OK: return "MDX_OK";
    case MDX_ERR_BAD_SHAPE: return "MDX_ERR_BAD_SHAPE";
    case MDX_ERR_NULL_POINTER: return "MDX_ERR_NULL_POINTER";
    case MDX_ERR_WORKSPACE: return "MDX_ERR_WORKSPACE";
    case MDX_ERR_LAUNCH: return "MDX_ERR_LAUNCH";
    case MDX_ERR_UNSUPPORTED: return "MDX_ERR_UNSUPPORTED";
    case MDX_ERR_MISALIGNED: return "MDX_ERR_MISALIGNED";
    default: return "MDX_ERR_UNKNOWN";
    }
}

MDX_EXPORT int mdx_desc_init(mdx_desc *d, int B, int H, int W, int h, int w, int S, int automask,
                             double min_depth, double max_depth)
{
    if (!d) return MDX_ERR_NULL_POINTER;
    if (!(min_depth > 0.0) || !(max_depth > min_depth)) return MDX_ERR_BAD_SHAPE;
    d->B = B; d->H = H; d->W = W; d->h = h; d->w = w; d->S = S;
    d->flags = (automask ? MDX_FLAG_AUTOMASK : 0u) | ((H + W <= 128) ? MDX_FLAG_UPSAMPLE_PREMUL : 0u);
    // warp.py:34-37 in Python doubles, rounded to f32 where they meet the tensor
    const double min_disp = 1.0 / max_depth, max_disp = 1.0 / min_depth;
    d->disp_a = (float)min_disp;
    d->disp_b = (float)(max_disp - min_disp);
    return validate_desc(d);
}

MDX_EXPORT int mdx_compose_projection(const float *K, const float *T, int B, float *P, void *stream)
{
    if (!K || !T || !P) return MDX_ERR_NULL_POINTER;
    if (B <= 0) return MDX_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(compose_projection_kernel, dim3((B * 12 + 63) / 64), dim3(64), 0, (hipStream_t)stream, K, T, B, P);
    return check_launch();
}

MDX_EXPORT int mdx_identity_loss(const mdx_desc *d, const float *target, const mdx_sources *src,
                                 float *ident, void *stream)
{
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!target || !src || !ident) return MDX_ERR_NULL_POINTER;
    for (int f = 0; f < d->S; ++f)
        if (!src->img[f]) return MDX_ERR_NULL_POINTER;
    FwdArgs a = {};
    a.d = *d; a.target = target; a.src = *src; a.reproj = ident;
    return launch_fwd<true>(a, (hipStream_t)stream);
}

MDX_EXPORT size_t mdx_photometric_workspace_bytes(const mdx_desc *d)
{
    return validate_desc(d) ? 0 : ws_total(d);
}

MDX_EXPORT int mdx_photometric_fwd(const mdx_desc *d, const float *disp, const float *target,
                                   const mdx_sources *src, const float *invK, const float *P,
                                   const float *ident, const float *noise, uint8_t *idx,
                                   float *loss_sum, float *to_opt, float *depth, float *warp,
                                   float *reproj, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!disp || !target || !src || !invK || !P || !idx) return MDX_ERR_NULL_POINTER;
    if ((d->flags & MDX_FLAG_AUTOMASK) && (!ident || !noise)) return MDX_ERR_NULL_POINTER;
    for (int f = 0; f < d->S; ++f)
        if (!src->img[f]) return MDX_ERR_NULL_POINTER;
    if (!workspace || workspace_bytes < ws_total(d)) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 8)) return MDX_ERR_MISALIGNED;
    FwdArgs a = {};
    a.d = *d; a.disp = disp; a.target = target; a.src = *src; a.invK = invK; a.P = P;
    a.ident = ident; a.noise = noise; a.idx = idx; a.to_opt = to_opt; a.depth = depth; a.warp = warp;
    a.reproj = reproj; a.partials = (double *)workspace;
    rc = launch_fwd<false>(a, (hipStream_t)stream);
    if (rc || !loss_sum) return rc;   // loss_sum == NULL: leave the per-tile partials in the workspace
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream,
                       (const double *)workspace, (int)num_tiles(d), loss_sum);
    return check_launch();
}

MDX_EXPORT int mdx_photometric_bwd(const mdx_desc *d, const float *disp, const float *target,
                                   const mdx_sources *src, const float *invK, const float *P,
                                   const uint8_t *idx, float g_const, const float *g_dev,
                                   float *gdisp, float *gP, void *workspace, size_t workspace_bytes,
                                   void *stream)
{
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!disp || !target || !src || !invK || !P || !idx || !gdisp || !gP) return MDX_ERR_NULL_POINTER;
    for (int f = 0; f < d->S; ++f)
        if (!src->img[f]) return MDX_ERR_NULL_POINTER;
    if (!workspace || workspace_bytes < ws_total(d)) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 8)) return MDX_ERR_MISALIGNED;
    hipStream_t st = (hipStream_t)stream;
    const bool same = (d->h == d->H && d->w == d->W);
    BwdArgs a = {};
    a.d = *d; a.disp = disp; a.target = target; a.src = *src; a.invK = invK; a.P = P; a.idx = idx;
    a.g_const = g_const; a.g_dev = g_dev;
    a.partP = (float *)((char *)workspace + ws_off_partP(d));
    a.gup = same ? gdisp : (float *)((char *)workspace + ws_off_gup(d));
    const dim3 grid = tile_grid(d);
    switch (d->S) {
    case 1: hipLaunchKernelGGL(photometric_bwd_kernel<1>, grid, dim3(NT), 0, st, a); break;
    case 2: hipLaunchKernelGGL(photometric_bwd_kernel<2>, grid, dim3(NT), 0, st, a); break;
    case 3: hipLaunchKernelGGL(photometric_bwd_kernel<3>, grid, dim3(NT), 0, st, a); break;
    case 4: hipLaunchKernelGGL(photometric_bwd_kernel<4>, grid, dim3(NT), 0, st, a); break;
    default: return MDX_ERR_BAD_SHAPE;
    }
    if ((rc = check_launch())) return rc;
    const int nP = d->S * d->B * 12;
    hipLaunchKernelGGL(finish_gP_kernel, dim3((nP + 63) / 64), dim3(64), 0, st, a.partP, d->S, d->B,
                       (int)(grid.x * grid.y), g_const, g_dev, gP);
    if ((rc = check_launch())) return rc;
    if (!same) {
        const size_t n = (size_t)d->B * d->h * d->w;
        hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st,
                           (const float *)a.gup, d->B, d->H, d->W, gdisp, d->h, d->w);
        rc = check_launch();
    }
    return rc;
}

MDX_EXPORT int mdx_interpolate_bilinear_bwd(const float *gout, int BC, int H, int W, float *gin, int h,
                                            int w, void *stream)
{
    if (!gout || !gin) return MDX_ERR_NULL_POINTER;
    if (BC <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return MDX_ERR_BAD_SHAPE;
    const size_t n = (size_t)BC * h * w;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                       (hipStream_t)stream, gout, BC, H, W, gin, h, w);
    return check_launch();
}
