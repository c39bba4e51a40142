// photo_fwd.hip -- fused photometric FORWARD kernel for gfx950 (MI355X).
//
// One launch per scale replaces, for that scale,
//   compute.image2warping  (model_tool/processor.py:141-162)  and the photometric half of
//   compute.compute_loss   (model_tool/processor.py:172-204,212):
// bilinear disparity upsample -> disparity2depth -> Depth2PointCloud -> PointCloud2Pixel ->
// grid_sample(border) -> SSIM+L1 ReprojectionLoss -> (+identity loss, noise) -> per-pixel min.
//
// One 256-thread block (4 x wave64) per 64x8 tile of one image; tiles are walked in an XCD-aware order.
//   stage 0  target tile + halo and the disparity region -> LDS with 16-byte coalesced loads;
//   stage 1  every thread warps up to three tile+halo pixels.  The three pixels are processed TOGETHER:
//            all their 8-byte tap loads (two per channel: north pair, south pair) are issued before the
//            first is consumed, so a block pays two memory round trips here instead of six;
//   stage 2  SSIM + L1 per pixel from LDS (3x3 windows never touch HBM again), min / arg-min in
//            registers, loss partial by wave64 shuffles -> one double per tile (summed by a second,
//            fixed-order pass: deterministic).
// IDENT = true evaluates the identity (auto-mask) losses: stage 1 is replaced by loading the un-warped
// source tiles (processor.py:187-191).
#include <type_traits>
#include "photo_common.hpp"

namespace mdx {

constexpr int NPIX = (FX * FY + NT - 1) / NT;   // tile+halo pixels per thread in stage 1 (3)
constexpr int ROWS = TY / (NT / 64);            // output rows per thread in stage 2 (2): ADJACENT rows, so that
                                                // the two 3x3 windows share two of their three LDS rows and products

struct HaloPx { int ly, lx, px, py; bool valid, interior; };

MDX_DEV HaloPx halo_px(int i, int x0, int y0, int H, int W)
{
    HaloPx h;
    h.valid = i < FX * FY;
    const int ii = h.valid ? i : 0;
    h.ly = ii / FX;
    h.lx = ii - h.ly * FX;
    const int gx = x0 + h.lx - 1, gy = y0 + h.ly - 1;
    h.valid = h.valid && gx <= W && gy <= H;          // beyond the reflected ring: never read
    h.px = h.valid ? reflect(gx, W) : 0;
    h.py = h.valid ? reflect(gy, H) : 0;
    h.interior = h.valid && gx == h.px && gy == h.py && h.lx >= 1 && h.lx <= TX && h.ly >= 1 && h.ly <= TY;
    return h;
}

// low-resolution disparity region that the tile's bilinear taps can touch
struct DispRegion { int ry0, rx0, nry, nrx; bool staged; };

MDX_DEV DispRegion disp_region(const mdx_desc &d, int x0, int y0)
{
    DispRegion r;
    const int ymin = max(y0 - 1, 0), ymax = min(y0 + TY, d.H - 1);
    const int xmin = max(x0 - 1, 0), xmax = min(x0 + TX, d.W - 1);
    const float sy = (float)d.h / (float)d.H, sx = (float)d.w / (float)d.W;
    r.ry0 = up_tap(sy, ymin, d.h).i0;
    r.rx0 = up_tap(sx, xmin, d.w).i0;
    r.nry = up_tap(sy, ymax, d.h).i1 - r.ry0 + 1;
    r.nrx = up_tap(sx, xmax, d.w).i1 - r.rx0 + 1;
    r.staged = r.nry * r.nrx <= FX * FY;
    return r;
}

MDX_DEV float upsample_staged(const float *s_d, const DispRegion &r, const mdx_desc &d, int py, int px)
{
    const UpTap ty = up_tap((float)d.h / (float)d.H, py, d.h);
    const UpTap tx = up_tap((float)d.w / (float)d.W, px, d.w);
    const float *r0 = s_d + (ty.i0 - r.ry0) * r.nrx - r.rx0, *r1 = s_d + (ty.i1 - r.ry0) * r.nrx - r.rx0;
    return up_combine(r0[tx.i0], r0[tx.i1], r1[tx.i0], r1[tx.i1], ty, tx, (d.flags & MDX_FLAG_UPSAMPLE_PREMUL) != 0);
}

// COEF: training variant -- besides the loss it emits, for every pixel whose arg-min is a reprojection channel,
// the three SSIM coefficient maps (alpha, beta, gamma per colour channel) of that frame: the window statistics
// are in registers here anyway, so the backward kernel (photo_bwd.hip, coefficient path) never rebuilds them.
template <int S, bool IDENT, bool COEF>
MDX_DEV void photometric_fwd_body(const FwdArgs &a)
{
    __shared__ float s_t[3][FY][FX];
    __shared__ float s_x[S][3][FY][FX];
    __shared__ float s_d[FY][FX];
    __shared__ double s_red[NT / 64];

    const mdx_desc &d = a.d;
    const int H = d.H, W = d.W;
    const size_t HW = (size_t)H * W;
    const TileId tile = tile_id();
    const int b = tile.b, x0 = tile.tx * TX, y0 = tile.ty * TY;
    const int tid = threadIdx.x;
    const float *tgt_b = a.target + (size_t)b * 3 * HW;
    const bool wide = ((W & 3) == 0) && (x0 + TX <= W);

    // ---- stage 0: coalesced tile loads ----
#pragma unroll
    for (int c = 0; c < 3; ++c) load_plane_tile<1>(s_t[c], tgt_b + c * HW, H, W, x0, y0, tid);
    if (IDENT) {
#pragma unroll
        for (int f = 0; f < S; ++f)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                load_plane_tile<1>(s_x[f][c], a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, x0, y0, tid);
    }
    const bool same_res = (d.h == H && d.w == W);
    DispRegion reg = {};
    if (!IDENT) {
        const float *disp_b = a.disp + (size_t)b * d.h * d.w;
        if (same_res) {
            load_plane_tile<1>(s_d, disp_b, H, W, x0, y0, tid);
        } else {
            reg = disp_region(d, x0, y0);
            if (reg.staged)
                for (int i = tid; i < reg.nry * reg.nrx; i += NT) {
                    const int ry = i / reg.nrx, rx = i - ry * reg.nrx;
                    (&s_d[0][0])[i] = disp_b[(size_t)(reg.ry0 + ry) * d.w + reg.rx0 + rx];
                }
        }
    }
    __syncthreads();

    // ---- stage 1: warp the tile + halo, three pixels per thread in flight together ----
    if (!IDENT) {
        const float *disp_b = a.disp + (size_t)b * d.h * d.w;
        const float *invK_b = a.invK + b * 16;
        const Norm2 nd = desc_norm(d);
        // The tile + halo holds FX*FY = 660 pixels = 2 full rounds of the block plus 148: only the waves that own
        // part of that remainder run a third pixel (wave-uniform choice, so it is a scalar branch)
        auto warp_stage = [&](auto np_tag) {
            constexpr int NP = decltype(np_tag)::value;
            HaloPx hp[NP];
            PixelGeom g[NP];
    #pragma unroll
            for (int k = 0; k < NP; ++k) {
                hp[k] = halo_px(tid + k * NT, x0, y0, H, W);
                float up;
                if (same_res) up = s_d[hp[k].ly][hp[k].lx];
                else if (reg.staged) up = upsample_staged(&s_d[0][0], reg, d, hp[k].py, hp[k].px);
                else up = upsample_at(disp_b, d.h, d.w, H, W, hp[k].py, hp[k].px, (d.flags & MDX_FLAG_UPSAMPLE_PREMUL) != 0);
                g[k] = geom_from_disp(d, up, invK_b, hp[k].px, hp[k].py);
                if (a.depth && hp[k].interior) at32(a.depth + (size_t)b * HW, (unsigned)(hp[k].py * W + hp[k].px)) = g[k].depth;
            }
    #pragma unroll
            for (int f = 0; f < S; ++f) {
                const float *Pf = a.P + ((size_t)f * d.B + b) * 12;
                Tap t[NP];
    #pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const Proj pr = project_point(Pf, g[k].X0, g[k].X1, g[k].X2, 1.0f, nd, 1e-7f);
                    t[k] = make_tap(pr.gx, pr.gy, H, W);
                }
                Corners cn[NP][3];
    #pragma unroll
                for (int k = 0; k < NP; ++k)
    #pragma unroll
                    for (int c = 0; c < 3; ++c)
                        cn[k][c] = load_corners(a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, t[k]);
    #pragma unroll
                for (int k = 0; k < NP; ++k)
    #pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = sample(cn[k][c], t[k]);
                        if (hp[k].valid) s_x[f][c][hp[k].ly][hp[k].lx] = v;
                        if (a.warp && !wide && hp[k].interior)
                            at32(a.warp + (((size_t)f * d.B + b) * 3 + c) * HW, (unsigned)(hp[k].py * W + hp[k].px)) = v;
                    }
            }
        };
        const int third = __builtin_amdgcn_readfirstlane((tid & ~63) + (NPIX - 1) * NT < FX * FY);
        if (third) warp_stage(std::integral_constant<int, NPIX>{});
        else warp_stage(std::integral_constant<int, NPIX - 1>{});
        __syncthreads();
        // warped colours of the tile -> HBM as 16-byte stores (kept for the backward kernel)
        if (a.warp && wide) {
            for (int i = tid; i < S * 3 * TY * (TX / 4); i += NT) {
                const int j = i % (TX / 4), r = (i / (TX / 4)) % TY, fc = i / ((TX / 4) * TY);
                if (y0 + r >= H) continue;
                const float *sp = &s_x[0][0][0][0] + (size_t)fc * FY * FX + (r + 1) * FX + 1 + 4 * j;
                const int f = fc / 3, c = fc - 3 * f;
                float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
                *reinterpret_cast<float4 *>(&at32(a.warp + (((size_t)f * d.B + b) * 3 + c) * HW, (unsigned)((y0 + r) * W + x0 + 4 * j))) = v;
            }
        }
    }

    // ---- stage 2: SSIM + L1 from LDS, min / arg-min, loss partial ----
    double acc = 0.0;
    const int tx = tid & 63;
    const int px = x0 + tx;
    const bool automask = (d.flags & MDX_FLAG_AUTOMASK) != 0;
    float idv[ROWS][S], nzv[ROWS][S];
    if (!IDENT && automask) {   // issue the per-pixel loads first: their latency hides under the SSIM math
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            const int py = y0 + ROWS * (tid >> 6) + q;
            const bool valid = px < W && py < H;
#pragma unroll
            for (int f = 0; f < S; ++f) {
                const unsigned o = (unsigned)((valid ? py : 0) * W + (valid ? px : 0));
                idv[q][f] = at32(a.ident + ((size_t)b * S + f) * HW, o);
                nzv[q][f] = at32(a.noise + ((size_t)b * S + f) * HW, o);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const int r = ROWS * (tid >> 6) + q;
        const int py = y0 + r;
        const bool valid = px < W && py < H;
        const unsigned p = (unsigned)((valid ? py : 0) * W + (valid ? px : 0));
        float y9[3][9];
        TargetStats ts[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int k = 0; k < 9; ++k) y9[c][k] = s_t[c][r + k / 3][tx + k % 3];
            ts[c] = target_stats(y9[c]);
        }
        float rl[S];
        SsimTerms st[COEF ? S : 1][3];
#pragma unroll
        for (int f = 0; f < S; ++f) {
            float ss[3], ad[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float x9[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) x9[k] = s_x[f][c][r + k / 3][tx + k % 3];
                const SsimTerms stc = pred_stats(x9, y9[c]);
                if (COEF) st[COEF ? f : 0][c] = stc;
                ss[c] = clamp01(ssim_raw(stc, ts[c]));
                ad[c] = fabsf(y9[c][4] - x9[4]);
            }
            rl[f] = reprojection_combine(ss, ad);
            if (a.reproj && valid) at32(a.reproj + ((size_t)b * S + f) * HW, p) = rl[f];
        }
        if (IDENT) continue;
        // concat [ident + 1e-5*noise, reproj] and torch.min's first-minimum rule (processor.py:194-204)
        float best = 0.f;
        int bi = 0;
        if (automask) {
#pragma unroll
            for (int f = 0; f < S; ++f) {
                const float t = 1e-5f * nzv[q][f];
                const float v = idv[q][f] + t;
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
        if (valid) {
            at32(a.idx + (size_t)b * HW, p) = (uint8_t)bi;
            if (a.to_opt) at32(a.to_opt + (size_t)b * HW, p) = best;
            acc += (double)best;
        }
        if (COEF) {
            // coefficient maps of the selected frame (zero where an identity channel won: the auto-mask)
            const int fsel = automask ? bi - S : bi;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                SsimTerms sel = st[0][c];
#pragma unroll
                for (int f = 1; f < S; ++f)
                    if (fsel == f) sel = st[COEF ? f : 0][c];
                SsimGrad sg = ssim_grad(sel, ts[c], 0.85f / 3.0f);
                if (fsel < 0) { sg.alpha = 0.f; sg.beta = 0.f; sg.gamma = 0.f; }
                if (valid) {   // [B,3,H,W,3]: one 12-byte store per colour channel (a store instruction costs the
                               // same issue slot for 4 or 12 bytes per lane)
                    char *o = reinterpret_cast<char *>(a.coef + ((size_t)b * 3 + c) * HW * 3) + p * 12u;
                    float3_a4 v;
                    v.x = sg.alpha; v.y = sg.beta; v.z = sg.gamma;
                    *reinterpret_cast<float3_a4 *>(o) = v;
                }
            }
        }
    }
    if (IDENT) return;
    acc = wave_sum(acc);
    if ((tid & 63) == 0) s_red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) t += s_red[k];
        a.partials[tile.linear] = t;
    }
}

template <int S, bool IDENT>
// blocks per CU: 5 by registers for S <= 2; the warped tiles of more frames make LDS the limit (4, then 3)
__global__ __launch_bounds__(NT, S <= 2 ? 5 : (S == 3 ? 4 : 3)) void photometric_fwd_kernel(FwdArgs a)
{
    photometric_fwd_body<S, IDENT, false>(a);
}

// training form: exactly four waves per SIMD (128 VGPRs) -- the window statistics of every frame stay live until
// the arg-min is known
template <int S>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(S <= 3 ? 4 : 3, S <= 3 ? 4 : 3))) void photometric_fwd_coef_kernel(FwdArgs a)
{
    photometric_fwd_body<S, false, true>(a);
}

template <bool IDENT, bool COEF>
static int launch_fwd_t(const FwdArgs &a, hipStream_t st)
{
    const dim3 grid = tile_grid(&a.d);
    switch (a.d.S) {
    case 1:
        if constexpr (COEF) hipLaunchKernelGGL((photometric_fwd_coef_kernel<1>), grid, dim3(NT), 0, st, a);
        else hipLaunchKernelGGL((photometric_fwd_kernel<1, IDENT>), grid, dim3(NT), 0, st, a);
        break;
    case 2:
        if constexpr (COEF) hipLaunchKernelGGL((photometric_fwd_coef_kernel<2>), grid, dim3(NT), 0, st, a);
        else hipLaunchKernelGGL((photometric_fwd_kernel<2, IDENT>), grid, dim3(NT), 0, st, a);
        break;
    case 3:
        if constexpr (COEF) hipLaunchKernelGGL((photometric_fwd_coef_kernel<3>), grid, dim3(NT), 0, st, a);
        else hipLaunchKernelGGL((photometric_fwd_kernel<3, IDENT>), grid, dim3(NT), 0, st, a);
        break;
    case 4:
        if constexpr (COEF) hipLaunchKernelGGL((photometric_fwd_coef_kernel<4>), grid, dim3(NT), 0, st, a);
        else hipLaunchKernelGGL((photometric_fwd_kernel<4, IDENT>), grid, dim3(NT), 0, st, a);
        break;
    default: return MDX_ERR_BAD_SHAPE;
    }
    return check_launch();
}

int launch_photometric_fwd(const FwdArgs &a, bool ident, hipStream_t st)
{
    if (ident) return launch_fwd_t<true, false>(a, st);
    return a.coef ? launch_fwd_t<false, true>(a, st) : launch_fwd_t<false, false>(a, st);
}

}  // namespace mdx
