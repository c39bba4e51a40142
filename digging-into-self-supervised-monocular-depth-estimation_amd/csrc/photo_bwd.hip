// photo_bwd.hip -- fused photometric BACKWARD kernel for gfx950 (MI355X).  SURVEY Appendix A.1 / A.2.
//
// For one scale: d(loss)/d(to_optimise) on every pixel  ->  d(loss)/d(disp_s) and d(loss)/d(P).
// torch.min routes the upstream gradient to the arg-min channel only (idx); identity channels end in
// constants, so masked pixels contribute nothing (that IS the auto-mask).
//   phase A  warped colours + target on the tile + 2-pixel halo -> LDS.  When the forward kept its warped
//            colours they are simply re-read with 16-byte loads; otherwise the warp is recomputed;
//   phase B  at every window centre whose arg-min is a reprojection channel: the three SSIM coefficient
//            maps (alpha, beta, gamma) of that frame -> LDS;
//   phase C  per pixel: 3x3 gather of the maps (with the reflection-padding fold), + L1 term -> d(warped
//            colour); chain through grid_sample -> projection -> depth: d(up-sampled disparity) and the
//            twelve d(P) sums, reduced through LDS to one partial per tile (fixed-order second pass).
#include "photo_common.hpp"

namespace mdx {

constexpr int NPIXB = (BX * BY + NT - 1) / NT;   // tile+2-halo pixels per thread when re-warping (4)

// SAVED: the forward's warped colours are available (a.warp) -- a separate instantiation so that the
// re-warp path's registers (four pixels of taps and corner values in flight) do not cap its occupancy.
//
// LDS budget: target + S warped images on the 2-pixel halo (9 planes at S = 2, 29 KB) plus ONE channel's
// (alpha, beta, gamma) maps (8 KB): the three colour channels are processed one after the other through
// the same three planes, which keeps four blocks resident per CU instead of two.
constexpr int ROWSB = TY / (NT / 64);   // output rows per thread (2)

constexpr int BWD_WAVES = 3;            // waves per SIMD the register allocation is held to for S <= 3 (2 and 4 measured slower, round 3)
template <int S, bool SAVED>
__global__ __launch_bounds__(NT, S <= 3 ? BWD_WAVES : 1) void photometric_bwd_kernel(BwdArgs a)
{
    constexpr int N_T = 3 * BY * BX, N_X = S * 3 * BY * BX, N_ABG = 3 * FY * FX, N_SEL = (FY * FX + 3) / 4;
    constexpr int N_POOL = N_T + N_X + N_ABG + N_SEL, N_RED = S * 12 * NT;
    __shared__ float pool[N_POOL > N_RED ? N_POOL : N_RED];
    float(*s_t)[BY][BX] = reinterpret_cast<float(*)[BY][BX]>(pool);
    float(*s_x)[3][BY][BX] = reinterpret_cast<float(*)[3][BY][BX]>(pool + N_T);
    float(*s_abg)[FY][FX] = reinterpret_cast<float(*)[FY][FX]>(pool + N_T + N_X);         // [alpha,beta,gamma]
    signed char(*s_sel)[FX] = reinterpret_cast<signed char(*)[FX]>(pool + N_T + N_X + N_ABG);   // frame or -1

    const mdx_desc &d = a.d;
    const int H = d.H, W = d.W;
    const size_t HW = (size_t)H * W;
    const TileId tile = tile_id();
    const int b = tile.b, x0 = tile.tx * TX, y0 = tile.ty * TY;
    const int tid = threadIdx.x;
    const float *tgt_b = a.target + (size_t)b * 3 * HW;
    const float *disp_b = a.disp + (size_t)b * d.h * d.w;
    const float *invK_b = a.invK + b * 16;
    const bool automask = (d.flags & MDX_FLAG_AUTOMASK) != 0;
    const Norm2 nd = desc_norm(d);

    // ---- phase A ----
#pragma unroll
    for (int c = 0; c < 3; ++c) load_plane_tile<2>(s_t[c], tgt_b + c * HW, H, W, x0, y0, tid);
    if (SAVED) {
#pragma unroll
        for (int f = 0; f < S; ++f)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                load_plane_tile<2>(s_x[f][c], a.warp + (((size_t)f * d.B + b) * 3 + c) * HW, H, W, x0, y0, tid);
    } else {
        int lys[NPIXB], lxs[NPIXB];
        bool val[NPIXB];
        PixelGeom g[NPIXB];
#pragma unroll
        for (int k = 0; k < NPIXB; ++k) {
            const int i = tid + k * NT;
            val[k] = i < BX * BY;
            const int ii = val[k] ? i : 0;
            lys[k] = ii / BX;
            lxs[k] = ii - lys[k] * BX;
            const int gx = x0 + lxs[k] - 2, gy = y0 + lys[k] - 2;
            val[k] = val[k] && gx >= -1 && gy >= -1 && gx <= W && gy <= H;
            g[k] = pixel_geom(d, disp_b, invK_b, val[k] ? reflect(gx, W) : 0, val[k] ? reflect(gy, H) : 0);
        }
#pragma unroll
        for (int f = 0; f < S; ++f) {
            const float *Pf = a.P + ((size_t)f * d.B + b) * 12;
            Tap t[NPIXB];
            Corners cn[NPIXB][3];
#pragma unroll
            for (int k = 0; k < NPIXB; ++k) {
                const Proj pr = project_point(Pf, g[k].X0, g[k].X1, g[k].X2, 1.0f, nd, 1e-7f);
                t[k] = make_tap(pr.gx, pr.gy, H, W);
            }
#pragma unroll
            for (int k = 0; k < NPIXB; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    cn[k][c] = load_corners(a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, t[k]);
#pragma unroll
            for (int k = 0; k < NPIXB; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (val[k]) s_x[f][c][lys[k]][lxs[k]] = sample(cn[k][c], t[k]);
        }
    }
    // arg-min selection of every window centre (frame index, -1 = masked / outside)
    for (int i = tid; i < FX * FY; i += NT) {
        const int ly = i / FX, lx = i - ly * FX;
        const int px = x0 + lx - 1, py = y0 + ly - 1;
        int f = -1;
        if (px >= 0 && px < W && py >= 0 && py < H) {
            const int sel = a.idx[(size_t)b * HW + (size_t)py * W + px];
            f = automask ? sel - S : sel;
            if (f < 0 || f >= S) f = -1;
        }
        s_sel[ly][lx] = (signed char)f;
    }
    __syncthreads();

    // ---- per-pixel sampling geometry of this thread's rows: the few numbers the channel passes need
    //      (tap corner, fractional offsets, projection) stay in registers, the rest is recomputed at the end
    const int tx = tid & 63;
    const int px = x0 + tx;
    const float g_scale = a.g_const * (a.g_dev ? a.g_dev[0] : 1.0f);
    bool valid[ROWSB];
    unsigned selpack[ROWSB];           // 9 window centres x 3 bits: frame + 1, 0 = none
    int tap_xy[ROWSB][S];              // x0 | y0 << 16
    float tap_dx0[ROWSB][S], tap_dy0[ROWSB][S];
    float gu[ROWSB][S], gv[ROWSB][S];
    unsigned flags[ROWSB];             // bit f: frame f selected somewhere in the window; bit 8+2f: inx; 9+2f: iny
#pragma unroll
    for (int q = 0; q < ROWSB; ++q) {
        const int r = (tid >> 6) + q * (NT / 64);
        const int py = y0 + r;
        valid[q] = px < W && py < H;
        unsigned pack = 0, fl = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int sel = s_sel[r + k / 3][tx + k % 3];
            pack |= (unsigned)(sel + 1) << (3 * k);
            if (sel >= 0 && valid[q]) fl |= 1u << sel;
        }
        selpack[q] = pack;
        const PixelGeom g = pixel_geom(d, disp_b, invK_b, valid[q] ? px : 0, valid[q] ? py : 0);
#pragma unroll
        for (int f = 0; f < S; ++f) {
            const Proj pr = project_point(a.P + ((size_t)f * d.B + b) * 12, g.X0, g.X1, g.X2, 1.0f, nd, 1e-7f);
            const Tap t = make_tap(pr.gx, pr.gy, H, W);
            tap_xy[q][f] = t.x0 | (t.y0 << 16);
            tap_dx0[q][f] = t.ix - (float)t.x0;
            tap_dy0[q][f] = t.iy - (float)t.y0;
            fl |= (t.inx ? 1u : 0u) << (8 + 2 * f);
            fl |= (t.iny ? 1u : 0u) << (9 + 2 * f);
            gu[q][f] = gv[q][f] = 0.f;
        }
        flags[q] = fl;
    }

    // ---- channel passes: phase B (coefficient maps of channel c) -> phase C (gather + sampling derivative) ----
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        for (int i = tid; i < FX * FY; i += NT) {
            const int ly = i / FX, lx = i - ly * FX;
            const int f = s_sel[ly][lx];
            SsimGrad sg = {0.f, 0.f, 0.f};
            if (f >= 0) {
                float x9[9], y9[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    x9[k] = s_x[f][c][ly + k / 3][lx + k % 3];
                    y9[k] = s_t[c][ly + k / 3][lx + k % 3];
                }
                sg = ssim_grad(pred_stats(x9, y9), target_stats(y9), 0.85f / 3.0f);
            }
            s_abg[0][ly][lx] = sg.alpha;
            s_abg[1][ly][lx] = sg.beta;
            s_abg[2][ly][lx] = sg.gamma;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ROWSB; ++q) {
            const int r = (tid >> 6) + q * (NT / 64);
            const int py = y0 + r;
            // reflection-pad fold: a pixel one step inside the border also receives the mirrored ring tap
            const float wxs[3] = {px == 1 ? 2.f : 1.f, 1.f, px == W - 2 ? 2.f : 1.f};
            const float wys[3] = {py == 1 ? 2.f : 1.f, 1.f, py == H - 2 ? 2.f : 1.f};
            float gA[S], gB[S], gC[S];
#pragma unroll
            for (int f = 0; f < S; ++f) gA[f] = gB[f] = gC[f] = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int ly = r + k / 3, lx = tx + k % 3;
                const float al = s_abg[0][ly][lx], be = s_abg[1][ly][lx], ga = s_abg[2][ly][lx];
                const int sel1 = (int)((selpack[q] >> (3 * k)) & 7u);
                const float wgt = wys[k / 3] * wxs[k % 3];
#pragma unroll
                for (int f = 0; f < S; ++f) {
                    const float m = sel1 == f + 1 ? wgt : 0.f;
                    gA[f] = __builtin_fmaf(m, al, gA[f]);
                    gB[f] = __builtin_fmaf(m, be, gB[f]);
                    gC[f] = __builtin_fmaf(m, ga, gC[f]);
                }
            }
            const int csel1 = (int)((selpack[q] >> 12) & 7u);   // window centre (k = 4)
            const float yq = s_t[c][r + 2][tx + 2];
#pragma unroll
            for (int f = 0; f < S; ++f) {
                if (!((flags[q] >> f) & 1u)) continue;
                const float xq = s_x[f][c][r + 2][tx + 2];
                float gx = (gA[f] + 2.0f * xq * gB[f] + yq * gC[f]) * (1.0f / 9.0f);
                if (csel1 == f + 1) gx -= 0.05f * ((yq > xq) ? 1.f : ((yq < xq) ? -1.f : 0.f));   // 0.15*mean_c|y-x|
                Tap t;
                t.x0 = tap_xy[q][f] & 0xffff;
                t.y0 = tap_xy[q][f] >> 16;
                const Corners cn = load_corners(a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, t);
                const float dx0 = tap_dx0[q][f], dy0 = tap_dy0[q][f], dx1 = 1.0f - dx0, dy1 = 1.0f - dy0;
                gu[q][f] += gx * ((cn.ne - cn.nw) * dy1 + (cn.se - cn.sw) * dy0);
                gv[q][f] += gx * ((cn.sw - cn.nw) * dx1 + (cn.se - cn.ne) * dx0);
            }
        }
        __syncthreads();
    }

    // ---- chain to depth / P ----
    float accP[S][12];
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int k = 0; k < 12; ++k) accP[f][k] = 0.f;
#pragma unroll
    for (int q = 0; q < ROWSB; ++q) {
        if (!valid[q]) continue;
        const int py = y0 + (tid >> 6) + q * (NT / 64);
        const PixelGeom g = pixel_geom(d, disp_b, invK_b, px, py);
        float gdepth = 0.f;
#pragma unroll
        for (int f = 0; f < S; ++f) {
            if (!((flags[q] >> f) & 1u)) continue;
            const float *Pf = a.P + ((size_t)f * d.B + b) * 12;
            const Proj pr = project_point(Pf, g.X0, g.X1, g.X2, 1.0f, nd, 1e-7f);
            // grid normalisation (2/(W-1)) and grid_sample's un-normalisation ((W-1)/2) cancel
            const float u_ = ((flags[q] >> (8 + 2 * f)) & 1u) ? gu[q][f] : 0.f;
            const float v_ = ((flags[q] >> (9 + 2 * f)) & 1u) ? gv[q][f] : 0.f;
            const float iz = 1.0f / pr.z;
            const float gq0 = u_ * iz, gq1 = v_ * iz, gq2 = -(u_ * pr.u + v_ * pr.v) * iz;
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

    // ---- d(P): block reduction through LDS (the pool is free: the last channel pass ended on a barrier) ----
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int k = 0; k < 12; ++k) pool[(f * 12 + k) * NT + tid] = accP[f][k];
    __syncthreads();
    for (int v = tid >> 6; v < S * 12; v += NT / 64) {
        const float *p = pool + v * NT + (tid & 63);
        float t = (p[0] + p[64]) + (p[128] + p[192]);
        t = wave_sum(t);
        if ((tid & 63) == 0) a.partP[(size_t)tile.linear * (S * 12) + v] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// Coefficient path: the training forward already emitted, per pixel, the (alpha, beta, gamma) maps of its
// arg-min frame (photo_fwd.hip, COEF).  What is left for the backward is phase C only:
// nine coefficient planes on the 1-pixel halo -> LDS (16-byte loads), one barrier, then per pixel the 3x3
// gather with the reflection-pad fold, the L1 term, and the grid_sample / projection / depth chain.
// No window statistics, no channel passes, no re-warp: ~60 % fewer instructions than the kernel above.
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(NT, 3) void photometric_bwd_coef_kernel(BwdArgs a)
{
    __shared__ float s_abg[9][FY][FX];             // [3*channel + {alpha, beta, gamma}] on the tile + 1-pixel halo
    __shared__ signed char s_sel[FY][FX + 2];      // arg-min frame of the pixel, or -1
    __shared__ float s_red[NT / 64][S * 12];       // d(P): one row per wave

    const mdx_desc &d = a.d;
    const int H = d.H, W = d.W;
    const size_t HW = (size_t)H * W;
    const TileId tile = tile_id();
    const int b = tile.b, x0 = tile.tx * TX, y0 = tile.ty * TY;
    const int tid = threadIdx.x;
    const float *disp_b = a.disp + (size_t)b * d.h * d.w;
    const float *invK_b = a.invK + b * 16;
    const bool automask = (d.flags & MDX_FLAG_AUTOMASK) != 0;
    const Norm2 nd = desc_norm(d);

    const float g_scale = a.g_const * (a.g_dev ? a.g_dev[0] : 1.0f);
    const int tx = tid & 63;
    const int px = x0 + tx;
    float gdepth[ROWSB];
    // frame-independent part of the geometry, once per row; its disparity taps are the first loads of the block so
    // the arithmetic behind them overlaps the staging traffic below
    PixelGeom geo[ROWSB];
#pragma unroll
    for (int q = 0; q < ROWSB; ++q) {
        gdepth[q] = 0.f;
        const int py = y0 + ROWSB * (tid >> 6) + q;
        const bool ok = px < W && py < H;
        geo[q] = pixel_geom(d, disp_b, invK_b, ok ? px : 0, ok ? py : 0);
    }
    // coefficient triplets of the tile + 1-pixel halo (reflection padded): one 12-byte load per pixel and colour
    // channel, parked as three planes (the 3x3 gather below then reads pairs of adjacent columns)
    for (int i = tid; i < 3 * FY * FX; i += NT) {
        const int c = i / (FY * FX), rem = i - c * (FY * FX);
        const int ly = rem / FX, lx = rem - ly * FX;
        const int hx = x0 + lx - 1, hy = y0 + ly - 1;
        if (hx > W || hy > H) continue;            // beyond the reflected ring: never read
        const unsigned o = (unsigned)(reflect(hy, H) * W + reflect(hx, W)) * 12u;
        const float3_a4 v = *reinterpret_cast<const float3_a4 *>(reinterpret_cast<const char *>(a.coef + ((size_t)b * 3 + c) * HW * 3) + o);
        s_abg[3 * c][ly][lx] = v.x; s_abg[3 * c + 1][ly][lx] = v.y; s_abg[3 * c + 2][ly][lx] = v.z;
    }
    for (int i = tid; i < FX * FY; i += NT) {
        const int ly = i / FX, lx = i - ly * FX;
        const int hx = x0 + lx - 1, hy = y0 + ly - 1;
        int f = -1;
        if (hx >= 0 && hx < W && hy >= 0 && hy < H) {
            const int sel = a.idx[(size_t)b * HW + (size_t)hy * W + hx];
            f = automask ? sel - S : sel;
            if (f < 0 || f >= S) f = -1;
        }
        s_sel[ly][lx] = (signed char)f;
    }
    __syncthreads();

    // frame by frame (rolled loop: one frame's taps, colours and sums live at a time -- 3 blocks/CU would
    // otherwise spill), both rows of this thread inside
#pragma unroll 1
    for (int f = 0; f < S; ++f) {
        const float *Pf = a.P + ((size_t)f * d.B + b) * 12;
        const float *src_b = a.src.img[f] + (size_t)b * 3 * HW;
        float accP[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) accP[k] = 0.f;
#pragma unroll
        for (int q = 0; q < ROWSB; ++q) {
            const int r = ROWSB * (tid >> 6) + q;
            const int py = y0 + r;
            if (px >= W || py >= H) continue;
            // window centres that selected this frame, with the reflection-pad fold (a pixel one step inside the
            // border also receives the mirrored ring tap)
            const float wxs[3] = {px == 1 ? 2.f : 1.f, 1.f, px == W - 2 ? 2.f : 1.f};
            const float wys[3] = {py == 1 ? 2.f : 1.f, 1.f, py == H - 2 ? 2.f : 1.f};
            float mk[9];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const bool hit = s_sel[r + k / 3][tx + k % 3] == f;
                mk[k] = hit ? wys[k / 3] * wxs[k % 3] : 0.f;
                any = any || hit;
            }
            if (!any) continue;
            const bool centre = s_sel[r + 1][tx + 1] == f;
            const unsigned pb = (unsigned)(py * W + px) * 4u;   // byte offset inside a plane
            float yq[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                yq[c] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.target + ((size_t)b * 3 + c) * HW) + pb);
            const PixelGeom &g = geo[q];
            const Proj pr = project_point(Pf, g.X0, g.X1, g.X2, 1.0f, nd, 1e-7f);
            const Tap t = make_tap(pr.gx, pr.gy, H, W);
            Corners cn[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) cn[c] = load_corners(src_b + c * HW, H, W, t);
            const float dy1 = (float)(t.y0 + 1) - t.iy, dy0 = t.iy - (float)t.y0;
            const float dx1 = (float)(t.x0 + 1) - t.ix, dx0 = t.ix - (float)t.x0;
            // 3x3 coefficient sums from LDS first: they do not need the gathers, which are still in flight
            float gA[3], gB[3], gC[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gA[c] = 0.f; gB[c] = 0.f; gC[c] = 0.f;
                const float *pa = &s_abg[3 * c][r][tx];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int o = (k / 3) * FX + k % 3;
                    gA[c] = __builtin_fmaf(mk[k], pa[o], gA[c]);
                    gB[c] = __builtin_fmaf(mk[k], pa[FY * FX + o], gB[c]);
                    gC[c] = __builtin_fmaf(mk[k], pa[2 * FY * FX + o], gC[c]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the sums above the first use of a gathered corner
            float gu = 0.f, gv = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // the warped colour of the pixel itself: the corners are here for the gradient anyway, so it is
                // re-sampled (bit-identical to the forward's value) instead of being stored and re-read
                const float xq = sample(cn[c], t);
                float gx = (gA[c] + 2.0f * xq * gB[c] + yq[c] * gC[c]) * (1.0f / 9.0f);
                if (centre) gx -= 0.05f * ((yq[c] > xq) ? 1.f : ((yq[c] < xq) ? -1.f : 0.f));   // 0.15*mean_c|y-x|
                gu += gx * ((cn[c].ne - cn[c].nw) * dy1 + (cn[c].se - cn[c].sw) * dy0);
                gv += gx * ((cn[c].sw - cn[c].nw) * dx1 + (cn[c].se - cn[c].ne) * dx0);
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
                // depth = 1/(a + b*disp)  ->  d depth / d disp = -b * depth^2 (applied here, per frame)
                gdepth[q] += gX * g.r[j] * (-d.disp_b * g.depth * g.depth);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accP[j] += gq0 * X[j];
                accP[4 + j] += gq1 * X[j];
                accP[8 + j] += gq2 * X[j];
            }
        }
        // wave totals in registers (DPP), lane 63 parks them for the cross-wave sum below
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float t = wave_sum_dpp_lane63(accP[k]);
            if ((tid & 63) == 63) s_red[tid >> 6][f * 12 + k] = t;
        }
    }
#pragma unroll
    for (int q = 0; q < ROWSB; ++q) {
        const int py = y0 + ROWSB * (tid >> 6) + q;
        if (px < W && py < H) a.gup[(size_t)b * HW + (size_t)py * W + px] = gdepth[q] * g_scale;
    }

    // ---- d(P): the four wave totals, fixed order ----
    __syncthreads();
    if (tid < S * 12)
        a.partP[(size_t)tile.linear * (S * 12) + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
}

// d(P)[f,b,:] = g * sum over the tiles of image b: one wave64 per output element, fixed order
__global__ __launch_bounds__(64) void finish_gP_kernel(const float *__restrict__ partP, int S, int B, int tiles,
                                                       float g_const, const float *__restrict__ g_dev,
                                                       float *__restrict__ gP)
{
    const int i = blockIdx.x;   // over S*B*12
    const int k = i % 12, bb = (i / 12) % B, f = i / (12 * B);
    double acc = 0.0;
    for (int t = threadIdx.x; t < tiles; t += 64)
        acc += (double)partP[((size_t)bb * tiles + t) * (S * 12) + f * 12 + k];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) {
        const float g = g_const * (g_dev ? g_dev[0] : 1.0f);
        gP[i] = (float)(acc * (double)g);
    }
}

int launch_photometric_bwd(const BwdArgs &a, hipStream_t st)
{
    const dim3 grid = tile_grid(&a.d);
#define MDX_BWD_CASE(SS)                                                                                  \
    case SS:                                                                                              \
        if (a.coef) hipLaunchKernelGGL((photometric_bwd_coef_kernel<SS>), grid, dim3(NT), 0, st, a); \
        else if (a.warp) hipLaunchKernelGGL((photometric_bwd_kernel<SS, true>), grid, dim3(NT), 0, st, a);     \
        else hipLaunchKernelGGL((photometric_bwd_kernel<SS, false>), grid, dim3(NT), 0, st, a);           \
        break;
    switch (a.d.S) {
        MDX_BWD_CASE(1) MDX_BWD_CASE(2) MDX_BWD_CASE(3) MDX_BWD_CASE(4)
    default: return MDX_ERR_BAD_SHAPE;
    }
#undef MDX_BWD_CASE
    return check_launch();
}

int launch_finish_gP(const float *partP, int S, int B, int tiles, float g_const, const float *g_dev, float *gP,
                     hipStream_t st)
{
    hipLaunchKernelGGL(finish_gP_kernel, dim3(S * B * 12), dim3(64), 0, st, partP, S, B, tiles, g_const, g_dev, gP);
    return check_launch();
}

}  // namespace mdx
