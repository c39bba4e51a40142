// photo_common.hpp -- shared pieces of the fused photometric kernels (gfx950).
#pragma once
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
    float *coef;         // optional [B,9,H,W]: (alpha,beta,gamma) x channel of the selected frame (training)
    double *partials;
};

struct BwdArgs {
    mdx_desc d;
    const float *disp, *target;
    mdx_sources src;
    const float *invK, *P;
    const uint8_t *idx;
    const float *warp;   // optional [S,B,3,H,W]: the forward's warped colours (skips the re-warp)
    const float *coef;   // optional [B,9,H,W]: the forward's SSIM coefficient maps (needs warp too)
    float g_const;
    const float *g_dev;
    float *gup;          // [B,H,W] d loss / d upsampled disparity
    float *partP;        // [tiles][S][12]
};

int launch_photometric_fwd(const FwdArgs &a, bool ident, hipStream_t st);
int launch_photometric_bwd(const BwdArgs &a, hipStream_t st);

// XCD-aware tile order (workgroups are dealt round-robin over the 8 XCDs, each with a private 4 MB L2):
// remap the linear workgroup id so that the blocks that share an XCD walk a CONTIGUOUS run of tiles.
// Neighbouring tiles re-read each other's halo rows/columns and gather footprints; with this order those
// re-reads hit the XCD's L2 instead of going back to the Infinity Cache / HBM.  Bijective for any grid.
struct TileId { int tx, ty, b; unsigned linear; };

MDX_DEV TileId tile_id()
{
    const unsigned nx = gridDim.x, ny = gridDim.y, nwg = nx * ny * gridDim.z;
    const unsigned orig = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    TileId t;
    t.linear = wg;
    t.tx = (int)(wg % nx);
    t.ty = (int)((wg / nx) % ny);
    t.b = (int)(wg / (nx * ny));
    return t;
}

MDX_DEV Norm2 desc_norm(const mdx_desc &d)
{
    Norm2 n;
    n.w = make_normdiv(d.W - 1, (d.flags & MDX_FLAG_FASTDIV_W) != 0);
    n.h = make_normdiv(d.H - 1, (d.flags & MDX_FLAG_FASTDIV_H) != 0);
    return n;
}

// geometry of one pixel: everything that does not depend on the source frame
struct PixelGeom { float depth, X0, X1, X2, r[3]; };

MDX_DEV PixelGeom geom_from_disp(const mdx_desc &d, float up, const float *__restrict__ invK_b, int px, int py)
{
    PixelGeom g;
    const float sd = scaled_disp(up, d.disp_a, d.disp_b);
    g.depth = 1.0f / sd;
    pixel_ray(invK_b, (float)px, (float)py, g.r);
    g.X0 = g.depth * g.r[0];
    g.X1 = g.depth * g.r[1];
    g.X2 = g.depth * g.r[2];
    return g;
}

MDX_DEV PixelGeom pixel_geom(const mdx_desc &d, const float *__restrict__ disp_b,
                             const float *__restrict__ invK_b, int px, int py)
{
    const float up = upsample_at(disp_b, d.h, d.w, d.H, d.W, py, px, (d.flags & MDX_FLAG_UPSAMPLE_PREMUL) != 0);
    return geom_from_disp(d, up, invK_b, px, py);
}

// ---------------------------------------------------------------------------------------------
// Cooperative load of one image plane's tile (+HALO, reflection padded) into LDS.
// Fast path (W % 4 == 0, tile fully inside in x): the 64-wide body goes through 16-byte loads
// (global_load_dwordx4, 1 KiB per wave instruction), only the 2*HALO halo columns are scalar.
// dst is [TY+2*HALO][TX+2*HALO]; slot (ly,lx) holds padded position (x0+lx-HALO, y0+ly-HALO).
// Positions beyond the one-pixel reflection ring (gx < -1, gx > W, ...) are never read and stay unset.
// ---------------------------------------------------------------------------------------------
template <int HALO>
MDX_DEV void load_plane_tile(float (*dst)[TX + 2 * HALO], const float *__restrict__ plane, int H, int W,
                             int x0, int y0, int tid)
{
    constexpr int NYT = TY + 2 * HALO, NXT = TX + 2 * HALO;
    const bool wide = ((W & 3) == 0) && (x0 + TX <= W);
    if (wide) {
        for (int i = tid; i < NYT * (TX / 4); i += NT) {
            const int ly = i / (TX / 4), j = i - ly * (TX / 4);
            const int gy = y0 + ly - HALO;
            if (gy < -1 || gy > H) continue;
            const int py = reflect(gy, H);
            const float4 v = *reinterpret_cast<const float4 *>(&at32(plane, (unsigned)(py * W + x0 + 4 * j)));
            float *o = &dst[ly][HALO + 4 * j];
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
        for (int i = tid; i < NYT * 2 * HALO; i += NT) {
            const int ly = i / (2 * HALO), e = i - ly * (2 * HALO);
            const int lx = e < HALO ? e : TX + e;
            const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
            if (gx < -1 || gx > W || gy < -1 || gy > H) continue;
            dst[ly][lx] = at32(plane, (unsigned)(reflect(gy, H) * W + reflect(gx, W)));
        }
    } else {
        for (int i = tid; i < NYT * NXT; i += NT) {
            const int ly = i / NXT, lx = i - ly * NXT;
            const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
            if (gx < -1 || gx > W || gy < -1 || gy > H) continue;
            dst[ly][lx] = at32(plane, (unsigned)(reflect(gy, H) * W + reflect(gx, W)));
        }
    }
}

// The same for NP planes at once, with EVERY load of the block issued before the first LDS write (the 64-wide body as 16-byte
// loads, the halo columns as scalars; rows beyond the reflection ring are clamped onto it: loaded, never read).  One plane at
// a time (load_plane_tile) each thread's load sits in its own exec-masked block with an s_waitcnt vmcnt(0) behind it: NP x 2
// memory round trips one after the other (tools/isa_loadwaits.py).  `planes[p]` are wave-uniform pointers.
template <int HALO, int NP>
MDX_DEV void load_plane_tiles(float (*const (&dst)[NP])[TX + 2 * HALO], const float *const (&planes)[NP], int H, int W, int x0,
                              int y0, int tid)
{
    constexpr int NYT = TY + 2 * HALO, NBODY = NYT * (TX / 4), NHALO = NYT * 2 * HALO;
    static_assert(NBODY <= NT && NHALO <= NT, "one body quad and one halo pixel per thread");
    const bool wide = ((W & 3) == 0) && (x0 + TX <= W);
    if (!wide) {
#pragma unroll
        for (int p = 0; p < NP; ++p) load_plane_tile<HALO>(dst[p], planes[p], H, W, x0, y0, tid);
        return;
    }
    const int bt = tid < NBODY ? tid : NBODY - 1, ht = tid < NHALO ? tid : NHALO - 1;       // (idle threads repeat a load)
    const int bly = bt / (TX / 4), bj = bt - bly * (TX / 4);
    const int hly = ht / (2 * HALO), he = ht - hly * (2 * HALO);
    const int hlx = he < HALO ? he : TX + he;
    const int bgy = min(max(y0 + bly - HALO, -1), H), hgy = min(max(y0 + hly - HALO, -1), H), hgx = min(max(x0 + hlx - HALO, -1), W);
    const unsigned bo = (unsigned)(reflect(bgy, H) * W + x0 + 4 * bj), ho = (unsigned)(reflect(hgy, H) * W + reflect(hgx, W));
    float4 bv[NP];
    float hv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        bv[p] = *reinterpret_cast<const float4 *>(&at32(planes[p], bo));
        hv[p] = at32(planes[p], ho);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (tid < NBODY) {
            float *o = &dst[p][bly][HALO + 4 * bj];
            o[0] = bv[p].x; o[1] = bv[p].y; o[2] = bv[p].z; o[3] = bv[p].w;
        }
        if (tid < NHALO) dst[p][hly][hlx] = hv[p];
    }
}

}  // namespace mdx
