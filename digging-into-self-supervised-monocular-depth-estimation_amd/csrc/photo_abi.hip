// photo_abi.hip -- C-ABI entry points of the fused photometric path (include/mdx.h) and the small
// finishing kernels around the two fused kernels (photo_fwd.hip, photo_bwd.hip).
#include "photo_common.hpp"

namespace mdx {

int launch_finish_gP(const float *partP, int S, int B, int tiles, float g_const, const float *g_dev, float *gP,
                     hipStream_t st);

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

// Transpose of the bilinear upsample (autograd of warp.py:18-20) as a gather: four lanes share one
// low-resolution pixel (rows y, y+4, ... each), the x weights of its footprint live in registers.
// NW bounds the footprint width (2*ratio + 3); wider footprints take the generic kernel below.
template <int NW>
__global__ __launch_bounds__(NT) void upsample_bwd4_kernel(const float *__restrict__ gout, int BC, int H, int W,
                                                           float *__restrict__ gin, int h, int w)
{
    const size_t gid = (size_t)blockIdx.x * NT + threadIdx.x;
    const size_t n = (size_t)BC * h * w;
    const size_t o = gid >> 2 < n ? gid >> 2 : n - 1;
    const int l = (int)(gid & 3);
    const int jx = (int)(o % w), iy = (int)((o / w) % h);
    const size_t bc = o / ((size_t)w * h);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int ya = max(0, (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1);
    const int yb = min(H - 1, (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1);
    const int xa = max(0, (int)floorf(((float)jx - 0.5f) / sx - 0.5f) - 1);
    const int xb = min(W - 1, (int)ceilf(((float)jx + 1.5f) / sx - 0.5f) + 1);
    float wx[NW];
#pragma unroll
    for (int t = 0; t < NW; ++t) {
        const int x = xa + t;
        const UpTap tx = up_tap(sx, x <= xb ? x : xb, w);
        wx[t] = x <= xb ? (tx.i0 == jx ? tx.l0 : 0.f) + (tx.i1 == jx ? tx.l1 : 0.f) : 0.f;
    }
    const float *g = gout + bc * (size_t)H * W;
    float acc = 0.f;
    for (int y = ya + l; y <= yb; y += 4) {
        const UpTap ty = up_tap(sy, y, h);
        const float wy = (ty.i0 == iy ? ty.l0 : 0.f) + (ty.i1 == iy ? ty.l1 : 0.f);
        if (wy == 0.f) continue;
        const float *row = g + (size_t)y * W;
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < NW; ++t) rs = __builtin_fmaf(wx[t], row[min(xa + t, W - 1)], rs);
        acc = __builtin_fmaf(wy, rs, acc);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (l == 0 && (gid >> 2) < n) gin[o] = acc;
}

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
            const float wxv = (tx.i0 == jx ? tx.l0 : 0.f) + (tx.i1 == jx ? tx.l1 : 0.f);
            row += wxv * g[(size_t)y * W + x];
        }
        acc += wy * row;
    }
    gin[i] = acc;
}

int launch_upsample_bwd(const float *gout, int BC, int H, int W, float *gin, int h, int w, hipStream_t st)
{
    const size_t n = (size_t)BC * h * w;
    const int foot = (int)ceilf(2.0f * (float)W / (float)w) + 3;
    if (foot <= 8)
        hipLaunchKernelGGL(upsample_bwd4_kernel<8>, dim3((unsigned)((4 * n + NT - 1) / NT)), dim3(NT), 0, st, gout, BC,
                           H, W, gin, h, w);
    else if (foot <= 20)
        hipLaunchKernelGGL(upsample_bwd4_kernel<20>, dim3((unsigned)((4 * n + NT - 1) / NT)), dim3(NT), 0, st, gout, BC,
                           H, W, gin, h, w);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, gout, BC, H, W,
                           gin, h, w);
    return check_launch();
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
    d->flags = (automask ? MDX_FLAG_AUTOMASK : 0u) | ((H + W <= 128) ? MDX_FLAG_UPSAMPLE_PREMUL : 0u) |
               (div_verified(W - 1) ? MDX_FLAG_FASTDIV_W : 0u) | (div_verified(H - 1) ? MDX_FLAG_FASTDIV_H : 0u);
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

static int check_sources(const mdx_desc *d, const mdx_sources *src)
{
    if (!src) return MDX_ERR_NULL_POINTER;
    for (int f = 0; f < d->S; ++f) {
        if (!src->img[f]) return MDX_ERR_NULL_POINTER;
        if (!aligned(src->img[f], 16)) return MDX_ERR_MISALIGNED;
    }
    return MDX_OK;
}

MDX_EXPORT int mdx_identity_loss(const mdx_desc *d, const float *target, const mdx_sources *src,
                                 float *ident, void *stream)
{
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!target || !ident) return MDX_ERR_NULL_POINTER;
    if ((rc = check_sources(d, src))) return rc;
    if (!aligned(target, 16)) return MDX_ERR_MISALIGNED;
    FwdArgs a = {};
    a.d = *d; a.target = target; a.src = *src; a.reproj = ident;
    return launch_photometric_fwd(a, true, (hipStream_t)stream);
}

MDX_EXPORT size_t mdx_photometric_workspace_bytes(const mdx_desc *d)
{
    return validate_desc(d) ? 0 : ws_total(d);
}

MDX_EXPORT void *mdx_event_create(void)
{
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? (void *)e : nullptr;
}

MDX_EXPORT void mdx_event_destroy(void *event)
{
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}

MDX_EXPORT int mdx_event_elapsed_us(void *start, void *stop, float *us)
{
    if (!start || !stop || !us) return MDX_ERR_NULL_POINTER;
    float ms = 0.f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess ||
        hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess)
        return MDX_ERR_LAUNCH;
    *us = 1e3f * ms;
    return MDX_OK;
}

static void mark(const mdx_timing *t, bool stop, hipStream_t st)
{
    void *e = t ? (stop ? t->stop : t->start) : nullptr;
    if (e) (void)hipEventRecord((hipEvent_t)e, st);
}

MDX_EXPORT int mdx_photometric_fwd_timed(const mdx_desc *d, const float *disp, const float *target,
                                         const mdx_sources *src, const float *invK, const float *P,
                                         const float *ident, const float *noise, uint8_t *idx,
                                         float *loss_sum, float *to_opt, float *depth, float *warp,
                                         float *reproj, float *coef, void *workspace, size_t workspace_bytes,
                                         void *stream, const mdx_timing *t)
{
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!disp || !target || !invK || !P || !idx) return MDX_ERR_NULL_POINTER;
    if ((d->flags & MDX_FLAG_AUTOMASK) && (!ident || !noise)) return MDX_ERR_NULL_POINTER;
    if ((rc = check_sources(d, src))) return rc;
    if (!workspace || workspace_bytes < ws_total(d)) return MDX_ERR_WORKSPACE;
    // 16-byte loads/stores on the image planes: torch allocations are 256-byte aligned
    if (!aligned(workspace, 8) || !aligned(target, 16) || !aligned(disp, 16) || (warp && !aligned(warp, 16)) ||
        (coef && !aligned(coef, 16)))
        return MDX_ERR_MISALIGNED;
    FwdArgs a = {};
    a.d = *d; a.disp = disp; a.target = target; a.src = *src; a.invK = invK; a.P = P;
    a.ident = ident; a.noise = noise; a.idx = idx; a.to_opt = to_opt; a.depth = depth; a.warp = warp;
    a.reproj = reproj; a.coef = coef; a.partials = (double *)workspace;
    mark(t, false, (hipStream_t)stream);
    rc = launch_photometric_fwd(a, false, (hipStream_t)stream);
    mark(t, true, (hipStream_t)stream);
    if (rc || !loss_sum) return rc;   // loss_sum == NULL: leave the per-tile partials in the workspace
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream,
                       (const double *)workspace, (int)num_tiles(d), loss_sum);
    return check_launch();
}

MDX_EXPORT int mdx_photometric_fwd(const mdx_desc *d, const float *disp, const float *target,
                                   const mdx_sources *src, const float *invK, const float *P,
                                   const float *ident, const float *noise, uint8_t *idx,
                                   float *loss_sum, float *to_opt, float *depth, float *warp,
                                   float *reproj, float *coef, void *workspace, size_t workspace_bytes,
                                   void *stream)
{
    return mdx_photometric_fwd_timed(d, disp, target, src, invK, P, ident, noise, idx, loss_sum, to_opt, depth, warp,
                                     reproj, coef, workspace, workspace_bytes, stream, nullptr);
}

MDX_EXPORT int mdx_photometric_bwd_timed(const mdx_desc *d, const float *disp, const float *target,
                                         const mdx_sources *src, const float *invK, const float *P,
                                         const uint8_t *idx, const float *warp, const float *coef,
                                         float g_const, const float *g_dev, float *gdisp, float *gP,
                                         void *workspace, size_t workspace_bytes, void *stream,
                                         const mdx_timing *t)
{
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!disp || !target || !invK || !P || !idx) return MDX_ERR_NULL_POINTER;
    if ((gdisp == nullptr) != (gP == nullptr)) return MDX_ERR_NULL_POINTER;
    if ((rc = check_sources(d, src))) return rc;
    if (!workspace || workspace_bytes < ws_total(d)) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 8) || !aligned(target, 16) || (warp && !aligned(warp, 16)) || (coef && !aligned(coef, 16)))
        return MDX_ERR_MISALIGNED;
    hipStream_t st = (hipStream_t)stream;
    // gdisp == gP == NULL: run the fused kernel only and leave its raw outputs (per-tile d(P) partials, the
    // full-resolution disparity gradient) in the workspace -- used to time that kernel in isolation
    const bool raw = gdisp == nullptr;
    const bool same = !raw && (d->h == d->H && d->w == d->W);
    BwdArgs a = {};
    a.d = *d; a.disp = disp; a.target = target; a.src = *src; a.invK = invK; a.P = P; a.idx = idx;
    a.warp = warp; a.coef = coef; a.g_const = g_const; a.g_dev = g_dev;
    a.partP = (float *)((char *)workspace + ws_off_partP(d));
    a.gup = same ? gdisp : (float *)((char *)workspace + ws_off_gup(d));
    mark(t, false, st);
    rc = launch_photometric_bwd(a, st);
    mark(t, true, st);
    if (rc || raw) return rc;
    const dim3 grid = tile_grid(d);
    if ((rc = launch_finish_gP(a.partP, d->S, d->B, (int)(grid.x * grid.y), g_const, g_dev, gP, st))) return rc;
    if (!same) rc = launch_upsample_bwd(a.gup, d->B, d->H, d->W, gdisp, d->h, d->w, st);
    return rc;
}

MDX_EXPORT int mdx_photometric_bwd(const mdx_desc *d, const float *disp, const float *target,
                                   const mdx_sources *src, const float *invK, const float *P,
                                   const uint8_t *idx, const float *warp, const float *coef,
                                   float g_const, const float *g_dev, float *gdisp, float *gP,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    return mdx_photometric_bwd_timed(d, disp, target, src, invK, P, idx, warp, coef, g_const, g_dev, gdisp, gP,
                                     workspace, workspace_bytes, stream, nullptr);
}

MDX_EXPORT int mdx_interpolate_bilinear_bwd(const float *gout, int BC, int H, int W, float *gin, int h,
                                            int w, void *stream)
{
    if (!gout || !gin) return MDX_ERR_NULL_POINTER;
    if (BC <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return MDX_ERR_BAD_SHAPE;
    return launch_upsample_bwd(gout, BC, H, W, gin, h, w, (hipStream_t)stream);
}
