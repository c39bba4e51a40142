// photo_train_finish.hip -- the finishing pass of the one-launch training kernel (photo_train.hip) for gfx950.
#include <cstdlib>
#include <cstring>
#include "photo_train.hpp"

namespace mdx {

// ---------------------------------------------------------------------------------------------
// Second pass, ONE launch, fixed summation orders (deterministic):
//   * transpose of the bilinear upsample (autograd of warp.py:18-20) of every scale below full resolution with an
//     integer ratio R in {2, 4, 8}: a block owns a tile of TH x TW low-resolution pixels, stages the full-resolution
//     gradient region their footprints cover in LDS with coalesced 16-byte loads, forms the x / y weights of the
//     tile once (up_tap: the forward's own tap arithmetic, so borders are exact) and then every pixel sums its
//     (2R+4)^2 footprint out of LDS (LPO lanes per pixel split the rows, a shuffle tree adds them).  Gathering the
//     footprints straight from global memory cost ~40 CU-cycles per load instruction (every lane another cache line);
//   * d(P)[scale][f][b][k] = sum over the items of (scale, b) -- one wave64 per output;
//   * loss_sum[scale].
// ---------------------------------------------------------------------------------------------
struct FinishArgs {
    const float *gup[MDX_MAX_SCALES];
    float *gin[MDX_MAX_SCALES];
    int h[MDX_MAX_SCALES], w[MDX_MAX_SCALES], ratio[MDX_MAX_SCALES], tiles_x[MDX_MAX_SCALES], tiles_y[MDX_MAX_SCALES];
    int up_first[MDX_MAX_SCALES + 1];   // first block of each scale's upsample job (equal = no job)
    int B, H, W, nscales, S, ipi;
    const float *partP;
    const double *loss_part;
    float *gP, *loss_sum;
    unsigned long long *rng;            // optional device {seed, offset}: the step is over, the next one draws new noise
};

// (an output pixel i of an even integer ratio R is touched by the full-resolution indices R*i - R/2 - 2 .. R*i + 3R/2 + 1:
// a footprint of 2R + 4 taps, margin of one included -- the weights decide, the margin only covers rounding)
constexpr int FIN_LDS_FLOATS = 44 * 144 + 64 * 20 + 4 * 20;     // largest configuration (R = 8)

template <int R, int TW, int TH, int LPO>
MDX_DEV void upsample_bwd_tile(const float *__restrict__ gout, int H, int W, float *__restrict__ gin, int h, int w,
                               int bc, int tile_x, int tile_y, float *lds)
{
    constexpr int NT_ = 2 * R + 4;                       // taps per axis
    constexpr int NR = R * TH + R + 4;                   // region rows
    constexpr int NC = ((R * TW + R + 4 + 3 + 3) / 4) * 4;   // region columns: + up to 3 for the 16-byte alignment
    static_assert(NR * NC + TW * NT_ + TH * NT_ <= FIN_LDS_FLOATS, "LDS");
    static_assert(TW * TH * LPO == NT, "one lane group per pixel");
    float *s_g = lds, *s_wx = lds + NR * NC, *s_wy = s_wx + TW * NT_;
    const int tid = threadIdx.x;
    const int jx0 = tile_x * TW, iy0 = tile_y * TH;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int ys = R * iy0 - R / 2 - 2;                  // region origin (may lie outside the image: zeros)
    const int xs = (R * jx0 - R / 2 - 2) & ~3;           // aligned down to 4 columns (arithmetic on negatives is fine: two's complement)
    const float *g = gout + (size_t)bc * H * W;
    // ---- stage the region: 16-byte loads where the four columns are inside the image, scalars at its edges ----
    const bool vec = (W & 3) == 0;
    if (vec) {
        // W % 4 == 0 and xs is a multiple of 4: a quad lies wholly inside the image or wholly outside.  Every quad of the thread
        // is loaded UNCONDITIONALLY (address clamped into the image, zeros selected afterwards), all of them in flight before the
        // first LDS write: with the load inside `if (inside)` each pass of the loop was one memory round trip -- up to seven
        // in a row at R = 8 (tools/isa_loadwaits.py)
        constexpr int NQ = NR * (NC / 4), NIT = (NQ + NT - 1) / NT;
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = min(tid + it * NT, NQ - 1);
            const int rr = e / (NC / 4), c4 = (e - rr * (NC / 4)) * 4;
            const int yc = min(max(ys + rr, 0), H - 1), xc = min(max(xs + c4, 0), W - 4);
            v[it] = *reinterpret_cast<const float4 *>(g + (size_t)yc * W + xc);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * NT;
            if (e >= NQ) break;
            const int rr = e / (NC / 4), c4 = (e - rr * (NC / 4)) * 4;
            const int y = ys + rr, x = xs + c4;
            const bool inside = y >= 0 && y < H && x >= 0 && x < W;
            *reinterpret_cast<float4 *>(s_g + rr * NC + c4) = inside ? v[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else
    for (int e = tid; e < NR * (NC / 4); e += NT) {
        const int rr = e / (NC / 4), c4 = (e - rr * (NC / 4)) * 4;
        const int y = ys + rr, x = xs + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H) {
            if (vec && x >= 0 && x + 3 < W) {
                v = *reinterpret_cast<const float4 *>(g + (size_t)y * W + x);
            } else {
                const float *row = g + (size_t)y * W;
                v.x = (x >= 0 && x < W) ? row[x] : 0.f;
                v.y = (x + 1 >= 0 && x + 1 < W) ? row[x + 1] : 0.f;
                v.z = (x + 2 >= 0 && x + 2 < W) ? row[x + 2] : 0.f;
                v.w = (x + 3 >= 0 && x + 3 < W) ? row[x + 3] : 0.f;
            }
        }
        *reinterpret_cast<float4 *>(s_g + rr * NC + c4) = v;
    }
    // ---- weights of the tile's columns and rows (zero for taps outside the image or not touching the pixel) ----
    for (int e = tid; e < TW * NT_ + TH * NT_; e += NT) {
        const bool isx = e < TW * NT_;
        const int ee = isx ? e : e - TW * NT_;
        const int p = ee / NT_, t = ee - p * NT_;
        const int i = (isx ? jx0 : iy0) + p, n_in = isx ? w : h, n_out = isx ? W : H;
        const int pos = R * i - R / 2 - 2 + t;           // full-resolution index of tap t
        float wgt = 0.f;
        if (i < n_in && pos >= 0 && pos < n_out) {
            const UpTap tp = up_tap(isx ? sx : sy, pos, n_in);
            wgt = (tp.i0 == i ? tp.l0 : 0.f) + (tp.i1 == i ? tp.l1 : 0.f);
        }
        (isx ? s_wx : s_wy)[ee] = wgt;
    }
    __syncthreads();
    // ---- footprint sums ----
    const int l = tid % LPO, pix = tid / LPO;
    const int jj = pix % TW, ii = pix / TW;
    const int c0 = R * (jx0 + jj) - R / 2 - 2 - xs;      // region column of tap 0
    float wxr[NT_];
#pragma unroll
    for (int t = 0; t < NT_; ++t) wxr[t] = s_wx[jj * NT_ + t];
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < (NT_ + LPO - 1) / LPO; ++it) {
        const int ty = l + it * LPO;
        if (ty < NT_) {
            const float *row = s_g + (R * ii + ty) * NC + c0;
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < NT_; ++t) rs = __builtin_fmaf(wxr[t], row[t], rs);
            acc = __builtin_fmaf(s_wy[ii * NT_ + ty], rs, acc);
        }
    }
#pragma unroll
    for (int m = 1; m < LPO; m <<= 1) acc += __shfl_xor(acc, m, 64);
    const int jx = jx0 + jj, iy = iy0 + ii;
    if (l == 0 && jx < w && iy < h) gin[((size_t)bc * h + iy) * w + jx] = acc;
}

// tile shape by ratio (host and device agree through these)
static int finish_tw(int r) { return r == 2 ? 64 : (r == 4 ? 32 : 16); }
static int finish_th(int r) { return 4; }


__global__ __launch_bounds__(NT) void train_finish_kernel(FinishArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_lds[FIN_LDS_FLOATS];
    const int blk = blockIdx.x;
    if (blk < a.up_first[MDX_MAX_SCALES]) {
        const int sc = blk >= a.up_first[3] ? 3 : (blk >= a.up_first[2] ? 2 : (blk >= a.up_first[1] ? 1 : 0));
        const float *gup = pick(a.gup, sc);
        float *gin = pick(a.gin, sc);
        const int h = pick(a.h, sc), w = pick(a.w, sc), r = pick(a.ratio, sc);
        const int tx_n = pick(a.tiles_x, sc), ty_n = pick(a.tiles_y, sc);
        const int rel = blk - pick4(a.up_first, sc);
        const int tile_x = rel % tx_n, tile_y = (rel / tx_n) % ty_n, bc = rel / (tx_n * ty_n);
        if (r == 2) upsample_bwd_tile<2, 64, 4, 1>(gup, a.H, a.W, gin, h, w, bc, tile_x, tile_y, s_lds);
        else if (r == 4) upsample_bwd_tile<4, 32, 4, 2>(gup, a.H, a.W, gin, h, w, bc, tile_x, tile_y, s_lds);
        else upsample_bwd_tile<8, 16, 4, 4>(gup, a.H, a.W, gin, h, w, bc, tile_x, tile_y, s_lds);
        return;
    }
    // reductions.  Loads are issued in groups of four independent ones (a dependent load -> add chain would pay one
    // memory round trip per element); the order of the additions is fixed.
    double *s_red = reinterpret_cast<double *>(s_lds);
    const int rblk = blk - a.up_first[MDX_MAX_SCALES];
    const int lane = threadIdx.x & 63;
    const int ngp = a.nscales * a.S * a.B * 12;
    const int ngp_blocks = (ngp + NT / 64 - 1) / (NT / 64);
    if (rblk < ngp_blocks) {            // d(P): one wave64 per output, four outputs per block
        const int i = rblk * (NT / 64) + (threadIdx.x >> 6);
        if (i >= ngp) return;
        const int k = i % 12, bb = (i / 12) % a.B, f = (i / (12 * a.B)) % a.S, sc = i / (12 * a.B * a.S);
        const float *p = a.partP + ((size_t)(sc * a.B + bb) * a.ipi) * (a.S * 12) + f * 12 + k;
        const size_t stride = (size_t)a.S * 12;
        double acc = 0.0;
        for (int t = lane; t < a.ipi; t += 256) {
            const float v0 = p[(size_t)t * stride];
            const float v1 = t + 64 < a.ipi ? p[(size_t)(t + 64) * stride] : 0.f;
            const float v2 = t + 128 < a.ipi ? p[(size_t)(t + 128) * stride] : 0.f;
            const float v3 = t + 192 < a.ipi ? p[(size_t)(t + 192) * stride] : 0.f;
            acc += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        acc = wave_sum(acc);
        if (lane == 0) a.gP[i] = (float)acc;
    } else {                            // loss_sum[scale]: one block per scale
        const int sc = rblk - ngp_blocks;
        if (sc >= a.nscales) return;
        const double *p = a.loss_part + (size_t)sc * a.B * a.ipi;
        const int cnt = a.B * a.ipi;
        double acc = 0.0;
        for (int t = threadIdx.x; t < cnt; t += 4 * NT) {
            const double v0 = p[t];
            const double v1 = t + NT < cnt ? p[t + NT] : 0.0;
            const double v2 = t + 2 * NT < cnt ? p[t + 2 * NT] : 0.0;
            const double v3 = t + 3 * NT < cnt ? p[t + 3 * NT] : 0.0;
            acc += (v0 + v1) + (v2 + v3);
        }
        acc = wave_sum(acc);
        if (lane == 0) s_red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            a.loss_sum[sc] = (float)((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
            if (sc == 0 && a.rng) a.rng[1] += 1ull;
        }
    }
}

int launch_upsample_bwd(const float *gout, int BC, int H, int W, float *gin, int h, int w, hipStream_t st);

int launch_train_finish(const mdx_train_desc *d, bool grad, int ipi, const float *partP, const double *loss_part,
                        float *const *gup, float *const *gdisp, float *gP, float *loss_sum, unsigned long long *rng,
                        hipStream_t st)
{
    int rc;
    FinishArgs fa = {};
    fa.B = d->B; fa.H = d->H; fa.W = d->W; fa.nscales = d->nscales; fa.S = grad ? d->S : 0; fa.ipi = ipi;
    fa.partP = partP; fa.loss_part = loss_part; fa.gP = gP; fa.loss_sum = loss_sum; fa.rng = rng;
    int nblk = 0;
    bool separate[MDX_MAX_SCALES] = {false, false, false, false};
    for (int s = 0; s < MDX_MAX_SCALES; ++s) {
        fa.up_first[s] = nblk;
        const int ss = s < d->nscales ? s : 0;
        fa.gup[s] = gup[ss]; fa.gin[s] = grad ? gdisp[ss] : nullptr; fa.h[s] = d->h[ss]; fa.w[s] = d->w[ss];
        fa.ratio[s] = 2; fa.tiles_x[s] = fa.tiles_y[s] = 1;
        if (!grad || s >= d->nscales || (d->h[s] == d->H && d->w[s] == d->W)) continue;
#ifdef MDX_TRAIN_STAMPS
        if (const char *e = getenv("MDX_FINISH_SKIP")) if (strchr(e, '0' + s)) continue;   // diagnostic: leave a scale out
#endif
        // the tiled pass takes the integer ratios 2, 4, 8 (same on both axes); anything else the per-scale kernels
        const int r = d->W / d->w[s];
        if (d->W != r * d->w[s] || d->H != r * d->h[s] || (r != 2 && r != 4 && r != 8)) { separate[s] = true; continue; }
        fa.ratio[s] = r;
        fa.tiles_x[s] = (d->w[s] + finish_tw(r) - 1) / finish_tw(r);
        fa.tiles_y[s] = (d->h[s] + finish_th(r) - 1) / finish_th(r);
        nblk += fa.tiles_x[s] * fa.tiles_y[s] * d->B;
    }
    fa.up_first[MDX_MAX_SCALES] = nblk;
    const int ngp_blocks = (d->nscales * fa.S * d->B * 12 + NT / 64 - 1) / (NT / 64);
    hipLaunchKernelGGL(train_finish_kernel, dim3(nblk + ngp_blocks + d->nscales), dim3(NT), 0, st, fa);
    if ((rc = check_launch())) return rc;
    for (int s = 0; s < d->nscales; ++s)
        if (separate[s] && (rc = launch_upsample_bwd(gup[s], d->B, d->H, d->W, gdisp[s], d->h[s], d->w[s], st))) return rc;
    return MDX_OK;
}

}  // namespace mdx
