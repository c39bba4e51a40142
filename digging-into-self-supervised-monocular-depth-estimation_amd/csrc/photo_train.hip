// photo_train.hip -- the TRAINING form of the photometric path for gfx950 (MI355X): forward AND gradient of every
// scale of
//   compute.image2warping  (model_tool/processor.py:139-163)  and the photometric half of
//   compute.compute_loss   (model_tool/processor.py:167-204,212)
// in ONE launch (SURVEY 8f N4 + the "gradient in the forward" lever): everything downstream of sum(to_optimise) is
// linear in the upstream gradient, so the kernel emits, besides idx and the loss partials, the gradient for a UNIT
// upstream (d sum / d upsampled disparity, d sum / d P); autograd's backward only scales them.
//
// Decomposition: one wave64 per work item = (scale, image, chunk of R rows, strip of 60 columns).  No workgroup
// barriers, no shared tiles:
//   * lane l owns column c0 + l - 2 of the strip (2 halo columns each side: warp on 64, SSIM on 62, gradient on 60);
//   * the wave marches down the rows; per row it (1) warps row r+2, (2) evaluates SSIM / L1 / min / arg-min and the
//     SSIM coefficient triplets of row r+1, (3) gathers the 3x3 coefficient sums and runs the
//     grid_sample -> projection -> depth chain of row r.  The three rows of history every stage needs live in
//     REGISTERS of the lane (vertical neighbours) and are read from the NEIGHBOUR LANES through DPP operands
//     (wave_shr:1 / wave_shl:1; horizontal neighbours): the 3x3 windows never touch LDS or HBM;
//   * row index, tile origin, image, scale are wave-uniform: address arithmetic is scalar, a lane's column-dependent
//     terms (bilinear x taps of the disparity, the x part of the pixel ray, reflection) are computed once per item;
//   * LDS is only a per-lane stash ring (3 rows) for the sampling derivatives a row's gradient needs two rows later.
// Arithmetic and its order are those of photo_fwd.hip / mdx_device.hpp (bit-exact per-pixel values, same arg-min).
#include "photo_common.hpp"

namespace mdx {

constexpr int SW = 60;   // output columns per wave (64 lanes - 2 x 2 halo lanes)

struct TrainArgs {
    int B, H, W, S, nscales;
    unsigned flags;
    float disp_a, disp_b;
    int h[MDX_MAX_SCALES], w[MDX_MAX_SCALES];
    int rows, nchunks, nstrips;
    const float *disp[MDX_MAX_SCALES];
    const float *P[MDX_MAX_SCALES];
    const float *noise[MDX_MAX_SCALES];
    uint8_t *idx[MDX_MAX_SCALES];
    float *gup[MDX_MAX_SCALES];
    float *to_opt[MDX_MAX_SCALES];
    const float *target, *ident, *invK;
    mdx_sources src;
    float *depth0;
    double *loss_part;   // [items]
    float *partP;        // [items][S][12]
};

// wave-uniform selection from a by-value kernel-argument array without dynamic indexing (which would send the
// argument block to scratch)
template <typename T> MDX_DEV T pick(const T (&v)[MDX_MAX_SCALES], int s)
{
    return s == 0 ? v[0] : (s == 1 ? v[1] : (s == 2 ? v[2] : v[3]));
}

template <int CTRL> MDX_DEV float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> MDX_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// value held by the lane to the left / right (0 at the ends of the wave).  EVERY lane must be active.
MDX_DEV float from_left(float v) { return dpp_f<0x138>(v); }    // wave_shr:1
MDX_DEV float from_right(float v) { return dpp_f<0x130>(v); }   // wave_shl:1

// AvgPool2d(3,1) at this lane's column over history rows (a0, a1, a2): the nine taps summed row-major,
// sequentially, then a true divide by 9 -- pool9()'s order with the side columns taken from the neighbour lanes.
MDX_DEV float pool3(float a0, float a1, float a2)
{
    float s = from_left(a0) + a0;
    s = s + from_right(a0);
    s = s + from_left(a1);
    s = s + a1;
    s = s + from_right(a1);
    s = s + from_left(a2);
    s = s + a2;
    s = s + from_right(a2);
    return div9(s);
}

template <int S>
__global__ __launch_bounds__(64) void photometric_train_kernel(TrainArgs a)
{
    // per-lane stash ring: [row slot][2f] = (d colour_c / du, u), [2f+1] = (d colour_c / dv, v) of frame f
    __shared__ float4 s_stash[3][2 * S][64];

    const int lane = threadIdx.x;
    // ---- work item, XCD-aware (photo_common.hpp tile_id): the blocks of one XCD walk a contiguous run of items ----
    const unsigned nwg = gridDim.x, orig = blockIdx.x;
    const unsigned qq = nwg / 8, rr = nwg % 8, xcd = orig % 8;
    const unsigned item = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + orig / 8;
    const int strip = (int)(item % (unsigned)a.nstrips);
    unsigned rest = item / (unsigned)a.nstrips;
    const int chunk = (int)(rest % (unsigned)a.nchunks);
    rest /= (unsigned)a.nchunks;
    const int b = (int)(rest % (unsigned)a.B);
    const int scale = (int)(rest / (unsigned)a.B);

    mdx_desc d;
    d.B = a.B; d.H = a.H; d.W = a.W; d.S = S; d.flags = a.flags; d.disp_a = a.disp_a; d.disp_b = a.disp_b;
    d.h = pick(a.h, scale);
    d.w = pick(a.w, scale);
    const int H = d.H, W = d.W;
    const size_t HW = (size_t)H * W;
    const float *disp_b = pick(a.disp, scale) + (size_t)b * d.h * d.w;
    const float *P_s = pick(a.P, scale);
    const float *noise_s = pick(a.noise, scale);
    uint8_t *idx_s = pick(a.idx, scale);
    float *gup_s = pick(a.gup, scale);
    float *to_opt_s = pick(a.to_opt, scale);
    const float *invK_b = a.invK + b * 16;
    const float *tgt_b = a.target + (size_t)b * 3 * HW;
    const bool automask = (d.flags & MDX_FLAG_AUTOMASK) != 0;
    const bool premul = (d.flags & MDX_FLAG_UPSAMPLE_PREMUL) != 0;
    const bool same_res = d.h == H && d.w == W;
    const Norm2 nd = desc_norm(d);

    // ---- per-lane constants: the column ----
    const int col = strip * SW + lane - 2;
    const int colc = min(max(col, -1), W);           // beyond the reflected ring: clamped, never used
    const int pxr = reflect(colc, W);
    const bool col_img = col >= 0 && col < W;
    const bool out_lane = lane >= 2 && lane < 2 + SW && col < W;
    const bool ssim_lane = col_img && lane >= 1 && lane <= 62;   // both neighbour lanes exist
    const float wx0 = col == 1 ? 2.f : 1.f, wx2 = col == W - 2 ? 2.f : 1.f;   // reflection-pad fold (x)
    const UpTap tx = up_tap((float)d.w / (float)W, pxr, d.w);
    const float fpx = (float)pxr;

    const int r0 = chunk * a.rows, r1 = min(r0 + a.rows, H);

    // ---- histories (registers) ----
    float xh[3][S][3];   // warped colours, rows wr-2 .. wr
    float yh[3][3];      // target colours, same rows
    float ch[3][3][3];   // [row][channel][alpha,beta,gamma] of the arg-min frame, rows sr-2 .. sr
    int selh[3];         // arg-min frame (or -1), same rows
    int flh[3];          // grid_sample pass flags (bit 2f: x inside, 2f+1: y inside), rows wr-2 .. wr
    float dph[3];        // depth, rows wr-2 .. wr
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        selh[j] = -1; flh[j] = 0; dph[j] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            yh[j][c] = 0.f;
#pragma unroll
            for (int f = 0; f < S; ++f) xh[j][f][c] = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) ch[j][c][k] = 0.f;
        }
    }
    float accP[S][12];
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int k = 0; k < 12; ++k) accP[f][k] = 0.f;
    double acc = 0.0;

    const int nsteps = (r1 - r0) + 4;
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        const int wr = r0 - 2 + t;        // row warped in this step
        const int sr = wr - 1;            // row whose SSIM / arg-min / coefficients are formed
        const int gr = wr - 2;            // row whose gradient is formed
        const int slot_w = t % 3, slot_r = (t + 1) % 3;

        // ================= (1) warp row wr =================
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            yh[0][c] = yh[1][c]; yh[1][c] = yh[2][c];
#pragma unroll
            for (int f = 0; f < S; ++f) { xh[0][f][c] = xh[1][f][c]; xh[1][f][c] = xh[2][f][c]; }
        }
        flh[0] = flh[1]; flh[1] = flh[2];
        dph[0] = dph[1]; dph[1] = dph[2];
        if (wr >= -1 && wr <= H) {
            const int pyr = reflect(wr, H);
            const unsigned po = (unsigned)(pyr * W + pxr);
#pragma unroll
            for (int c = 0; c < 3; ++c) yh[2][c] = at32(tgt_b + c * HW, po);
            float up;
            if (same_res) {
                up = at32(disp_b, po);
            } else {
                const UpTap ty = up_tap((float)d.h / (float)H, pyr, d.h);
                const float *row0 = disp_b + ty.i0 * d.w, *row1 = disp_b + ty.i1 * d.w;
                up = up_combine(row0[tx.i0], row0[tx.i1], row1[tx.i0], row1[tx.i1], ty, tx, premul);
            }
            const PixelGeom g = geom_from_disp(d, up, invK_b, pxr, pyr);
            dph[2] = g.depth;
            if (a.depth0 && scale == 0 && out_lane && wr >= r0 && wr < r1) at32(a.depth0 + (size_t)b * HW, po) = g.depth;
            int fl = 0;
#pragma unroll
            for (int f = 0; f < S; ++f) {
                const float *Pf = P_s + ((size_t)f * d.B + b) * 12;
                const Proj pr = project_point(Pf, g.X0, g.X1, g.X2, 1.0f, nd, 1e-7f);
                const Tap tp = make_tap(pr.gx, pr.gy, H, W);
                const float *src_b = a.src.img[f] + (size_t)b * 3 * HW;
                Corners cn[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) cn[c] = load_corners(src_b + c * HW, H, W, tp);
                const float dy1 = (float)(tp.y0 + 1) - tp.iy, dy0 = tp.iy - (float)tp.y0;
                const float dx1 = (float)(tp.x0 + 1) - tp.ix, dx0 = tp.ix - (float)tp.x0;
                float du[3], dv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    xh[2][f][c] = sample(cn[c], tp);
                    du[c] = (cn[c].ne - cn[c].nw) * dy1 + (cn[c].se - cn[c].sw) * dy0;
                    dv[c] = (cn[c].sw - cn[c].nw) * dx1 + (cn[c].se - cn[c].ne) * dx0;
                }
                s_stash[slot_w][2 * f][lane] = make_float4(du[0], du[1], du[2], pr.u);
                s_stash[slot_w][2 * f + 1][lane] = make_float4(dv[0], dv[1], dv[2], pr.v);
                fl |= (tp.inx ? 1 : 0) << (2 * f);
                fl |= (tp.iny ? 1 : 0) << (2 * f + 1);
            }
            flh[2] = fl;
        }

        // ================= (2) SSIM + L1, min / arg-min, coefficient triplets of row sr =================
        if (t < 2) continue;
        selh[0] = selh[1]; selh[1] = selh[2];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) { ch[0][c][k] = ch[1][c][k]; ch[1][c][k] = ch[2][c][k]; }
        if (sr >= 0 && sr < H) {
            const unsigned po = (unsigned)(sr * W + pxr);
            float idv[S], nzv[S];
            if (automask) {
#pragma unroll
                for (int f = 0; f < S; ++f) {
                    idv[f] = at32(a.ident + ((size_t)b * S + f) * HW, po);
                    nzv[f] = at32(noise_s + ((size_t)b * S + f) * HW, po);
                }
            }
            TargetStats ts[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float y0 = yh[0][c], y1 = yh[1][c], y2 = yh[2][c];
                ts[c].mu = pool3(y0, y1, y2);
                ts[c].e2 = pool3(y0 * y0, y1 * y1, y2 * y2);
                ts[c].mu2 = ts[c].mu * ts[c].mu;
            }
            float rl[S];
            SsimTerms st[S][3];
#pragma unroll
            for (int f = 0; f < S; ++f) {
                float ss[3], ad[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float x0 = xh[0][f][c], x1 = xh[1][f][c], x2 = xh[2][f][c];
                    st[f][c].mu_x = pool3(x0, x1, x2);
                    st[f][c].ex2 = pool3(x0 * x0, x1 * x1, x2 * x2);
                    st[f][c].exy = pool3(x0 * yh[0][c], x1 * yh[1][c], x2 * yh[2][c]);
                    ss[c] = clamp01(ssim_raw(st[f][c], ts[c]));
                    ad[c] = fabsf(yh[1][c] - x1);
                }
                rl[f] = reprojection_combine(ss, ad);
            }
            // concat [ident + 1e-5*noise, reproj] and torch.min's first-minimum rule (processor.py:194-204)
            float best = 0.f;
            int bi = 0;
            if (automask) {
#pragma unroll
                for (int f = 0; f < S; ++f) {
                    const float tn = 1e-5f * nzv[f];
                    const float v = idv[f] + tn;
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
            int fsel = automask ? bi - S : bi;
            if (!ssim_lane) fsel = -1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                SsimTerms sel = st[0][c];
#pragma unroll
                for (int f = 1; f < S; ++f)
                    if (fsel == f) sel = st[f][c];
                SsimGrad sg = ssim_grad(sel, ts[c], 0.85f / 3.0f);
                const bool keep = fsel >= 0;
                ch[2][c][0] = keep ? sg.alpha : 0.f;
                ch[2][c][1] = keep ? sg.beta : 0.f;
                ch[2][c][2] = keep ? sg.gamma : 0.f;
            }
            selh[2] = fsel;
            if (out_lane && sr >= r0 && sr < r1) {
                at32(idx_s + (size_t)b * HW, po) = (uint8_t)bi;
                if (to_opt_s) at32(to_opt_s + (size_t)b * HW, po) = best;
                acc += (double)best;
            }
        } else {
            selh[2] = -1;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) ch[2][c][k] = 0.f;
        }

        // ================= (3) gradient of row gr =================
        if (t < 4) continue;
        const float wy0 = gr == 1 ? 2.f : 1.f, wy2 = gr == H - 2 ? 2.f : 1.f;   // reflection-pad fold (y)
        float r3[3];
        pixel_ray(invK_b, fpx, (float)gr, r3);
        const float depth = dph[0];
        const float X[4] = {depth * r3[0], depth * r3[1], depth * r3[2], 1.0f};
        float gdepth = 0.f;
#pragma unroll
        for (int f = 0; f < S; ++f) {
            const int own = (selh[0] == f) | (selh[1] == f) | (selh[2] == f);
            const int hit = own | dpp_i<0x138>(own) | dpp_i<0x130>(own);
            if (__builtin_amdgcn_ballot_w64(hit != 0 && out_lane) == 0) continue;   // wave-uniform
            const float w0 = selh[0] == f ? wy0 : 0.f, w1 = selh[1] == f ? 1.f : 0.f, w2 = selh[2] == f ? wy2 : 0.f;
            const float4 sa = s_stash[slot_r][2 * f][lane], sb = s_stash[slot_r][2 * f + 1][lane];
            const float du[3] = {sa.x, sa.y, sa.z}, dv[3] = {sb.x, sb.y, sb.z};
            const bool centre = selh[1] == f;
            float gu = 0.f, gv = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float sum3[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // vertical (own column, registers) then horizontal (neighbour lanes) 3-tap sums
                    float v = w0 * ch[0][c][k];
                    v = __builtin_fmaf(w1, ch[1][c][k], v);
                    v = __builtin_fmaf(w2, ch[2][c][k], v);
                    float s = __builtin_fmaf(from_left(v), wx0, v);
                    sum3[k] = __builtin_fmaf(from_right(v), wx2, s);
                }
                const float xq = xh[0][f][c], yq = yh[0][c];
                float gx = (sum3[0] + 2.0f * xq * sum3[1] + yq * sum3[2]) * (1.0f / 9.0f);
                if (centre) gx -= 0.05f * ((yq > xq) ? 1.f : ((yq < xq) ? -1.f : 0.f));   // 0.15*mean_c|y-x|
                gu += gx * du[c];
                gv += gx * dv[c];
            }
            // grid normalisation (2/(W-1)) and grid_sample's un-normalisation ((W-1)/2) cancel
            gu = (((flh[0] >> (2 * f)) & 1) && out_lane) ? gu : 0.f;
            gv = (((flh[0] >> (2 * f + 1)) & 1) && out_lane) ? gv : 0.f;
            const float *Pf = P_s + ((size_t)f * d.B + b) * 12;
            float z = Pf[8] * X[0];
            z = __builtin_fmaf(Pf[9], X[1], z);
            z = __builtin_fmaf(Pf[10], X[2], z);
            z = __builtin_fmaf(Pf[11], 1.0f, z) + 1e-7f;
            const float iz = __builtin_amdgcn_rcpf(z);
            const float gq0 = gu * iz, gq1 = gv * iz, gq2 = -(gu * sa.w + gv * sb.w) * iz;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float gX = gq0 * Pf[j] + gq1 * Pf[4 + j] + gq2 * Pf[8 + j];
                gdepth += gX * r3[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accP[f][j] += gq0 * X[j];
                accP[f][4 + j] += gq1 * X[j];
                accP[f][8 + j] += gq2 * X[j];
            }
        }
        // depth = 1/(a + b*disp)  ->  d depth / d disp = -b * depth^2
        if (out_lane) at32(gup_s + (size_t)b * HW, (unsigned)(gr * W + pxr)) = gdepth * (-d.disp_b * depth * depth);
    }

    // ---- per-item partials: d(P) (register reduction, total in lane 63) and the loss sum ----
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float tot = wave_sum_dpp_lane63(accP[f][k]);
            if (lane == 63) a.partP[(size_t)item * (S * 12) + f * 12 + k] = tot;
        }
    acc = wave_sum(acc);
    if (lane == 0) a.loss_part[item] = acc;
}

// Second pass, fixed order (deterministic): one wave64 per output.
//   blocks [0, nscales*S*B*12): gP[scale][f][b][k] = sum over the items of (scale, b)
//   blocks [nscales*S*B*12, +nscales): loss_sum[scale]
__global__ __launch_bounds__(64) void train_finish_kernel(const float *__restrict__ partP,
                                                          const double *__restrict__ loss_part, int nscales, int S,
                                                          int B, int ipi, float *__restrict__ gP,
                                                          float *__restrict__ loss_sum)
{
    const int i = blockIdx.x, ngp = nscales * S * B * 12;
    double acc = 0.0;
    if (i < ngp) {
        const int k = i % 12, bb = (i / 12) % B, f = (i / (12 * B)) % S, sc = i / (12 * B * S);
        const float *p = partP + ((size_t)(sc * B + bb) * ipi) * (S * 12) + f * 12 + k;
        for (int t = threadIdx.x; t < ipi; t += 64) acc += (double)p[(size_t)t * (S * 12)];
        acc = wave_sum(acc);
        if (threadIdx.x == 0) gP[i] = (float)acc;
    } else {
        const int sc = i - ngp;
        const double *p = loss_part + (size_t)sc * B * ipi;
        for (int t = threadIdx.x; t < B * ipi; t += 64) acc += p[t];
        acc = wave_sum(acc);
        if (threadIdx.x == 0) loss_sum[sc] = (float)acc;
    }
}

static int default_rows(const mdx_train_desc *d)
{
    // enough items to fill the 256 CUs a few waves deep, few enough that the 4 halo rows of a chunk stay cheap
    const int nstrips = (d->W + SW - 1) / SW;
    int rows = 32;
    while (rows > 8 && (long long)d->nscales * d->B * ((d->H + rows - 1) / rows) * nstrips < 6144) rows /= 2;
    return rows;
}

struct TrainPlan { int rows, nchunks, nstrips; size_t items, off_partP, off_gup, total; };

static TrainPlan plan(const mdx_train_desc *d)
{
    TrainPlan p;
    p.rows = d->rows_per_chunk > 0 ? d->rows_per_chunk : default_rows(d);
    p.nchunks = (d->H + p.rows - 1) / p.rows;
    p.nstrips = (d->W + SW - 1) / SW;
    p.items = (size_t)d->nscales * d->B * p.nchunks * p.nstrips;
    p.off_partP = p.items * sizeof(double);
    p.off_gup = p.off_partP + ((p.items * d->S * 12 * sizeof(float) + 15) & ~(size_t)15);
    p.total = p.off_gup + (size_t)d->nscales * d->B * d->H * d->W * sizeof(float);
    return p;
}

static int validate_train(const mdx_train_desc *d)
{
    if (!d) return MDX_ERR_NULL_POINTER;
    if (d->nscales < 1 || d->nscales > MDX_MAX_SCALES) return MDX_ERR_BAD_SHAPE;
    if (d->rows_per_chunk < 0) return MDX_ERR_BAD_SHAPE;
    for (int s = 0; s < d->nscales; ++s) {
        mdx_desc one = {d->B, d->H, d->W, d->h[s], d->w[s], d->S, d->flags, d->disp_a, d->disp_b};
        const int rc = validate_desc(&one);
        if (rc) return rc;
    }
    return MDX_OK;
}

int launch_upsample_bwd(const float *gout, int BC, int H, int W, float *gin, int h, int w, hipStream_t st);

}  // namespace mdx

using namespace mdx;

MDX_EXPORT int mdx_train_desc_init(mdx_train_desc *d, int B, int H, int W, int S, int nscales, const int32_t *h,
                                   const int32_t *w, int automask, double min_depth, double max_depth,
                                   int rows_per_chunk)
{
    if (!d || !h || !w) return MDX_ERR_NULL_POINTER;
    if (nscales < 1 || nscales > MDX_MAX_SCALES) return MDX_ERR_BAD_SHAPE;
    mdx_desc one;
    const int rc = mdx_desc_init(&one, B, H, W, h[0], w[0], S, automask, min_depth, max_depth);
    if (rc) return rc;
    d->B = B; d->H = H; d->W = W; d->S = S; d->nscales = nscales; d->flags = one.flags;
    d->disp_a = one.disp_a; d->disp_b = one.disp_b; d->rows_per_chunk = rows_per_chunk;
    for (int s = 0; s < MDX_MAX_SCALES; ++s) { d->h[s] = s < nscales ? h[s] : 0; d->w[s] = s < nscales ? w[s] : 0; }
    return validate_train(d);
}

MDX_EXPORT size_t mdx_photometric_train_workspace_bytes(const mdx_train_desc *d)
{
    return validate_train(d) ? 0 : plan(d).total;
}

MDX_EXPORT int mdx_photometric_train(const mdx_train_desc *d, const float *const *disp, const float *target,
                                     const mdx_sources *src, const float *invK, const float *const *P,
                                     const float *ident, const float *const *noise, uint8_t *const *idx,
                                     float *loss_sum, float *const *gdisp, float *gP, float *depth0,
                                     float *const *to_opt, void *workspace, size_t workspace_bytes, void *stream,
                                     const mdx_timing *t)
{
    int rc = validate_train(d);
    if (rc) return rc;
    if (!disp || !target || !src || !invK || !P || !idx || !loss_sum || !gdisp || !gP) return MDX_ERR_NULL_POINTER;
    const bool automask = (d->flags & MDX_FLAG_AUTOMASK) != 0;
    if (automask && (!ident || !noise)) return MDX_ERR_NULL_POINTER;
    for (int f = 0; f < d->S; ++f)
        if (!src->img[f]) return MDX_ERR_NULL_POINTER;
    const TrainPlan p = plan(d);
    if (!workspace || workspace_bytes < p.total) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 16)) return MDX_ERR_MISALIGNED;
    if (p.items >= (1ull << 31)) return MDX_ERR_BAD_SHAPE;
    TrainArgs a = {};
    a.B = d->B; a.H = d->H; a.W = d->W; a.S = d->S; a.nscales = d->nscales; a.flags = d->flags;
    a.disp_a = d->disp_a; a.disp_b = d->disp_b;
    a.rows = p.rows; a.nchunks = p.nchunks; a.nstrips = p.nstrips;
    a.target = target; a.ident = ident; a.invK = invK; a.src = *src; a.depth0 = depth0;
    a.loss_part = (double *)workspace;
    a.partP = (float *)((char *)workspace + p.off_partP);
    float *gup_ws = (float *)((char *)workspace + p.off_gup);
    const size_t n = (size_t)d->B * d->H * d->W;
    for (int s = 0; s < d->nscales; ++s) {
        if (!disp[s] || !P[s] || !idx[s] || !gdisp[s] || (automask && !noise[s])) return MDX_ERR_NULL_POINTER;
        a.h[s] = d->h[s]; a.w[s] = d->w[s];
        a.disp[s] = disp[s]; a.P[s] = P[s]; a.noise[s] = automask ? noise[s] : nullptr; a.idx[s] = idx[s];
        a.to_opt[s] = to_opt ? to_opt[s] : nullptr;
        const bool same = d->h[s] == d->H && d->w[s] == d->W;
        a.gup[s] = same ? gdisp[s] : gup_ws + s * n;
    }
    for (int s = d->nscales; s < MDX_MAX_SCALES; ++s) {   // never selected; keep the picks well defined
        a.h[s] = a.h[0]; a.w[s] = a.w[0]; a.disp[s] = a.disp[0]; a.P[s] = a.P[0]; a.noise[s] = a.noise[0];
        a.idx[s] = a.idx[0]; a.gup[s] = a.gup[0]; a.to_opt[s] = a.to_opt[0];
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)p.items), block(64);
    if (t && t->start) (void)hipEventRecord((hipEvent_t)t->start, st);
    switch (d->S) {
    case 1: hipLaunchKernelGGL((photometric_train_kernel<1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((photometric_train_kernel<2>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((photometric_train_kernel<3>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((photometric_train_kernel<4>), grid, block, 0, st, a); break;
    default: return MDX_ERR_BAD_SHAPE;
    }
    if (t && t->stop) (void)hipEventRecord((hipEvent_t)t->stop, st);
    if ((rc = check_launch())) return rc;
    const int ngp = d->nscales * d->S * d->B * 12;
    hipLaunchKernelGGL(train_finish_kernel, dim3(ngp + d->nscales), dim3(64), 0, st, a.partP, a.loss_part, d->nscales,
                       d->S, d->B, p.nchunks * p.nstrips, gP, loss_sum);
    if ((rc = check_launch())) return rc;
    for (int s = 0; s < d->nscales; ++s) {
        const bool same = d->h[s] == d->H && d->w[s] == d->W;
        if (!same && (rc = launch_upsample_bwd(a.gup[s], d->B, d->H, d->W, gdisp[s], d->h[s], d->w[s], st))) return rc;
    }
    return MDX_OK;
}
