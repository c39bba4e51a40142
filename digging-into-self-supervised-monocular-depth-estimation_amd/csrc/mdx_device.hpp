// mdx_device.hpp -- per-pixel device math of the photometric path, gfx950.
//
// Every function states the reference line it reproduces and the OPERATION ORDER that makes the
// float32 result bit-identical to the reference's CPU (ATen) execution.  The translation unit is
// compiled with -ffp-contract=off: a fused multiply-add happens only where __builtin_fmaf is written,
// every other mul/add is rounded separately, and `/` is the correctly rounded IEEE divide.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MDX_DEV __device__ __forceinline__

namespace mdx {

// Element `i` of an array whose base pointer is wave-uniform (a plane of one image): the byte offset is
// formed in 32 bits, so the access uses the scalar-base + 32-bit-VGPR-offset form (one VGPR, no 64-bit VALU
// address arithmetic).  A plane is far below 4 GB.
template <typename T> MDX_DEV const T &at32(const T *base, unsigned i)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + i * (unsigned)sizeof(T));
}
template <typename T> MDX_DEV T &at32(T *base, unsigned i)
{
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + i * (unsigned)sizeof(T));
}

// ---------------------------------------------------------------------------------------------
// A1  bilinear upsample, align_corners=False  (warp.py:18-20 via processor.py:142)
// ---------------------------------------------------------------------------------------------
struct UpTap { int i0, i1; float l0, l1; };

MDX_DEV UpTap up_tap(float scale, int dst, int in_size)
{
    // ATen area_pixel_compute_source_index: scale*(dst+0.5)-0.5, clamped at 0
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    int a = (int)src;
    a = a > in_size - 1 ? in_size - 1 : a;
    UpTap t;
    t.i0 = a;
    t.i1 = a + (a < in_size - 1 ? 1 : 0);
    t.l1 = src - (float)a;
    t.l0 = 1.0f - t.l1;
    return t;
}

// premul: ATen's small-output kernel (H+W <= 128); else the generic separable kernel
MDX_DEV float up_combine(float v00, float v01, float v10, float v11, const UpTap &ty, const UpTap &tx,
                         bool premul)
{
    if (premul) {
        float w00 = ty.l0 * tx.l0, w01 = ty.l0 * tx.l1, w10 = ty.l1 * tx.l0, w11 = ty.l1 * tx.l1;
        float acc = w01 * v01;
        acc = __builtin_fmaf(w00, v00, acc);
        acc = __builtin_fmaf(w10, v10, acc);
        return __builtin_fmaf(w11, v11, acc);
    }
    float top = __builtin_fmaf(tx.l0, v00, tx.l1 * v01);
    float bot = __builtin_fmaf(tx.l0, v10, tx.l1 * v11);
    return __builtin_fmaf(ty.l0, top, ty.l1 * bot);
}

MDX_DEV float upsample_at(const float *__restrict__ disp, int h, int w, int H, int W, int y, int x,
                          bool premul)
{
    if (h == H && w == W) return disp[y * w + x];
    UpTap ty = up_tap((float)h / (float)H, y, h);
    UpTap tx = up_tap((float)w / (float)W, x, w);
    return up_combine(disp[ty.i0 * w + tx.i0], disp[ty.i0 * w + tx.i1], disp[ty.i1 * w + tx.i0],
                      disp[ty.i1 * w + tx.i1], ty, tx, premul);
}

// ---------------------------------------------------------------------------------------------
// A2  disparity2depth (warp.py:34-39): sd = a + b*disp (mul, then add); depth = 1/sd
// ---------------------------------------------------------------------------------------------
MDX_DEV float scaled_disp(float disp, float a, float b)
{
    float t = b * disp;
    return a + t;
}

// ---------------------------------------------------------------------------------------------
// A4  Depth2PointCloud (warp.py:238-242): r = invK[:3,:3] @ (x,y,1) as MKL's k-ordered FMA chain
// ---------------------------------------------------------------------------------------------
MDX_DEV void pixel_ray(const float *__restrict__ invK, float x, float y, float r[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float t = invK[i * 4 + 0] * x;
        t = __builtin_fmaf(invK[i * 4 + 1], y, t);
        r[i] = __builtin_fmaf(invK[i * 4 + 2], 1.0f, t);
    }
}

// ---------------------------------------------------------------------------------------------
// Correctly rounded division by a constant in three instructions instead of the ~10 of the generic
// IEEE sequence:  q = x*r;  e = fma(-b, q, x);  q' = fma(e, r, q)   with r = RN(1/b).
// tools/check_constdiv.c proves q' == x/b bit for bit for EVERY finite float x (subnormals
// included) for b = 9 and 3 (and the benchmark sizes' W-1, H-1); tools/gen_divtable.c proves it for all
// normal x for every integer 2 <= b < 4096 (csrc/mdx_divtable.inc).  Precondition: x finite.
// ---------------------------------------------------------------------------------------------
MDX_DEV float div_by_const(float x, float b, float r)
{
    const float q = x * r;
    const float e = __builtin_fmaf(-b, q, x);
    return __builtin_fmaf(e, r, q);
}
MDX_DEV float div9(float x) { return div_by_const(x, 9.0f, 1.0f / 9.0f); }
MDX_DEV float div3(float x) { return div_by_const(x, 3.0f, 1.0f / 3.0f); }

// division by the run-time constants W-1 / H-1: fast form only when the host found the divisor in the
// verified table, and only for normal finite x; everything else takes the IEEE divide
struct NormDiv { float b, r; bool fast; };

MDX_DEV NormDiv make_normdiv(int n_minus_1, bool fast)
{
    NormDiv d;
    d.b = (float)n_minus_1;
    d.r = 1.0f / d.b;
    d.fast = fast;
    return d;
}

MDX_DEV float div_norm(float x, const NormDiv &d)
{
    const float ax = fabsf(x);
    if (d.fast && ax > 1e-30f && ax < 3.0e38f) return div_by_const(x, d.b, d.r);
    return x / d.b;
}

// ---------------------------------------------------------------------------------------------
// A5  PointCloud2Pixel (warp.py:261-268)
// ---------------------------------------------------------------------------------------------
struct Proj { float u, v, z, gx, gy; };

struct Norm2 { NormDiv w, h; };

MDX_DEV Proj project_point(const float *__restrict__ P, float X0, float X1, float X2, float X3,
                           const Norm2 &nd, float eps)
{
    float q[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float t = P[i * 4 + 0] * X0;
        t = __builtin_fmaf(P[i * 4 + 1], X1, t);
        t = __builtin_fmaf(P[i * 4 + 2], X2, t);
        q[i] = __builtin_fmaf(P[i * 4 + 3], X3, t);
    }
    Proj p;
    p.z = q[2] + eps;
    p.u = q[0] / p.z;
    p.v = q[1] / p.z;
    float nx = div_norm(p.u, nd.w);
    float ny = div_norm(p.v, nd.h);
    p.gx = (nx - 0.5f) * 2.0f;
    p.gy = (ny - 0.5f) * 2.0f;
    return p;
}

// ---------------------------------------------------------------------------------------------
// A6  grid_sample bilinear / border / align_corners=True (warp.py:12-14)
// ---------------------------------------------------------------------------------------------
struct Tap {
    int x0, y0;
    float nw, ne, sw, se;
    float ix, iy;
    bool inx, iny;   // unclipped coordinate strictly inside -> gradient passes
};

MDX_DEV Tap make_tap(float gx, float gy, int H, int W)
{
    Tap t;
    float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f);
    float iy = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
    t.inx = (ix > 0.f) && (ix < (float)(W - 1));
    t.iny = (iy > 0.f) && (iy < (float)(H - 1));
    ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    float xw = floorf(ix), yn = floorf(iy);
    float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
    t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
    t.x0 = (int)xw; t.y0 = (int)yn;
    t.ix = ix; t.iy = iy;
    return t;
}

struct Corners { float nw, ne, sw, se; };

// out-of-range corners read as 0 (ATen's masked gather); x0,y0 are always in range after clipping
// The two taps of a row are adjacent floats: ONE 8-byte load per row (global_load_dwordx2; 4-byte
// alignment is enough on gfx950) instead of two dword gathers -- the memory pipeline of a CU is paced
// by load INSTRUCTIONS here, not by bytes.  The pair is anchored at min(x0, W-2) so it never leaves
// the row; the row below is clamped to H-1 and zeroed when it is out of range.  Needs W >= 2.
typedef float float2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float float3_a4 __attribute__((ext_vector_type(3), aligned(4)));

MDX_DEV Corners load_corners(const float *__restrict__ img, int H, int W, const Tap &t)
{
    const int xl = t.x0 < W - 1 ? t.x0 : W - 2;
    const bool shifted = xl != t.x0;            // x0 == W-1: the east taps are out of range
    const bool ys = t.y0 + 1 < H;
    const int y1 = ys ? t.y0 + 1 : t.y0;
    // 32-bit unsigned element offsets from a wave-uniform plane pointer: the loads then use the
    // scalar-base + 32-bit-VGPR-offset addressing form instead of a 64-bit address pair per load
    // (the BYTE offset is formed in 32 bits: only then can the compiler keep it out of 64-bit arithmetic)
    const unsigned o0 = (unsigned)(t.y0 * W + xl) * 4u, o1 = (unsigned)(y1 * W + xl) * 4u;
    const char *base = reinterpret_cast<const char *>(img);
    const float2_a4 top = *reinterpret_cast<const float2_a4 *>(base + o0);
    const float2_a4 bot = *reinterpret_cast<const float2_a4 *>(base + o1);
    Corners c;
    c.nw = shifted ? top.y : top.x;
    c.ne = shifted ? 0.f : top.y;
    c.sw = ys ? (shifted ? bot.y : bot.x) : 0.f;
    c.se = ys ? (shifted ? 0.f : bot.y) : 0.f;
    return c;
}

MDX_DEV float sample(const Corners &c, const Tap &t)
{
    float acc = c.nw * t.nw;
    acc = __builtin_fmaf(c.ne, t.ne, acc);
    acc = __builtin_fmaf(c.sw, t.sw, acc);
    return __builtin_fmaf(c.se, t.se, acc);
}

// ---------------------------------------------------------------------------------------------
// A7/A8  SSIM + L1 (model_loss.py:28-41, 97-103)
// ---------------------------------------------------------------------------------------------
#define MDX_C1 0.0001f   // 0.01 ** 2
#define MDX_C2 0.0009f   // 0.03 ** 2

MDX_DEV int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// AvgPool2d(3,1): row-major sequential sum of the nine taps, then a true divide by 9
MDX_DEV float pool9(const float v[9])
{
    float s = v[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) s = s + v[k];
    return div9(s);
}

struct TargetStats { float mu, e2, mu2; };   // mu_y, pool(y*y), mu_y*mu_y

MDX_DEV TargetStats target_stats(const float y[9])
{
    float yy[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) yy[k] = y[k] * y[k];
    TargetStats t;
    t.mu = pool9(y);
    t.e2 = pool9(yy);
    t.mu2 = t.mu * t.mu;
    return t;
}

struct SsimTerms { float mu_x, ex2, exy; };

MDX_DEV SsimTerms pred_stats(const float x[9], const float y[9])
{
    float xx[9], xy[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { xx[k] = x[k] * x[k]; xy[k] = x[k] * y[k]; }
    SsimTerms s;
    s.mu_x = pool9(x);
    s.ex2 = pool9(xx);
    s.exy = pool9(xy);
    return s;
}

// returns the un-clamped (1 - n/d)/2; the SSIM map value is clamp(raw, 0, 1)
MDX_DEV float ssim_raw(const SsimTerms &s, const TargetStats &t)
{
    float mxx = s.mu_x * s.mu_x;
    float mxy = s.mu_x * t.mu;
    float sig_x = s.ex2 - mxx;
    float sig_y = t.e2 - t.mu2;
    float sig_xy = s.exy - mxy;
    float a = 2.0f * s.mu_x;
    a = a * t.mu;
    float A1 = a + MDX_C1;
    float A2 = 2.0f * sig_xy;
    A2 = A2 + MDX_C2;
    float n = A1 * A2;
    float B1 = (mxx + t.mu2) + MDX_C1;
    float B2 = (sig_x + sig_y) + MDX_C2;
    float d = B1 * B2;
    return (1.0f - n / d) / 2.0f;
}

MDX_DEV float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// ReprojectionLoss from per-channel SSIM and |y-x|: ((c0+c1)+c2)/3 means; 0.85f*ssim + 0.15f*l1
MDX_DEV float reprojection_combine(const float ssim[3], const float ad[3])
{
    float l1 = (ad[0] + ad[1]) + ad[2];
    float ss = (ssim[0] + ssim[1]) + ssim[2];
    l1 = div3(l1);
    ss = div3(ss);
    float a = 0.85f * ss, b = 0.15f * l1;
    return a + b;
}

// closed-form d(clamped SSIM)/d(window sums) for one channel
//   grad wrt padded x at a tap with values (xq, yq): (alpha + 2*xq*beta + yq*gamma) / 9
struct SsimGrad { float alpha, beta, gamma; };

MDX_DEV SsimGrad ssim_grad(const SsimTerms &s, const TargetStats &t, float g)
{
    float mx = s.mu_x, my = t.mu;
    float sig_x = s.ex2 - mx * mx, sig_y = t.e2 - t.mu2, sig_xy = s.exy - mx * my;
    float A1 = 2.0f * mx * my + MDX_C1, A2 = 2.0f * sig_xy + MDX_C2;
    float B1 = mx * mx + t.mu2 + MDX_C1, B2 = sig_x + sig_y + MDX_C2;
    float n = A1 * A2, d = B1 * B2;
    float raw = (1.0f - n / d) * 0.5f;
    SsimGrad r;
    if (!(raw >= 0.f && raw <= 1.f)) { r.alpha = r.beta = r.gamma = 0.f; return r; }
    float inv_d = 1.0f / d;
    float Ln = -0.5f * inv_d, Ld = 0.5f * n * inv_d * inv_d;
    float dA1 = Ln * A2, dA2 = Ln * A1, dB1 = Ld * B2, dB2 = Ld * B1;
    r.alpha = g * 2.0f * (my * (dA1 - dA2) + mx * (dB1 - dB2));
    r.beta = g * dB2;
    r.gamma = g * 2.0f * dA2;
    return r;
}

// ---------------------------------------------------------------------------------------------
// reductions: wave64 shuffle tree, then LDS across the waves of the block
// ---------------------------------------------------------------------------------------------
MDX_DEV double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

MDX_DEV float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// wave64 sum that stays in the VALU (no ds_bpermute): row_shr 1/2/4/8 inside the four 16-lane rows, then
// row_bcast:15 and row_bcast:31 across rows.  The total is valid in lane 63 only.  Every lane of the wave must be
// active at the call.
MDX_DEV float wave_sum_dpp_lane63(float v)
{
#define MDX_DPP_STEP(ctrl, rowmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rowmask, 0xf, true))
    MDX_DPP_STEP(0x111, 0xf);
    MDX_DPP_STEP(0x112, 0xf);
    MDX_DPP_STEP(0x114, 0xf);
    MDX_DPP_STEP(0x118, 0xf);
    MDX_DPP_STEP(0x142, 0xa);
    MDX_DPP_STEP(0x143, 0xc);
#undef MDX_DPP_STEP
    return v;
}

}  // namespace mdx
