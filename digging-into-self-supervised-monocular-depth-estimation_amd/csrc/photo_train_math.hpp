// photo_train_math.hpp -- the lane-level building blocks of the marching-wave training kernel (photo_train.hip), gfx950:
// neighbour-lane reads through DPP operands, the 3x3 pools, the SSIM value (bit-exact, three channels in lock step) and
// its gradient coefficients, the geometry's written-out divisions, and the item's matrices as transient scalars.
// Arithmetic and its order are those of mdx_device.hpp (the per-pixel values are bit-identical to the per-scale kernels').
#pragma once
#include "photo_train.hpp"

namespace mdx {

template <int CTRL> MDX_DEV float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> MDX_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// value held by the lane to the left / right (0 at the ends of the wave).  EVERY lane must be active.
MDX_DEV float from_left(float v) { return dpp_f<0x138>(v); }    // wave_shr:1
MDX_DEV float from_right(float v) { return dpp_f<0x130>(v); }   // wave_shl:1

// AvgPool2d(3,1) at this lane's column for N independent quantities at once; a[i][j] = quantity i at history row j.
// Per quantity the nine taps are summed row-major, sequentially, then truly divided by 9 -- pool9()'s order, with
// the side columns read from the neighbour lanes.  The N chains advance in lock step: a DPP instruction that reads a
// register written by one of the two preceding VALU instructions costs wait states (s_nop), and a VALU instruction
// that consumes its predecessor's result issues at half rate (profiles/r02_micro_valu_dep.txt); with N >= 3 chains
// interleaved neither happens.
template <int N> MDX_DEV void pool3_n(const float (&a)[N][3], float (&out)[N])
{
    float s[N];
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = from_left(a[i][0]) + a[i][0];
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = s[i] + from_right(a[i][0]);
#pragma unroll
    for (int j = 1; j < 3; ++j) {
#pragma unroll
        for (int i = 0; i < N; ++i) s[i] = s[i] + from_left(a[i][j]);
#pragma unroll
        for (int i = 0; i < N; ++i) s[i] = s[i] + a[i][j];
#pragma unroll
        for (int i = 0; i < N; ++i) s[i] = s[i] + from_right(a[i][j]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = div9(s[i]);
}

// n / d and 1 / d together.  The correctly rounded float32 quotient as the compiler expands `n / d` for gfx950 is
//   s = div_scale(...); r0 = rcp(d_s); e0 = fma(-d_s, r0, 1); r1 = fma(e0, r0, r0); q0 = n_s * r1;
//   e1 = fma(-d_s, q0, n_s); q1 = fma(e1, r1, q0); e2 = fma(-d_s, q1, n_s); q = div_fixup(div_fmas(e2, r1, q1))
// (11 instructions), where div_scale / div_fmas / div_fixup only act when an operand or the quotient is near the ends
// of the exponent range (|d| or |n / d| beyond 2^+-96, subnormals, infinities).  Written out without those three -- the
// SAME operations on the same values whenever no scaling applies, hence the same bits -- the sequence is 8 instructions and
// leaves r1 = 1/d to 1 ulp, which the gradient coefficients need anyway (they used a second rcp + Newton step).
// Domain here: d = B1 * B2 >= C1 * C2 = 9e-8, |n| = |A1 * A2| is 0 or >= 1e-4 * 2^-34 for colours in [0, 255]
// (tools/check_fastdiv.hip compares 2^32 pairs of that domain against `/` on the GPU: profiles/r03_fastdiv_check.txt).
struct QuotRcp { float q, r; };
MDX_DEV QuotRcp quot_rcp(float n, float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = n * r1;
    const float e1 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-d, q1, n);
    QuotRcp o;
    o.q = __builtin_fmaf(e2, r1, q1);
    o.r = r1;
    return o;
}

// the target's window statistics as the SSIM quotient consumes them: sig_y = pool(y*y) - mu_y^2 (ssim_raw()'s own
// subtraction, formed here once per pixel -- or once per STEP by photo_prologue.hip)
struct TStat { float mu, mu2, sig_y; };

// SSIM of one colour channel of one frame in two halves.  The VALUE now: ssim_raw()'s operations in ssim_raw()'s order
// (bit-exact), keeping the intermediates the gradient needs.  The COEFFICIENT triplet of SURVEY appendix A.1 later, only in
// waves where the frame can still be some lane's arg-min (in auto-masked regions never): it re-uses those intermediates
// and quot_rcp's refined reciprocal (gradients carry a 1e-4 tolerance, not bit-exactness).
struct SsimMid { float A1, A2, B1, B2, q, inv_d, mu_x, raw; };

// The value for the THREE colour channels of a frame in lock step: every operation of ssim_raw() (mdx_device.hpp) for channel
// 0, 1, 2 in turn (the same operations on the same values: the same bits), the quotient by quot_rcp()'s sequence.  A VALU instruction that reads its predecessor's result
// costs ~2 extra cycles which other waves do not fill (profiles/r02_micro_valu_dep.txt); one channel's value is a chain of
// ~25 such instructions (the 8 of the quotient back to back), three channels interleaved have none.  The statements are
// written round-robin and pinned with MDX_LOCKSTEP (an empty asm the three values pass through): the scheduler, which
// orders for register pressure here, would otherwise put each chain back together.
#define MDX_LOCKSTEP(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
MDX_DEV void ssim_value_mid3(const float (&o)[3][3], const TStat (&t)[3], SsimMid (&m)[3], float (&val)[3])
{
    float mxx[3], mxy[3], sig_x[3], sig_xy[3], a[3], n[3], d[3], r0[3], e0[3], r1[3], q0[3], e1[3], q1[3], e2[3];
#define MDX_EACH(stmt) { const int c = 0; stmt; } { const int c = 1; stmt; } { const int c = 2; stmt; }
    MDX_EACH(mxx[c] = o[c][0] * o[c][0])
    MDX_EACH(mxy[c] = o[c][0] * t[c].mu)
    MDX_EACH(sig_x[c] = o[c][1] - mxx[c])
    MDX_EACH(sig_xy[c] = o[c][2] - mxy[c])
    MDX_EACH(a[c] = 2.0f * o[c][0])
    MDX_EACH(a[c] = a[c] * t[c].mu)
    MDX_LOCKSTEP(a[0], a[1], a[2]);
    MDX_EACH(m[c].A1 = a[c] + MDX_C1)
    MDX_EACH(m[c].A2 = 2.0f * sig_xy[c])
    MDX_EACH(m[c].A2 = m[c].A2 + MDX_C2)
    MDX_EACH(m[c].B1 = (mxx[c] + t[c].mu2))
    MDX_EACH(m[c].B1 = m[c].B1 + MDX_C1)
    MDX_EACH(m[c].B2 = (sig_x[c] + t[c].sig_y))
    MDX_EACH(m[c].B2 = m[c].B2 + MDX_C2)
    MDX_EACH(n[c] = m[c].A1 * m[c].A2)
    MDX_EACH(d[c] = m[c].B1 * m[c].B2)
    MDX_LOCKSTEP(d[0], d[1], d[2]);
    // quot_rcp(n, d), three at a time
    MDX_EACH(r0[c] = __builtin_amdgcn_rcpf(d[c]))
    MDX_LOCKSTEP(r0[0], r0[1], r0[2]);
    MDX_EACH(e0[c] = __builtin_fmaf(-d[c], r0[c], 1.0f))
    MDX_LOCKSTEP(e0[0], e0[1], e0[2]);
    MDX_EACH(r1[c] = __builtin_fmaf(e0[c], r0[c], r0[c]))
    MDX_LOCKSTEP(r1[0], r1[1], r1[2]);
    MDX_EACH(q0[c] = n[c] * r1[c])
    MDX_LOCKSTEP(q0[0], q0[1], q0[2]);
    MDX_EACH(e1[c] = __builtin_fmaf(-d[c], q0[c], n[c]))
    MDX_LOCKSTEP(e1[0], e1[1], e1[2]);
    MDX_EACH(q1[c] = __builtin_fmaf(e1[c], r1[c], q0[c]))
    MDX_LOCKSTEP(q1[0], q1[1], q1[2]);
    MDX_EACH(e2[c] = __builtin_fmaf(-d[c], q1[c], n[c]))
    MDX_LOCKSTEP(e2[0], e2[1], e2[2]);
    MDX_EACH(m[c].q = __builtin_fmaf(e2[c], r1[c], q1[c]))
    MDX_LOCKSTEP(m[0].q, m[1].q, m[2].q);
    MDX_EACH(m[c].inv_d = r1[c])
    MDX_EACH(m[c].mu_x = o[c][0])
    MDX_EACH(m[c].raw = 1.0f - m[c].q)
    MDX_LOCKSTEP(m[0].raw, m[1].raw, m[2].raw);
    MDX_EACH(m[c].raw = m[c].raw / 2.0f)
    MDX_LOCKSTEP(m[0].raw, m[1].raw, m[2].raw);
    MDX_EACH(val[c] = clamp01(m[c].raw))
#undef MDX_EACH
}

// gradient-only arithmetic may fuse multiply-adds (tolerance 1e-4; the VALUES stay unfused: -ffp-contract=off)
#define MDX_GRAD_FP _Pragma("clang fp contract(fast)")
MDX_DEV SsimGrad ssim_coef_mid(const SsimMid &m, const TStat &t, float gscale)
{
    MDX_GRAD_FP
    const float Ln = -0.5f * m.inv_d, Ld = 0.5f * m.q * m.inv_d;
    const float dA1 = Ln * m.A2, dA2 = Ln * m.A1, dB1 = Ld * m.B2, dB2 = Ld * m.B1;
    const bool pass = m.raw >= 0.f && m.raw <= 1.f;   // clamp passes the gradient on the closed interval
    const float gs = pass ? gscale : 0.f;
    SsimGrad g;
    g.alpha = gs * 2.0f * (t.mu * (dA1 - dA2) + m.mu_x * (dB1 - dB2));
    g.beta = gs * dB2;
    g.gamma = gs * 2.0f * dA2;
    return g;
}

// ---- the geometry's divisions with the same written-out sequence ----
// depth = 1 / sd, and u = q0 / z, v = q1 / z sharing ONE refined reciprocal of z (22 -> 11 instructions per frame).
// Bit-equal to the IEEE `/` whenever no operand needs div_scale's rescaling; a zero, subnormal, infinite or NaN
// divisor anywhere in the wave sends the whole wave through `/` (one v_cmp_class + a scalar branch that is never taken
// on real data: z = q2 + 1e-7 is 0 or >= 2^-47 in magnitude).
MDX_DEV bool wave_all_normal(float v)
{
    return __builtin_amdgcn_ballot_w64(__builtin_amdgcn_classf(v, 0x2F7)) == 0;   // anything but +-normal (classf: the float form)
}
MDX_DEV float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
}
MDX_DEV float quot_with(float n, float d, float r1)
{
    const float q0 = n * r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r1, q1);
}

// project_point() (mdx_device.hpp) with the shared-reciprocal divisions
MDX_DEV Proj project_point_train(const float *__restrict__ P, float X0, float X1, float X2, const Norm2 &nd, float eps)
{
    float q[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float t = P[i * 4 + 0] * X0;
        t = __builtin_fmaf(P[i * 4 + 1], X1, t);
        t = __builtin_fmaf(P[i * 4 + 2], X2, t);
        q[i] = __builtin_fmaf(P[i * 4 + 3], 1.0f, t);
    }
    Proj p;
    p.z = q[2] + eps;
    if (wave_all_normal(p.z)) {
        const float r1 = refined_rcp(p.z);
        p.u = quot_with(q[0], p.z, r1);
        p.v = quot_with(q[1], p.z, r1);
    } else {
        p.u = q[0] / p.z;
        p.v = q[1] / p.z;
    }
    // u / (W-1), v / (H-1): div_norm() picks per LANE between the verified 3-instruction constant division and the IEEE
    // divide (a select: both run).  A per-WAVE choice was measured in round 3 and is not faster.
    p.gx = (div_norm(p.u, nd.w) - 0.5f) * 2.0f;
    p.gy = (div_norm(p.v, nd.h) - 0.5f) * 2.0f;
    return p;
}

// load_corners() (mdx_device.hpp) for FINITE images, with two selects per colour channel instead of six.  A corner that lies
// outside the image reads as 0 in the reference (ATen's masked gather).  It lies outside only when the clipped coordinate
// sits exactly on the last column / row: x0 = floor(ix) = W-1 forces ix = W-1, hence the east weights s*w, n*w are s*(+0),
// n*(+0) = +0 -- and likewise the south weights when y0 = H-1.  A finite value times a zero weight adds a signed zero to
// the fma chain of sample(), which leaves every accumulator value as it is (x + (+-0) = x; the one sign-of-zero case, -0 + +0,
// comes out +0 here and in the reference alike), so the out-of-range corner may hold ANY finite number: the neighbouring
// in-range pixel the 8-byte load brings along anyway.  The sampling derivatives use such a corner only in terms that are
// multiplied by dx0 = ix - x0 = 0 (dy0 = 0) or masked out by the coordinate's "strictly inside" flag.  Only NaN / Inf pixels
// would tell the difference; the colours here are bytes / 255.
MDX_DEV Corners load_corners_finite(const float *__restrict__ img, int H, int W, const Tap &t)
{
    const int xl = t.x0 < W - 1 ? t.x0 : W - 2;
    const bool shifted = xl != t.x0;            // x0 == W-1: the pair is anchored one column to the left
    const int y1 = t.y0 + 1 < H ? t.y0 + 1 : t.y0;
    const unsigned o0 = (unsigned)(t.y0 * W + xl) * 4u, o1 = (unsigned)(y1 * W + xl) * 4u;
    const char *base = reinterpret_cast<const char *>(img);
    const float2_a4 top = *reinterpret_cast<const float2_a4 *>(base + o0);
    const float2_a4 bot = *reinterpret_cast<const float2_a4 *>(base + o1);
    Corners c;
    c.nw = shifted ? top.y : top.x;
    c.ne = top.y;
    c.sw = shifted ? bot.y : bot.x;
    c.se = bot.y;
    return c;
}

// geom_from_disp() (photo_common.hpp) with the written-out reciprocal
MDX_DEV PixelGeom geom_from_disp_train(const mdx_desc &d, float up, const float *__restrict__ invK_b, int px, int py)
{
    PixelGeom g;
    const float sd = scaled_disp(up, d.disp_a, d.disp_b);
    if (wave_all_normal(sd)) g.depth = quot_with(1.0f, sd, refined_rcp(sd));
    else g.depth = 1.0f / sd;
    pixel_ray(invK_b, (float)px, (float)py, g.r);
    g.X0 = g.depth * g.r[0];
    g.X1 = g.depth * g.r[1];
    g.X2 = g.depth * g.r[2];
    return g;
}

// ---- the item's 3x4 matrices as TRANSIENT scalars ----
// P (S x 12) and invK (12) are wave-uniform.  Held in scalar registers for the whole item they take 36 of the ~100
// SGPRs; with the row pointers of a step on top the allocator spilled 61 values to VGPR lanes and paid ~90
// v_readlane / v_writelane (VALU issue slots, plus hazard s_nops) per step.  They are re-read through the scalar cache
// where they are used (s_load_dwordx4 x3 per matrix, scalar unit, no VALU slot) so that they are dead in between:
// 61 -> 41 spilled SGPRs, 120 -> 45 v_readlane, 165 -> 116 s_nop cycles in the kernel, -1.5 % time.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef unsigned u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
struct SRows {
    f32x4 a, b, c;
};
static __device__ __forceinline__ const float *uniform_ptr(const float *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const float *)(((unsigned long long)hi << 32) | lo);
}
static __device__ __forceinline__ void sload12(const float *p, SRows &m)
{
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20"
                 : "=&s"(m.a), "=&s"(m.b), "=&s"(m.c)
                 : "s"(p));
}
static __device__ __forceinline__ void swait12(SRows &m, float *o)     // the loads above have landed; o = the 12 floats
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(m.a), "+s"(m.b), "+s"(m.c));
    o[0] = m.a.x; o[1] = m.a.y; o[2] = m.a.z; o[3] = m.a.w;
    o[4] = m.b.x; o[5] = m.b.y; o[6] = m.b.z; o[7] = m.b.w;
    o[8] = m.c.x; o[9] = m.c.y; o[10] = m.c.z; o[11] = m.c.w;
}

}  // namespace mdx
