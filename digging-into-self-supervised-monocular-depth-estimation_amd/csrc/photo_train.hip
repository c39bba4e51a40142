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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "photo_common.hpp"

namespace mdx {

constexpr int SW = 60;   // output columns per wave (64 lanes - 2 x 2 halo lanes)

// A/B switch of the builder (build.py passes -D...; the shipped library never reads the environment):
// 1 = the divisions of the row loop as written-out sequences that share reciprocals (see quot_rcp), 0 = plain `/`
#ifndef MDX_TRAIN_FASTDIV
#define MDX_TRAIN_FASTDIV 1
#endif
#ifndef MDX_TRAIN_OPAQUE_HIST
#define MDX_TRAIN_OPAQUE_HIST 1
#endif
#ifndef MDX_TRAIN_LAZY_COEF
#define MDX_TRAIN_LAZY_COEF 1     // SSIM gradient coefficients of a frame only in waves where it can be some lane's arg-min
#endif
#ifndef MDX_TRAIN_WAVE_UNIFORM
#define MDX_TRAIN_WAVE_UNIFORM 0  // per-wave (not per-lane) choice of the u/(W-1), v/(H-1) division form and of the corner border
                                  // handling: ~70 fewer VALU instructions per step on paper, +1 % time measured (tools/ab_bench_repeat.sh)
#endif
#ifndef MDX_TRAIN_WU_NORM
#define MDX_TRAIN_WU_NORM MDX_TRAIN_WAVE_UNIFORM
#endif
#ifndef MDX_TRAIN_WU_CORNER
#define MDX_TRAIN_WU_CORNER MDX_TRAIN_WAVE_UNIFORM
#endif
#ifndef MDX_EVAL_WAVES
#define MDX_EVAL_WAVES 5          // waves per SIMD of the forward-only form for S <= 2 (96 VGPRs)
#endif
#ifndef MDX_TRAIN_LOSS_LDS
#define MDX_TRAIN_LOSS_LDS 1      // loss partial of a lane: LDS cell + ds_add_f32 (1) or a float register (0)
#endif
#ifndef MDX_TRAIN_DEPTH_LDS
#define MDX_TRAIN_DEPTH_LDS 0     // depth of the rows awaiting their gradient: LDS ring (1) or three registers (0).
#endif                            // Measured inside the step (tools/ab_bench.sh): the ring costs +8 us (its read sits at the head
                                  // of the gradient phase, exposed), the registers cost no spill that matters

#ifndef MDX_TRAIN_DEFER_GUP
#define MDX_TRAIN_DEFER_GUP 0     // 1: the gradient row is stored in the NEXT step (one register across the loop edge)
#endif

struct TrainArgs {
    int B, H, W, S, nscales;
    unsigned flags;
    float disp_a, disp_b;
    int h[MDX_MAX_SCALES], w[MDX_MAX_SCALES];
    // guided schedule: every column (scale, image, strip) is cut into lev_n[0] chunks of lev_r[0] rows, then lev_n[1] of
    // lev_r[1], then lev_n[2] of lev_r[2] (the last one may be ragged); items are numbered level by level, so the
    // large chunks are dispatched first and the small ones fill the tail of the launch
    int lev_n[3], lev_r[3], lev_item0[3], lev_row0[3], lev_k0[3];
    int nchunks, nstrips, ncols;
    const float *disp[MDX_MAX_SCALES];
    const float *P[MDX_MAX_SCALES];
    const float *noise[MDX_MAX_SCALES];
    uint8_t *idx[MDX_MAX_SCALES];
    float *gup[MDX_MAX_SCALES];
    float *to_opt[MDX_MAX_SCALES];
    const float *target, *ident, *invK;
    // PRE (photo_prologue.hip ran this step): the target's window statistics [B,6,H,W] and, per scale, the best identity
    // channel of every pixel (value, index) instead of the identity and noise maps
    const float *tstat;                       // [B,H,W,6]: (mu_y, sigma_y) x channel, interleaved (24 B per pixel: two loads)
    const float *bidfi[MDX_MAX_SCALES];       // [B,H,W,2]: (best identity value, its channel index as int32) per scale
    mdx_sources src;
    float *depth0;
    double *loss_part;   // [items]
    float *partP;        // [items][S][12]
    unsigned long long *stamps;   // diagnostic builds (-DMDX_TRAIN_STAMPS) only: [items][8] phase cycle sums
};

// In-kernel phase clock of the diagnostic build (cdna_hip_programming.md section 7, "In-kernel stamps"): the values go to a
// buffer of their own and no output is computed from them; the product build compiles none of it.
#ifdef MDX_TRAIN_STAMPS
#define MDX_STAMP(k)                                                         \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        st_acc[k] += now_ - st_last;                                         \
        st_last = now_;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                   \
    } while (0)
#else
#define MDX_STAMP(k) do { } while (0)
#endif

// wave-uniform selection from a by-value kernel-argument array without dynamic indexing (which would send the
// argument block to scratch)
template <typename T> MDX_DEV T pick(const T (&v)[MDX_MAX_SCALES], int s)
{
    return s == 0 ? v[0] : (s == 1 ? v[1] : (s == 2 ? v[2] : v[3]));
}

MDX_DEV int pick4(const int (&v)[MDX_MAX_SCALES + 1], int s)
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

// SSIM of one colour channel of one frame, value AND gradient coefficients in one go.  The value follows ssim_raw()
// operation for operation (bit-exact); the coefficient triplet of SURVEY appendix A.1 re-uses its intermediates and a
// hardware reciprocal (gradients carry a 1e-4 tolerance, not bit-exactness).
struct SsimBoth { float val; SsimGrad g; };

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

// The same in two halves (MDX_TRAIN_LAZY_COEF): the value now -- keeping what the coefficients need -- and the coefficient
// triplet later, only in waves where the frame can still be some lane's arg-min.
struct SsimMid { float A1, A2, B1, B2, q, inv_d, mu_x, raw; };

MDX_DEV float ssim_value_mid(const SsimTerms &s, const TStat &t, SsimMid &m)
{
    const float mxx = s.mu_x * s.mu_x;
    const float mxy = s.mu_x * t.mu;
    const float sig_x = s.ex2 - mxx;
    const float sig_xy = s.exy - mxy;
    float a = 2.0f * s.mu_x;
    a = a * t.mu;
    m.A1 = a + MDX_C1;
    float A2 = 2.0f * sig_xy;
    m.A2 = A2 + MDX_C2;
    const float n = m.A1 * m.A2;
    m.B1 = (mxx + t.mu2) + MDX_C1;
    m.B2 = (sig_x + t.sig_y) + MDX_C2;
    const float d = m.B1 * m.B2;
    const QuotRcp qr = quot_rcp(n, d);
    m.q = qr.q;
    m.inv_d = qr.r;
    m.mu_x = s.mu_x;
    m.raw = (1.0f - qr.q) / 2.0f;
    return clamp01(m.raw);
}

MDX_DEV SsimGrad ssim_coef_mid(const SsimMid &m, const TStat &t, float gscale)
{
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

MDX_DEV SsimBoth ssim_both(const SsimTerms &s, const TStat &t, float gscale)
{
    const float mxx = s.mu_x * s.mu_x;
    const float mxy = s.mu_x * t.mu;
    const float sig_x = s.ex2 - mxx;
    const float sig_y = t.sig_y;
    const float sig_xy = s.exy - mxy;
    float a = 2.0f * s.mu_x;
    a = a * t.mu;
    const float A1 = a + MDX_C1;
    float A2 = 2.0f * sig_xy;
    A2 = A2 + MDX_C2;
    const float n = A1 * A2;
    const float B1 = (mxx + t.mu2) + MDX_C1;
    const float B2 = (sig_x + sig_y) + MDX_C2;
    const float d = B1 * B2;
#if MDX_TRAIN_FASTDIV
    const QuotRcp qr = quot_rcp(n, d);
    const float q = qr.q, inv_d = qr.r;
#else
    const float q = n / d;
    float inv_d = __builtin_amdgcn_rcpf(d);                              // 1 ulp
    inv_d = __builtin_fmaf(__builtin_fmaf(-d, inv_d, 1.0f), inv_d, inv_d);   // one Newton step
#endif
    const float raw = (1.0f - q) / 2.0f;
    SsimBoth r;
    r.val = clamp01(raw);
    const float Ln = -0.5f * inv_d, Ld = 0.5f * q * inv_d;
    const float dA1 = Ln * A2, dA2 = Ln * A1, dB1 = Ld * B2, dB2 = Ld * B1;
    const bool pass = raw >= 0.f && raw <= 1.f;   // clamp passes the gradient on the closed interval
    const float gs = pass ? gscale : 0.f;
    r.g.alpha = gs * 2.0f * (t.mu * (dA1 - dA2) + s.mu_x * (dB1 - dB2));
    r.g.beta = gs * dB2;
    r.g.gamma = gs * 2.0f * dA2;
    return r;
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
#if MDX_TRAIN_FASTDIV
    if (wave_all_normal(p.z)) {
        const float r1 = refined_rcp(p.z);
        p.u = quot_with(q[0], p.z, r1);
        p.v = quot_with(q[1], p.z, r1);
    } else
#endif
    {
        p.u = q[0] / p.z;
        p.v = q[1] / p.z;
    }
    // u / (W-1), v / (H-1): div_norm() picks per LANE between the verified 3-instruction constant division and the IEEE
    // divide -- as a select, so both run.  Here the choice is made once per wave (a lane outside the verified range, i.e.
    // |x| <= 1e-30 or >= 3e38, sends the wave through `/`): the 2 x 11 instructions of the unused divides are gone.
    float nx, ny;
    const float au = fabsf(p.u), av = fabsf(p.v);
    const bool fast_ok = nd.w.fast && nd.h.fast && au > 1e-30f && au < 3.0e38f && av > 1e-30f && av < 3.0e38f;
    if (MDX_TRAIN_WU_NORM && __builtin_amdgcn_ballot_w64(!fast_ok) == 0) {
        nx = div_by_const(p.u, nd.w.b, nd.w.r);
        ny = div_by_const(p.v, nd.h.b, nd.h.r);
    } else {
        nx = div_norm(p.u, nd.w);
        ny = div_norm(p.v, nd.h);
    }
    p.gx = (nx - 0.5f) * 2.0f;
    p.gy = (ny - 0.5f) * 2.0f;
    return p;
}

// load_corners() (mdx_device.hpp) with the border handling decided per wave: a tap on the image's last column / last row
// needs the pair shifted / the lower pair zeroed (four selects per channel); only waves that hold such a tap pay for them.
MDX_DEV Corners load_corners_train(const float *__restrict__ img, int H, int W, const Tap &t, bool border_wave)
{
    if (!MDX_TRAIN_WU_CORNER || border_wave) return load_corners(img, H, W, t);
    const unsigned o0 = (unsigned)(t.y0 * W + t.x0) * 4u, o1 = o0 + (unsigned)W * 4u;
    const char *base = reinterpret_cast<const char *>(img);
    const float2_a4 top = *reinterpret_cast<const float2_a4 *>(base + o0);
    const float2_a4 bot = *reinterpret_cast<const float2_a4 *>(base + o1);
    Corners c;
    c.nw = top.x; c.ne = top.y; c.sw = bot.x; c.se = bot.y;
    return c;
}

// geom_from_disp() (photo_common.hpp) likewise
MDX_DEV PixelGeom geom_from_disp_train(const mdx_desc &d, float up, const float *__restrict__ invK_b, int px, int py)
{
    PixelGeom g;
    const float sd = scaled_disp(up, d.disp_a, d.disp_b);
#if MDX_TRAIN_FASTDIV
    if (wave_all_normal(sd)) g.depth = quot_with(1.0f, sd, refined_rcp(sd));
    else
#endif
        g.depth = 1.0f / sd;
    pixel_ray(invK_b, (float)px, (float)py, g.r);
    g.X0 = g.depth * g.r[0];
    g.X1 = g.depth * g.r[1];
    g.X2 = g.depth * g.r[2];
    return g;
}

// ---- the item's 3x4 matrices as TRANSIENT scalars ----
// P (S x 12) and invK (12) are wave-uniform.  Held in scalar registers for the whole item they take 36 of the ~100
// SGPRs; with the row pointers of a step on top the allocator spilled 61 values to VGPR lanes and paid ~90
// v_readlane / v_writelane (VALU issue slots, plus hazard s_nops) per step.  MDX_TRAIN_SLOAD re-reads them through
// the scalar cache where they are used (s_load_dwordx4 x3 per matrix, scalar unit, no VALU slot) so that they are dead
// in between: 61 -> 41 spilled SGPRs, 120 -> 45 v_readlane, 165 -> 116 s_nop cycles in the kernel, -1.5 % time.
// (=0 keeps the old form for A/B builds.)
#ifndef MDX_TRAIN_SLOAD
#define MDX_TRAIN_SLOAD 1
#endif
#ifndef MDX_TRAIN_OPAQUE_HW
#define MDX_TRAIN_OPAQUE_HW 1
#endif
#ifndef MDX_TRAIN_XCD_GROUP
#define MDX_TRAIN_XCD_GROUP 1
#endif
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

// SSIM value alone (the no-gradient form of the kernel): ssim_raw()'s operation order, bit-equal to ssim_both().val
MDX_DEV float ssim_val(const SsimTerms &s, const TStat &t)
{
    const float mxx = s.mu_x * s.mu_x;
    const float mxy = s.mu_x * t.mu;
    const float sig_x = s.ex2 - mxx;
    const float sig_y = t.sig_y;
    const float sig_xy = s.exy - mxy;
    float a = 2.0f * s.mu_x;
    a = a * t.mu;
    const float A1 = a + MDX_C1;
    float A2 = 2.0f * sig_xy;
    A2 = A2 + MDX_C2;
    const float n = A1 * A2;
    const float B1 = (mxx + t.mu2) + MDX_C1;
    const float B2 = (sig_x + sig_y) + MDX_C2;
    const float d = B1 * B2;
#if MDX_TRAIN_FASTDIV
    const float q = quot_rcp(n, d).q;
#else
    const float q = n / d;
#endif
    return clamp01((1.0f - q) / 2.0f);
}

// GRAD = true: the training form (loss, indices AND the unit-upstream gradients).  GRAD = false: validation /
// torch.no_grad() (model_train.py:75-79 runs the loss on the validation split every epoch): the same marching wave
// without the coefficient histories, the stash and the gradient phase -- every scale's forward in ONE launch.
// PRE = true: the step's prologue kernel has evaluated what the scales share (target statistics, best identity channel):
// they are loaded (8 dwords and a byte per pixel) instead of re-derived (6 pools) / re-read (2 S dwords) per scale.
template <int S, bool GRAD, bool PRE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GRAD ? (S <= 2 ? 3 : 2) : (S <= 3 ? 4 : 3), GRAD ? (S <= 2 ? 3 : 2) : 8))) void photometric_train_kernel(TrainArgs a)
{
    // per-lane stash ring: [row slot][2f] = (d colour_c / du, u), [2f+1] = (d colour_c / dv, v) of frame f
    __shared__ float4 s_stash[GRAD ? 3 : 1][GRAD ? 2 * S : 1][GRAD ? 64 : 1];
    // loss partial of each lane: an LDS cell, updated by one ds_add_f32 per row (own address: sequential, deterministic).
    // As a register it was the value the allocator spilled in the <2, GRAD, PRE> instantiation -- a scratch load,
    // s_waitcnt vmcnt(0), add, scratch store in every step.  float32: at most 44 addends in [0, 1] per lane (error ~1e-7
    // of the lane's sum, independent between the ~1.5 M lanes); lanes, items and scales are summed in float64.
    __shared__ float s_loss[64];
#if MDX_TRAIN_DEPTH_LDS
    __shared__ float s_depth[GRAD ? 3 : 1][GRAD ? 64 : 1];     // depth of the rows in the stash ring (three registers less)
#endif

    const int lane = threadIdx.x;
    // ---- work item: level-major order (see TrainArgs), dispatched in block order.  An XCD-contiguous order inside each
    //      level (neighbouring strips on one XCD) was measured and is WORSE (fabric-side fetch 1.36 GB instead of 1.0 GB
    //      per launch, 349 us instead of 320 us on tools/kbench.py's data: a contiguous run holds items of equal cost,
    //      and the XCDs drift apart); grouping the SCALES of a region on one XCD (below) keeps the balance and cuts the
    //      fetch to 0.65 GB ----
    const int item_d = (int)blockIdx.x;
    const int lev = item_d >= a.lev_item0[2] ? 2 : (item_d >= a.lev_item0[1] ? 1 : 0);
    const int lev_first = lev == 2 ? a.lev_item0[2] : (lev == 1 ? a.lev_item0[1] : a.lev_item0[0]);
    const int lev_rows = lev == 2 ? a.lev_r[2] : (lev == 1 ? a.lev_r[1] : a.lev_r[0]);
#if MDX_TRAIN_XCD_GROUP
    // Workgroup i runs on XCD i % 8.  The scales of one (chunk, image, strip) read the same target / source / identity
    // rows: their ids share the residue mod 8 and are adjacent (start together), so three of the four find those
    // lines in the XCD's L2.  Groups go round the XCDs one by one -- every XCD keeps the same mix of costs (what the
    // XCD-contiguous order lost).
    const int j = item_d - lev_first;
    const int per_scale = a.nstrips * a.B, ns = a.ncols / per_scale;
    const int ngroups = (lev == 2 ? a.lev_n[2] : (lev == 1 ? a.lev_n[1] : a.lev_n[0])) * per_scale;
    const int full = ngroups / 8, blk = 8 * ns;
    int grp, scale;
    if (j < full * blk) {
        const int q = j / blk, r = j - q * blk;
        scale = r >> 3;
        grp = q * 8 + (r & 7);
    } else {
        const int jj = j - full * blk, rem = ngroups - full * 8;
        scale = jj / rem;
        grp = full * 8 + jj % rem;
    }
    const int kk = grp / per_scale, gcol = grp % per_scale;
    const int strip = gcol % a.nstrips;
    const int b = gcol / a.nstrips;
#else
    const int kk = (item_d - lev_first) / a.ncols, colid = (item_d - lev_first) % a.ncols;
    const int strip = colid % a.nstrips;
    const int b = (colid / a.nstrips) % a.B;
    const int scale = colid / (a.nstrips * a.B);
#endif
    const int chunk = (lev == 2 ? a.lev_k0[2] : (lev == 1 ? a.lev_k0[1] : a.lev_k0[0])) + kk;
    // slot of the item's partials: the items of one (scale, image) contiguous, whatever the dispatch order
    const unsigned item = (unsigned)(((scale * a.B + b) * a.nchunks + chunk) * a.nstrips + strip);

    mdx_desc d;
    d.B = a.B; d.H = a.H; d.W = a.W; d.S = S; d.flags = a.flags; d.disp_a = a.disp_a; d.disp_b = a.disp_b;
    d.h = pick(a.h, scale);
    d.w = pick(a.w, scale);
    const int H = d.H, W = d.W;
    const size_t HW0 = (size_t)H * W;
    // Inside the row loop the plane size is made opaque once per step (MDX_TRAIN_OPAQUE_HW): otherwise the dozen plane
    // bases `tensor + (b*planes + c)*HW` are hoisted out of the loop as 64-bit scalars -- and spilled to VGPR lanes;
    // recomputing them per step costs scalar-unit instructions only.
    unsigned HW = (unsigned)HW0;
    const float *disp_b = pick(a.disp, scale) + (size_t)b * d.h * d.w;
    const float *P_s = pick(a.P, scale);
    const float *noise_s = pick(a.noise, scale);
    // MDX_TRAIN_LATE_PICK: the per-scale output pointers (and the best-identity map) are fetched from the kernel-argument
    // segment WHERE THEY ARE USED -- one s_load_dwordx2 at (array offset + 8 * scale), two scalar registers for a few
    // instructions.  Held across the row loop they were loop-invariant 64-bit scalars the allocator spilled to VGPR lanes,
    // and every use paid two v_readlane (8.5 cycles each on the vector ALU that bounds this kernel).  (Re-deriving them with
    // pick() -- four pointer loads and a select chain -- at the use sites made it worse: 28 -> 49 spilled scalars.)
#ifndef MDX_TRAIN_LATE_PICK
#define MDX_TRAIN_LATE_PICK 1
#endif
#if MDX_TRAIN_LATE_PICK
    const char *const kargs = (const char *)__builtin_amdgcn_kernarg_segment_ptr();     // TrainArgs is the kernel's only argument
    const unsigned scale8 = 8u * (unsigned)scale;
#define MDX_LATE(T, arr)                                                                                          \
    ([&]() {                                                                                                      \
        unsigned long long v_;                                                                                    \
        asm volatile("s_load_dwordx2 %0, %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"                                \
                     : "=s"(v_) : "s"(kargs), "s"(scale8), "n"(__builtin_offsetof(TrainArgs, arr)));              \
        typedef __attribute__((address_space(1))) char *gp_;       /* a global pointer: saddr stores, not flat ones */ \
        return (T)(gp_)v_;                                                                                        \
    }())
#else
    uint8_t *idx_s = pick(a.idx, scale);
    float *gup_s = pick(a.gup, scale);
    float *to_opt_s = pick(a.to_opt, scale);
    const float *bidfi_l = pick(a.bidfi, scale);
#define MDX_LATE(T, arr) ((T)MDX_LATE_##arr)
#define MDX_LATE_idx idx_s
#define MDX_LATE_gup gup_s
#define MDX_LATE_to_opt to_opt_s
#define MDX_LATE_bidfi bidfi_l
#endif
    const float *invK_b = a.invK + b * 16;
    const float *tgt_b = a.target + (size_t)b * 3 * HW0;
#ifndef MDX_TRAIN_LATE_FLAGS
#define MDX_TRAIN_LATE_FLAGS 1
#endif
#if MDX_TRAIN_LATE_FLAGS
    // the item's wave-uniform conditions, re-derived from one scalar WHERE THEY ARE USED (a bit test on the scalar unit):
    // as loop-invariant booleans they lived in 64-bit lane masks, which the allocator spilled to VGPR lanes and fetched
    // back with two v_readlane per use
    const unsigned item_bits = ((d.flags & MDX_FLAG_AUTOMASK) ? 1u : 0u) | ((d.flags & MDX_FLAG_UPSAMPLE_PREMUL) ? 2u : 0u) |
                               ((d.h == d.H && d.w == d.W) ? 4u : 0u);
    auto item_bit = [&](unsigned m) {
        unsigned v = item_bits;
        asm volatile("" : "+s"(v));
        return (v & m) != 0;
    };
#define automask item_bit(1u)
#define premul item_bit(2u)
#define same_res item_bit(4u)
#else
    const bool automask = (d.flags & MDX_FLAG_AUTOMASK) != 0;
    const bool premul = (d.flags & MDX_FLAG_UPSAMPLE_PREMUL) != 0;
    const bool same_res = d.h == H && d.w == W;
#endif
    const Norm2 nd = desc_norm(d);

    // ---- per-lane constants: the column ----
    const int col = strip * SW + lane - 2;
    const int colc = min(max(col, -1), W);           // beyond the reflected ring: clamped, never used
    const int pxr = reflect(colc, W);
    const bool col_img = col >= 0 && col < W;
    const bool out_lane = lane >= 2 && lane < 2 + SW && col < W;
    const bool ssim_lane = col_img && lane >= 1 && lane <= 62;   // both neighbour lanes exist
    const bool edge_strip = strip == 0 || strip * SW + 61 >= W - 2;          // the strips that hold column 1 or W-2
    // the column's bilinear x tap: kept as (i0, l1); i1 and l0 follow in one instruction each where they are used
    // (two registers less across the row loop)
    const UpTap tx_full = up_tap((float)d.w / (float)W, pxr, d.w);
    const int tx_i0 = tx_full.i0;
    const float tx_l1 = tx_full.l1;
    auto tx_tap = [&]() {
        UpTap t;
        t.i0 = tx_i0;
        t.i1 = tx_i0 + (tx_i0 < d.w - 1 ? 1 : 0);
        t.l1 = tx_l1;
        t.l0 = 1.0f - tx_l1;
        return t;
    };

    const int r0 = (lev == 2 ? a.lev_row0[2] : (lev == 1 ? a.lev_row0[1] : a.lev_row0[0])) + kk * lev_rows;
    const int r1 = min(r0 + lev_rows, H);

    // ---- the item's matrices, once, as wave-uniform scalars (inside the loop they would be re-fetched through vector
    //      memory every row and consumed at once: two exposed cache round trips per row) ----
#if MDX_TRAIN_SLOAD
    const float *Pp[S];
#pragma unroll
    for (int f = 0; f < S; ++f) Pp[f] = uniform_ptr(P_s + ((size_t)f * d.B + b) * 12);
    const float *iKp = uniform_ptr(invK_b);
#else
    float Pm[S][12], iK[12];
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int k = 0; k < 12; ++k)
            Pm[f][k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                __builtin_bit_cast(int, P_s[((size_t)f * d.B + b) * 12 + k])));
#pragma unroll
    for (int k = 0; k < 12; ++k)
        iK[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, invK_b[k])));
#endif

    // ---- histories (registers) ----
    float xh[3][S][3];   // warped colours, rows wr-2 .. wr
    float yh[3][3];      // target colours, same rows
    float ch[3][3][3];   // [row][channel][alpha,beta,gamma] of the arg-min frame, rows sr-2 .. sr
    int selp = 0;        // arg-min frame + 1 (0 = none) of rows sr-2, sr-1, sr in bits 0-3, 4-7, 8-11
    int flp = 0;         // grid_sample pass flags (bit 2f: x inside, 2f+1: y inside) of rows wr-2, wr-1, wr in bytes 0-2
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            yh[j][c] = 0.f;
#pragma unroll
            for (int f = 0; f < S; ++f) xh[j][f][c] = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) ch[j][c][k] = 0.f;
        }
    }
#if !MDX_TRAIN_DEPTH_LDS
    float dph[3] = {0.f, 0.f, 0.f};
#endif
    // d(P) accumulators.  With X_j = depth * r_j and the pixel ray r = invK (px, py, 1) linear in the pixel, the twelve
    // sums of a frame follow from nine: A_i = sum gq_i*depth, Bv_i = sum py*gq_i*depth, C_i = sum gq_i (px is a
    // constant of the lane and is applied at the end).
    float accA[S][3], accB[S][3], accC[S][3];
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int i = 0; i < 3; ++i) accA[f][i] = accB[f][i] = accC[f][i] = 0.f;
#if MDX_TRAIN_LOSS_LDS
    s_loss[lane] = 0.f;
#else
    float acc_reg = 0.f;
#endif

    // ---- prefetch registers: the loads of the NEXT step whose addresses do not depend on computed data ----
    float pf_y[3], pf_d[4], pf_id[S], pf_nz[S];
    f32x4 pf_ts4 = {0.f, 0.f, 0.f, 0.f};   // PRE: mu_y[0..2], sigma_y[0]
    float2_a4 pf_ts2 = {0.f, 0.f};         //      sigma_y[1..2]
    u32x2_a4 pf_bf = {0u, 0u};             //      best identity value (float bits), its index
    auto prefetch_warp_row = [&](int wr) {      // target colours + disparity taps of row wr
        const int pyr = reflect(min(max(wr, -1), H), H);
        const unsigned po = (unsigned)(pyr * W + pxr);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // the plane's base as an opaque SCALAR: left to the compiler the plane offset is added per lane in 64 bits
            // (v_lshl_add_u64, 8.5 cycles against 2.8 for a 32-bit add -- tools/int_rate.hip) and the load loses its
            // scalar-base form
            unsigned long long pv = (unsigned long long)(tgt_b + (size_t)c * HW);
            asm volatile("" : "+s"(pv));
            typedef __attribute__((address_space(1))) const float *gptr;      // (a GLOBAL pointer: an integer cast to a generic one
            pf_y[c] = at32((const float *)(gptr)pv, po);                       //  would turn the load into a flat one)
        }
        if (same_res) {
            pf_d[0] = at32(disp_b, po);
        } else {
            // the four disparity taps at 32-bit element offsets from the (scalar) map base: row pointers formed from the
            // per-lane row index were 64-bit VALU address arithmetic
            const UpTap ty = up_tap((float)d.h / (float)H, pyr, d.h);
            const unsigned o0 = (unsigned)(ty.i0 * d.w), o1 = (unsigned)(ty.i1 * d.w);
            const UpTap tx = tx_tap();
            pf_d[0] = at32(disp_b, o0 + (unsigned)tx.i0); pf_d[1] = at32(disp_b, o0 + (unsigned)tx.i1);
            pf_d[2] = at32(disp_b, o1 + (unsigned)tx.i0); pf_d[3] = at32(disp_b, o1 + (unsigned)tx.i1);
        }
    };
    auto prefetch_ssim_row = [&](int sr) {      // identity loss + noise of row sr (PRE: target statistics, best identity)
        const unsigned po = (unsigned)(min(max(sr, 0), H - 1) * W + pxr);
        if constexpr (PRE) {
            const char *tp = reinterpret_cast<const char *>(a.tstat + (size_t)b * 6 * HW) + po * 24u;
            pf_ts4 = *reinterpret_cast<const f32x4_a8 *>(tp);
            pf_ts2 = *reinterpret_cast<const float2_a4 *>(tp + 16);
            if (automask) pf_bf = *reinterpret_cast<const u32x2_a4 *>(reinterpret_cast<const char *>(MDX_LATE(const float *, bidfi) + (size_t)b * 2 * HW) + po * 8u);
        } else {
            if (!automask) return;
#pragma unroll
            for (int f = 0; f < S; ++f) {
                pf_id[f] = at32(a.ident + ((size_t)b * S + f) * HW, po);
                pf_nz[f] = at32(noise_s + ((size_t)b * S + f) * HW, po);
            }
        }
    };
    prefetch_warp_row(r0 - (GRAD ? 2 : 1));
#ifndef MDX_TRAIN_PRE_LATE
#define MDX_TRAIN_PRE_LATE 1      // PRE: fetch the SSIM row's statistics inside the step (behind the corner loads), not a step ahead
#endif
    if (!(PRE && MDX_TRAIN_PRE_LATE)) prefetch_ssim_row(r0 - (GRAD ? 3 : 2));

#if MDX_TRAIN_DEFER_GUP
    float gup_val = 0.f;
    int gup_row = -1;
#endif
#ifdef MDX_TRAIN_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_last;
#endif
    // without the gradient phase a chunk needs one halo row each side (SSIM of rows r0 .. r1-1) instead of two
    const int nsteps = (r1 - r0) + (GRAD ? 4 : 2);
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        const int wr = r0 - (GRAD ? 2 : 1) + t;        // row warped in this step
        const int sr = wr - 1;            // row whose SSIM / arg-min / coefficients are formed
        const int gr = wr - 2;            // row whose gradient is formed
        const int slot_w = t % 3, slot_r = (t + 1) % 3;
#if MDX_TRAIN_OPAQUE_HW
        asm volatile("" : "+s"(HW));
#endif
#if MDX_TRAIN_SLOAD
        SRows s_iK, s_P[S];
        sload12(iKp, s_iK);
#pragma unroll
        for (int f = 0; f < S; ++f) sload12(Pp[f], s_P[f]);
#endif

        // ================= (1) warp row wr =================
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            yh[0][c] = yh[1][c]; yh[1][c] = yh[2][c];
#pragma unroll
            for (int f = 0; f < S; ++f) { xh[0][f][c] = xh[1][f][c]; xh[1][f][c] = xh[2][f][c]; }
        }
        flp = (unsigned)flp >> 8;
        {   // rows beyond the reflected ring (only the first step of a chunk at the image's top, the last at its
            // bottom) are warped at the clamped row and never used: no branch, one code path for the memory counters
            const int pyr = reflect(min(max(wr, -1), H), H);
#pragma unroll
            for (int c = 0; c < 3; ++c) yh[2][c] = pf_y[c];
            float up;
            if (same_res) {
                up = pf_d[0];
            } else {
                const UpTap ty = up_tap((float)d.h / (float)H, pyr, d.h);
                up = up_combine(pf_d[0], pf_d[1], pf_d[2], pf_d[3], ty, tx_tap(), premul);
            }
#if MDX_TRAIN_SLOAD
            float iK[12], Pm[S][12];
            swait12(s_iK, iK);
#pragma unroll
            for (int f = 0; f < S; ++f) swait12(s_P[f], Pm[f]);
#endif
            const PixelGeom g = geom_from_disp_train(d, up, iK, pxr, pyr);
#if MDX_TRAIN_DEPTH_LDS
            if constexpr (GRAD) s_depth[slot_w][lane] = g.depth;      // the gradient of this row (two steps on) needs it
#else
            dph[0] = dph[1]; dph[1] = dph[2]; dph[2] = g.depth;
#endif
            if (a.depth0 && scale == 0 && out_lane && wr >= r0 && wr < r1)
                at32(a.depth0 + (size_t)b * HW, (unsigned)(pyr * W + pxr)) = g.depth;
            // taps of every frame first, then ALL corner loads, then (while they fly) the next step's prefetches
            Proj pr[S];
            Tap tp[S];
            Corners cn[S][3];
#pragma unroll
            for (int f = 0; f < S; ++f) {
                pr[f] = project_point_train(Pm[f], g.X0, g.X1, g.X2, nd, 1e-7f);
                tp[f] = make_tap(pr[f].gx, pr[f].gy, H, W);
            }
#pragma unroll
            for (int f = 0; f < S; ++f) {
                // does any lane of the wave tap the last column or the last row of this frame?  (wave-uniform)
                const bool border_wave = __builtin_amdgcn_ballot_w64(tp[f].x0 >= W - 1 || tp[f].y0 >= H - 1) != 0;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    cn[f][c] = load_corners_train(a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, tp[f], border_wave);
            }
            prefetch_warp_row(wr + 1);
#if MDX_TRAIN_DEFER_GUP
            if constexpr (GRAD) {
                if (gup_row >= 0 && out_lane) at32(MDX_LATE(float *, gup) + (size_t)b * HW, (unsigned)(gup_row * W + pxr)) = gup_val;
                gup_row = -1;
            }
#endif
            // PRE: this step's SSIM row -- three loads whose latency the sampling arithmetic below covers; fetched a step
            // ahead they would hold eight more registers across the loop edge (6 spills at 168 VGPRs)
            if (PRE && MDX_TRAIN_PRE_LATE) prefetch_ssim_row(sr);
            MDX_STAMP(0);   // geometry, taps, load issue
            int fl = 0;
#pragma unroll
            for (int f = 0; f < S; ++f) {
                const float dy1 = (float)(tp[f].y0 + 1) - tp[f].iy, dy0 = tp[f].iy - (float)tp[f].y0;
                const float dx1 = (float)(tp[f].x0 + 1) - tp[f].ix, dx0 = tp[f].ix - (float)tp[f].x0;
                float du[3], dv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    xh[2][f][c] = sample(cn[f][c], tp[f]);
                    if constexpr (GRAD) {
                        du[c] = (cn[f][c].ne - cn[f][c].nw) * dy1 + (cn[f][c].se - cn[f][c].sw) * dy0;
                        dv[c] = (cn[f][c].sw - cn[f][c].nw) * dx1 + (cn[f][c].se - cn[f][c].ne) * dx0;
                    }
                }
                if constexpr (GRAD) {
                    s_stash[slot_w][2 * f][lane] = make_float4(du[0], du[1], du[2], pr[f].u);
                    s_stash[slot_w][2 * f + 1][lane] = make_float4(dv[0], dv[1], dv[2], pr[f].v);
                    fl |= (tp[f].inx ? 1 : 0) << (2 * f);
                    fl |= (tp[f].iny ? 1 : 0) << (2 * f + 1);
                }
            }
            flp |= fl << 16;
            MDX_STAMP(1);   // corner data arrives, samples, derivatives, stash
        }

        // ================= (2) SSIM + L1, min / arg-min, coefficient triplets of row sr =================
        if constexpr (GRAD) {
            selp = (unsigned)selp >> 4;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) { ch[0][c][k] = ch[1][c][k]; ch[1][c][k] = ch[2][c][k]; }
        }
        if (t >= 2 && sr >= 0 && sr < H) {
            // every product a pool reads through DPP is formed in a block of its own, pinned ahead of the pools by a
            // scheduling barrier (see pool3_n); three chains at a time keep the transient registers low
            TStat ts[3];
            if constexpr (PRE) {
                const float pf_ts[6] = {pf_ts4.x, pf_ts4.y, pf_ts4.z, pf_ts4.w, pf_ts2.x, pf_ts2.y};
#pragma unroll
                for (int c = 0; c < 3; ++c) { ts[c].mu = pf_ts[c]; ts[c].mu2 = pf_ts[c] * pf_ts[c]; ts[c].sig_y = pf_ts[3 + c]; }
            } else {
                float q[3][3], o[3];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int j = 0; j < 3; ++j) q[c][j] = yh[j][c];
                pool3_n<3>(q, o);
#pragma unroll
                for (int c = 0; c < 3; ++c) { ts[c].mu = o[c]; ts[c].mu2 = o[c] * o[c]; }
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int j = 0; j < 3; ++j) q[c][j] = yh[j][c] * yh[j][c];
                __builtin_amdgcn_sched_barrier(0);
                pool3_n<3>(q, o);
#pragma unroll
                for (int c = 0; c < 3; ++c) ts[c].sig_y = o[c] - ts[c].mu2;
            }
            // reprojection channels in order; the coefficient triplets of the best reprojection frame so far are
            // kept as the candidate (strict <: torch.min's first-minimum rule among equal values)
            // the best identity channel first (concat [ident + 1e-5*noise, reproj], processor.py:194-204): a reprojection
            // frame that does not beat it in any lane of the wave needs no gradient coefficients
            float bid = 0.f;
            int fi = 0;
            if (automask) {
                if constexpr (PRE) {
                    bid = __builtin_bit_cast(float, pf_bf.x);
                    fi = (int)pf_bf.y;
                } else {
#pragma unroll
                    for (int f = 0; f < S; ++f) {
                        const float tn = 1e-5f * pf_nz[f];
                        const float v = pf_id[f] + tn;
                        if (f == 0 || v < bid) { bid = v; fi = f; }
                    }
                }
            }
            float best_r = 0.f;
            int fr = 0;
            float cand[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
            for (int f = 0; f < S; ++f) {
                float ss[3], ad[3];
                SsimGrad sg[3];
#if MDX_TRAIN_LAZY_COEF
                SsimMid mid[3];
#endif
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float q[3][3], o[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        q[0][j] = xh[j][f][c];
                        q[1][j] = xh[j][f][c] * xh[j][f][c];
                        q[2][j] = xh[j][f][c] * yh[j][c];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    pool3_n<3>(q, o);
                    SsimTerms st;
                    st.mu_x = o[0]; st.ex2 = o[1]; st.exy = o[2];
                    if constexpr (GRAD) {
#if MDX_TRAIN_LAZY_COEF
                        ss[c] = ssim_value_mid(st, ts[c], mid[c]);
#else
                        const SsimBoth sb = ssim_both(st, ts[c], 0.85f / 3.0f);
                        ss[c] = sb.val;
                        sg[c] = sb.g;
#endif
                    } else {
                        ss[c] = ssim_val(st, ts[c]);
                    }
                    ad[c] = fabsf(yh[1][c] - xh[1][f][c]);
                }
                const float rl = reprojection_combine(ss, ad);
                const bool better = f == 0 || rl < best_r;
                best_r = better ? rl : best_r;
                fr = better ? f : fr;
                if constexpr (GRAD) {
#if MDX_TRAIN_LAZY_COEF
                    // the frame can end as a lane's arg-min only where it beats the frames before it AND the best identity
                    // channel (later frames can only take lanes away); where it ends as the arg-min, `better` held here, so
                    // the candidate selected below is the one the gradient phase needs.  Wave-uniform skip otherwise: in
                    // auto-masked regions (static scene, sky) no frame's coefficients are formed at all.
                    const bool can_win = better && ssim_lane && (!automask || rl < bid);
                    if (__builtin_amdgcn_ballot_w64(can_win) != 0) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) sg[c] = ssim_coef_mid(mid[c], ts[c], 0.85f / 3.0f);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            cand[c][0] = better ? sg[c].alpha : cand[c][0];
                            cand[c][1] = better ? sg[c].beta : cand[c][1];
                            cand[c][2] = better ? sg[c].gamma : cand[c][2];
                        }
                    }
#else
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        cand[c][0] = better ? sg[c].alpha : cand[c][0];
                        cand[c][1] = better ? sg[c].beta : cand[c][1];
                        cand[c][2] = better ? sg[c].gamma : cand[c][2];
                    }
#endif
                }
            }
            // concat [ident + 1e-5*noise, reproj] and torch.min's first-minimum rule (processor.py:194-204)
            float best = best_r;
            int bi = fr;
            if (automask) {
                const bool reproj_wins = best_r < bid;     // identity channels come first: ties go to them
                best = reproj_wins ? best_r : bid;
                bi = reproj_wins ? S + fr : fi;
                fr = reproj_wins ? fr : -1;
            }
            if constexpr (GRAD) {
                if (!ssim_lane) fr = -1;
                const bool keep = fr >= 0;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) ch[2][c][k] = keep ? cand[c][k] : 0.f;
                selp |= (fr + 1) << 8;
            }
#if MDX_TRAIN_OPAQUE_HIST
            // `best` and `bi` are pinned in front of the store branch by an empty volatile asm.  In the forward-only form
            // nothing but the stores below consumes them, and the optimiser then SINKS the whole SSIM evaluation into that
            // lane-divergent branch -- except the neighbour-lane reads, which are convergent and stay outside; the backend
            // folds a DPP read into the add that consumes it only within one basic block, so all 108 of them became
            // separate v_mov_dpp (and 76 values were spilled).
            asm volatile("" : "+v"(best), "+v"(bi));
#endif
            if (out_lane && sr >= r0 && sr < r1) {
                const unsigned po = (unsigned)(sr * W + pxr);
                at32(MDX_LATE(uint8_t *, idx) + (size_t)b * HW, po) = (uint8_t)bi;
                float *const to_opt_p = MDX_LATE(float *, to_opt);
                if (to_opt_p) at32(to_opt_p + (size_t)b * HW, po) = best;
#if MDX_TRAIN_LOSS_LDS
                // the cell's address from a lane id formed HERE (two v_mbcnt, volatile so that they are not hoisted): as a
                // loop-invariant register it was the next value to be spilled
                unsigned lid;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
                (void)__hip_atomic_fetch_add(&s_loss[lid], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
                acc_reg += best;
#endif
            }
        } else if constexpr (GRAD) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) ch[2][c][k] = 0.f;
        }
        if (!(PRE && MDX_TRAIN_PRE_LATE)) prefetch_ssim_row(sr + 1);   // behind its last use: one step of cover, no second copy of the values

        MDX_STAMP(2);   // SSIM phase
        // ================= (3) gradient of row gr =================
        if constexpr (GRAD) {
        if (t >= 4) {
        const float wy0 = gr == 1 ? 2.f : 1.f, wy2 = gr == H - 2 ? 2.f : 1.f;   // reflection-pad fold (y)
#if MDX_TRAIN_SLOAD
        float iK[12], Pm[S][12];
        {   // re-read right here: issued before the SSIM phase, or kept from the top of the step, the longer live
            // ranges cost more in spills than the exposed scalar-cache latency (300 us against 314 / 312 us, kbench)
            SRows g_iK, g_P[S];
            sload12(iKp, g_iK);
#pragma unroll
            for (int f = 0; f < S; ++f) sload12(Pp[f], g_P[f]);
            swait12(g_iK, iK);
#pragma unroll
            for (int f = 0; f < S; ++f) swait12(g_P[f], Pm[f]);
        }
#endif
        float r3[3];
        pixel_ray(iK, (float)pxr, (float)gr, r3);     // (float)pxr: one conversion per step instead of a register per item
#if MDX_TRAIN_DEPTH_LDS
        const float depth = s_depth[slot_r][lane];
#else
        const float depth = dph[0];
#endif
        const float fgr = (float)gr;
        const int sel0 = selp & 15, sel1 = (selp >> 4) & 15, sel2 = (selp >> 8) & 15;
        float gdepth = 0.f;
#pragma unroll
        for (int f = 0; f < S; ++f) {
            const int own = (sel0 == f + 1) | (sel1 == f + 1) | (sel2 == f + 1);
            const int hit = own | dpp_i<0x138>(own) | dpp_i<0x130>(own);
            if (__builtin_amdgcn_ballot_w64(hit != 0 && out_lane) == 0) continue;   // wave-uniform
            const float w0 = sel0 == f + 1 ? wy0 : 0.f, w1 = sel1 == f + 1 ? 1.f : 0.f, w2 = sel2 == f + 1 ? wy2 : 0.f;
            const float4 sa = s_stash[slot_r][2 * f][lane], sb = s_stash[slot_r][2 * f + 1][lane];
            const float du[3] = {sa.x, sa.y, sa.z}, dv[3] = {sb.x, sb.y, sb.z};
            const bool centre = sel1 == f + 1;
            float gu = 0.f, gv = 0.f;
            // vertical 3-tap sums (own column, registers) of all nine coefficient maps first, then -- behind a
            // scheduling barrier, see the note on DPP operands above -- the horizontal ones (neighbour lanes)
            float vs[3][3], sum3[3][3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float v = w0 * ch[0][c][k];
                    v = __builtin_fmaf(w1, ch[1][c][k], v);
                    vs[c][k] = __builtin_fmaf(w2, ch[2][c][k], v);
                }
            __builtin_amdgcn_sched_barrier(0);
            // own + left + right as two DPP-operand adds (an fma cannot take a DPP operand: it cost a v_mov_dpp each);
            // the reflection-pad fold -- column 0's window counts column 1 twice, column W-1's counts W-2 twice -- is
            // added on top in the strips that hold those columns only (wave-uniform branch)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) sum3[c][k] = vs[c][k] + from_left(vs[c][k]);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) sum3[c][k] = sum3[c][k] + from_right(vs[c][k]);
            if (edge_strip) {
                // extra weight of the reflection-pad fold: 1 at column 1 (left neighbour) / W-2 (right neighbour), else 0;
                // formed here (the column from the lane id) rather than kept in two registers for the whole item
                const int ecol = strip * SW + lane - 2;
                const float ex0 = ecol == 1 ? 1.f : 0.f, ex2 = ecol == W - 2 ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float s = __builtin_fmaf(from_left(vs[c][k]), ex0, sum3[c][k]);
                        sum3[c][k] = __builtin_fmaf(from_right(vs[c][k]), ex2, s);
                    }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float xq = xh[0][f][c], yq = yh[0][c];
                float gx = (sum3[c][0] + 2.0f * xq * sum3[c][1] + yq * sum3[c][2]) * (1.0f / 9.0f);
                if (centre) gx -= 0.05f * ((yq > xq) ? 1.f : ((yq < xq) ? -1.f : 0.f));   // 0.15*mean_c|y-x|
                gu += gx * du[c];
                gv += gx * dv[c];
            }
            // grid normalisation (2/(W-1)) and grid_sample's un-normalisation ((W-1)/2) cancel
            gu = (((flp >> (2 * f)) & 1) && out_lane) ? gu : 0.f;
            gv = (((flp >> (2 * f + 1)) & 1) && out_lane) ? gv : 0.f;
            const float *Pf = Pm[f];
            const float X0 = depth * r3[0], X1 = depth * r3[1], X2 = depth * r3[2];
            float z = Pf[8] * X0;
            z = __builtin_fmaf(Pf[9], X1, z);
            z = __builtin_fmaf(Pf[10], X2, z);
            z = __builtin_fmaf(Pf[11], 1.0f, z) + 1e-7f;
            const float iz = __builtin_amdgcn_rcpf(z);
            const float gq[3] = {gu * iz, gv * iz, -(gu * sa.w + gv * sb.w) * iz};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float gX = gq[0] * Pf[j] + gq[1] * Pf[4 + j] + gq[2] * Pf[8 + j];
                gdepth += gX * r3[j];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float gd = gq[i] * depth;
                accA[f][i] += gd;
                accB[f][i] = __builtin_fmaf(fgr, gd, accB[f][i]);
                accC[f][i] += gq[i];
            }
        }
        // depth = 1/(a + b*disp)  ->  d depth / d disp = -b * depth^2
#if MDX_TRAIN_DEFER_GUP
        gup_val = gdepth * (-d.disp_b * depth * depth);
        gup_row = gr;
#else
        if (out_lane) at32(MDX_LATE(float *, gup) + (size_t)b * HW, (unsigned)(gr * W + pxr)) = gdepth * (-d.disp_b * depth * depth);
#endif
        MDX_STAMP(3);   // gradient phase
        }
        }
    }
#if MDX_TRAIN_DEFER_GUP
    if constexpr (GRAD)
        if (gup_row >= 0 && out_lane) at32(MDX_LATE(float *, gup) + (size_t)b * HW, (unsigned)(gup_row * W + pxr)) = gup_val;
#endif
#ifdef MDX_TRAIN_STAMPS
    if (lane == 0 && a.stamps) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a.stamps[(size_t)item * 8 + k] = st_acc[k];
        a.stamps[(size_t)item * 8 + 4] = __builtin_amdgcn_s_memtime() - st_begin;
        a.stamps[(size_t)item * 8 + 5] = (unsigned long long)nsteps |
                                         ((unsigned long long)__builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4) << 32);   // HW_ID[15:0]
        a.stamps[(size_t)item * 8 + 6] = st_begin;
        a.stamps[(size_t)item * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // XCC_ID
    }
#endif

    // ---- per-item partials: d(P) (register reduction, total in lane 63) and the loss sum ----
    // d(P)[i][j] = sum gq_i * depth * r_j,  r_j = k_j0*px + k_j1*py + k_j2  (j < 3);  d(P)[i][3] = sum gq_i
    if constexpr (GRAD)
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float sA = wave_sum_dpp_lane63(accA[f][i]);
            const float sAx = wave_sum_dpp_lane63(accA[f][i] * (float)pxr);
            const float sB = wave_sum_dpp_lane63(accB[f][i]);
            const float sC = wave_sum_dpp_lane63(accC[f][i]);
            if (lane == 63) {
#if MDX_TRAIN_SLOAD
                const float *iK = invK_b;
#endif
                float *o = a.partP + (size_t)item * (S * 12) + f * 12 + i * 4;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    o[j] = iK[j * 4 + 0] * sAx + iK[j * 4 + 1] * sB + iK[j * 4 + 2] * sA;
                o[3] = sC;
            }
        }
#if MDX_TRAIN_LOSS_LDS
    const double acc_item = wave_sum((double)s_loss[lane]);
#else
    const double acc_item = wave_sum((double)acc_reg);
#endif
    if (lane == 0) a.loss_part[item] = acc_item;
}

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

// first / last output index whose bilinear source index scale*(dst+0.5)-0.5 can fall in (i-1, i+1), with a margin of
// one (the weights decide; the margin only has to cover the rounding of this estimate).  For an even integer ratio R
// these are R*i - R/2 - 2 and R*i + 3R/2 + 1: a footprint of 2R + 4 taps.
MDX_DEV int foot_lo(int i, float inv_scale) { return (int)floorf(((float)i - 0.5f) * inv_scale - 0.5f) - 1; }
MDX_DEV int foot_hi(int i, float inv_scale) { return (int)ceilf(((float)i + 1.5f) * inv_scale - 0.5f) + 1; }

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
#if MDX_TRAIN_LATE_FLAGS
#undef automask
#undef premul
#undef same_res
#endif


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

struct TrainPlan {
    int lev_n[3], lev_r[3], lev_item0[3], lev_row0[3], lev_k0[3];
    int nchunks, nstrips, ncols;
    size_t items, off_partP, off_gup, total;
};

// Chunk schedule of a column of H rows.  rows_per_chunk > 0: uniform chunks of that many rows (tests, sweeps).
// 0: guided -- about 60 % of the rows in large chunks (4 halo rows per chunk cost little), 25 % in chunks half as
// tall, the rest in small ones that end the launch without a long ragged tail (a work item is one wave walking
// rows + 4 steps; their cost also varies with the auto-mask pattern).  In -DMDX_DEV_SWITCHES builds
// MDX_TRAIN_SCHEDULE="r1,f1,r2,f2,r3" overrides (rows of the levels, fractions of H in levels 1 and 2) for tuning.
static void choose_levels(const mdx_train_desc *d, TrainPlan &p, bool grad)
{
    const int H = d->H;
    int r[3] = {0, 0, 0};
    double f1 = 0.6, f2 = 0.25;
    if (d->rows_per_chunk > 0) {
        r[0] = r[1] = r[2] = d->rows_per_chunk;
        f1 = f2 = 0.0;
    } else {
        r[0] = H >= 160 ? 40 : (H >= 64 ? 24 : 16);
        r[1] = r[0] / 2;
        r[2] = r[1] / 2 > 4 ? r[1] / 2 : 4;
        if (H >= 160 && H < 256) {
            // re-swept in round 3 for the BASELINE height (tools/sweep_schedule.sh, timing inside the real step): with the
            // gradient phase skipped in auto-masked regions a halo step costs relatively more, and fewer, taller chunks win --
            // 3 x 40 + 2 x 24 + 2 x 12 rows (7 chunks, 220 steps per column) 210.5 us against 223.7 us for round 2's
            // 2 x 40 + 2 x 20 + 8 x 10 (12 chunks, 240 steps); 3 x 40 + 3 x 24 (6 chunks) is unbalanced again (224.6 us)
            r[1] = 24; r[2] = 12; f1 = 0.63; f2 = 0.25;
            // the forward-only form (validation, torch.no_grad()): two halo rows per chunk instead of four, 4 waves per SIMD
            // instead of 3, no gradient phase whose cost varies with the mask -- many short chunks balance better than few
            // tall ones (tools/sweep_schedule_eval.sh on bench.py's batch: 16/8/4 rows 116.5 us, 20/10/5 117.5, 24/12/6
            // 122.5, the training schedule 127.5, 48/24/12 153)
            if (!grad) { r[0] = 16; r[1] = 8; r[2] = 4; f1 = 0.6; f2 = 0.3; }
        }
#ifdef MDX_DEV_SWITCHES      // sweeps of the builder only (MDX_BUILD_DEFINES=-DMDX_DEV_SWITCHES): the shipped library never reads the environment
        if (const char *e = getenv("MDX_TRAIN_SCHEDULE")) {
            int a0, a1, a2;
            double g1, g2;
            if (sscanf(e, "%d,%lf,%d,%lf,%d", &a0, &g1, &a1, &g2, &a2) == 5 && a0 > 0 && a1 > 0 && a2 > 0 && g1 >= 0 &&
                g2 >= 0 && g1 + g2 <= 1.0) {
                r[0] = a0; r[1] = a1; r[2] = a2; f1 = g1; f2 = g2;
            }
        }
#endif
    }
    int row = 0, k = 0;
    for (int l = 0; l < 3; ++l) {
        int n;
        if (l == 2) {
            n = (H - row + r[l] - 1) / r[l];                 // the rest, last chunk ragged
        } else {
            n = (int)((l == 0 ? f1 : f2) * H / r[l]);
            if (row + n * r[l] > H) n = (H - row) / r[l];
        }
        p.lev_n[l] = n; p.lev_r[l] = r[l]; p.lev_row0[l] = row; p.lev_k0[l] = k;
        row += n * r[l];
        k += n;
    }
    p.nchunks = k;
}

static TrainPlan plan(const mdx_train_desc *d, bool grad)
{
    TrainPlan p;
    choose_levels(d, p, grad);
    p.nstrips = (d->W + SW - 1) / SW;
    p.ncols = d->nscales * d->B * p.nstrips;
    int it = 0;
    for (int l = 0; l < 3; ++l) { p.lev_item0[l] = it; it += p.lev_n[l] * p.ncols; }
    p.items = (size_t)p.ncols * p.nchunks;
    p.off_partP = p.items * sizeof(double);
    p.off_gup = p.off_partP + ((p.items * d->S * 12 * sizeof(float) + 15) & ~(size_t)15);
    p.total = p.off_gup + (size_t)d->nscales * d->B * d->H * d->W * sizeof(float);
#ifdef MDX_TRAIN_STAMPS
    p.total += p.items * 8 * sizeof(unsigned long long);
#endif
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
    if (validate_train(d)) return 0;
    const size_t a = plan(d, true).total, b = plan(d, false).total;      // one workspace serves both forms of the launch
    return a > b ? a : b;
}

struct PreInputs {                      // what photo_prologue.hip wrote for this step (all null: ident / noise form)
    const float *tstat;
    const float *const *bidfi;
    unsigned long long *rng;
};

static int train_launch(const mdx_train_desc *d, const float *const *disp, const float *target,
                        const mdx_sources *src, const float *invK, const float *const *P,
                        const float *ident, const float *const *noise, const PreInputs &pre, uint8_t *const *idx,
                        float *loss_sum, float *const *gdisp, float *gP, float *depth0,
                        float *const *to_opt, void *workspace, size_t workspace_bytes, void *stream,
                        const mdx_timing *t)
{
    int rc = validate_train(d);
    if (rc) return rc;
    if (!disp || !target || !src || !invK || !P || !idx || !loss_sum) return MDX_ERR_NULL_POINTER;
    if ((gdisp == nullptr) != (gP == nullptr)) return MDX_ERR_NULL_POINTER;
    const bool grad = gdisp != nullptr;      // both null: every scale's forward alone (validation, torch.no_grad())
    const bool automask = (d->flags & MDX_FLAG_AUTOMASK) != 0;
    const bool use_pre = pre.tstat != nullptr;
    if (use_pre) {
        if (automask && !pre.bidfi) return MDX_ERR_NULL_POINTER;
        if (!aligned(pre.tstat, 8)) return MDX_ERR_MISALIGNED;
    } else if (automask && (!ident || !noise)) return MDX_ERR_NULL_POINTER;
    for (int f = 0; f < d->S; ++f)
        if (!src->img[f]) return MDX_ERR_NULL_POINTER;
    const TrainPlan p = plan(d, grad);
    if (!workspace || workspace_bytes < p.total) return MDX_ERR_WORKSPACE;
    if (!aligned(workspace, 16)) return MDX_ERR_MISALIGNED;
    if (p.items >= (1ull << 31)) return MDX_ERR_BAD_SHAPE;
    TrainArgs a = {};
    a.B = d->B; a.H = d->H; a.W = d->W; a.S = d->S; a.nscales = d->nscales; a.flags = d->flags;
    a.disp_a = d->disp_a; a.disp_b = d->disp_b;
    for (int l = 0; l < 3; ++l) {
        a.lev_n[l] = p.lev_n[l]; a.lev_r[l] = p.lev_r[l]; a.lev_item0[l] = p.lev_item0[l];
        a.lev_row0[l] = p.lev_row0[l]; a.lev_k0[l] = p.lev_k0[l];
    }
    a.nchunks = p.nchunks; a.nstrips = p.nstrips; a.ncols = p.ncols;
    a.target = target; a.ident = ident; a.invK = invK; a.src = *src; a.depth0 = depth0;
    a.loss_part = (double *)workspace;
    a.partP = (float *)((char *)workspace + p.off_partP);
    float *gup_ws = (float *)((char *)workspace + p.off_gup);
    const size_t n = (size_t)d->B * d->H * d->W;
    for (int s = 0; s < d->nscales; ++s) {
        if (!disp[s] || !P[s] || !idx[s] || (grad && !gdisp[s])) return MDX_ERR_NULL_POINTER;
        if (automask && (use_pre ? !pre.bidfi[s] : !noise[s])) return MDX_ERR_NULL_POINTER;
        a.h[s] = d->h[s]; a.w[s] = d->w[s];
        a.disp[s] = disp[s]; a.P[s] = P[s]; a.noise[s] = (automask && !use_pre) ? noise[s] : nullptr; a.idx[s] = idx[s];
        a.bidfi[s] = (automask && use_pre) ? pre.bidfi[s] : nullptr;
        a.to_opt[s] = to_opt ? to_opt[s] : nullptr;
        const bool same = d->h[s] == d->H && d->w[s] == d->W;
        a.gup[s] = !grad ? nullptr : (same ? gdisp[s] : gup_ws + s * n);
    }
    for (int s = d->nscales; s < MDX_MAX_SCALES; ++s) {   // never selected; keep the picks well defined
        a.h[s] = a.h[0]; a.w[s] = a.w[0]; a.disp[s] = a.disp[0]; a.P[s] = a.P[0]; a.noise[s] = a.noise[0];
        a.idx[s] = a.idx[0]; a.gup[s] = a.gup[0]; a.to_opt[s] = a.to_opt[0]; a.bidfi[s] = a.bidfi[0];
    }
    a.tstat = pre.tstat;
#ifdef MDX_TRAIN_STAMPS
    a.stamps = (unsigned long long *)((char *)workspace + p.total - p.items * 8 * sizeof(unsigned long long));
#endif
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)p.items), block(64);
    if (t && t->start) (void)hipEventRecord((hipEvent_t)t->start, st);
#define MDX_TRAIN_CASE(SS)                                                                                            \
    case SS:                                                                                                          \
        if (use_pre) {                                                                                                \
            if (grad) hipLaunchKernelGGL((photometric_train_kernel<SS, true, true>), grid, block, 0, st, a);          \
            else hipLaunchKernelGGL((photometric_train_kernel<SS, false, true>), grid, block, 0, st, a);              \
        } else {                                                                                                      \
            if (grad) hipLaunchKernelGGL((photometric_train_kernel<SS, true, false>), grid, block, 0, st, a);         \
            else hipLaunchKernelGGL((photometric_train_kernel<SS, false, false>), grid, block, 0, st, a);             \
        }                                                                                                             \
        break;
    switch (d->S) {
        MDX_TRAIN_CASE(1)
        MDX_TRAIN_CASE(2)
        MDX_TRAIN_CASE(3)
        MDX_TRAIN_CASE(4)
    default: return MDX_ERR_BAD_SHAPE;
    }
#undef MDX_TRAIN_CASE
    if (t && t->stop) (void)hipEventRecord((hipEvent_t)t->stop, st);
    if ((rc = check_launch())) return rc;
    FinishArgs fa = {};
    fa.B = d->B; fa.H = d->H; fa.W = d->W; fa.nscales = d->nscales; fa.S = grad ? d->S : 0; fa.ipi = p.nchunks * p.nstrips;
    fa.partP = a.partP; fa.loss_part = a.loss_part; fa.gP = gP; fa.loss_sum = loss_sum; fa.rng = pre.rng;
    int nblk = 0;
    bool separate[MDX_MAX_SCALES] = {false, false, false, false};
    for (int s = 0; s < MDX_MAX_SCALES; ++s) {
        fa.up_first[s] = nblk;
        const int ss = s < d->nscales ? s : 0;
        fa.gup[s] = a.gup[ss]; fa.gin[s] = grad ? gdisp[ss] : nullptr; fa.h[s] = d->h[ss]; fa.w[s] = d->w[ss];
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
        if (separate[s] && (rc = launch_upsample_bwd(a.gup[s], d->B, d->H, d->W, gdisp[s], d->h[s], d->w[s], st))) return rc;
    return MDX_OK;
}

MDX_EXPORT int mdx_photometric_train(const mdx_train_desc *d, const float *const *disp, const float *target,
                                     const mdx_sources *src, const float *invK, const float *const *P,
                                     const float *ident, const float *const *noise, uint8_t *const *idx,
                                     float *loss_sum, float *const *gdisp, float *gP, float *depth0,
                                     float *const *to_opt, void *workspace, size_t workspace_bytes, void *stream,
                                     const mdx_timing *t)
{
    const PreInputs none = {nullptr, nullptr, nullptr};
    return train_launch(d, disp, target, src, invK, P, ident, noise, none, idx, loss_sum, gdisp, gP, depth0, to_opt,
                        workspace, workspace_bytes, stream, t);
}

MDX_EXPORT int mdx_photometric_train_pre(const mdx_train_desc *d, const float *const *disp, const float *target,
                                         const mdx_sources *src, const float *invK, const float *const *P,
                                         const float *tstat, const float *const *bidfi,
                                         unsigned long long *rng_state, uint8_t *const *idx, float *loss_sum,
                                         float *const *gdisp, float *gP, float *depth0, float *const *to_opt,
                                         void *workspace, size_t workspace_bytes, void *stream, const mdx_timing *t)
{
    if (!tstat) return MDX_ERR_NULL_POINTER;
    const PreInputs pre = {tstat, bidfi, rng_state};
    return train_launch(d, disp, target, src, invK, P, nullptr, nullptr, pre, idx, loss_sum, gdisp, gP, depth0, to_opt,
                        workspace, workspace_bytes, stream, t);
}
