// pose.hip -- param2matrix (model_layer/warp.py:43-153: vector2translation, angle2rotation, param2matrix), gfx950.
//
// The reference builds the 4x4 camera-to-camera matrix of a predicted (axis-angle, translation) pair out of ~40
// element-wise torch ops on [N,1,1] tensors; with autograd that is ~250 kernel launches of 2-3 us per training step
// for 2 x 96 floats.  Here it is one launch forward and one backward, one thread per pose.
//   forward   the reference's operation sequence: angle = |a|, axis = a / (angle + 1e-5), Rodrigues matrix from
//             x*xC + cos, xyC - z*sin, ...; invert: M = R^T @ T(-t), else M = T(t) @ R (the matmul rows are summed in
//             index order, terms that are exactly zero included).
//   backward  forward-mode dual numbers over the six inputs (value + 6 partials through +, -, *, /, sqrt, sin, cos),
//             contracted with the upstream gradient of the 12 non-constant entries -- no hand-derived Jacobian.
#include "mdx_common.hpp"

namespace mdx {

template <int ND> struct Dual {
    float v;
    float d[ND > 0 ? ND : 1];
};

template <int ND> __device__ __forceinline__ Dual<ND> mk(float v)
{
    Dual<ND> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = 0.f;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator+(const Dual<ND> &a, const Dual<ND> &b)
{
    Dual<ND> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator-(const Dual<ND> &a, const Dual<ND> &b)
{
    Dual<ND> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator*(const Dual<ND> &a, const Dual<ND> &b)
{
    Dual<ND> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator/(const Dual<ND> &a, const Dual<ND> &b)
{
    Dual<ND> r;
    r.v = a.v / b.v;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> dsqrt(const Dual<ND> &a)
{
    Dual<ND> r;
    r.v = sqrtf(a.v);
    // torch: d|a| = a / |a|, 0 at the origin
    const float k = r.v > 0.f ? 0.5f / r.v : 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * k;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> dsin(const Dual<ND> &a)
{
    Dual<ND> r;
    r.v = sinf(a.v);
    const float c = cosf(a.v);
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * c;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> dcos(const Dual<ND> &a)
{
    Dual<ND> r;
    r.v = cosf(a.v);
    const float s = -sinf(a.v);
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * s;
    return r;
}

// M[0..11] = the three non-constant rows of the 4x4 matrix (row 3 is 0 0 0 1)
template <int ND> __device__ __forceinline__ void pose_matrix(const Dual<ND> a[3], const Dual<ND> t[3], bool invert,
                                                               Dual<ND> M[12])
{
    const Dual<ND> angle = dsqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    const Dual<ND> den = angle + mk<ND>(1e-5f);
    const Dual<ND> x = a[0] / den, y = a[1] / den, z = a[2] / den;
    const Dual<ND> cs = dcos(angle), sn = dsin(angle), Cc = mk<ND>(1.0f) - cs;
    const Dual<ND> xs = x * sn, ys = y * sn, zs = z * sn;
    const Dual<ND> xC = x * Cc, yC = y * Cc, zC = z * Cc;
    const Dual<ND> xyC = x * yC, yzC = y * zC, zxC = z * xC;
    Dual<ND> R[3][3] = {{x * xC + cs, xyC - zs, zxC + ys}, {xyC + zs, y * yC + cs, yzC - xs}, {zxC - ys, yzC + xs, z * zC + cs}};
    if (!invert) {   // T(t) @ R: rotation block R, translation column t
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) M[4 * i + j] = R[i][j];
            M[4 * i + 3] = t[i];
        }
    } else {         // R^T @ T(-t): rotation block R^T, translation column R^T (-t)
        const Dual<ND> zero = mk<ND>(0.f);
        const Dual<ND> nt[3] = {zero - t[0], zero - t[1], zero - t[2]};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) M[4 * i + j] = R[j][i];
            M[4 * i + 3] = (R[0][i] * nt[0] + R[1][i] * nt[1]) + R[2][i] * nt[2];
        }
    }
}

__global__ __launch_bounds__(64) void param2matrix_fwd_kernel(const float *__restrict__ aa, const float *__restrict__ tr,
                                                              int N, int invert, float *__restrict__ M)
{
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    Dual<0> a[3], t[3], m[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) { a[i].v = aa[3 * n + i]; t[i].v = tr[3 * n + i]; }
    pose_matrix<0>(a, t, invert != 0, m);
    float *o = M + 16 * (size_t)n;
#pragma unroll
    for (int i = 0; i < 12; ++i) o[i] = m[i].v;
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.0f;
}

__global__ __launch_bounds__(64) void param2matrix_bwd_kernel(const float *__restrict__ aa, const float *__restrict__ tr,
                                                              const float *__restrict__ gM, int N, int invert,
                                                              float *__restrict__ gaa, float *__restrict__ gtr)
{
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    Dual<6> a[3], t[3], m[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = mk<6>(aa[3 * n + i]);
        a[i].d[i] = 1.0f;
        t[i] = mk<6>(tr[3 * n + i]);
        t[i].d[3 + i] = 1.0f;
    }
    pose_matrix<6>(a, t, invert != 0, m);
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *go = gM + 16 * (size_t)n;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float gi = go[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = __builtin_fmaf(gi, m[i].d[k], g[k]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { gaa[3 * n + i] = g[i]; gtr[3 * n + i] = g[3 + i]; }
}

// ---- the pose network's output -> camera-to-camera matrices AND projections, one launch each way ---------------------------
// processor.py:61-83 + :143-160: per source frame a row slice, a select, param2matrix, K @ T and a stack -- ~10 launches forward
// and ~30 backward (every slice / select backward is a zero-fill + a copy), all in front of the pose network's backward.
// raw [M][F][6] = the pose head's output (axis-angle | translation); source s reads rows row0[s] .. row0[s]+B-1, entry frame[s].
struct PoseSel {
    int row0[MDX_MAX_SRC], frame[MDX_MAX_SRC], invert[MDX_MAX_SRC];
};

__global__ __launch_bounds__(64) void pose_projection_fwd_kernel(const float *__restrict__ raw, int F, const float *__restrict__ K,
                                                                 int B, int S, PoseSel sel, float *__restrict__ T,
                                                                 float *__restrict__ P)
{
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= S * B) return;
    const int s = n / B, b = n - s * B;
    int row0 = sel.row0[0], frame = sel.frame[0], inv = sel.invert[0];
#pragma unroll
    for (int k = 1; k < MDX_MAX_SRC; ++k)
        if (s == k) { row0 = sel.row0[k]; frame = sel.frame[k]; inv = sel.invert[k]; }
    const float *p = raw + ((size_t)(row0 + b) * F + frame) * 6;
    Dual<0> a[3], t[3], m[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) { a[i].v = p[i]; t[i].v = p[3 + i]; }
    pose_matrix<0>(a, t, inv != 0, m);
    float Tm[16];
#pragma unroll
    for (int i = 0; i < 12; ++i) Tm[i] = m[i].v;
    Tm[12] = 0.f; Tm[13] = 0.f; Tm[14] = 0.f; Tm[15] = 1.0f;
    float *o = T + 16 * (size_t)n;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = Tm[i];
    // (K @ T)[:3] in compose_projection_kernel's order: acc = 0; acc += K[r][k] * T[k][j], product and sum rounded separately
    const float *Kb = K + 16 * (size_t)b;
    float *q = P + 12 * (size_t)n;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float prod = Kb[r * 4 + k] * Tm[k * 4 + j];
                acc = acc + prod;
            }
            q[r * 4 + j] = acc;
        }
}

// one thread per entry (m, f) of raw: the sources that read it add their gradients in source order, every other entry gets zeros
__global__ __launch_bounds__(64) void pose_projection_bwd_kernel(const float *__restrict__ raw, int M, int F,
                                                                 const float *__restrict__ K, int B, int S, PoseSel sel,
                                                                 const float *__restrict__ gP, const float *__restrict__ gT,
                                                                 float *__restrict__ graw)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= M * F) return;
    const int mrow = e / F, f = e - mrow * F;
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *p = raw + (size_t)e * 6;
    // NOT unrolled: the body is the whole dual-number evaluation; this kernel runs once per step with a cold instruction cache, and
    // four copies of it cost more than the loop (45 us -> see profiles/)
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        int row0 = sel.row0[0], frame = sel.frame[0], inv = sel.invert[0];
#pragma unroll
        for (int k = 1; k < MDX_MAX_SRC; ++k)
            if (s == k) { row0 = sel.row0[k]; frame = sel.frame[k]; inv = sel.invert[k]; }
        if (f != frame || mrow < row0 || mrow >= row0 + B) continue;
        const int b = mrow - row0;
        const size_t n = (size_t)s * B + b;
        Dual<6> a[3], t[3], m[12];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i] = mk<6>(p[i]);
            a[i].d[i] = 1.0f;
            t[i] = mk<6>(p[3 + i]);
            t[i].d[3 + i] = 1.0f;
        }
        pose_matrix<6>(a, t, inv != 0, m);
        const float *Kb = K + 16 * (size_t)b;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // d/dT[k][j] of (K @ T)[:3] = sum_r K[r][k] * gP[r][j]   (+ what arrives for the matrix itself)
                float gi = 0.f;
                if (gP) {
                    const float *gp = gP + 12 * n;
                    gi = (Kb[0 * 4 + k] * gp[0 * 4 + j] + Kb[1 * 4 + k] * gp[1 * 4 + j]) + Kb[2 * 4 + k] * gp[2 * 4 + j];
                }
                if (gT) gi = gi + gT[16 * n + k * 4 + j];
#pragma unroll
                for (int q = 0; q < 6; ++q) g[q] = __builtin_fmaf(gi, m[k * 4 + j].d[q], g[q]);
            }
    }
    float *o = graw + (size_t)e * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[q] = g[q];
}

}  // namespace mdx

using namespace mdx;

MDX_EXPORT int mdx_param2matrix_fwd(const float *axisangle, const float *translation, int N, int invert, float *M,
                                    void *stream)
{
    if (!axisangle || !translation || !M) return MDX_ERR_NULL_POINTER;
    if (N <= 0) return MDX_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(param2matrix_fwd_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, axisangle,
                       translation, N, invert, M);
    return check_launch();
}

MDX_EXPORT int mdx_param2matrix_bwd(const float *axisangle, const float *translation, const float *gM, int N, int invert,
                                    float *gaxisangle, float *gtranslation, void *stream)
{
    if (!axisangle || !translation || !gM || !gaxisangle || !gtranslation) return MDX_ERR_NULL_POINTER;
    if (N <= 0) return MDX_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(param2matrix_bwd_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, axisangle,
                       translation, gM, N, invert, gaxisangle, gtranslation);
    return check_launch();
}

static int fill_sel(int M, int F, int B, int S, const int32_t *row0, const int32_t *frame, const int32_t *invert, PoseSel *sel)
{
    if (M <= 0 || F <= 0 || B <= 0 || S < 1 || S > MDX_MAX_SRC) return MDX_ERR_BAD_SHAPE;
    if (!row0 || !frame || !invert) return MDX_ERR_NULL_POINTER;
    for (int s = 0; s < MDX_MAX_SRC; ++s) {
        if (s < S) {
            if (row0[s] < 0 || row0[s] + B > M || frame[s] < 0 || frame[s] >= F) return MDX_ERR_BAD_SHAPE;
            sel->row0[s] = row0[s]; sel->frame[s] = frame[s]; sel->invert[s] = invert[s] ? 1 : 0;
        } else {
            sel->row0[s] = 0; sel->frame[s] = -1; sel->invert[s] = 0;
        }
    }
    return MDX_OK;
}

MDX_EXPORT int mdx_pose_projection_fwd(const float *raw, int M, int F, const float *K, int B, int S, const int32_t *row0,
                                       const int32_t *frame, const int32_t *invert, float *T, float *P, void *stream)
{
    PoseSel sel;
    int rc = fill_sel(M, F, B, S, row0, frame, invert, &sel);
    if (rc) return rc;
    if (!raw || !K || !T || !P) return MDX_ERR_NULL_POINTER;
    hipLaunchKernelGGL(pose_projection_fwd_kernel, dim3((S * B + 63) / 64), dim3(64), 0, (hipStream_t)stream, raw, F, K, B, S,
                       sel, T, P);
    return check_launch();
}

MDX_EXPORT int mdx_pose_projection_bwd(const float *raw, int M, int F, const float *K, int B, int S, const int32_t *row0,
                                       const int32_t *frame, const int32_t *invert, const float *gP, const float *gT,
                                       float *graw, void *stream)
{
    PoseSel sel;
    int rc = fill_sel(M, F, B, S, row0, frame, invert, &sel);
    if (rc) return rc;
    if (!raw || !K || !graw) return MDX_ERR_NULL_POINTER;
    hipLaunchKernelGGL(pose_projection_bwd_kernel, dim3((M * F + 63) / 64), dim3(64), 0, (hipStream_t)stream, raw, M, F, K, B, S,
                       sel, gP, gT, graw);
    return check_launch();
}
