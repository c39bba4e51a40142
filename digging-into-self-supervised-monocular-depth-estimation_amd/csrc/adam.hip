// adam.hip -- the optimiser step of the training loop (reference model_tool/loader.py:93-97: torch.optim.Adam), gfx950.
//
// torch.optim.Adam(fused=True) walks its tensor lists with multi_tensor_apply: at most 320 blocks of 512 threads per launch, each
// looping over a 65536-element chunk -- five launches of 54 us for the 27.8 M parameters of the two ResNet-18 networks, 3 TB/s, at
// the one point of a step where nothing else can run.  Here the step is ONE launch: a block owns 4096 elements of one tensor
// (7000 blocks), reads parameter, gradient and both moments once and writes the three that change.  The arithmetic is
// fused_adam_utils.cuh's expression by expression, including where it is carried out in double (beta * moment, lr / bias
// correction, + eps): the results agree with torch's kernel to the last bit wherever its compiler did not contract a
// multiply-add, and to an ulp otherwise.
#include "mdx_common.hpp"

namespace mdx {

struct AdamTensor {
    float *p, *m, *v;
    const float *step;        // this tensor's step count (float32, already incremented for this step)
    long long n;
};
constexpr int ADAM_MAX_GRADS = 384;          // gradient pointers travel as kernel arguments (they change from step to step)
struct AdamGrads { const float *g[ADAM_MAX_GRADS]; };
constexpr int ADAM_CHUNK = 4096;             // elements per block: 256 threads x 4 float4

__device__ __forceinline__ void adam_one(float &param, float grad, float &exp_avg, float &exp_avg_sq, double beta1, double beta2,
                                         double eps, float step_size, float bias_correction2_sqrt)
{
    exp_avg = (float)(beta1 * exp_avg + (1 - beta1) * grad);
    exp_avg_sq = (float)(beta2 * exp_avg_sq + (1 - beta2) * grad * grad);
    const float denom = (float)((sqrtf(exp_avg_sq) / bias_correction2_sqrt) + eps);
    param -= step_size * exp_avg / denom;
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamTensor *__restrict__ tab, AdamGrads grads, int first,
                                                        const int2 *__restrict__ blockmap, const float *__restrict__ lr_ptr, double lr,
                                                        double beta1, double beta2, double eps)
{
    const int2 bm = blockmap[blockIdx.x];                 // (tensor of this launch, chunk)
    const AdamTensor t = tab[first + bm.x];
    const float *__restrict__ g = grads.g[bm.x];
    const float step_count = *t.step;
    const double bc1 = 1 - pow(beta1, (double)step_count);
    const double bc2 = 1 - pow(beta2, (double)step_count);
    const float bias_correction1 = (float)bc1, bias_correction2_sqrt = (float)sqrt(bc2);
    const double lr_double = lr_ptr ? (double)*lr_ptr : lr;
    const float step_size = (float)(lr_double / bias_correction1);
    const long long base = (long long)bm.y * ADAM_CHUNK;
    const bool vec = ((((uintptr_t)t.p | (uintptr_t)t.m | (uintptr_t)t.v | (uintptr_t)g) & 15) == 0);
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / (256 * 4); ++k) {
        const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
        if (i >= t.n) break;
        if (vec && i + 3 < t.n) {
            float4 P = *reinterpret_cast<const float4 *>(t.p + i), G = *reinterpret_cast<const float4 *>(g + i);
            float4 M = *reinterpret_cast<const float4 *>(t.m + i), V = *reinterpret_cast<const float4 *>(t.v + i);
            adam_one(P.x, G.x, M.x, V.x, beta1, beta2, eps, step_size, bias_correction2_sqrt);
            adam_one(P.y, G.y, M.y, V.y, beta1, beta2, eps, step_size, bias_correction2_sqrt);
            adam_one(P.z, G.z, M.z, V.z, beta1, beta2, eps, step_size, bias_correction2_sqrt);
            adam_one(P.w, G.w, M.w, V.w, beta1, beta2, eps, step_size, bias_correction2_sqrt);
            *reinterpret_cast<float4 *>(t.p + i) = P;
            *reinterpret_cast<float4 *>(t.m + i) = M;
            *reinterpret_cast<float4 *>(t.v + i) = V;
        } else {
            for (long long j = i; j < i + 4 && j < t.n; ++j) {
                float P = t.p[j], M = t.m[j], V = t.v[j];
                adam_one(P, g[j], M, V, beta1, beta2, eps, step_size, bias_correction2_sqrt);
                t.p[j] = P; t.m[j] = M; t.v[j] = V;
            }
        }
    }
}

}  // namespace mdx

using namespace mdx;

MDX_EXPORT int mdx_adam_max_tensors(void) { return ADAM_MAX_GRADS; }
MDX_EXPORT int mdx_adam_chunk(void) { return ADAM_CHUNK; }
MDX_EXPORT size_t mdx_adam_table_entry_bytes(void) { return sizeof(AdamTensor); }

// table: DEVICE array of {float *param, *exp_avg, *exp_avg_sq; const float *step; int64 numel} (mdx_adam_table_entry_bytes() each);
// this launch takes entries first .. first + count - 1 (count <= mdx_adam_max_tensors()); grads: HOST array of count device
// pointers; blockmap: DEVICE array of nblocks {int32 tensor (0-based within the launch), int32 chunk} covering every tensor in
// chunks of mdx_adam_chunk() elements; lr_ptr: device float32 (a captured step's learning rate) or NULL (then lr).
// Adam without weight decay, amsgrad or maximize -- what the reference configures (loader.py:93-95).
MDX_EXPORT int mdx_adam_step(const void *table, int first, int count, const float *const *grads, const void *blockmap, int nblocks,
                             const float *lr_ptr, double lr, double beta1, double beta2, double eps, void *stream)
{
    if (!table || !grads || !blockmap) return MDX_ERR_NULL_POINTER;
    if (first < 0 || count < 1 || count > ADAM_MAX_GRADS || nblocks < 1) return MDX_ERR_BAD_SHAPE;
    AdamGrads G;
    for (int i = 0; i < count; ++i) {
        if (!grads[i]) return MDX_ERR_NULL_POINTER;
        G.g[i] = grads[i];
    }
    for (int i = count; i < ADAM_MAX_GRADS; ++i) G.g[i] = nullptr;
    hipLaunchKernelGGL(adam_step_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const AdamTensor *)table, G, first,
                       (const int2 *)blockmap, lr_ptr, lr, beta1, beta2, eps);
    return check_launch();
}
