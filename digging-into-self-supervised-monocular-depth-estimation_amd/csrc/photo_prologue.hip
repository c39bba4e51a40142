// photo_prologue.hip -- what the four scales of a step have in common, evaluated ONCE per step (gfx950, MI355X).
//
// compute.compute_loss (model_tool/processor.py:166-218) forms, inside its loop over the scales, three things that do
// not depend on the scale's disparity:
//   * the identity losses ReprojectionLoss(color_f, target) of every source frame         (processor.py:187-191);
//   * the SSIM window statistics of the TARGET image, mu_y and sigma_y per colour channel  (model_loss.py:28-33, once
//     per SSIM call: 4 scales x S frames times per step in the reference);
//   * identity + 1e-5 * randn, and -- because the identity channels come FIRST in the concatenation and torch.min keeps
//     the first minimum (processor.py:194-204) -- the best identity channel of every pixel: value `bid` and index `fi`.
//     The auto-mask of scale s then is "min over the reprojection channels < bid_s".
// Round 2's training kernel re-derived the target statistics in each of its four per-scale work items and read the
// identity and noise maps (16 B per pixel and scale); this kernel writes them once: tstat [B,H,W,6] (mu_y x 3, sigma_y x 3,
// interleaved: the training kernel fetches a pixel's six values with two load instructions), per scale bidfi_s
// [B,H,W,2] = (bid as float32, fi as int32: one 8-byte load), and optionally the identity maps themselves.
//
// The noise: either injected (noise[s] [B,S,H,W]: parity tests hand over the reference's captured torch.randn draws) or
// drawn here -- Philox4x32-10 keyed by a device-resident {seed, offset} pair + Box-Muller on the hardware log2 / sqrt /
// sin / cos -- so that a training step neither launches a generator kernel nor moves 4*S*4 bytes per pixel twice
// ("in-kernel Philox for training, injected tensor for parity").  The reference draws on the HOST
// (CPU generator, processor.py:195): no device stream can equal it; what is kept is the distribution -- N(0,1),
// independent over pixels, frames, scales and steps (tests/test_gpu_prologue.py).  The offset is advanced on the device
// by the step's finishing kernel (photo_train.hip), so a hipGraph replay draws fresh numbers.
//
// Tiling as photo_fwd.hip's identity pass: one 256-thread block per 64x8 tile, target and source tiles + 1-pixel
// reflected halo staged in LDS with 16-byte loads, two adjacent rows per thread.  Arithmetic and its order are those
// of mdx_device.hpp: every value written is bit-identical to what the per-scale kernels derive.
#include "photo_common.hpp"

namespace mdx {

struct PrologueArgs {
    mdx_desc d;                            // B, H, W, S, flags (h, w unused)
    int nscales;
    const float *target;
    mdx_sources src;
    const float *noise[MDX_MAX_SCALES];    // injected noise per scale, or all null: drawn here
    const unsigned long long *rng;         // device {seed, offset} (drawn noise)
    float *ident;                          // optional [B,S,H,W]
    float *tstat;                          // [B,H,W,6]
    float *bidfi[MDX_MAX_SCALES];          // [B,H,W,2]
};

// Philox4x32-10 (Salmon et al., SC'11): counter c, key k -> four 32-bit words
struct U4 { unsigned x, y, z, w; };

MDX_DEV U4 philox4x32_10(U4 c, unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// two N(0,1) from two words: r = sqrt(-2 ln u1), (r cos 2 pi u2, r sin 2 pi u2); u1 in (0,1), u2 in [0,1).
// v_log_f32 is log2, v_sin_f32 / v_cos_f32 take revolutions: no range reduction, no multiplications by 2 pi
MDX_DEV void box_muller(unsigned a, unsigned b, float &n0, float &n1)
{
    const float u1 = __builtin_fmaf((float)(a >> 8), 5.9604644775390625e-8f, 2.98023223876953125e-8f);   // (k + 0.5) 2^-24
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));           // -2 ln 2 log2(u1)
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}

template <int S>
__global__ __launch_bounds__(NT, S <= 2 ? 5 : (S == 3 ? 4 : 3)) void photometric_prologue_kernel(PrologueArgs a)
{
    __shared__ float s_t[3][FY][FX];
    __shared__ float s_x[S][3][FY][FX];

    const mdx_desc &d = a.d;
    const int H = d.H, W = d.W;
    const size_t HW = (size_t)H * W;
    const TileId tile = tile_id();
    const int b = tile.b, x0 = tile.tx * TX, y0 = tile.ty * TY;
    const int tid = threadIdx.x;
    const float *tgt_b = a.target + (size_t)b * 3 * HW;
    const bool automask = (d.flags & MDX_FLAG_AUTOMASK) != 0;      // off: only the target statistics are wanted

    // every load of the block in flight before the first LDS write (photo_common.hpp, load_plane_tiles)
    if (automask) {
        float (*dst[3 + 3 * S])[FX];
        const float *planes[3 + 3 * S];
#pragma unroll
        for (int c = 0; c < 3; ++c) { dst[c] = s_t[c]; planes[c] = tgt_b + c * HW; }
#pragma unroll
        for (int f = 0; f < S; ++f)
#pragma unroll
            for (int c = 0; c < 3; ++c) { dst[3 + 3 * f + c] = s_x[f][c]; planes[3 + 3 * f + c] = a.src.img[f] + ((size_t)b * 3 + c) * HW; }
        load_plane_tiles<1, 3 + 3 * S>(dst, planes, H, W, x0, y0, tid);
    } else {
        float (*dst[3])[FX] = {s_t[0], s_t[1], s_t[2]};
        const float *planes[3] = {tgt_b, tgt_b + HW, tgt_b + 2 * HW};
        load_plane_tiles<1, 3>(dst, planes, H, W, x0, y0, tid);
    }
    __syncthreads();

    const int tx = tid & 63;
    const int px = x0 + tx;
    const bool drawn = a.noise[0] == nullptr;
    unsigned k0 = 0, k1 = 0, o0 = 0, o1 = 0;
    if (automask && drawn) {
        const unsigned long long seed = a.rng[0], off = a.rng[1];
        k0 = (unsigned)seed; k1 = (unsigned)(seed >> 32); o0 = (unsigned)off; o1 = (unsigned)(off >> 32);
    }
    constexpr int ROWS = TY / (NT / 64);
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const int r = ROWS * (tid >> 6) + q;
        const int py = y0 + r;
        const bool valid = px < W && py < H;
        const unsigned p = (unsigned)((valid ? py : 0) * W + (valid ? px : 0));
        float y9[3][9];
        TargetStats ts[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int k = 0; k < 9; ++k) y9[c][k] = s_t[c][r + k / 3][tx + k % 3];
            ts[c] = target_stats(y9[c]);
        }
        if (valid) {      // sigma_y = pool(y*y) - mu_y^2: ssim_raw()'s own subtraction
            char *tp = reinterpret_cast<char *>(a.tstat + (size_t)b * 6 * HW) + p * 24u;
            float3_a4 m, sg;
            m.x = ts[0].mu; m.y = ts[1].mu; m.z = ts[2].mu;
            sg.x = ts[0].e2 - ts[0].mu2; sg.y = ts[1].e2 - ts[1].mu2; sg.z = ts[2].e2 - ts[2].mu2;
            *reinterpret_cast<float3_a4 *>(tp) = m;
            *reinterpret_cast<float3_a4 *>(tp + 12) = sg;
        }
        if (!automask) continue;
        float rl[S];
#pragma unroll
        for (int f = 0; f < S; ++f) {
            float ss[3], ad[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float x9[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) x9[k] = s_x[f][c][r + k / 3][tx + k % 3];
                ss[c] = clamp01(ssim_raw(pred_stats(x9, y9[c]), ts[c]));
                ad[c] = fabsf(y9[c][4] - x9[4]);
            }
            rl[f] = reprojection_combine(ss, ad);
            if (a.ident && valid) at32(a.ident + ((size_t)b * S + f) * HW, p) = rl[f];
        }
        // the noise of this pixel: normal j = scale * S + frame
        float nz[MDX_MAX_SCALES * S];
        if (drawn) {
            const unsigned pix = (unsigned)((size_t)b * HW) + p;
#pragma unroll
            for (int call = 0; call < (MDX_MAX_SCALES * S + 3) / 4; ++call) {
                if (call * 4 >= a.nscales * S) break;
                const U4 w = philox4x32_10(U4{pix, (unsigned)call, o0, o1}, k0, k1);
                float n0, n1, n2, n3;
                box_muller(w.x, w.y, n0, n1);
                box_muller(w.z, w.w, n2, n3);
                if (call * 4 + 0 < MDX_MAX_SCALES * S) nz[call * 4 + 0] = n0;
                if (call * 4 + 1 < MDX_MAX_SCALES * S) nz[call * 4 + 1] = n1;
                if (call * 4 + 2 < MDX_MAX_SCALES * S) nz[call * 4 + 2] = n2;
                if (call * 4 + 3 < MDX_MAX_SCALES * S) nz[call * 4 + 3] = n3;
            }
        }
#pragma unroll
        for (int s = 0; s < MDX_MAX_SCALES; ++s) {
            if (s >= a.nscales) break;
            float best = 0.f;
            int bi = 0;
#pragma unroll
            for (int f = 0; f < S; ++f) {
                const float n = drawn ? nz[s * S + f] : at32(a.noise[s] + ((size_t)b * S + f) * HW, p);
                const float t = 1e-5f * n;                 // identity_loss + 0.00001 * randn: mul, then add
                const float v = rl[f] + t;
                if (f == 0 || v < best) { best = v; bi = f; }    // torch.min: the first minimum
            }
            if (valid) {
                float2_a4 o;
                o.x = best;
                o.y = __builtin_bit_cast(float, bi);
                *reinterpret_cast<float2_a4 *>(reinterpret_cast<char *>(a.bidfi[s] + (size_t)b * 2 * HW) + p * 8u) = o;
            }
        }
    }
}

// the step's finishing kernel advances the offset; a prologue used on its own (tests) advances it here
__global__ void rng_advance_kernel(unsigned long long *rng) { rng[1] += 1ull; }

}  // namespace mdx

using namespace mdx;

MDX_EXPORT int mdx_photometric_prologue(const mdx_train_desc *td, const float *target, const mdx_sources *src,
                                        const float *const *noise, unsigned long long *rng_state, int advance_rng,
                                        float *ident, float *tstat, float *const *bidfi, void *stream)
{
    if (!td || !target || !tstat) return MDX_ERR_NULL_POINTER;
    if (td->nscales < 1 || td->nscales > MDX_MAX_SCALES) return MDX_ERR_BAD_SHAPE;
    mdx_desc d = {td->B, td->H, td->W, td->H, td->W, td->S, td->flags, td->disp_a, td->disp_b};
    int rc = validate_desc(&d);
    if (rc) return rc;
    if (!aligned(target, 16) || !aligned(tstat, 8)) return MDX_ERR_MISALIGNED;
    const bool automask = (d.flags & MDX_FLAG_AUTOMASK) != 0;
    PrologueArgs a = {};
    a.d = d; a.nscales = td->nscales; a.target = target; a.tstat = tstat; a.ident = ident; a.rng = rng_state;
    if (automask) {
        if (!src || !bidfi) return MDX_ERR_NULL_POINTER;
        for (int f = 0; f < d.S; ++f) {
            if (!src->img[f]) return MDX_ERR_NULL_POINTER;
            if (!aligned(src->img[f], 16)) return MDX_ERR_MISALIGNED;
        }
        a.src = *src;
        const bool injected = noise != nullptr && noise[0] != nullptr;
        if (!injected && !rng_state) return MDX_ERR_NULL_POINTER;
        for (int s = 0; s < td->nscales; ++s) {
            if (!bidfi[s]) return MDX_ERR_NULL_POINTER;
            if (injected && !noise[s]) return MDX_ERR_NULL_POINTER;
            a.bidfi[s] = bidfi[s]; a.noise[s] = injected ? noise[s] : nullptr;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = tile_grid(&d);
    switch (d.S) {
    case 1: hipLaunchKernelGGL((photometric_prologue_kernel<1>), grid, dim3(NT), 0, st, a); break;
    case 2: hipLaunchKernelGGL((photometric_prologue_kernel<2>), grid, dim3(NT), 0, st, a); break;
    case 3: hipLaunchKernelGGL((photometric_prologue_kernel<3>), grid, dim3(NT), 0, st, a); break;
    case 4: hipLaunchKernelGGL((photometric_prologue_kernel<4>), grid, dim3(NT), 0, st, a); break;
    default: return MDX_ERR_BAD_SHAPE;
    }
    if ((rc = check_launch())) return rc;
    if (advance_rng && rng_state && automask && !(noise && noise[0])) {
        hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, st, rng_state);
        rc = check_launch();
    }
    return rc;
}
