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
#include "photo_train_math.hpp"

namespace mdx {

constexpr int SW = 60;   // output columns per wave (64 lanes - 2 x 2 halo lanes)

// Round 4: the A/B switches of rounds 2-3 (MDX_TRAIN_FASTDIV, LAZY_COEF, SLOAD, OPAQUE_*, LATE_*, LOSS_LDS, XCD_GROUP,
// PRE_LATE on; WU_NORM, WU_CORNER, DEPTH_LDS, DEFER_GUP off) are resolved in the source: the winners are the code, the
// losers' measurements live in DESIGN.md section 4.1.  What remains switchable: -DMDX_TRAIN_STAMPS (diagnostic build with
// the in-kernel phase clock) and -DMDX_DEV_SWITCHES (chunk-schedule override for sweeps).

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
    float *stash;        // LOW form: [items][3 rows][S][64 lanes] float2 -- projected positions (u, v) of the last three warped rows
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


// GRAD = true: the training form (loss, indices AND the unit-upstream gradients).  GRAD = false: validation /
// torch.no_grad() (model_train.py:75-79 runs the loss on the validation split every epoch): the same marching wave
// without the coefficient histories, the stash and the gradient phase -- every scale's forward in ONE launch.
// PRE = true: the step's prologue kernel has evaluated what the scales share (target statistics, best identity channel):
// they are loaded (8 dwords and a byte per pixel) instead of re-derived (6 pools) / re-read (2 S dwords) per scale.
// Round 4, the LOW form of the training kernel (S >= 3): one more wave per SIMD.  The launch time follows the resident
// waves (S = 2, same code: 2 waves 261 us, 3 waves 204 us), and S = 3 (mono + stereo, BASELINE configs[4]) sat at 2 waves:
// 203 registers and 18.7 KB of LDS per wave.  State that is touched once per step or less leaves the registers / the LDS:
//   * the d(P) accumulators (9 S values per lane, touched only in the gradient phase): LDS cells, read - add - write at the
//     lane's own address (ds_add_f32, the unit's atomic path, DOUBLED the launch time: 18 per frame and step);
//   * the two older rows of the SSIM coefficient history (18 values per lane, read only by the gradient phase): an LDS
//     ring; the newest row stays in registers through the step that forms it and is written behind its gradient phase;
//   * the stash ring (sampling derivatives + projected position of the last three warped rows, 18 KB per wave at S = 3: what
//     capped the LDS occupancy) is gone.  Only the projected position (u, v) of a warped pixel is kept -- 8 bytes per frame
//     in a per-item ring in GLOBAL memory (4.6 KB per wave: L2-resident) -- and the gradient phase re-derives the tap from it
//     (the forward's own arithmetic on the forward's own bits: the same tap), re-reads the corners (cache hits) and forms
//     the sampling derivatives itself: only in waves where the frame is some lane's arg-min, not for every frame and step.
// S = 3: 203 -> 146 registers, 18.7 -> 11.8 KB LDS: 3 waves per SIMD, 340 -> 297 us on bench.py's mono + stereo batch.
// S <= 2 keeps the register form: there the same machinery costs more than a fourth wave gives (198 us in
// registers at 3 waves; LOW 251 us at 4 waves, 241 us with the full stash in the global ring, 223 us at 3 waves).
constexpr bool train_low(int S, bool grad) { return grad && S >= 3; }
constexpr int train_waves(int S, bool grad)
{
    return grad ? (S == 1 ? 4 : (S <= 3 ? 3 : 2)) : (S <= 3 ? 4 : 3);
}
template <int S, bool GRAD, bool PRE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(train_waves(S, GRAD), GRAD ? train_waves(S, GRAD) : 8))) void photometric_train_kernel(TrainArgs a)
{
    constexpr bool LOW = train_low(S, GRAD);
    // per-lane stash ring: [row slot][2f] = (d colour_c / du, u), [2f+1] = (d colour_c / dv, v) of frame f
    // (LOW: in global memory, a.stash)
    __shared__ float4 s_stash[(GRAD && !LOW) ? 3 : 1][(GRAD && !LOW) ? 2 * S : 1][(GRAD && !LOW) ? 64 : 1];
    __shared__ float s_acc[LOW ? 9 * S : 1][LOW ? 64 : 1];      // LOW: d(P) accumulators [f * 9 + {A, B, C} * 3 + i][lane]
    __shared__ float s_ch[LOW ? 2 : 1][LOW ? 9 : 1][LOW ? 64 : 1];   // LOW: coefficient rows sr-2, sr-1 (slot = step parity)
    // loss partial of each lane: an LDS cell, updated by one ds_add_f32 per row (own address: sequential, deterministic).
    // As a register it was the value the allocator spilled in the <2, GRAD, PRE> instantiation -- a scratch load,
    // s_waitcnt vmcnt(0), add, scratch store in every step.  float32: at most 44 addends in [0, 1] per lane (error ~1e-7
    // of the lane's sum, independent between the ~1.5 M lanes); lanes, items and scales are summed in float64.
    __shared__ float s_loss[64];

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
    const int chunk = (lev == 2 ? a.lev_k0[2] : (lev == 1 ? a.lev_k0[1] : a.lev_k0[0])) + kk;
    // slot of the item's partials: the items of one (scale, image) contiguous, whatever the dispatch order
    const unsigned item = (unsigned)(((scale * a.B + b) * a.nchunks + chunk) * a.nstrips + strip);

    mdx_desc d;
    d.B = a.B; d.H = a.H; d.W = a.W; d.S = S; d.flags = a.flags; d.disp_a = a.disp_a; d.disp_b = a.disp_b;
    d.h = pick(a.h, scale);
    d.w = pick(a.w, scale);
    const int H = d.H, W = d.W;
    const size_t HW0 = (size_t)H * W;
    // Inside the row loop the plane size is made opaque once per step: otherwise the dozen plane
    // bases `tensor + (b*planes + c)*HW` are hoisted out of the loop as 64-bit scalars -- and spilled to VGPR lanes;
    // recomputing them per step costs scalar-unit instructions only.
    unsigned HW = (unsigned)HW0;
    const float *disp_b = pick(a.disp, scale) + (size_t)b * d.h * d.w;
    const float *P_s = pick(a.P, scale);
    const float *noise_s = pick(a.noise, scale);
    // The per-scale output pointers (and the best-identity map) are fetched from the kernel-argument
    // segment WHERE THEY ARE USED -- one s_load_dwordx2 at (array offset + 8 * scale), two scalar registers for a few
    // instructions.  Held across the row loop they were loop-invariant 64-bit scalars the allocator spilled to VGPR lanes,
    // and every use paid two v_readlane (8.5 cycles each on the vector ALU that bounds this kernel).  (Re-deriving them with
    // pick() -- four pointer loads and a select chain -- at the use sites made it worse: 28 -> 49 spilled scalars.)
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
    const float *invK_b = a.invK + b * 16;
    const float *tgt_b = a.target + (size_t)b * 3 * HW0;
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
    const float *Pp[S];
#pragma unroll
    for (int f = 0; f < S; ++f) Pp[f] = uniform_ptr(P_s + ((size_t)f * d.B + b) * 12);
    const float *iKp = uniform_ptr(invK_b);

    // ---- histories (registers) ----
    float xh[3][S][3];   // warped colours, rows wr-2 .. wr
    float yh[3][3];      // target colours, same rows
    constexpr int CR = LOW ? 1 : 3, CN = CR - 1;   // coefficient rows held in registers; index of the newest
    float ch[CR][3][3];  // [row][channel][alpha,beta,gamma] of the arg-min frame, rows sr-2 .. sr (LOW: row sr only)
    int selp = 0;        // arg-min frame + 1 (0 = none) of rows sr-2, sr-1, sr in bits 0-3, 4-7, 8-11
    int flp = 0;         // grid_sample pass flags (bit 2f: x inside, 2f+1: y inside) of rows wr-2, wr-1, wr in bytes 0-2
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            yh[j][c] = 0.f;
#pragma unroll
            for (int f = 0; f < S; ++f) xh[j][f][c] = 0.f;
            if (j < CR)
#pragma unroll
                for (int k = 0; k < 3; ++k) ch[j][c][k] = 0.f;
        }
    }
    if constexpr (LOW) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_ch[0][k][lane] = s_ch[1][k][lane] = 0.f;
#pragma unroll
        for (int k = 0; k < 9 * S; ++k) s_acc[k][lane] = 0.f;
    }
    float dph[3] = {0.f, 0.f, 0.f};
    // d(P) accumulators.  With X_j = depth * r_j and the pixel ray r = invK (px, py, 1) linear in the pixel, the twelve
    // sums of a frame follow from nine: A_i = sum gq_i*depth, Bv_i = sum py*gq_i*depth, C_i = sum gq_i (px is a
    // constant of the lane and is applied at the end).
    constexpr int AS = LOW ? 1 : S;
    float accA[AS][3], accB[AS][3], accC[AS][3];
    if constexpr (!LOW)
#pragma unroll
    for (int f = 0; f < S; ++f)
#pragma unroll
        for (int i = 0; i < 3; ++i) accA[f][i] = accB[f][i] = accC[f][i] = 0.f;
    // LOW: the item's (u, v) ring in global memory; a slot is S planes of 64 float2.  A slot's base is handed to the
    // accesses as an OPAQUE SCALAR (as the target planes are, below): left to the compiler, `ring + 16 * lane` is hoisted
    // out of the loop as a 64-bit per-lane pointer (two registers, spilled) and the slot offset added per lane in 64 bits
    const unsigned long long stash_item =
        LOW ? (unsigned long long)(reinterpret_cast<char *>(a.stash) + (size_t)item * (3u * S * 512u)) : 0ull;
    auto stash_slot = [&](int slot, int plane) {
        unsigned long long v = stash_item + (unsigned)slot * (S * 512u) + (unsigned)plane * 512u;
        asm volatile("" : "+s"(v));
        typedef __attribute__((address_space(1))) float2 *gptr;
        return (float2 *)(gptr)v;
    };
    // ... and the lane's element index as an opaque 32-bit value formed next to the access: hoisted out of the loop it becomes a
    // zero-extended 64-bit pair, and the access loses its scalar-base + 32-bit-offset form (v_lshl_add_u64 per access)
    auto stash_lane = [&](unsigned plane_off) {
        unsigned v = (unsigned)lane;
        asm volatile("" : "+v"(v));
        return v + plane_off;
    };
    s_loss[lane] = 0.f;

    // ---- prefetch registers: the loads of the NEXT step whose addresses do not depend on computed data ----
    float pf_y[3], pf_d[4], pf_id[S], pf_nz[S];
    float pf_tyl1 = 0.f;                   // the prefetched row's bilinear y weight (wave-uniform; one register instead of
                                           // ten dependent instructions to form it again at the head of the next step)
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
            pf_tyl1 = ty.l1;
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
    // (PRE fetches the SSIM row's statistics inside the step, behind the corner loads, not a step ahead)
    if (!PRE) prefetch_ssim_row(r0 - (GRAD ? 3 : 2));

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
        asm volatile("" : "+s"(HW));
        SRows s_iK, s_P[S];
        sload12(iKp, s_iK);
#pragma unroll
        for (int f = 0; f < S; ++f) sload12(Pp[f], s_P[f]);

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
                UpTap ty;                  // (i0, i1 served the prefetch's addresses; the weights are all that is left to use)
                ty.i0 = ty.i1 = 0;
                ty.l1 = pf_tyl1;
                ty.l0 = 1.0f - pf_tyl1;
                up = up_combine(pf_d[0], pf_d[1], pf_d[2], pf_d[3], ty, tx_tap(), premul);
            }
            float iK[12], Pm[S][12];
            swait12(s_iK, iK);
#pragma unroll
            for (int f = 0; f < S; ++f) swait12(s_P[f], Pm[f]);
            const PixelGeom g = geom_from_disp_train(d, up, iK, pxr, pyr);
            dph[0] = dph[1]; dph[1] = dph[2]; dph[2] = g.depth;
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
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    cn[f][c] = load_corners_finite(a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, tp[f]);
            }
            prefetch_warp_row(wr + 1);
            // PRE: this step's SSIM row -- three loads whose latency the sampling arithmetic below covers; fetched a step
            // ahead they would hold eight more registers across the loop edge (6 spills at 168 VGPRs)
            if (PRE) prefetch_ssim_row(sr);
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
                    if constexpr (GRAD && !LOW) {
                        du[c] = (cn[f][c].ne - cn[f][c].nw) * dy1 + (cn[f][c].se - cn[f][c].sw) * dy0;
                        dv[c] = (cn[f][c].sw - cn[f][c].nw) * dx1 + (cn[f][c].se - cn[f][c].ne) * dx0;
                    }
                }
                if constexpr (LOW) {
                    at32(stash_slot(slot_w, f), stash_lane(0u)) = make_float2(pr[f].u, pr[f].v);
                } else if constexpr (GRAD) {
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
                for (int k = 0; k < 3; ++k)
                    if constexpr (!LOW) { ch[0][c][k] = ch[1][c][k]; ch[1][c][k] = ch[2][c][k]; }
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
                SsimMid mid[3];
                float po[3][3];
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
                    po[c][0] = o[0]; po[c][1] = o[1]; po[c][2] = o[2];
                    ad[c] = fabsf(yh[1][c] - xh[1][f][c]);
                }
                ssim_value_mid3(po, ts, mid, ss);      // (forward-only form: the intermediates kept in `mid` are dead code)
                const float rl = reprojection_combine(ss, ad);
                const bool better = f == 0 || rl < best_r;
                best_r = better ? rl : best_r;
                fr = better ? f : fr;
                if constexpr (GRAD) {
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
                // the candidate as it stands: where no reprojection frame is this lane's arg-min (fr = -1) the row's
                // selection code below is 0 and the gradient phase weights the row with 0 for every frame -- the (finite)
                // values need no clearing (nine selects per step)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) ch[CN][c][k] = cand[c][k];
                selp |= (fr + 1) << 8;
            }
            // `best` and `bi` are pinned in front of the store branch by an empty volatile asm.  In the forward-only form
            // nothing but the stores below consumes them, and the optimiser then SINKS the whole SSIM evaluation into that
            // lane-divergent branch -- except the neighbour-lane reads, which are convergent and stay outside; the backend
            // folds a DPP read into the add that consumes it only within one basic block, so all 108 of them became
            // separate v_mov_dpp (and 76 values were spilled).
            asm volatile("" : "+v"(best), "+v"(bi));
            if (out_lane && sr >= r0 && sr < r1) {
                const unsigned po = (unsigned)(sr * W + pxr);
                at32(MDX_LATE(uint8_t *, idx) + (size_t)b * HW, po) = (uint8_t)bi;
                float *const to_opt_p = MDX_LATE(float *, to_opt);
                if (to_opt_p) at32(to_opt_p + (size_t)b * HW, po) = best;
                // the cell's address from a lane id formed HERE (two v_mbcnt, volatile so that they are not hoisted): as a
                // loop-invariant register it was the next value to be spilled
                unsigned lid;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
                (void)__hip_atomic_fetch_add(&s_loss[lid], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        } else if constexpr (GRAD) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) ch[CN][c][k] = 0.f;
        }
        if (!PRE) prefetch_ssim_row(sr + 1);   // behind its last use: one step of cover, no second copy of the values

        MDX_STAMP(2);   // SSIM phase
        // ================= (3) gradient of row gr =================
        if constexpr (GRAD) {
        if (t >= 4) {
        // everything in this block feeds gradients only (bar: 1e-4 against the oracle, not bit-exactness): the compiler may
        // fuse its multiply-adds -- the file is built with -ffp-contract=off for the sake of the VALUES
        MDX_GRAD_FP
        const float wy0 = gr == 1 ? 2.f : 1.f, wy2 = gr == H - 2 ? 2.f : 1.f;   // reflection-pad fold (y)
        float iK[12], Pm[LOW ? 1 : S][12];
        {   // re-read right here: issued before the SSIM phase, or kept from the top of the step, the longer live
            // ranges cost more in spills than the exposed scalar-cache latency (300 us against 314 / 312 us, kbench).
            // LOW (S >= 3): a frame's matrix inside its own pass of the frame loop -- three at once, on top of what the
            // re-derived tap needs, spilled 44 scalars to VGPR lanes (78 v_readlane / v_writelane per step)
            SRows g_iK, g_P[LOW ? 1 : S];
            sload12(iKp, g_iK);
            if constexpr (!LOW)
#pragma unroll
                for (int f = 0; f < S; ++f) sload12(Pp[f], g_P[f]);
            swait12(g_iK, iK);
            if constexpr (!LOW)
#pragma unroll
                for (int f = 0; f < S; ++f) swait12(g_P[f], Pm[f]);
        }
        float r3[3];
        pixel_ray(iK, (float)pxr, (float)gr, r3);     // (float)pxr: one conversion per step instead of a register per item
        const float depth = dph[0];
        const float fgr = (float)gr;
        const int sel0 = selp & 15, sel1 = (selp >> 4) & 15, sel2 = (selp >> 8) & 15;
        float gdepth = 0.f;
        // LOW: the two older coefficient rows come out of the LDS ring (row sr-2: this step's slot, row sr-1: the other);
        // not at all in waves where no lane selected any frame in the three rows (auto-masked regions)
        // (read per frame and colour channel right where the vertical sums are formed: 6 values in flight, not 18)
        bool wave_any = true;
        if constexpr (LOW) wave_any = __builtin_amdgcn_ballot_w64((selp & 0xfff) != 0) != 0;
        if (wave_any)
#pragma unroll
        for (int f = 0; f < S; ++f) {
            const int own = (sel0 == f + 1) | (sel1 == f + 1) | (sel2 == f + 1);
            const int hit = own | dpp_i<0x138>(own) | dpp_i<0x130>(own);
            if (__builtin_amdgcn_ballot_w64(hit != 0 && out_lane) == 0) continue;   // wave-uniform
            const float w0 = sel0 == f + 1 ? wy0 : 0.f, w1 = sel1 == f + 1 ? 1.f : 0.f, w2 = sel2 == f + 1 ? wy2 : 0.f;
            float4 sa, sb;
            float du[3], dv[3];
            Tap gt;
            Corners gc[3];
            if constexpr (LOW) {
                // the row's projected position out of the ring; from it the forward's tap (project_point_train's last two
                // lines and make_tap on the same bits), the corners again, and -- further down, behind the coefficient sums
                // that cover the loads -- the sampling derivatives
                const float2 uv = at32(stash_slot(slot_r, f), stash_lane(0u));
                sa.w = uv.x; sb.w = uv.y;
                gt = make_tap((div_norm(uv.x, nd.w) - 0.5f) * 2.0f, (div_norm(uv.y, nd.h) - 0.5f) * 2.0f, H, W);
#pragma unroll
                for (int c = 0; c < 3; ++c) gc[c] = load_corners_finite(a.src.img[f] + ((size_t)b * 3 + c) * HW, H, W, gt);
            } else {
                sa = s_stash[slot_r][2 * f][lane];
                sb = s_stash[slot_r][2 * f + 1][lane];
                du[0] = sa.x; du[1] = sa.y; du[2] = sa.z; dv[0] = sb.x; dv[1] = sb.y; dv[2] = sb.z;
            }
            const bool centre = sel1 == f + 1;
            float gu = 0.f, gv = 0.f;
            // vertical 3-tap sums (own column, registers) of all nine coefficient maps first, then -- behind a
            // scheduling barrier, see the note on DPP operands above -- the horizontal ones (neighbour lanes)
            float vs[3][3], sum3[3][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float ca[3], cb[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    ca[k] = LOW ? s_ch[LOW ? (t & 1) : 0][LOW ? c * 3 + k : 0][LOW ? lane : 0] : ch[0][c][k];
                    cb[k] = LOW ? s_ch[LOW ? ((t + 1) & 1) : 0][LOW ? c * 3 + k : 0][LOW ? lane : 0] : ch[LOW ? 0 : 1][c][k];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float v = w0 * ca[k];
                    v = __builtin_fmaf(w1, cb[k], v);
                    vs[c][k] = __builtin_fmaf(w2, ch[CN][c][k], v);
                }
                if constexpr (LOW) __builtin_amdgcn_sched_barrier(0);
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
            if constexpr (LOW) {
                const float dy1 = (float)(gt.y0 + 1) - gt.iy, dy0 = gt.iy - (float)gt.y0;
                const float dx1 = (float)(gt.x0 + 1) - gt.ix, dx0 = gt.ix - (float)gt.x0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    du[c] = (gc[c].ne - gc[c].nw) * dy1 + (gc[c].se - gc[c].sw) * dy0;
                    dv[c] = (gc[c].sw - gc[c].nw) * dx1 + (gc[c].se - gc[c].ne) * dx0;
                }
            }
            // the three colour channels in lock step (three independent chains instead of one after the other)
            float gxc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gxc[c] = 2.0f * xh[0][f][c];
#pragma unroll
            for (int c = 0; c < 3; ++c) gxc[c] = __builtin_fmaf(gxc[c], sum3[c][1], sum3[c][0]);
#pragma unroll
            for (int c = 0; c < 3; ++c) gxc[c] = __builtin_fmaf(yh[0][c], sum3[c][2], gxc[c]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float xq = xh[0][f][c], yq = yh[0][c];
                const float l1 = centre ? 0.05f * ((yq > xq) ? 1.f : ((yq < xq) ? -1.f : 0.f)) : 0.f;   // 0.15*mean_c|y-x|
                gxc[c] = __builtin_fmaf(gxc[c], 1.0f / 9.0f, -l1);
            }
            gu = __builtin_fmaf(gxc[2], du[2], __builtin_fmaf(gxc[1], du[1], gxc[0] * du[0]));
            gv = __builtin_fmaf(gxc[2], dv[2], __builtin_fmaf(gxc[1], dv[1], gxc[0] * dv[0]));
            // grid normalisation (2/(W-1)) and grid_sample's un-normalisation ((W-1)/2) cancel
            gu = ((LOW ? gt.inx : (((flp >> (2 * f)) & 1) != 0)) && out_lane) ? gu : 0.f;
            gv = ((LOW ? gt.iny : (((flp >> (2 * f + 1)) & 1) != 0)) && out_lane) ? gv : 0.f;
            if constexpr (LOW) {
                SRows g_P;
                sload12(Pp[f], g_P);
                swait12(g_P, Pm[0]);
            }
            const float *Pf = Pm[LOW ? 0 : f];
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
                if constexpr (LOW) {
                    // (the cell's address from a lane id formed here, as for the loss cell: two v_mbcnt, not a register
                    // held across the loop)
                    unsigned lid;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
                    // read - add - write of the lane's own cells (nobody else touches them): plain LDS traffic; ds_add_f32, the
                    // unit's atomic path, doubled the launch time at 9 per frame and step
                    s_acc[f * 9 + i][lid] = s_acc[f * 9 + i][lid] + gd;
                    s_acc[f * 9 + 3 + i][lid] = __builtin_fmaf(fgr, gd, s_acc[f * 9 + 3 + i][lid]);
                    s_acc[f * 9 + 6 + i][lid] = s_acc[f * 9 + 6 + i][lid] + gq[i];
                } else {
                    accA[f][i] += gd;
                    accB[f][i] = __builtin_fmaf(fgr, gd, accB[f][i]);
                    accC[f][i] += gq[i];
                }
            }
        }
        // depth = 1/(a + b*disp)  ->  d depth / d disp = -b * depth^2
        if (out_lane) at32(MDX_LATE(float *, gup) + (size_t)b * HW, (unsigned)(gr * W + pxr)) = gdepth * (-d.disp_b * depth * depth);
        MDX_STAMP(3);   // gradient phase
        }
        if constexpr (LOW) {
            // the row formed in this step ages into the ring (over row sr-2, which the gradient phase above was the last to read)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) s_ch[t & 1][c * 3 + k][lane] = ch[CN][c][k];
        }
        }
    }
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
            const float vA = LOW ? s_acc[LOW ? f * 9 + i : 0][LOW ? lane : 0] : accA[LOW ? 0 : f][i];
            const float vB = LOW ? s_acc[LOW ? f * 9 + 3 + i : 0][LOW ? lane : 0] : accB[LOW ? 0 : f][i];
            const float vC = LOW ? s_acc[LOW ? f * 9 + 6 + i : 0][LOW ? lane : 0] : accC[LOW ? 0 : f][i];
            const float sA = wave_sum_dpp_lane63(vA);
            const float sAx = wave_sum_dpp_lane63(vA * (float)pxr);
            const float sB = wave_sum_dpp_lane63(vB);
            const float sC = wave_sum_dpp_lane63(vC);
            if (lane == 63) {
                const float *iK = invK_b;
                float *o = a.partP + (size_t)item * (S * 12) + f * 12 + i * 4;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    o[j] = iK[j * 4 + 0] * sAx + iK[j * 4 + 1] * sB + iK[j * 4 + 2] * sA;
                o[3] = sC;
            }
        }
    const double acc_item = wave_sum((double)s_loss[lane]);
    if (lane == 0) a.loss_part[item] = acc_item;
}

#undef automask
#undef premul
#undef same_res

struct TrainPlan {
    int lev_n[3], lev_r[3], lev_item0[3], lev_row0[3], lev_k0[3];
    int nchunks, nstrips, ncols;
    size_t items, off_partP, off_gup, off_stash, total;
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
        if (H >= 256 && grad) {
            // BASELINE configs[3] (320 x 1024), swept in round 4 (tools/r4_sweep.sh, inside the real step): 4 x 48 + 3 x 24 + 5 x 12
            // rows 378.5 us against 390.9 us for 40 / 20 / 10 at 0.6 / 0.25 (16 chunks); 64 / 32 / 16: 387.5; 56 / 28 / 14: 397.8
            // (48 / 28 / 9 at 0.6 / 0.27 -- the taller-middle, shorter-tail pattern that won at H = 192 -- is another 1 %:
            // 387.8 / 390.5 us against 391.2 / 394.3 us in one run)
            r[0] = 48; r[1] = 28; r[2] = 9; f1 = 0.6; f2 = 0.27;
        }
        if (H >= 160 && H < 256) {
            // re-swept in round 3 for the BASELINE height (tools/sweep_schedule.sh, timing inside the real step): with the
            // gradient phase skipped in auto-masked regions a halo step costs relatively more, and fewer, taller chunks win --
            // 3 x 40 + 2 x 24 + 2 x 12 rows (7 chunks, 220 steps per column) 210.5 us against 223.7 us for round 2's
            // 2 x 40 + 2 x 20 + 8 x 10 (12 chunks, 240 steps); 3 x 40 + 3 x 24 (6 chunks) is unbalanced again (224.6 us)
            r[1] = 24; r[2] = 12; f1 = 0.63; f2 = 0.25;
            // round 4, after the row loop got cheaper where a frame wins (contraction, lock-step): a taller middle level and a
            // shorter last one -- 3 x 40 + 2 x 26 + 2 x 10 rows 192.3 us (mean of five runs) against 195.8 us for 24 / 12 in the
            // same runs; configs[4] (S = 3) 276.6 against 279.8 us (tools/r4_sweep.sh, profiles/r04_schedule_sweep.txt)
            if (grad) { r[1] = 26; r[2] = 10; f2 = 0.28; }
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
    p.off_stash = p.off_gup + (size_t)d->nscales * d->B * d->H * d->W * sizeof(float);
    // LOW form (training only): the items' (u, v) rings, 3 rows x S planes of 64 float2
    p.total = p.off_stash + (train_low(d->S, grad) ? p.items * 3 * (size_t)d->S * 512 : 0);
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
    a.stash = (float *)((char *)workspace + p.off_stash);
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
    return launch_train_finish(d, grad, p.nchunks * p.nstrips, a.partP, a.loss_part, a.gup, gdisp, gP, loss_sum, pre.rng, st);
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
