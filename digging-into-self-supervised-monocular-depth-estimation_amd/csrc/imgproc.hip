// imgproc.hip -- the per-sample image preparation of the KITTI loaders on the GPU, for gfx950.
//
// Reference call sites: model_loader/kitti_mono.py:288-291, 349-350 (transforms.Resize((H>>s, W>>s), ANTIALIAS) of the
// ORIGINAL image for every scale), 302-303 (FLIP_LEFT_RIGHT), 284-285, 352-353 (ColorJitter), 283, 351 (ToTensor).
// The arithmetic is Pillow's (not under /root/reference); these kernels reproduce it bit for bit:
//   resample    Resample.c: Lanczos-3 weights in double, normalised, rounded to 22-bit fixed point (host: mdx_resample_plan);
//               horizontal pass -> uint8 -> vertical pass -> uint8, each  clip8((2^21 + sum px*k) >> 22)
//   jitter      ImageEnhance.Brightness / Contrast / Color = Blend.c's  a + alpha*(b - a)  in float32 with truncation
//               (alpha in [0,1]) or clipping (outside); Contrast's grey level = int(mean(L) + 0.5) over the image;
//               hue = Convert.c's rgb2hsv / hsv2rgb round trip (float/double mix) with the H byte shifted modulo 256
//   ToTensor    float(u8) / 255.0f (IEEE division)
// Byte / integer work, HBM- and L2-bound: no MFMA.  A CPU worker spends 3.7 ms decoding one 1242x375 JPEG and 41 ms
// on the four resizes and the jitter of that frame (tools/loader_cost.py); with these kernels the workers only decode.
//
// Jobs travel BY VALUE in the kernel arguments (<= MDX_IMG_JOBS per launch): no device-side job table to keep alive,
// nothing to copy, capturable.  Layouts: source = interleaved RGB rows as Pillow / the JPEG decoder hand them over
// ([h][w][3] uint8, row stride in bytes); everything downstream is planar ([3][h][w]), the step's layout.
#include <algorithm>
#include <cmath>
#include <vector>
#include <cstdlib>
#include <cstring>
#include "mdx_common.hpp"

namespace mdx {

constexpr int RS_BITS = 32 - 8 - 2;          // Resample.c PRECISION_BITS

// What a launch carries (kernel arguments, by value): the distinct plans of its jobs and the jobs themselves, packed --
// a batch of 12 samples (36 frames at scale 0 + the target's three smaller scales per sample = 72 jobs) is ONE
// horizontal and ONE vertical launch.  (Round 2 passed 32 caller-layout jobs of 108 bytes per launch: three launches of
// each pass per batch, each with its own tail.)
constexpr int IMG_PLANS = 16;                // distinct (size -> size) plans per launch and axis
constexpr int IMG_JOBS = MDX_IMG_JOBS;       // jobs per launch (72)
struct PlanX {
    const int *bounds, *kk;                  // [out_w][2], [ksize][out_w]
    const int *kc;                           // rows form: mdx_resample_plan_cols table [2][out_w][kc_row]
    uint16_t in_w, out_w, ksize;             // (sizes <= 16384: validate_resample)
    uint16_t pitch;                          // rows form: LDS row pitch in bytes (4 x an odd number)
    uint16_t kc_lead, kc_row;
    uint8_t lg_cols;                         // rows form: log2(output columns per block); HW_GATHER = gather form
    uint8_t pad[3];
};
struct PlanY {
    const int *bounds, *kk;                  // [out_h][2], [ksize][out_h]
    int in_h, out_h, ksize, pad;
};
struct PackedJob {
    const uint8_t *src;
    uint8_t *inter, *dst_u8;
    float *dst_f32;
    int in_stride;
    uint8_t px, py, flags, pad;              // plan indices; flags: JOB_FLIP | JOB_VEC4 | JOB_VEC8
};
struct ResampleLaunch {
    PlanX x[IMG_PLANS];
    PlanY y[IMG_PLANS];
    PackedJob j[IMG_JOBS];
};
static_assert(sizeof(PlanX) == 40 && sizeof(PlanY) == 32 && sizeof(PackedJob) == 40, "packed launch layout");
static_assert(sizeof(ResampleLaunch) <= 4096, "kernel arguments are limited to 4 KB");
constexpr int JOB_FLIP = 1, JOB_VEC4 = 2, JOB_VEC8 = 4;
constexpr int HW_GATHER = 255;
struct JitterJobs {
    mdx_jitter_job j[MDX_JITTER_JOBS];
};
static_assert(sizeof(JitterJobs) <= 4096, "kernel arguments are limited to 4 KB");

static __device__ __forceinline__ uint8_t clip8(int v)
{
    v >>= RS_BITS;
    return (uint8_t)min(max(v, 0), 255);
}

// ToTensor's float32(byte) / 255 without the division sequence: q0 = b * rc, r = fma(-255, q0, b), q = fma(r, rc, q0)
// with rc = fl(1/255) is the correctly rounded quotient for every byte (all 256 checked in exact rational arithmetic,
// tests/test_imgproc_cpu.py; on the GPU the bit-exact tests of tests/test_gpu_imgproc.py go through it) -- 3 instructions
// instead of ~12 per value, in kernels that are bound by the vector ALU.
static __device__ __forceinline__ float unit_from_byte(unsigned b)
{
    const float rc = __builtin_bit_cast(float, 0x3b808081u);       // fl(1 / 255)
    const float fb = (float)b;
    const float q0 = fb * rc;
    const float r = __builtin_fmaf(-255.0f, q0, fb);
    return __builtin_fmaf(r, rc, q0);
}

// One byte per lane -> dwords: lane 4q gathers the bytes of lanes 4q..4q+3 (DPP row shifts; groups of four never
// straddle a 16-lane row) and stores them as one dword.  Byte-wide global stores are what bounded the first version
// of the horizontal pass (64 one-byte writes per wave instruction: 385 us for the stores alone against 194 us with the
// arithmetic spreading them out).
static __device__ __forceinline__ void store_bytes_packed(uint8_t *dst, int byte, bool vec4, bool ok, int lane)
{
    if (vec4) {
        const int b1 = __builtin_amdgcn_update_dpp(0, byte, 0x101, 0xf, 0xf, true);     // row_shl:1  (lane i <- lane i+1)
        const int b2 = __builtin_amdgcn_update_dpp(0, byte, 0x102, 0xf, 0xf, true);
        const int b3 = __builtin_amdgcn_update_dpp(0, byte, 0x103, 0xf, 0xf, true);
        if (ok && (lane & 3) == 0)
            *(unsigned *)dst = (unsigned)byte | ((unsigned)b1 << 8) | ((unsigned)b2 << 16) | ((unsigned)b3 << 24);
    } else if (ok) {
        *dst = (uint8_t)byte;
    }
}

static __device__ __forceinline__ unsigned load32_unaligned(const uint8_t *p)
{
    unsigned v;
    __builtin_memcpy(&v, p, 4);          // one global_load_dword: gfx950 global accesses need no alignment
    return v;
}

// ---- horizontal pass, GATHER form (the general fallback: any filter width, any size) ----
// interleaved RGB [in_h][in_w][3] -> planar uint8 inter [3][in_h][out_w].  A wave owns 64 consecutive output columns and
// HR rows (weights, bounds and address arithmetic are shared by the rows; HR x 2 independent loads are in flight); a
// pixel's three bytes arrive as ONE dword: bytes [3px, 3px+3], or for the last pixel of a row [3px-1, 3px+2] shifted down
// (never a byte past the row).  Every tap is a gather at a stride of 3 * in_w / out_w bytes: the address path bounds it
// (~18 CU-cycles per load instruction, profiles/r02_imgproc_kernel_pmc.txt).  Rounds 2 and 3 built three more forms with
// lanes along the columns (16-byte window loads; spans staged in LDS and read back as bytes -- ~3 bank passes per read;
// one wave per column with the taps along the lanes for 65..128-tap filters): each within 20 % of this one.  The ROWS
// form below replaced them all; this one serves what does not fit its LDS tile.
constexpr int HR = 4;
template <bool FLIP, bool EDGE>
static __device__ __forceinline__ void resample_h_body(const PackedJob &J, const PlanX &X, int in_h, int xo, int y0, bool ok, int lane)
{
    const int xmin = X.bounds[2 * xo], n = X.bounds[2 * xo + 1];
    const int *k = X.kk + xo;                      // [ksize][out_w]: a wave reads 256 consecutive bytes per tap
    const uint8_t *row[HR];
    int s[HR][3];
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        row[r] = J.src + (size_t)min(y0 + r, in_h - 1) * J.in_stride;        // y0 is wave-uniform
        s[r][0] = s[r][1] = s[r][2] = 1 << (RS_BITS - 1);
    }
    const int last = X.in_w - 1;
    unsigned off = 3u * (unsigned)(FLIP ? last - xmin : xmin);
#pragma unroll 2
    for (int t = 0; t < n; ++t) {
        const int c = k[(size_t)t * X.out_w];
        unsigned o = off, sh = 0;
        if (EDGE) {
            const bool edge = off == 3u * (unsigned)last;
            o = off - (edge ? 1u : 0u);
            sh = edge ? 8u : 0u;
        }
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            unsigned v = load32_unaligned(row[r] + o);
            if (EDGE) v >>= sh;
            s[r][0] += __mul24((int)(v & 255u), c);             // |weight| < 2^22: v_mad_i32_i24
            s[r][1] += __mul24((int)((v >> 8) & 255u), c);
            s[r][2] += __mul24((int)((v >> 16) & 255u), c);
        }
        off = FLIP ? off - 3u : off + 3u;
    }
    const size_t plane = (size_t)in_h * X.out_w;
    const bool vec4 = (J.flags & JOB_VEC4) != 0;
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        const bool row_ok = ok && y0 + r < in_h;                       // wave-uniform apart from the column tail
        const size_t o = (size_t)min(y0 + r, in_h - 1) * X.out_w + xo;
        store_bytes_packed(J.inter + o, clip8(s[r][0]), vec4, row_ok, lane);
        store_bytes_packed(J.inter + plane + o, clip8(s[r][1]), vec4, row_ok, lane);
        store_bytes_packed(J.inter + 2 * plane + o, clip8(s[r][2]), vec4, row_ok, lane);
    }
}

__global__ __launch_bounds__(256) void resample_h_kernel(ResampleLaunch L)
{
    const PackedJob &J = L.j[blockIdx.z];
    const PlanX &X = L.x[J.px];
    if (X.lg_cols != HW_GATHER) return;                              // this job runs in resample_h_rows_kernel
    const int in_h = L.y[J.py].in_h;
    const int lane = threadIdx.x & 63;
    const int xo0 = blockIdx.x * 64;
    // the wave index as a scalar: taken from threadIdx.x alone the compiler treats the row addresses as lane-varying
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int y0 = (blockIdx.y * 4 + wave) * HR;
    if (xo0 >= X.out_w || y0 >= in_h) return;                        // wave-uniform: the packed stores need whole waves
    const bool ok = xo0 + lane < X.out_w;
    const int xo = min(xo0 + lane, X.out_w - 1);
    // does any lane tap the last source pixel?  (bounds are monotonic: the last column reaches furthest)
    const int xl = min(xo0 + 63, X.out_w - 1);
    const bool flip = (J.flags & JOB_FLIP) != 0;
    const bool edge = flip ? X.bounds[2 * xo0] == 0 : X.bounds[2 * xl] + X.bounds[2 * xl + 1] == X.in_w;
    if (flip) {
        if (edge) resample_h_body<true, true>(J, X, in_h, xo, y0, ok, lane);
        else resample_h_body<true, false>(J, X, in_h, xo, y0, ok, lane);
    } else {
        if (edge) resample_h_body<false, true>(J, X, in_h, xo, y0, ok, lane);
        else resample_h_body<false, false>(J, X, in_h, xo, y0, ok, lane);
    }
}

// ---- horizontal pass, ROWS form ----
// With lanes along the output columns every tap is a gather at a ~6-byte stride, through the address path (gather form)
// or across the LDS banks (the staged form of this round's first rebuild: ~3 passes per byte read).  Here the lanes of a
// wave are 64 consecutive source ROWS and the wave walks over output columns: a tap's position and weight are then
// WAVE-UNIFORM, and the lanes read the same byte column of 64 different rows -- with a row pitch of 4 x an odd number
// of bytes those are 64 different banks, so a ds_read_u8 takes its 2.3 LDS cycles and no more (tools/int_rate.hip).  Per
// tap and channel: one byte read (offset in the instruction) and ONE v_mad_i32_i24 with the weight as the scalar operand.
// A block of 4 waves stages the span its `cols` columns touch (coalesced 16-byte global loads, every source byte crosses
// the address path once) and the columns' weights (transposed to [column][tap]); each wave takes cols / 4 columns, and
// the results go back through LDS (transposed there) so that they leave as 16-byte row segments.  Flipped images stage
// the mirrored span and take a column's weights in reverse order.  One launch serves every filter width (the tap loop is
// wave-uniform).  Measured (rocprofv3, 32 / 12 KITTI frames 1242x375, us): 640 columns (13 taps) 38 against 46.5 for the
// staged form; 320 (25 taps) 20.6 against 28.2 (gather); 160 (49 taps) 20 against 34 (staged); 80 (95 taps) 22 against 31
// (taps along the lanes).  Its bound is the vector ALU and the LDS together (per column and wave 39 multiply-adds, 13
// v_readlane of 8 cycles each and ~25 instructions of bookkeeping against 43 LDS instructions in that first form; see
// hw_taps below for what round 4 made of the tap loop).
constexpr int HW_ROWS = 64;
constexpr int HW_ROWT = 1;                  // 64-row tiles per block, one after the other (2 / 3 / 6 tiles measured no faster: below)
constexpr int HW_LDS_BUDGET = 32 * 1024;    // preferred LDS per block (5-6 blocks per CU).  Re-swept on the round-4 kernel (tools/r4_rows_lds.sh,
                                            // stage time per batch): 24 KB 151 us, 32 KB 132-134 (32 / 16 / 8 / 4 columns per block at the four
                                            // scales), 40 KB 134-137 (32 / 32 / 16 / 4), 45-56 KB 133-135 (64 / 32 / 16 / 8), 64 KB 153
constexpr int HW_LDS_MAX = 64 * 1024;       // a 4-column block may take this much; beyond it the job keeps the gather form
typedef volatile uint8_t __attribute__((address_space(3))) *lds_v_u8;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));
// a * w + c with a, w in 24 bits (|weight| < 2^22, byte < 2^8): ONE instruction, the weight in a scalar register.  Written
// as asm because the compiler forms v_mad_i32_i24 from __mul24 + add in a few places only.
static __device__ __forceinline__ int mad24s(int a, int w, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(w), "v"(c));
    return d;
}
// sixteen consecutive weights of a column's row of the column-major table, through the scalar cache (any dword alignment:
// tools/smem_test.hip).  The compiler does not see the load: the wait is part of the statement.
static __device__ __forceinline__ i32x16_t sload16(const int *p)
{
    // a uniform pointer into the CONSTANT address space: the compiler selects s_load_dwordx16 itself and accounts for it in its
    // own s_waitcnt bookkeeping, so the load may stay in flight across the LDS reads that follow (an inline-asm s_load with its
    // own s_waitcnt lgkmcnt(0), round 4's first form, could not)
    typedef const i32x16_t __attribute__((address_space(4), aligned(4))) *cptr;
    return *(cptr)(unsigned long long)p;
}

// Round 4 (late): TWO neighbouring columns per pass, weights in scalar registers.  The windows of neighbouring output columns
// overlap almost entirely (1242 -> 640: 13 taps, 1.94 pixels apart), so a wave that evaluates them together over the UNION of
// the two windows reads each source byte once for both: per column 24 byte reads instead of 39 at 13 taps.  The weights come
// as sixteen consecutive entries of the column's row of the column-major table (mdx_resample_plan_cols: zero-padded on both
// sides, so the second column's row is simply read at an offset and taps outside a window multiply by 0) with ONE scalar load
// per column and sixteen union pixels -- no LDS weight table, no v_readlane (8.5 cycles each, a quarter of the tap loop's
// vector-ALU time), no weights in the set-up.  Flipped images: the union runs over the mirrored windows with the reversed rows
// (table direction 1).  Exact integer sums: the order of the taps is free.
template <bool PAIR>
static __device__ __forceinline__ void hw_taps(lds_v_u8 q, const int *rf, const int *rs, int U, int (&sf)[3], int (&ss)[3])
{
    for (int c0 = 0; c0 < U; c0 += 16, q += 48) {
        const i32x16_t wf = sload16(rf + c0);
        i32x16_t ws = wf;
        if (PAIR) ws = sload16(rs + c0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (c0 + 4 * g >= U) break;                                        // wave-uniform
            int b[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) b[e] = (int)q[12 * g + e];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    sf[c] = mad24s(b[3 * t + c], wf[4 * g + t], sf[c]);
                    if (PAIR) ss[c] = mad24s(b[3 * t + c], ws[4 * g + t], ss[c]);
                }
            }
        }
    }
}

template <bool FLIP>
static __device__ __forceinline__ void resample_h_rows(const PackedJob &J, const PlanX &X, int in_h, uint8_t *lds)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = X.lg_cols, cols = 1 << lg, pitch = X.pitch;
    const int xo0 = blockIdx.x << lg;
    const int xl = min(xo0 + cols, (int)X.out_w) - 1;
    const int last = X.in_w - 1;
    // first / last source pixel the block's columns tap (bounds are monotonic in the column)
    const int lo_min = X.bounds[2 * xo0], hi_max = X.bounds[2 * xl] + X.bounds[2 * xl + 1] - 1;
    const int px_lo = FLIP ? last - hi_max : lo_min, px_hi = FLIP ? last - lo_min : hi_max;
    const int a0 = (3 * px_lo) & ~15;
    const int nch = min((3 * px_hi + 3 - a0 + 15) >> 4, (pitch - 12) >> 4);
    // readable bytes of the image: the LAST row ends at 3 * in_w, not at in_stride (a bottom-right crop view of a larger image has
    // nothing behind it; the chunk that straddles the end is repaired by the shift below)
    const unsigned total = (unsigned)(in_h - 1) * (unsigned)J.in_stride + 3u * (unsigned)X.in_w;
    // Every global load of the set-up is issued before the first LDS write that depends on one.  The first form staged chunk by
    // chunk -- `if (inside) 16-byte load else byte-wise tail`, then the LDS write, per pass of a 4 x (1..4)-pass loop nest -- and
    // the compiler put s_waitcnt vmcnt(0) behind each load: 4 to 16 memory round trips in a row per block
    // (tools/isa_loadwaits.py; SQ_WAIT_ANY was 51 % of the wave-cycles with neither the vector ALU nor the memory busy).  Now: the
    // wave's column bounds first, then per pass over the chunk columns FOUR 16-byte loads (the thread's four rows) at addresses
    // clamped into the image -- the one chunk that would cross the image's last byte is repaired afterwards under a branch that
    // is hardly ever taken.
    const int opitch = cols + 4;                                               // (cols + 4) / 4 is odd: the lanes' result bytes fall in 64 banks
    uint8_t *s_out = lds + HW_ROWS * pitch;                                    // [3][HW_ROWS][opitch]
    const int per = cols >> 2;
    const int xw0 = xo0 + wave * per;
    const int bcol = min(xw0 + min(lane, per - 1), (int)X.out_w - 1);          // lane j: the bounds of the wave's column j
    const int bx = X.bounds[2 * bcol], bn = X.bounds[2 * bcol + 1];
    const int nper = min(per, (int)X.out_w - xw0);                             // wave-uniform (may be <= 0)
    const int krow = X.kc_row, klead = X.kc_lead;
    const int *ktab = X.kc + (FLIP ? (size_t)X.out_w * krow : 0) + klead;      // row of column x: ktab + x * krow
    const size_t plane = (size_t)in_h * X.out_w;
    const bool wide = (cols & 15) == 0 && (X.out_w & 15) == 0 && (((size_t)J.inter) & 15) == 0;
    // A block walks HW_ROWT tiles of 64 rows one after the other (same columns: bounds, span and table rows are set up once).
    // Measured twice, before and after the set-up's loads were batched (image preparation per batch, graph replay): 1 / 2 / 3 / 6
    // tiles per block 141.9 / 143.7 / 143.5 / 148.4 us -- fewer, longer blocks buy nothing, so a block takes ONE tile; the loop
    // stays.  (No barrier is needed between tiles: the staging of tile k + 1 writes the span rows, which every
    // wave finished reading before the barrier in front of tile k's stores; its tap loop writes the result tile behind the
    // barrier that follows the staging, i.e. after every wave's stores of tile k.)
    // Phase by phase (development builds without the staging loads / with at most n taps; LABNOTES.md, round 4), per batch:
    // block skeleton + stores 11 us, + staging 21 us (162 MB of spans through the L2: the source is re-read 3.2 x), + taps 44 us,
    // all three 67 us: staging and taps overlap by 9 us only.  Loading tile k + 1's first two chunk columns into registers BEFORE
    // tile k's tap loop and writing them to the span rows behind it (2 / 3 tiles per block) measured 146.1 / 141.4 us against
    // 142.7 us: the 40 registers it holds across the tap loop cost the two resident blocks per CU that would have overlapped
    // anyway (110 registers against 39).  Not kept.
    // the four 16-byte loads of chunk column `ch` (the thread's four rows of tile yb) / their way into the span rows
    auto stage_load = [&](int yb, int ch, u32x4_t (&v)[HW_ROWS / 16], unsigned (&off)[HW_ROWS / 16]) {
#pragma unroll
        for (int k = 0; k < HW_ROWS / 16; ++k) {
            const int y = min(yb + (tid >> 4) + 16 * k, in_h - 1);
            off[k] = (unsigned)y * (unsigned)J.in_stride + (unsigned)a0 + 16u * (unsigned)ch;
            __builtin_memcpy(&v[k], J.src + min(off[k], total - 16u), 16);              // one unaligned global_load_dwordx4
        }
    };
    auto stage_write = [&](int ch, u32x4_t (&v)[HW_ROWS / 16], const unsigned (&off)[HW_ROWS / 16]) {
#pragma unroll
        for (int k = 0; k < HW_ROWS / 16; ++k) {
            if (off[k] + 16u > total) {
                // the image's last bytes: the load was moved back to end at the last byte (nothing past it is read), the wanted
                // bytes sit `sh` bytes further up in the 16 loaded ones: a 128-bit right shift, zeros coming in (never used)
                const unsigned sh = min(off[k] + 16u - total, 16u) * 8u;                // 8 .. 128 bits
                const unsigned long long lo = (unsigned long long)v[k].x | ((unsigned long long)v[k].y << 32);
                const unsigned long long hi = (unsigned long long)v[k].z | ((unsigned long long)v[k].w << 32);
                const unsigned long long nlo = sh >= 128u ? 0ull : (sh >= 64u ? hi >> (sh - 64u) : (lo >> sh) | (hi << (64u - sh)));
                const unsigned long long nhi = sh >= 64u ? 0ull : hi >> sh;
                v[k].x = (unsigned)nlo; v[k].y = (unsigned)(nlo >> 32); v[k].z = (unsigned)nhi; v[k].w = (unsigned)(nhi >> 32);
            }
            unsigned *d = reinterpret_cast<unsigned *>(lds + ((tid >> 4) + 16 * k) * pitch + 16 * ch);  // 4-byte aligned only: the pitch is 4 x odd
            d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
        }
    };
    // 16 lanes per row, 16 rows per sweep (no division by run-time values anywhere in this kernel: an emulated integer
    // division is ~40 vector instructions -- the first version spent more of them on its index arithmetic than on the taps)
    auto stage_tile = [&](int yb, int ch_first) {
        if (total >= 16u) {
            for (int ch = ch_first; ch < nch; ch += 16) {
                u32x4_t v[HW_ROWS / 16];
                unsigned off[HW_ROWS / 16];
                stage_load(yb, ch, v, off);
                stage_write(ch, v, off);
            }
        } else if (ch_first < 16) {
            for (int r = tid >> 4; r < HW_ROWS; r += 16) {                     // an image of fewer than 16 bytes
                const unsigned row_off = (unsigned)min(yb + r, in_h - 1) * (unsigned)J.in_stride + (unsigned)a0;
                for (int ch = tid & 15; ch < nch; ch += 16) {
                    const unsigned off = row_off + 16u * (unsigned)ch;
                    for (int e = 0; e < 16; ++e) lds[r * pitch + 16 * ch + e] = off + e < total ? J.src[off + e] : 0;
                }
            }
        }
    };
#pragma unroll 1
    for (int rt = 0; rt < HW_ROWT; ++rt) {
    const int yb = ((int)blockIdx.y * HW_ROWT + rt) * HW_ROWS;
    if (yb >= in_h) break;                                                     // block-uniform
    stage_tile(yb, tid & 15);
    __syncthreads();
    for (int j = 0; j < nper; j += 2) {
        const bool pair = j + 1 < nper;
        const int ja = j, jb = pair ? j + 1 : j;
        const int xa = __builtin_amdgcn_readlane(bx, ja), na = __builtin_amdgcn_readlane(bn, ja);
        const int xb = __builtin_amdgcn_readlane(bx, jb), nb = __builtin_amdgcn_readlane(bn, jb);
        // window starts in the (mirrored, when flipped) source row.  Unflipped: column a starts first (bounds are monotonic);
        // flipped: column b does.  `first` starts at p0, `second` d pixels later
        const int pa = FLIP ? last - xa - (na - 1) : xa, pb = FLIP ? last - xb - (nb - 1) : xb;
        const int p0 = FLIP ? pb : pa, d = min(max(FLIP ? pa - pb : pb - pa, 0), klead);   // (0 <= d <= lead by the table's construction;
                                                                                           //  clamped: a wrong table must not read outside it)
        const int nf = FLIP ? nb : na, ns = FLIP ? na : nb;
        const int cf = xw0 + (FLIP ? jb : ja), cs = xw0 + (FLIP ? ja : jb);
        lds_v_u8 q = (lds_v_u8)(lds + lane * pitch + (3 * p0 - a0));
        int sf[3] = {1 << (RS_BITS - 1), 1 << (RS_BITS - 1), 1 << (RS_BITS - 1)};
        int ss[3] = {1 << (RS_BITS - 1), 1 << (RS_BITS - 1), 1 << (RS_BITS - 1)};
        const int *rf = ktab + (size_t)cf * krow;
        if (pair) hw_taps<true>(q, rf, ktab + (size_t)cs * krow - d, max(nf, d + ns), sf, ss);
        else hw_taps<false>(q, rf, rf, nf, sf, ss);
        const int xcf = cf - xo0, xcs = cs - xo0;
#pragma unroll
        for (int c = 0; c < 3; ++c) s_out[(c * HW_ROWS + lane) * opitch + xcf] = clip8(sf[c]);
        if (pair)
#pragma unroll
            for (int c = 0; c < 3; ++c) s_out[(c * HW_ROWS + lane) * opitch + xcs] = clip8(ss[c]);
    }
    __syncthreads();
    if (wide) {                                                                // 16-byte row segments
        const int lgs = lg - 4;
        for (int i = tid; i < (3 * HW_ROWS) << lgs; i += 256) {
            const int row = i >> lgs, sg = i & ((1 << lgs) - 1);               // row = plane * HW_ROWS + r
            const int pl = row >> 6, r = row & 63;
            if (yb + r < in_h && xo0 + 16 * sg < X.out_w) {
                const unsigned *sp = reinterpret_cast<const unsigned *>(s_out + row * opitch + 16 * sg);
                u32x4_t v = {sp[0], sp[1], sp[2], sp[3]};
                *reinterpret_cast<u32x4_t *>(J.inter + pl * plane + (size_t)(yb + r) * X.out_w + xo0 + 16 * sg) = v;
            }
        }
    } else if (J.flags & JOB_VEC4) {                                           // dwords (out_w % 4 == 0, 4-byte aligned planes)
        const int lgs = lg - 2;
        for (int i = tid; i < (3 * HW_ROWS) << lgs; i += 256) {
            const int row = i >> lgs, sg = i & ((1 << lgs) - 1);
            const int pl = row >> 6, r = row & 63;
            if (yb + r < in_h && xo0 + 4 * sg < X.out_w)
                *reinterpret_cast<unsigned *>(J.inter + pl * plane + (size_t)(yb + r) * X.out_w + xo0 + 4 * sg) =
                    *reinterpret_cast<const unsigned *>(s_out + row * opitch + 4 * sg);
        }
    } else {
        for (int i = tid; i < (3 * HW_ROWS) << lg; i += 256) {
            const int row = i >> lg, xc = i & (cols - 1);
            const int pl = row >> 6, r = row & 63;
            if (yb + r < in_h && xo0 + xc < X.out_w)
                J.inter[pl * plane + (size_t)(yb + r) * X.out_w + xo0 + xc] = s_out[row * opitch + xc];
        }
    }
    }   // rt
}

__global__ __launch_bounds__(256) void resample_h_rows_kernel(ResampleLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_span[];
    const PackedJob &J = L.j[blockIdx.z];
    const PlanX &X = L.x[J.px];
    if (X.lg_cols == HW_GATHER) return;
    const int in_h = L.y[J.py].in_h;
    if ((int)(blockIdx.x << X.lg_cols) >= X.out_w || (int)blockIdx.y * HW_ROWS * HW_ROWT >= in_h) return;      // block-uniform
    if (J.flags & JOB_FLIP) resample_h_rows<true>(J, X, in_h, s_span);
    else resample_h_rows<false>(J, X, in_h, s_span);
}

// ---- vertical pass ----
// planar inter [3][in_h][out_w] -> planar [3][out_h][out_w] uint8 and / or float32 (= u8 / 255).
// Large outputs (JOB_VEC8): a thread owns eight consecutive bytes of an output row (two dword loads per tap); the threads of
// a block cover whole rows, 256 / (out_w / 8) of them (a 640-byte row takes 80 threads: three rows per block, 94 % of the
// lanes busy -- the first form gave a row a whole block and left 38 % of the lanes idle in a kernel bound by the vector
// ALU).  Per tap and byte: one v_mul_i32_i24 with the byte selected in the instruction (SDWA) and half a three-operand add,
// instead of an extraction and a multiply-add.  Small outputs and any other shape: a thread per byte (the 24x80 scale has
// 5760 bytes per image and 95 taps each: it needs the threads, not the width).
static __device__ __forceinline__ int mul_byte0(unsigned v, int w)
{
    int d;
    asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(d) : "v"(v), "v"(w));
    return d;
}
static __device__ __forceinline__ int mul_byte1(unsigned v, int w)
{
    int d;
    asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(d) : "v"(v), "v"(w));
    return d;
}
static __device__ __forceinline__ int mul_byte2(unsigned v, int w)
{
    int d;
    asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(d) : "v"(v), "v"(w));
    return d;
}
static __device__ __forceinline__ int mul_byte3(unsigned v, int w)
{
    int d;
    asm("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(d) : "v"(v), "v"(w));
    return d;
}
static __device__ __forceinline__ int add3(int a, int b, int c)
{
    int d;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void resample_v_kernel(ResampleLaunch L)
{
    const PackedJob &J = L.j[blockIdx.z / 3];
    const int c = blockIdx.z % 3;
    const PlanX &X = L.x[J.px];
    const PlanY &Y = L.y[J.py];
    const int out_w = X.out_w, out_h = Y.out_h, in_h = Y.in_h;
    if (J.flags & JOB_VEC8) {                      // block-uniform
        // Round 4 (late): a thread owns eight bytes of TWO neighbouring output rows.  Their windows overlap almost entirely (375 ->
        // 192 rows: 13 taps, 1.95 rows apart), so the union is walked once -- 15 row loads for two rows instead of 26 -- and a
        // launch has half the blocks: the pass was bound by blocks x (chain of dependent loads in front of a block's first tap),
        // not by its arithmetic or its bytes.  Row a starts first (bounds are monotonic), row b `d` rows later; a tap outside a
        // row's window gets weight 0 (the plan's table is zero beyond a row's taps; beyond the table the index is clamped and
        // the weight selected to 0 -- no conditional load anywhere).
        const int tpr = out_w >> 3;                // threads per row
        int xo, yp;
        if (tpr >= 256) {
            xo = (blockIdx.x * 256 + threadIdx.x) * 8;
            yp = blockIdx.y;
        } else {
            if (blockIdx.x) return;
            const int rpb = 256 / tpr;             // row PAIRS per block (block-uniform)
            const int r = threadIdx.x / tpr;
            xo = (threadIdx.x - r * tpr) * 8;
            yp = blockIdx.y * rpb + r;
            if (r >= rpb) return;
        }
        const int ya = 2 * yp;
        if (xo >= out_w || ya >= out_h) return;
        const bool two = ya + 1 < out_h;
        const int yb = two ? ya + 1 : ya;
        const int ymin = Y.bounds[2 * ya], na = Y.bounds[2 * ya + 1];
        const int yminb = Y.bounds[2 * yb], nb = Y.bounds[2 * yb + 1];
        const int d = yminb - ymin;
        const int U = max(na, d + nb), ks = Y.ksize;
        const int *ka = Y.kk + ya, *kb = Y.kk + yb; // [ksize][out_h]
        const uint8_t *p = J.inter + ((size_t)c * in_h + ymin) * out_w + xo;
        const size_t o = ((size_t)c * out_h + ya) * out_w + xo;
        int s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 1 << (RS_BITS - 1);
        const size_t ps = (size_t)out_w, ksz = (size_t)out_h;
        const int last_row = in_h - 1 - ymin;      // union rows beyond the image (the pad of an odd union) are read at the last row, weight 0
        for (int t = 0; t < U; t += 2) {
            const u32x2_t va = *reinterpret_cast<const u32x2_t *>(p + (size_t)min(t, last_row) * ps);
            const u32x2_t vb = *reinterpret_cast<const u32x2_t *>(p + (size_t)min(t + 1, last_row) * ps);
            const int t1 = t + 1, u0 = t - d, u1 = t + 1 - d;
            const int wa0r = ka[(size_t)min(t, ks - 1) * ksz], wa1r = ka[(size_t)min(t1, ks - 1) * ksz];
            const int wb0r = kb[(size_t)min(max(u0, 0), ks - 1) * ksz], wb1r = kb[(size_t)min(max(u1, 0), ks - 1) * ksz];
            const int wa0 = t < na ? wa0r : 0, wa1 = t1 < na ? wa1r : 0;
            const int wb0 = (u0 >= 0 && u0 < nb) ? wb0r : 0, wb1 = (u1 >= 0 && u1 < nb) ? wb1r : 0;
            s[0] = add3(s[0], mul_byte0(va.x, wa0), mul_byte0(vb.x, wa1)); q[0] = add3(q[0], mul_byte0(va.x, wb0), mul_byte0(vb.x, wb1));
            s[1] = add3(s[1], mul_byte1(va.x, wa0), mul_byte1(vb.x, wa1)); q[1] = add3(q[1], mul_byte1(va.x, wb0), mul_byte1(vb.x, wb1));
            s[2] = add3(s[2], mul_byte2(va.x, wa0), mul_byte2(vb.x, wa1)); q[2] = add3(q[2], mul_byte2(va.x, wb0), mul_byte2(vb.x, wb1));
            s[3] = add3(s[3], mul_byte3(va.x, wa0), mul_byte3(vb.x, wa1)); q[3] = add3(q[3], mul_byte3(va.x, wb0), mul_byte3(vb.x, wb1));
            s[4] = add3(s[4], mul_byte0(va.y, wa0), mul_byte0(vb.y, wa1)); q[4] = add3(q[4], mul_byte0(va.y, wb0), mul_byte0(vb.y, wb1));
            s[5] = add3(s[5], mul_byte1(va.y, wa0), mul_byte1(vb.y, wa1)); q[5] = add3(q[5], mul_byte1(va.y, wb0), mul_byte1(vb.y, wb1));
            s[6] = add3(s[6], mul_byte2(va.y, wa0), mul_byte2(vb.y, wa1)); q[6] = add3(q[6], mul_byte2(va.y, wb0), mul_byte2(vb.y, wb1));
            s[7] = add3(s[7], mul_byte3(va.y, wa0), mul_byte3(vb.y, wa1)); q[7] = add3(q[7], mul_byte3(va.y, wb0), mul_byte3(vb.y, wb1));
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half && !two) break;
            unsigned v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = clip8(half ? q[e] : s[e]);
            const size_t oo = o + (half ? ps : 0);
            if (J.dst_u8) {
                u32x2_t pk = {v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24), v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24)};
                *reinterpret_cast<u32x2_t *>(J.dst_u8 + oo) = pk;
            }
            if (J.dst_f32) {
                *reinterpret_cast<float4 *>(J.dst_f32 + oo) = make_float4(unit_from_byte(v[0]), unit_from_byte(v[1]), unit_from_byte(v[2]), unit_from_byte(v[3]));
                *reinterpret_cast<float4 *>(J.dst_f32 + oo + 4) = make_float4(unit_from_byte(v[4]), unit_from_byte(v[5]), unit_from_byte(v[6]), unit_from_byte(v[7]));
            }
        }
    } else {
        const int xo = blockIdx.x * 256 + threadIdx.x;
        const int yo = blockIdx.y;
        if (xo >= out_w || yo >= out_h) return;
        const int ymin = Y.bounds[2 * yo], n = Y.bounds[2 * yo + 1];
        const int *k = Y.kk + yo;
        const uint8_t *p = J.inter + ((size_t)c * in_h + ymin) * out_w + xo;
        const size_t o = ((size_t)c * out_h + yo) * out_w + xo;
        int s = 1 << (RS_BITS - 1);
        int t = 0;
        for (; t + 4 <= n; t += 4, p += 4 * (size_t)out_w, k += 4 * (size_t)out_h) {      // four loads in flight: these are the
            int b[4], w[4];                                                              // small outputs with the long filters
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                b[u] = (int)p[(size_t)u * out_w];
                w[u] = k[(size_t)u * out_h];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += __mul24(b[u], w[u]);
        }
        for (; t < n; ++t, p += out_w, k += out_h) s += __mul24((int)p[0], k[0]);
        const uint8_t v = clip8(s);
        if (J.dst_u8) J.dst_u8[o] = v;
        if (J.dst_f32) J.dst_f32[o] = unit_from_byte(v);
    }
}

// ---- per-pixel colour maps (Pillow's Convert.c / Blend.c arithmetic) ----
struct RGB8 {
    int r, g, b;
};

static __device__ __forceinline__ int rgb_to_L(const RGB8 &p)
{
    return (int)(((unsigned)p.r * 19595u + (unsigned)p.g * 38470u + (unsigned)p.b * 7471u + 0x8000u) >> 16);
}

// ImagingBlend(a, b, alpha) for one byte: float32 a + alpha*(b - a); `inside` = alpha in [0,1]
// (inside: (UINT8)t with t in [0, 255]; outside: t <= 0 ? 0 : t >= 255 ? 255 : (UINT8)t -- both are the truncated t clamped)
static __device__ __forceinline__ int blend1(int a, int b, float alpha)
{
    const float t = (float)a + alpha * (float)(b - a);
    return min(max((int)t, 0), 255);
}

// Convert.c's rgb2hsv / hsv2rgb, value for value, with the divisions written out: a / b as q0 = a * rcp, one residual and
// one correction (the IEEE quotient when rcp is close enough -- these operands are small integers, the divisors 6 and 255
// constants); fmod(x, 1) for x > 0 as x - floor(x) (exact); round() for x >= 0 as trunc + (fraction >= 0.5).  The compiler's
// own division sequences and the library fmod made the hue step ~250 vector instructions per pixel, most of them double
// precision, in kernels the vector ALU bounds.  tests/test_gpu_imgproc.py checks all 2^24 RGB and all 2^24 HSV triples
// against Pillow's convert().
static __device__ __forceinline__ float div_small(float a, float b, float rb)       // rb = v_rcp_f32(b)
{
    const float q0 = a * rb;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, rb, q0);
}
static __device__ __forceinline__ double div_const(double x, double d, double rd)   // rd = fl(1 / d)
{
    const double q0 = x * rd;
    const double r = __builtin_fma(-d, q0, x);
    return __builtin_fma(r, rd, q0);
}
static __device__ __forceinline__ int round_pos(double x)                            // (int)round(x), x >= 0
{
    const double t = __builtin_trunc(x);
    return (int)t + ((x - t) >= 0.5 ? 1 : 0);
}

static __device__ __forceinline__ RGB8 rgb2hsv(const RGB8 &p)
{
    // no per-pixel branches anywhere in the chain: a divergent branch costs its scalar bookkeeping in every wave (the first
    // form spent more scalar than vector instructions); grey pixels (minc == maxc) run the arithmetic on a stand-in range
    const int maxc = max(p.r, max(p.g, p.b)), minc = min(p.r, min(p.g, p.b));
    const bool grey_px = minc == maxc;
    const float cr = grey_px ? 1.0f : (float)(maxc - minc), fm = grey_px ? 1.0f : (float)maxc;
    const float rcr = __builtin_amdgcn_rcpf(cr);
    const float s = div_small(cr, fm, __builtin_amdgcn_rcpf(fm));
    // Convert.c: rc, gc, bc = (maxc - channel) / cr;  h = bc - gc | 2 + rc - bc | 4 + gc - rc  (the first in float, the
    // others in double, rounded to float once: base + x1 - x2 in double, rounded once, is all three)
    const bool rmax = p.r == maxc, gmax = p.g == maxc;
    const int n1 = maxc - (rmax ? p.b : (gmax ? p.r : p.g)), n2 = maxc - (rmax ? p.g : (gmax ? p.b : p.r));
    const double base = rmax ? 0.0 : (gmax ? 2.0 : 4.0);
    const float x1 = div_small((float)n1, cr, rcr), x2 = div_small((float)n2, cr, rcr);
    float h = (float)(base + (double)x1 - (double)x2);
    const double d = div_const((double)h, 6.0, 1.0 / 6.0) + 1.0;                    // in [5/6, 2)
    h = (float)(d - __builtin_floor(d));                                             // fmod(d, 1.0)
    RGB8 o;
    o.r = grey_px ? 0 : min(max((int)((double)h * 255.0), 0), 255);
    o.g = grey_px ? 0 : min(max((int)((double)s * 255.0), 0), 255);
    o.b = maxc;
    return o;
}

static __device__ __forceinline__ RGB8 hsv2rgb(const RGB8 &q)      // q.r = H, q.g = S, q.b = V
{
    const int v = q.b;                                              // (S == 0: fs = 0 and p = qq = t = v below, as Convert.c returns)
    const double hf = div_const((double)(float)q.r * 6.0, 255.0, 1.0 / 255.0);
    const int i = (int)__builtin_floor(hf);
    const double f = (double)(float)(hf - (double)(float)i);
    const double fs = (double)(float)div_const((double)(float)q.g, 255.0, 1.0 / 255.0);
    const double vf = (double)(float)v;
    const int p = min(round_pos(vf * (1.0 - fs)), 255);
    const int qq = min(round_pos(vf * (1.0 - fs * f)), 255);
    const int t = min(round_pos(vf * (1.0 - fs * (1.0 - f))), 255);
    const int k = i >= 6 ? i - 6 : i;                               // i % 6, i in 0..6
    // k:  0 (v,t,p)  1 (q,v,p)  2 (p,v,t)  3 (p,q,v)  4 (t,p,v)  5 (v,p,q)
    RGB8 o;
    o.r = (k == 0 || k == 5) ? v : (k == 1 ? qq : (k == 4 ? t : p));
    o.g = (k == 1 || k == 2) ? v : (k == 0 ? t : (k == 3 ? qq : p));
    o.b = (k == 3 || k == 4) ? v : (k == 2 ? t : (k == 5 ? qq : p));
    return o;
}

// the adjustments order[first..last) applied to one pixel; `grey` = Contrast's degenerate level
template <int NP>
static __device__ __forceinline__ void jitter_ops_n(RGB8 (&p)[NP], const mdx_jitter_job &J, int first, int last, int grey)
{
    for (int i = first; i < last; ++i) {              // the op is the job's (wave-uniform): one scalar branch per op and NP pixels
        const int op = J.order[i];
        if (op == 0) {
#pragma unroll
            for (int e = 0; e < NP; ++e)
                p[e] = RGB8{blend1(0, p[e].r, J.brightness), blend1(0, p[e].g, J.brightness), blend1(0, p[e].b, J.brightness)};
        } else if (op == 1) {
#pragma unroll
            for (int e = 0; e < NP; ++e)
                p[e] = RGB8{blend1(grey, p[e].r, J.contrast), blend1(grey, p[e].g, J.contrast), blend1(grey, p[e].b, J.contrast)};
        } else if (op == 2) {
#pragma unroll
            for (int e = 0; e < NP; ++e) {
                const int L = rgb_to_L(p[e]);
                p[e] = RGB8{blend1(L, p[e].r, J.saturation), blend1(L, p[e].g, J.saturation), blend1(L, p[e].b, J.saturation)};
            }
        } else if (op == 3) {
#pragma unroll
            for (int e = 0; e < NP; ++e) {
                RGB8 q = rgb2hsv(p[e]);
                q.r = (q.r + J.hue_shift) & 255;
                p[e] = hsv2rgb(q);
            }
        }
    }
}
static __device__ __forceinline__ RGB8 jitter_ops(RGB8 p, const mdx_jitter_job &J, int first, int last, int grey)
{
    RGB8 a[1] = {p};
    jitter_ops_n<1>(a, J, first, last, grey);
    return a[0];
}

static __device__ __forceinline__ int contrast_slot(const mdx_jitter_job &J)
{
    for (int i = 0; i < 4; ++i)
        if (J.order[i] == 1) return i;
    return 4;
}

// Pass 1: sum of L over the image as it stands when Contrast is reached: exact integer partial sums, one per block.
__global__ __launch_bounds__(256) void jitter_mean_kernel(JitterJobs jobs)
{
    const mdx_jitter_job &J = jobs.j[blockIdx.y];
    const int n = J.h * J.w, slot = contrast_slot(J);
    unsigned sum = 0;
    if (slot < 4) {
        // four pixels per thread (one dword per plane) when the planes allow it, the rest pixel by pixel
        const bool vec = (n & 3) == 0 && (((size_t)J.src) & 3) == 0;
        const int body = vec ? n : 0;
        for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < body; i += gridDim.x * 1024) {
            const unsigned r4 = *reinterpret_cast<const unsigned *>(J.src + i), g4 = *reinterpret_cast<const unsigned *>(J.src + (size_t)n + i),
                           b4 = *reinterpret_cast<const unsigned *>(J.src + 2 * (size_t)n + i);
            RGB8 p[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = RGB8{(int)((r4 >> (8 * e)) & 255u), (int)((g4 >> (8 * e)) & 255u), (int)((b4 >> (8 * e)) & 255u)};
            jitter_ops_n<4>(p, J, 0, slot, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += (unsigned)rgb_to_L(p[e]);
        }
        for (int i = body + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            RGB8 p = {J.src[i], J.src[(size_t)n + i], J.src[2 * (size_t)n + i]};
            p = jitter_ops(p, J, 0, slot, 0);
            sum += (unsigned)rgb_to_L(p);
        }
    }
    __shared__ unsigned s_part[4];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0)          // one word per block, every launch: nothing to clear, no atomics
        J.lsum[blockIdx.x] = (unsigned long long)s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// Pass 2: the whole chain, planar uint8 in -> planar uint8 and / or float32 out.
__global__ __launch_bounds__(256) void jitter_apply_kernel(JitterJobs jobs)
{
    const mdx_jitter_job &J = jobs.j[blockIdx.y];
    const int n = J.h * J.w, slot = contrast_slot(J);
    // ImageEnhance.Contrast: int(sum / count + 0.5) in double
    int grey = 0;
    if (slot < 4) {                                                 // block-uniform
        // the partial sums of pass 1 (<= MDX_JITTER_PARTIALS = 128 of them): one per thread, reduced in the block
        __shared__ unsigned long long s_tot[2];
        unsigned long long part = threadIdx.x < gridDim.x ? J.lsum[threadIdx.x] : 0ull;
        if (threadIdx.x < 128) {
            for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
            if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = part;
        }
        __syncthreads();
        const unsigned long long total = s_tot[0] + s_tot[1];
        grey = (int)((double)total / (double)n + 0.5);
    }
    if (slot == 4 && J.order[0] == 4 && J.order[1] == 4 && J.order[2] == 4 && J.order[3] == 4) {
        // the empty chain (a sample without a draw in a batch where another has one): ToTensor alone, four bytes at a time
        const unsigned total = 3u * (unsigned)n;
        const bool vec = (((size_t)J.src) & 3) == 0 && (!J.dst_u8 || (((size_t)J.dst_u8) & 3) == 0) &&
                         (!J.dst_f32 || (((size_t)J.dst_f32) & 15) == 0);
        const unsigned body = vec ? total & ~3u : 0u;
        for (unsigned i = (blockIdx.x * 256 + threadIdx.x) * 4; i < body; i += gridDim.x * 1024) {
            const unsigned v = *reinterpret_cast<const unsigned *>(J.src + i);
            if (J.dst_u8) *reinterpret_cast<unsigned *>(J.dst_u8 + i) = v;
            if (J.dst_f32)
                *reinterpret_cast<float4 *>(J.dst_f32 + i) = make_float4(unit_from_byte(v & 255u), unit_from_byte((v >> 8) & 255u),
                                                                         unit_from_byte((v >> 16) & 255u), unit_from_byte(v >> 24));
        }
        for (unsigned i = body + blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            const uint8_t v = J.src[i];
            if (J.dst_u8) J.dst_u8[i] = v;
            if (J.dst_f32) J.dst_f32[i] = unit_from_byte(v);
        }
        return;
    }
    const bool vec4 = (n & 3) == 0 && (((size_t)J.src) & 3) == 0 && (!J.dst_u8 || (((size_t)J.dst_u8) & 3) == 0) &&
                      (!J.dst_f32 || (((size_t)J.dst_f32) & 15) == 0);
    const int body4 = vec4 ? n : 0;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < body4; i += gridDim.x * 1024) {      // four pixels per thread
        const unsigned r4 = *reinterpret_cast<const unsigned *>(J.src + i), g4 = *reinterpret_cast<const unsigned *>(J.src + (size_t)n + i),
                       b4 = *reinterpret_cast<const unsigned *>(J.src + 2 * (size_t)n + i);
        unsigned ro = 0, go = 0, bo = 0;
        RGB8 p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = RGB8{(int)((r4 >> (8 * e)) & 255u), (int)((g4 >> (8 * e)) & 255u), (int)((b4 >> (8 * e)) & 255u)};
        jitter_ops_n<4>(p, J, 0, 4, grey);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ro |= (unsigned)p[e].r << (8 * e); go |= (unsigned)p[e].g << (8 * e); bo |= (unsigned)p[e].b << (8 * e);
        }
        if (J.dst_u8) {
            *reinterpret_cast<unsigned *>(J.dst_u8 + i) = ro;
            *reinterpret_cast<unsigned *>(J.dst_u8 + (size_t)n + i) = go;
            *reinterpret_cast<unsigned *>(J.dst_u8 + 2 * (size_t)n + i) = bo;
        }
        if (J.dst_f32) {
            *reinterpret_cast<float4 *>(J.dst_f32 + i) = make_float4(unit_from_byte(ro & 255u), unit_from_byte((ro >> 8) & 255u),
                                                                     unit_from_byte((ro >> 16) & 255u), unit_from_byte(ro >> 24));
            *reinterpret_cast<float4 *>(J.dst_f32 + (size_t)n + i) = make_float4(unit_from_byte(go & 255u), unit_from_byte((go >> 8) & 255u),
                                                                                 unit_from_byte((go >> 16) & 255u), unit_from_byte(go >> 24));
            *reinterpret_cast<float4 *>(J.dst_f32 + 2 * (size_t)n + i) = make_float4(unit_from_byte(bo & 255u), unit_from_byte((bo >> 8) & 255u),
                                                                                     unit_from_byte((bo >> 16) & 255u), unit_from_byte(bo >> 24));
        }
    }
    for (int i = body4 + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        RGB8 p = {J.src[i], J.src[(size_t)n + i], J.src[2 * (size_t)n + i]};
        p = jitter_ops(p, J, 0, 4, grey);
        if (J.dst_u8) {
            J.dst_u8[i] = (uint8_t)p.r; J.dst_u8[(size_t)n + i] = (uint8_t)p.g; J.dst_u8[2 * (size_t)n + i] = (uint8_t)p.b;
        }
        if (J.dst_f32) {
            J.dst_f32[i] = unit_from_byte((unsigned)p.r);
            J.dst_f32[(size_t)n + i] = unit_from_byte((unsigned)p.g);
            J.dst_f32[2 * (size_t)n + i] = unit_from_byte((unsigned)p.b);
        }
    }
}

// mode 0: RGB -> HSV, 1: HSV -> RGB, 2: RGB -> L (one plane out); planar [3][n] in
__global__ __launch_bounds__(256) void color_convert_kernel(const uint8_t *src, uint8_t *dst, size_t n, int mode)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const RGB8 p = {src[i], src[n + i], src[2 * n + i]};
    if (mode == 2) {
        dst[i] = (uint8_t)rgb_to_L(p);
        return;
    }
    const RGB8 o = mode == 0 ? rgb2hsv(p) : hsv2rgb(p);
    dst[i] = (uint8_t)o.r; dst[n + i] = (uint8_t)o.g; dst[2 * n + i] = (uint8_t)o.b;
}

// ToTensor (kitti_mono.py:283,351): float32(byte) / 255 as the IEEE division ToTensor's CPU kernel performs (a bare
// multiplication by the rounded reciprocal -- what a generic tensor / scalar kernel does -- is off by one ulp for 126 of
// the 256 byte values; unit_from_byte adds the one correction step that makes it exact).  16 bytes per thread: one 16-byte load, four 16-byte stores; the tail byte by byte.
__global__ __launch_bounds__(256) void to_tensor_kernel(const uint8_t *__restrict__ src, float *__restrict__ dst, size_t n)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i >= n) return;
    if (i + 16 <= n && ((uintptr_t)(src + i) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 o;
            o.x = unit_from_byte(w[k] & 255u);
            o.y = unit_from_byte((w[k] >> 8) & 255u);
            o.z = unit_from_byte((w[k] >> 16) & 255u);
            o.w = unit_from_byte(w[k] >> 24);
            *reinterpret_cast<float4 *>(dst + i + 4 * k) = o;
        }
    } else {
        for (size_t k = i; k < n && k < i + 16; ++k) dst[k] = unit_from_byte(src[k]);
    }
}

// ---- host side ----
static double sinc_filter(double x)
{
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static double lanczos_filter(double x)
{
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

static int ksize_of(int in_size, int out_size)
{
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(3.0 * filterscale) * 2 + 1;
}

// does the plan (a device or host table we cannot read here) fit the job's sizes?  shapes only
static int validate_resample(const mdx_resample_job &J)
{
    if (!J.src || !J.xbounds || !J.xkk || !J.ybounds || !J.ykk || !J.inter) return MDX_ERR_NULL_POINTER;
    if (!J.dst_u8 && !J.dst_f32) return MDX_ERR_NULL_POINTER;
    if (J.in_h <= 0 || J.in_w < 2 || J.out_h <= 0 || J.out_w <= 0) return MDX_ERR_BAD_SHAPE;   // dword taps: >= 2 columns
    if (J.in_h > 16384 || J.in_w > 16384 || J.out_h > 16384 || J.out_w > 16384) return MDX_ERR_BAD_SHAPE;
    if (J.in_stride < 3 * J.in_w) return MDX_ERR_BAD_SHAPE;
    if (J.xksize != ksize_of(J.in_w, J.out_w) || J.yksize != ksize_of(J.in_h, J.out_h)) return MDX_ERR_BAD_SHAPE;
    if (J.dst_f32 && !aligned(J.dst_f32, 4)) return MDX_ERR_MISALIGNED;
    if (!aligned(J.xbounds, 4) || !aligned(J.xkk, 4) || !aligned(J.ybounds, 4) || !aligned(J.ykk, 4))
        return MDX_ERR_MISALIGNED;
    if (J.xkc) {                       // column-major table (mdx_resample_plan_cols): shapes only, as for the plan itself
        if (!aligned(J.xkc, 4)) return MDX_ERR_MISALIGNED;
        if (J.xkc_lead < 0 || J.xkc_lead > 65535 || J.xkc_row > 65535 ||
            J.xkc_row < J.xkc_lead + ((J.xksize + J.xkc_lead + 15) / 16) * 16 + 16) return MDX_ERR_BAD_SHAPE;
    }
    return MDX_OK;
}

}  // namespace mdx

using namespace mdx;

MDX_EXPORT int mdx_resample_ksize(int in_size, int out_size)
{
    if (in_size <= 0 || out_size <= 0) return MDX_ERR_BAD_SHAPE;
    return ksize_of(in_size, out_size);
}

// Resample.c precompute_coeffs (box = the whole axis) + normalize_coeffs_8bpc, into HOST arrays; the weights are stored
// tap-major (kk[tap][out]), the transpose of Pillow's table: a wave of consecutive outputs reads consecutive weights.
MDX_EXPORT int mdx_resample_plan(int in_size, int out_size, int *bounds, int *kk)
{
    if (!bounds || !kk) return MDX_ERR_NULL_POINTER;
    if (in_size <= 0 || out_size <= 0) return MDX_ERR_BAD_SHAPE;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    double *w = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            w[x] = lanczos_filter((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int *k = kk + xx;                          // kk[tap][xx]
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            k[(size_t)x * out_size] = v < 0 ? (int)(-0.5 + v * (1 << RS_BITS)) : (int)(0.5 + v * (1 << RS_BITS));
        }
        for (int x = xmax; x < ksize; ++x) k[(size_t)x * out_size] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
    return MDX_OK;
}

// The plan's weights column-major, zero-padded, both directions (include/mdx.h): what the rows form of the horizontal pass reads
// with scalar loads.  HOST arrays.
MDX_EXPORT int mdx_resample_plan_cols(int in_size, int out_size, int *lead_out, int *row_out, int *table)
{
    if (!lead_out || !row_out) return MDX_ERR_NULL_POINTER;
    if (in_size <= 0 || out_size <= 0) return MDX_ERR_BAD_SHAPE;
    const int ksize = ksize_of(in_size, out_size);
    std::vector<int> bounds((size_t)out_size * 2), kk((size_t)ksize * out_size);
    if (int rc = mdx_resample_plan(in_size, out_size, bounds.data(), kk.data())) return rc;
    int lead = 0;
    for (int x = 0; x + 1 < out_size; ++x) {
        const int d0 = bounds[2 * x + 2] - bounds[2 * x];                                              // window starts
        const int d1 = (bounds[2 * x + 2] + bounds[2 * x + 3]) - (bounds[2 * x] + bounds[2 * x + 1]);  // window ends (flipped: starts)
        if (d0 < 0 || d1 < 0) return MDX_ERR_UNSUPPORTED;                                              // (Pillow's bounds are monotonic)
        lead = d0 > lead ? d0 : lead;
        lead = d1 > lead ? d1 : lead;
    }
    const int row = lead + ((ksize + lead + 15) / 16) * 16 + 16;
    *lead_out = lead;
    *row_out = row;
    if (!table) return MDX_OK;
    memset(table, 0, sizeof(int) * 2 * (size_t)out_size * row);
    for (int x = 0; x < out_size; ++x) {
        const int n = bounds[2 * x + 1];
        int *f = table + (size_t)x * row + lead, *r = table + ((size_t)out_size + x) * row + lead;
        for (int t = 0; t < n; ++t) {
            f[t] = kk[(size_t)t * out_size + x];
            r[t] = kk[(size_t)(n - 1 - t) * out_size + x];
        }
    }
    return MDX_OK;
}

MDX_EXPORT int mdx_resample_lanczos_u8(const mdx_resample_job *jobs, int njobs, void *stream)
{
    if (!jobs) return MDX_ERR_NULL_POINTER;
    if (njobs <= 0) return MDX_ERR_BAD_SHAPE;
    for (int i = 0; i < njobs; ++i)
        if (int rc = validate_resample(jobs[i])) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool no_rows = false;
    const int rows_budget = HW_LDS_BUDGET, rows_cols_max = 64;      // (swept in round 4: the comment at HW_LDS_BUDGET)
    // long filters first: their blocks run longest (95 taps for the 80-column scale of a KITTI frame), dispatched last they
    // are the launch's tail (measured: vertical pass of a batch 62 us in caller order)
    std::vector<int> order(njobs);
    for (int i = 0; i < njobs; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int u, int v) {
        return jobs[u].xksize + jobs[u].yksize > jobs[v].xksize + jobs[v].yksize;
    });
    int first = 0;
    while (first < njobs) {
        ResampleLaunch a;
        memset(&a, 0, sizeof(a));
        int nx = 0, ny = 0, n = 0;
        int rows_lds = 0, rows_gx = 0, n_rows = 0, gather_w = 0, max_in_h = 0;
        int v8_gx = 0, v8_gy = 0, v1_gx = 0, v1_gy = 0;
        for (; first + n < njobs && n < IMG_JOBS; ++n) {
            const mdx_resample_job &S = jobs[order[first + n]];
            int ix = 0, iy = 0;
            while (ix < nx && !(a.x[ix].bounds == S.xbounds && a.x[ix].kk == S.xkk && a.x[ix].kc == S.xkc && a.x[ix].in_w == S.in_w && a.x[ix].out_w == S.out_w)) ++ix;
            while (iy < ny && !(a.y[iy].bounds == S.ybounds && a.y[iy].kk == S.ykk && a.y[iy].in_h == S.in_h && a.y[iy].out_h == S.out_h)) ++iy;
            if (ix == IMG_PLANS || iy == IMG_PLANS) break;          // a 17th plan: this job starts the next launch
            if (ix == nx) {
                PlanX &X = a.x[nx++];
                X.bounds = S.xbounds; X.kk = S.xkk; X.in_w = (uint16_t)S.in_w; X.out_w = (uint16_t)S.out_w; X.ksize = (uint16_t)S.xksize;
                X.kc = S.xkc; X.kc_lead = (uint16_t)S.xkc_lead; X.kc_row = (uint16_t)S.xkc_row;
                // rows form: the widest block (64, 32, ... 4 columns) whose staged span + result tile + weights fit the LDS
                // budget (measured on 1242 -> 640: 32 columns 38 us, 16 columns 45 us); gather form when even 4 do not fit
                X.lg_cols = HW_GATHER;
                const double scale = (double)S.in_w / S.out_w;
                for (int lg = 6; lg >= 2 && !no_rows && S.xkc && X.lg_cols == HW_GATHER; --lg) {
                    const int cw = 1 << lg;
                    if (cw > rows_cols_max) continue;
                    const int span_px = (int)ceil((cw - 1) * scale) + S.xksize + 2;
                    const int pitch = ((3 * span_px + 16 + 15) / 16) * 16 + 12;          // pitch / 4 is odd
                    const int lds = HW_ROWS * pitch + 3 * HW_ROWS * (cw + 4);
                    if (lds <= rows_budget || (lg == 2 && lds <= HW_LDS_MAX)) {
                        X.lg_cols = (uint8_t)lg;
                        X.pitch = (uint16_t)pitch;
                    }
                }
            }
            if (iy == ny) {
                PlanY &Y = a.y[ny++];
                Y.bounds = S.ybounds; Y.kk = S.ykk; Y.in_h = S.in_h; Y.out_h = S.out_h; Y.ksize = S.yksize;
            }
            const PlanX &X = a.x[ix];
            PackedJob &J = a.j[n];
            J.src = S.src; J.inter = S.inter; J.dst_u8 = S.dst_u8; J.dst_f32 = S.dst_f32; J.in_stride = S.in_stride;
            J.px = (uint8_t)ix; J.py = (uint8_t)iy;
            const bool vec4 = S.out_w % 4 == 0 && aligned(S.inter, 4);
            const bool vec8 = S.out_w % 8 == 0 && aligned(S.inter, 8) && (!S.dst_u8 || aligned(S.dst_u8, 8)) &&
                              (!S.dst_f32 || aligned(S.dst_f32, 16)) && S.out_w * S.out_h >= 16384;
            J.flags = (uint8_t)((S.flip ? JOB_FLIP : 0) | (vec4 ? JOB_VEC4 : 0) | (vec8 ? JOB_VEC8 : 0));
            if (X.lg_cols != HW_GATHER) {
                const int cw = 1 << X.lg_cols;
                const int lds = HW_ROWS * X.pitch + 3 * HW_ROWS * (cw + 4);
                rows_lds = lds > rows_lds ? lds : rows_lds;
                const int gx = (S.out_w + cw - 1) / cw;
                rows_gx = gx > rows_gx ? gx : rows_gx;
                ++n_rows;
            } else {
                gather_w = S.out_w > gather_w ? S.out_w : gather_w;
            }
            max_in_h = S.in_h > max_in_h ? S.in_h : max_in_h;
            if (vec8) {
                const int tpr = S.out_w / 8, pairs = (S.out_h + 1) / 2;     // a thread owns eight bytes of TWO rows
                const int gx = tpr >= 256 ? (tpr + 255) / 256 : 1;
                const int gy = tpr >= 256 ? pairs : (pairs + 256 / tpr - 1) / (256 / tpr);
                v8_gx = gx > v8_gx ? gx : v8_gx;
                v8_gy = gy > v8_gy ? gy : v8_gy;
            } else {
                const int gx = (S.out_w + 255) / 256;
                v1_gx = gx > v1_gx ? gx : v1_gx;
                v1_gy = S.out_h > v1_gy ? S.out_h : v1_gy;
            }
        }
        if (n_rows)
            hipLaunchKernelGGL(resample_h_rows_kernel, dim3(rows_gx, (max_in_h + HW_ROWS * HW_ROWT - 1) / (HW_ROWS * HW_ROWT), n), dim3(256),
                               (size_t)rows_lds, st, a);
        if (n_rows < n)
            hipLaunchKernelGGL(resample_h_kernel, dim3((gather_w + 63) / 64, (max_in_h + 4 * HR - 1) / (4 * HR), n), dim3(256), 0, st, a);
        hipLaunchKernelGGL(resample_v_kernel, dim3(v8_gx > v1_gx ? v8_gx : v1_gx, v8_gy > v1_gy ? v8_gy : v1_gy, 3 * n), dim3(256), 0,
                           st, a);
        first += n;
    }
    return check_launch();
}

MDX_EXPORT int mdx_color_jitter_u8(const mdx_jitter_job *jobs, int njobs, void *stream)
{
    if (!jobs) return MDX_ERR_NULL_POINTER;
    if (njobs <= 0) return MDX_ERR_BAD_SHAPE;
    for (int i = 0; i < njobs; ++i) {
        const mdx_jitter_job &J = jobs[i];
        if (!J.src || !J.lsum || (!J.dst_u8 && !J.dst_f32)) return MDX_ERR_NULL_POINTER;
        if (J.h <= 0 || J.w <= 0 || (long long)J.h * J.w > (1ll << 24)) return MDX_ERR_BAD_SHAPE;   // L sums fit uint32 per thread
        if (!aligned(J.lsum, 8) || (J.dst_f32 && !aligned(J.dst_f32, 4))) return MDX_ERR_MISALIGNED;
        unsigned seen = 0;
        for (int k = 0; k < 4; ++k) {
            if (J.order[k] < 0 || J.order[k] > 4) return MDX_ERR_UNSUPPORTED;     // 4 = skip this slot
            if (J.order[k] < 4 && (seen >> J.order[k] & 1u)) return MDX_ERR_UNSUPPORTED;
            if (J.order[k] < 4) seen |= 1u << J.order[k];
        }
    }
    hipStream_t st = (hipStream_t)stream;
    for (int first = 0; first < njobs; first += MDX_JITTER_JOBS) {
        const int n = njobs - first < MDX_JITTER_JOBS ? njobs - first : MDX_JITTER_JOBS;
        JitterJobs a;
        memset(&a, 0, sizeof(a));
        int max_px = 0;
        for (int i = 0; i < n; ++i) {
            a.j[i] = jobs[first + i];
            max_px = a.j[i].h * a.j[i].w > max_px ? a.j[i].h * a.j[i].w : max_px;
        }
        int blocks = (max_px + 255) / 256;
        if (blocks > MDX_JITTER_PARTIALS) blocks = MDX_JITTER_PARTIALS;    // x <= 32 jobs: fills the chip; the loops stride
        hipLaunchKernelGGL(jitter_mean_kernel, dim3(blocks, n), dim3(256), 0, st, a);
        hipLaunchKernelGGL(jitter_apply_kernel, dim3(blocks, n), dim3(256), 0, st, a);
    }
    return check_launch();
}

MDX_EXPORT int mdx_color_convert_u8(int mode, const uint8_t *src, uint8_t *dst, size_t npix, void *stream)
{
    if (!src || !dst) return MDX_ERR_NULL_POINTER;
    if (mode < 0 || mode > 2) return MDX_ERR_UNSUPPORTED;
    if (npix == 0 || npix > ((size_t)1 << 31)) return MDX_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(color_convert_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, npix, mode);
    return check_launch();
}

MDX_EXPORT int mdx_to_tensor_u8(const uint8_t *src, float *dst, size_t n, void *stream)
{
    if (!src || !dst) return MDX_ERR_NULL_POINTER;
    if (n == 0 || n > ((size_t)1 << 36)) return MDX_ERR_BAD_SHAPE;
    if (!aligned(dst, 16)) return MDX_ERR_MISALIGNED;
    hipLaunchKernelGGL(to_tensor_kernel, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return check_launch();
}
