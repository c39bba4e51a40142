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
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "mdx_common.hpp"

namespace mdx {

constexpr int RS_BITS = 32 - 8 - 2;          // Resample.c PRECISION_BITS

struct ResampleJob : mdx_resample_job {
    int vec4;                 // the vertical pass may use dword accesses (alignment and row length checked on the host)
    int h_taps;               // horizontal pass: 0 = lanes along the columns, C > 0 = resample_h_taps_kernel<C>
    int h_window;             // lanes along the columns: 0 = one gather per tap and row, KT > 0 = staged through LDS,
                              // KT taps evaluated (13, 16, 26, 49, 53)
};
struct ResampleJobs {
    ResampleJob j[MDX_IMG_JOBS];
};
struct JitterJobs {
    mdx_jitter_job j[MDX_IMG_JOBS];
};

static __device__ __forceinline__ uint8_t clip8(int v)
{
    v >>= RS_BITS;
    return (uint8_t)min(max(v, 0), 255);
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

// Horizontal pass: interleaved RGB [in_h][in_w][3] -> planar uint8 inter [3][in_h][out_w].
// A wave owns 64 consecutive output columns and HR rows (weights, bounds and address arithmetic are shared by the
// rows; HR x 2 independent loads are in flight); a pixel's three bytes arrive as ONE dword: bytes [3px, 3px+3], or for
// the last pixel of a row [3px-1, 3px+2] shifted down (never a byte past the row).
constexpr int HR = 4;
// EDGE: some lane of the wave taps the last pixel of a row (only the last columns do) -- the general form with the
// shifted load; otherwise every tap is the plain dword at 3*px.  Row bases are wave-uniform (scalar registers), the
// per-lane part of an address is one 32-bit offset.
template <bool FLIP, bool EDGE>
static __device__ __forceinline__ void resample_h_body(const ResampleJob &J, int xo, int y0, bool ok, int lane)
{
    const int xmin = J.xbounds[2 * xo], n = J.xbounds[2 * xo + 1];
    const int *k = J.xkk + xo;                     // [ksize][out_w]: a wave reads 256 consecutive bytes per tap
    const uint8_t *row[HR];
    int s[HR][3];
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        row[r] = J.src + (size_t)min(y0 + r, J.in_h - 1) * J.in_stride;      // y0 is wave-uniform
        s[r][0] = s[r][1] = s[r][2] = 1 << (RS_BITS - 1);
    }
    const int last = J.in_w - 1;
    unsigned off = 3u * (unsigned)(FLIP ? last - xmin : xmin);
#pragma unroll 2
    for (int t = 0; t < n; ++t) {
        const int c = k[(size_t)t * J.out_w];
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
            s[r][0] += __mul24((int)(v & 255u), c);             // |weight| < 2^22: v_mad_i32_i24, full rate
            s[r][1] += __mul24((int)((v >> 8) & 255u), c);
            s[r][2] += __mul24((int)((v >> 16) & 255u), c);
        }
        off = FLIP ? off - 3u : off + 3u;
    }
    const size_t plane = (size_t)J.in_h * J.out_w;
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        const bool row_ok = ok && y0 + r < J.in_h;                     // wave-uniform apart from the column tail
        const size_t o = (size_t)min(y0 + r, J.in_h - 1) * J.out_w + xo;
        store_bytes_packed(J.inter + o, clip8(s[r][0]), J.vec4 != 0, row_ok, lane);
        store_bytes_packed(J.inter + plane + o, clip8(s[r][1]), J.vec4 != 0, row_ok, lane);
        store_bytes_packed(J.inter + 2 * plane + o, clip8(s[r][2]), J.vec4 != 0, row_ok, lane);
    }
}

// Taps come straight from global memory (L1 / L2 serve the overlap of neighbouring columns and tiles).  An LDS-staged
// form (row segments copied with 16-byte loads, taps as unaligned ds_read_b32) was measured at 166 us against 60 us for
// this one on 32 KITTI frames: lanes 6 bytes apart reading unaligned dwords serialise on the LDS banks.
__global__ __launch_bounds__(256) void resample_h_kernel(ResampleJobs jobs)
{
    const ResampleJob &J = jobs.j[blockIdx.z];
    if (J.h_taps) return;                                            // this job runs in resample_h_taps_kernel
    const int lane = threadIdx.x & 63;
    const int xo0 = blockIdx.x * 64;
    // the wave index as a scalar: taken from threadIdx.x alone the compiler treats the row addresses as lane-varying
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int y0 = (blockIdx.y * 4 + wave) * HR;
    if (xo0 >= J.out_w || y0 >= J.in_h) return;                      // wave-uniform: the packed stores need whole waves
    const bool ok = xo0 + lane < J.out_w;
    const int xo = min(xo0 + lane, J.out_w - 1);
    // does any lane tap the last source pixel?  (bounds are monotonic: the last column reaches furthest)
    const int xl = min(xo0 + 63, J.out_w - 1);
    const bool edge = J.flip ? J.xbounds[2 * xo0] == 0 : J.xbounds[2 * xl] + J.xbounds[2 * xl + 1] == J.in_w;
    if (J.h_window) return;                                          // this job runs in resample_h_staged_kernel
    if (J.flip) {
        if (edge) resample_h_body<true, true>(J, xo, y0, ok, lane);
        else resample_h_body<true, false>(J, xo, y0, ok, lane);
    } else {
        if (edge) resample_h_body<false, true>(J, xo, y0, ok, lane);
        else resample_h_body<false, false>(J, xo, y0, ok, lane);
    }
}

// STAGED form of the horizontal pass (round 3).  The gather form above issues one 4-byte load per tap and row -- 13 load
// instructions per output for the 640-column scale -- and the address path, 64 lanes each fetching its own unaligned
// dword at a ~6-byte stride, is what bounds it (~18 CU-cycles per instruction, profiles/r02_imgproc_kernel_pmc.txt; a form
// that fetched each lane's 39-byte window with three unaligned 16-byte loads measured the same 57 us: the path is paced by
// the lanes' scattered addresses, not by the instruction count).  Here a block of 64 output columns x 16 rows first copies
// the source bytes its columns touch -- ONE contiguous span per row, ~420 bytes for the 640-column scale -- into LDS with
// coalesced 16-byte loads (each source byte crosses the address path once), then every lane reads its taps from LDS as
// BYTES (ds_read_u8 with the tap's offset as the instruction's immediate: no address arithmetic, no extraction, no
// alignment issue -- the first LDS version of round 2 read unaligned dwords and serialised on the banks) and multiplies.
// A span that runs past the image's last byte is completed with zeros (those taps' weights are zero): nothing is read
// beyond the image's rows.
// Flipped images stage the mirrored span and take the weights in reverse.  KT = taps evaluated (>= the filter's ksize).
// Measured (32 frames 1242x375 -> 640 columns): 47.7 us against 58 us for the gather form.  What bounds it now is the LDS:
// 64 lanes 6 bytes apart touch ~93 different dwords per read instruction -- three per bank -- so every ds_read_u8 takes
// ~three passes (156 of them per wave).  Reading dwords instead and picking the bytes apart costs two VALU instructions
// per byte and tap again (the compiler's own choice for the non-volatile form: 52 us); a form that fetched each lane's
// window with three unaligned 16-byte GLOBAL loads measured 57 us.  All three sit within 20 % of each other: the pass is
// a 6-byte-stride gather whichever way it is served.
constexpr int HS_ROWS = 16;                 // rows per block = 4 waves x HR
typedef const volatile uint8_t __attribute__((address_space(3))) *lds_cv_u8;
// a * b + c with a, b in 24 bits (|weight| < 2^22, byte < 2^8): ONE instruction.  Written as asm because the compiler
// forms v_mad_i32_i24 from __mul24 + add in a few places only and leaves a v_mul_i32_i24 + v_add pair elsewhere.
static __device__ __forceinline__ int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int KT, bool FLIP>
static __device__ __forceinline__ void resample_h_staged(const ResampleJob &J, uint8_t *lds, int pitch_max)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xo0 = blockIdx.x * 64, yb = blockIdx.y * HS_ROWS;
    const int xl = min(xo0 + 63, J.out_w - 1);
    const int last = J.in_w - 1;
    // first / last source pixel the block's columns tap (bounds are monotonic in the column)
    const int lo_min = J.xbounds[2 * xo0], hi_max = J.xbounds[2 * xl] + J.xbounds[2 * xl + 1] - 1;
    const int px_lo = FLIP ? last - hi_max : lo_min, px_hi = FLIP ? last - lo_min : hi_max;
    const int a0 = (3 * px_lo) & ~15;
    const int nch = min((3 * px_hi + 3 - a0 + 15) >> 4, pitch_max >> 4);
    const int pitch = nch << 4;
    const unsigned total = (unsigned)J.in_h * (unsigned)J.in_stride;          // bytes of the image's rows (its slot may be larger)
    for (int i = tid; i < HS_ROWS * nch; i += 256) {
        const int r = i / nch, ch = i - r * nch;
        const int y = min(yb + r, J.in_h - 1);
        const unsigned off = (unsigned)y * (unsigned)J.in_stride + (unsigned)(a0 + 16 * ch);
        u32x4_t v;
        if (off + 16u <= total) {
            __builtin_memcpy(&v, J.src + off, 16);                             // one unaligned global_load_dwordx4
        } else {                                                               // the image's last bytes: nothing past them is read
            uint8_t b[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) b[e] = off + e < total ? J.src[off + e] : 0;
            __builtin_memcpy(&v, b, 16);
        }
        *reinterpret_cast<u32x4_t *>(lds + r * pitch + 16 * ch) = v;
    }
    const bool ok = xo0 + lane < J.out_w;
    const int xo = min(xo0 + lane, J.out_w - 1);
    const int xmin = J.xbounds[2 * xo], n = J.xbounds[2 * xo + 1];
    const int *k = J.xkk + xo;
    const int px0 = FLIP ? last - xmin - (n - 1) : xmin;
    int w[KT];
#pragma unroll
    for (int p = 0; p < KT; ++p) {
        const int t = FLIP ? n - 1 - p : p;
        w[p] = p < n ? k[(size_t)t * J.out_w] : 0;
    }
    __syncthreads();
    const size_t plane = (size_t)J.in_h * J.out_w;
    const uint8_t *col = lds + (3 * px0 - a0);
    // results leave through LDS too when the block's 64 columns are whole and 16-byte aligned in the output rows: the
    // 3 x 16 row segments of 64 bytes go out as 192 16-byte stores (3 store instructions per block instead of 48 that
    // each write 16 scattered dwords)
    uint8_t *s_out = lds + HS_ROWS * pitch_max;                                // [3][HS_ROWS][64]
    const bool wide_out = J.vec4 && (J.out_w & 15) == 0 && xo0 + 64 <= J.out_w && (((size_t)J.inter) & 15) == 0 &&
                          (plane & 15) == 0;
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        const int row = wave * HR + r;
        // volatile: one ds_read_u8 per tap byte, as written.  Left alone the compiler merges the bytes into unaligned
        // 16-byte LDS reads and picks them apart with v_mul_i32_i24_sdwa + v_add (two VALU instructions per byte and tap,
        // ~900 per wave: the kernel then runs at the VALU's pace, 52 us -- no faster than the gather form); a byte read
        // delivers the operand ready for ONE v_mad_i32_i24
        const lds_cv_u8 q = (lds_cv_u8)(col + row * pitch);
        int s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0;
        int bytes[3 * KT];
#pragma unroll
        for (int e = 0; e < 3 * KT; ++e) bytes[e] = (int)q[e];
#pragma unroll
        for (int p = 0; p < KT; ++p) {
            s0 = mad24(bytes[3 * p], w[p], s0);
            s1 = mad24(bytes[3 * p + 1], w[p], s1);
            s2 = mad24(bytes[3 * p + 2], w[p], s2);
        }
        if (wide_out) {
            s_out[(0 * HS_ROWS + row) * 64 + lane] = clip8(s0);
            s_out[(1 * HS_ROWS + row) * 64 + lane] = clip8(s1);
            s_out[(2 * HS_ROWS + row) * 64 + lane] = clip8(s2);
        } else {
            const int y = min(yb + row, J.in_h - 1);
            const bool row_ok = ok && yb + row < J.in_h;
            const size_t o = (size_t)y * J.out_w + xo;
            store_bytes_packed(J.inter + o, clip8(s0), J.vec4 != 0, row_ok, lane);
            store_bytes_packed(J.inter + plane + o, clip8(s1), J.vec4 != 0, row_ok, lane);
            store_bytes_packed(J.inter + 2 * plane + o, clip8(s2), J.vec4 != 0, row_ok, lane);
        }
    }
    if (wide_out) {
        __syncthreads();
        if (tid < 3 * HS_ROWS * 4) {
            const int seg = tid >> 2, quarter = tid & 3;                       // seg = plane * HS_ROWS + row
            const int pl = seg / HS_ROWS, row = seg - pl * HS_ROWS;
            if (yb + row < J.in_h)
                *reinterpret_cast<u32x4_t *>(J.inter + pl * plane + (size_t)(yb + row) * J.out_w + xo0 + 16 * quarter) =
                    *reinterpret_cast<const u32x4_t *>(s_out + seg * 64 + 16 * quarter);
        }
    }
}

template <int KT>
__global__ __launch_bounds__(256) void resample_h_staged_kernel(ResampleJobs jobs, int pitch_max)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_span[];
    const ResampleJob &J = jobs.j[blockIdx.z];
    if (J.h_window != KT) return;
    if ((int)blockIdx.x * 64 >= J.out_w || (int)blockIdx.y * HS_ROWS >= J.in_h) return;      // block-uniform
    if (J.flip) resample_h_staged<KT, true>(J, s_span, pitch_max);
    else resample_h_staged<KT, false>(J, s_span, pitch_max);
}

// Horizontal pass for strong reductions (65..128 taps: the 80-column scale of a 1242-wide frame has 95).  With lanes along the
// output columns a tap load gathers at a stride of 3*in/out bytes -- 46 bytes for the 80-column scale: 64 cache lines
// per instruction, 6 % of each used.  Here a WAVE owns one output column: each 16-lane DPP row takes one source row,
// each lane a contiguous chunk of C taps (16*C >= taps; the row's taps are 3*n contiguous bytes), and four row-local
// DPP adds finish the sum -- 3 reduction instructions per output instead of 18 for a whole-wave reduction.  The
// column's weights stay in registers while the wave walks down the rows.
constexpr int HT_ROWS = 96;            // rows per block (4 at a time)
template <int C>
__global__ __launch_bounds__(256) void resample_h_taps_kernel(ResampleJobs jobs)
{
    const ResampleJob &J = jobs.j[blockIdx.z];
    if (J.h_taps != C) return;
    const int lane = threadIdx.x & 63, xo = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int y0 = blockIdx.y * HT_ROWS;
    if (xo >= J.out_w || y0 >= J.in_h) return;                         // wave-uniform
    const int sub = lane & 15, rr = lane >> 4;
    const int xmin = J.xbounds[2 * xo], n = J.xbounds[2 * xo + 1];
    const int last = J.in_w - 1;
    int w[C], off[C], sh[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const int t = sub * C + j, tt = min(t, n - 1);                  // past the last tap: its address, weight 0
        w[j] = t < n ? J.xkk[(size_t)t * J.out_w + xo] : 0;
        const int px = J.flip ? last - (xmin + tt) : xmin + tt;
        const int edge = px == last ? 1 : 0;
        off[j] = 3 * px - edge;
        sh[j] = 8 * edge;
    }
    const size_t plane = (size_t)J.in_h * J.out_w;
    const int y1 = min(y0 + HT_ROWS, J.in_h);
    for (int yb = y0; yb < y1; yb += 4) {                               // every lane runs every iteration (DPP below)
        const int y = yb + rr;
        const uint8_t *row = J.src + (size_t)min(y, J.in_h - 1) * J.in_stride;
        int s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const unsigned v = load32_unaligned(row + off[j]) >> sh[j];
            s0 += __mul24((int)(v & 255u), w[j]);
            s1 += __mul24((int)((v >> 8) & 255u), w[j]);
            s2 += __mul24((int)((v >> 16) & 255u), w[j]);
        }
        // sum over the 16 lanes of the DPP row: lane 15 of each row ends with the total (row_shr, zero fill)
#define MDX_ROW_SHR_ADD(d)                                                     \
        s0 += __builtin_amdgcn_update_dpp(0, s0, 0x110 + d, 0xf, 0xf, true);   \
        s1 += __builtin_amdgcn_update_dpp(0, s1, 0x110 + d, 0xf, 0xf, true);   \
        s2 += __builtin_amdgcn_update_dpp(0, s2, 0x110 + d, 0xf, 0xf, true);
        MDX_ROW_SHR_ADD(8) MDX_ROW_SHR_ADD(4) MDX_ROW_SHR_ADD(2) MDX_ROW_SHR_ADD(1)
#undef MDX_ROW_SHR_ADD
        if (sub == 15 && y < J.in_h) {
            const size_t o = (size_t)y * J.out_w + xo;
            J.inter[o] = clip8(s0 + (1 << (RS_BITS - 1)));
            J.inter[plane + o] = clip8(s1 + (1 << (RS_BITS - 1)));
            J.inter[2 * plane + o] = clip8(s2 + (1 << (RS_BITS - 1)));
        }
    }
}

// Vertical pass: planar inter [3][in_h][out_w] -> planar [3][out_h][out_w] uint8 and / or float32 (= u8 / 255).
// VEC = 4: a thread owns four consecutive bytes of an output row (one dword load per tap; out_w % 4 == 0 and 4-byte
// aligned planes -- every size of the KITTI pyramids); VEC = 1: any shape.  The taps' weights are wave-uniform.
template <int VEC>
__global__ __launch_bounds__(256) void resample_v_kernel(ResampleJobs jobs)
{
    const ResampleJob &J = jobs.j[blockIdx.z / 3];
    const int c = blockIdx.z % 3;
    const int xo = (blockIdx.x * 256 + threadIdx.x) * VEC;
    const int yo = blockIdx.y;
    if (xo >= J.out_w || yo >= J.out_h) return;
    if (VEC == 4 && !J.vec4) return;               // this job runs in the VEC = 1 launch
    if (VEC == 1 && J.vec4) return;
    const int ymin = J.ybounds[2 * yo], n = J.ybounds[2 * yo + 1];
    const int *k = J.ykk + yo;                     // [ksize][out_h]
    const uint8_t *p = J.inter + ((size_t)c * J.in_h + ymin) * J.out_w + xo;
    const size_t o = ((size_t)c * J.out_h + yo) * J.out_w + xo;
    if (VEC == 4) {
        int s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
        const size_t ps = (size_t)J.out_w, ks = (size_t)J.out_h;
        int t = 0;
        for (; t + 4 <= n; t += 4, p += 4 * ps, k += 4 * ks) {         // four loads in flight
            unsigned v[4];
            int w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *(const unsigned *)(p + u * ps);
                w[u] = k[u * ks];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 += __mul24((int)(v[u] & 255u), w[u]); s1 += __mul24((int)((v[u] >> 8) & 255u), w[u]);
                s2 += __mul24((int)((v[u] >> 16) & 255u), w[u]); s3 += __mul24((int)(v[u] >> 24), w[u]);
            }
        }
        for (; t < n; ++t, p += ps, k += ks) {
            const unsigned v = *(const unsigned *)p;
            const int w = k[0];
            s0 += __mul24((int)(v & 255u), w); s1 += __mul24((int)((v >> 8) & 255u), w);
            s2 += __mul24((int)((v >> 16) & 255u), w); s3 += __mul24((int)(v >> 24), w);
        }
        const unsigned v0 = clip8(s0), v1 = clip8(s1), v2 = clip8(s2), v3 = clip8(s3);
        if (J.dst_u8) *(unsigned *)(J.dst_u8 + o) = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
        if (J.dst_f32)
            *(float4 *)(J.dst_f32 + o) = make_float4((float)v0 / 255.0f, (float)v1 / 255.0f, (float)v2 / 255.0f, (float)v3 / 255.0f);
    } else {
        int s = 1 << (RS_BITS - 1);
        for (int t = 0; t < n; ++t, p += J.out_w, k += J.out_h) s += __mul24((int)p[0], k[0]);
        const uint8_t v = clip8(s);
        if (J.dst_u8) J.dst_u8[o] = v;
        if (J.dst_f32) J.dst_f32[o] = (float)v / 255.0f;
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
static __device__ __forceinline__ int blend1(int a, int b, float alpha, bool inside)
{
    const float t = (float)a + alpha * (float)(b - a);
    if (inside) return (int)(uint8_t)t;
    return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}

static __device__ __forceinline__ RGB8 rgb2hsv(const RGB8 &p)
{
    const int maxc = max(p.r, max(p.g, p.b)), minc = min(p.r, min(p.g, p.b));
    RGB8 o = {0, 0, maxc};
    if (minc == maxc) return o;
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - p.r) / cr, gc = (float)(maxc - p.g) / cr, bc = (float)(maxc - p.b) / cr;
    float h;
    if (p.r == maxc) h = bc - gc;
    else if (p.g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    o.r = min(max((int)((double)h * 255.0), 0), 255);
    o.g = min(max((int)((double)s * 255.0), 0), 255);
    return o;
}

static __device__ __forceinline__ RGB8 hsv2rgb(const RGB8 &q)      // q.r = H, q.g = S, q.b = V
{
    const int v = q.b;
    if (q.g == 0) return RGB8{v, v, v};
    const double hf = (double)(float)q.r * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const double f = (double)(float)(hf - (double)(float)i);
    const double fs = (double)(float)((double)(float)q.g / 255.0);
    const double vf = (double)(float)v;
    const int p = min(max((int)round(vf * (1.0 - fs)), 0), 255);
    const int qq = min(max((int)round(vf * (1.0 - fs * f)), 0), 255);
    const int t = min(max((int)round(vf * (1.0 - fs * (1.0 - f))), 0), 255);
    switch (i % 6) {
    case 0: return RGB8{v, t, p};
    case 1: return RGB8{qq, v, p};
    case 2: return RGB8{p, v, t};
    case 3: return RGB8{p, qq, v};
    case 4: return RGB8{t, p, v};
    default: return RGB8{v, p, qq};
    }
}

// the adjustments order[first..last) applied to one pixel; `grey` = Contrast's degenerate level
static __device__ __forceinline__ RGB8 jitter_ops(RGB8 p, const mdx_jitter_job &J, int first, int last, int grey)
{
    for (int i = first; i < last; ++i) {
        const int op = J.order[i];
        if (op == 0) {
            const bool in = J.brightness >= 0.f && J.brightness <= 1.f;
            p = RGB8{blend1(0, p.r, J.brightness, in), blend1(0, p.g, J.brightness, in), blend1(0, p.b, J.brightness, in)};
        } else if (op == 1) {
            const bool in = J.contrast >= 0.f && J.contrast <= 1.f;
            p = RGB8{blend1(grey, p.r, J.contrast, in), blend1(grey, p.g, J.contrast, in), blend1(grey, p.b, J.contrast, in)};
        } else if (op == 2) {
            const bool in = J.saturation >= 0.f && J.saturation <= 1.f;
            const int L = rgb_to_L(p);
            p = RGB8{blend1(L, p.r, J.saturation, in), blend1(L, p.g, J.saturation, in), blend1(L, p.b, J.saturation, in)};
        } else if (op == 3) {
            RGB8 q = rgb2hsv(p);
            q.r = (q.r + J.hue_shift) & 255;
            p = hsv2rgb(q);
        }
    }
    return p;
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
    if (slot < 4)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            RGB8 p = {J.src[i], J.src[(size_t)n + i], J.src[2 * (size_t)n + i]};
            p = jitter_ops(p, J, 0, slot, 0);
            sum += (unsigned)rgb_to_L(p);
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
    if (slot < 4) {
        unsigned long long total = 0;
        for (int b = 0; b < (int)gridDim.x; ++b) total += J.lsum[b];
        grey = (int)((double)total / (double)n + 0.5);
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        RGB8 p = {J.src[i], J.src[(size_t)n + i], J.src[2 * (size_t)n + i]};
        p = jitter_ops(p, J, 0, 4, grey);
        if (J.dst_u8) {
            J.dst_u8[i] = (uint8_t)p.r; J.dst_u8[(size_t)n + i] = (uint8_t)p.g; J.dst_u8[2 * (size_t)n + i] = (uint8_t)p.b;
        }
        if (J.dst_f32) {
            J.dst_f32[i] = (float)p.r / 255.0f;
            J.dst_f32[(size_t)n + i] = (float)p.g / 255.0f;
            J.dst_f32[2 * (size_t)n + i] = (float)p.b / 255.0f;
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

// ToTensor (kitti_mono.py:283,351): float32(byte) / 255 with the IEEE divide ToTensor's CPU division performs (a
// multiplication by the rounded reciprocal -- what a generic tensor / scalar kernel does -- is off by one ulp for 126 of
// the 256 byte values).  16 bytes per thread: one 16-byte load, four 16-byte stores; the tail byte by byte.
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
            o.x = (float)(w[k] & 255u) / 255.0f;
            o.y = (float)((w[k] >> 8) & 255u) / 255.0f;
            o.z = (float)((w[k] >> 16) & 255u) / 255.0f;
            o.w = (float)(w[k] >> 24) / 255.0f;
            *reinterpret_cast<float4 *>(dst + i + 4 * k) = o;
        }
    } else {
        for (size_t k = i; k < n && k < i + 16; ++k) dst[k] = (float)src[k] / 255.0f;
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

MDX_EXPORT int mdx_resample_lanczos_u8(const mdx_resample_job *jobs, int njobs, void *stream)
{
    if (!jobs) return MDX_ERR_NULL_POINTER;
    if (njobs <= 0) return MDX_ERR_BAD_SHAPE;
    for (int i = 0; i < njobs; ++i)
        if (int rc = validate_resample(jobs[i])) return rc;
    hipStream_t st = (hipStream_t)stream;
    for (int first = 0; first < njobs; first += MDX_IMG_JOBS) {
        const int n = njobs - first < MDX_IMG_JOBS ? njobs - first : MDX_IMG_JOBS;
        ResampleJobs a;
        memset(&a, 0, sizeof(a));
        int max_in_h = 0, min_in_h = 1 << 30, max_out_w = 0, max_out_h = 0, n4 = 0, n_gather = 0, cols_out_w = 0, taps_out_w = 0;
        unsigned taps_used = 0, win_used = 0;
        int win_pitch[5] = {0, 0, 0, 0, 0};
#ifdef MDX_DEV_SWITCHES      // A/B builds only (build.py MDX_BUILD_DEFINES=-DMDX_DEV_SWITCHES): the shipped library never reads the environment
        const bool force_cols = getenv("MDX_RESAMPLE_COLUMNS") != nullptr;
        const bool force_gather = getenv("MDX_RESAMPLE_GATHER") != nullptr;
        const bool force_staged = getenv("MDX_RESAMPLE_STAGED") != nullptr;
#else
        const bool force_cols = false, force_gather = false, force_staged = false;
#endif
        for (int i = 0; i < n; ++i) {
            static_cast<mdx_resample_job &>(a.j[i]) = jobs[first + i];
            ResampleJob &J = a.j[i];
            J.vec4 = J.out_w % 4 == 0 && aligned(J.inter, 4) && (!J.dst_u8 || aligned(J.dst_u8, 4)) &&
                     (!J.dst_f32 || aligned(J.dst_f32, 16));
            n4 += J.vec4;
            // horizontal form by filter width (measured on 12 KITTI frames, columns / taps form: 25 taps 28 / 45 us,
            // 49 taps 36 / 43 us, 95 taps 82 / 31 us): chunked taps for 65..128 taps, lanes along the columns otherwise
            const int ks = J.xksize;
            J.h_taps = force_cols || ks <= 64 || ks > 128 ? 0 : (ks <= 96 ? 6 : 8);
            // staged form where it measured faster than the gather form (32 / 12 KITTI frames, rocprofv3): 13 taps 47.7 us
            // against 58, 49 taps 33.7 against 36; at 25 taps the gather form wins (28.5 against 30.8) and keeps the job;
            // tiny sources keep the gather form too
            J.h_window = (J.h_taps || force_gather || J.in_w < 64) ? 0
                         : (ks <= 13 ? 13 : (ks <= 16 ? 16 : (ks <= 26 ? (force_staged ? 26 : 0) : (ks <= 49 ? 49 : (ks <= 53 ? 53 : 0)))));
            if (J.h_window) {
                const int slot = J.h_window == 13 ? 0 : (J.h_window == 16 ? 1 : (J.h_window == 26 ? 2 : (J.h_window == 49 ? 3 : 4)));
                const double scale = (double)J.in_w / J.out_w;
                const int span_px = (int)ceil(63.0 * scale) + ks + 2;
                const int pitch = ((3 * span_px + 16 + 15) / 16) * 16;
                if (HS_ROWS * pitch > 60 * 1024) {
                    J.h_window = 0;                      // (cannot happen for <= 53 taps: scale < 9, span < 640 pixels)
                } else {
                    win_used |= 1u << slot;
                    win_pitch[slot] = pitch > win_pitch[slot] ? pitch : win_pitch[slot];
                }
            }
            if (J.h_taps) {
                taps_used |= J.h_taps == 6 ? 4u : 8u;
                taps_out_w = J.out_w > taps_out_w ? J.out_w : taps_out_w;
            } else {
                cols_out_w = J.out_w > cols_out_w ? J.out_w : cols_out_w;
            }
            n_gather += (!J.h_taps && !J.h_window) ? 1 : 0;
            max_in_h = J.in_h > max_in_h ? J.in_h : max_in_h;
            min_in_h = J.in_h < min_in_h ? J.in_h : min_in_h;
            max_out_w = J.out_w > max_out_w ? J.out_w : max_out_w;
            max_out_h = J.out_h > max_out_h ? J.out_h : max_out_h;
        }
        const dim3 cg((cols_out_w + 63) / 64, (max_in_h + 4 * HR - 1) / (4 * HR), n);
        if (n_gather) hipLaunchKernelGGL(resample_h_kernel, cg, dim3(256), 0, st, a);          // gather-form jobs
        // staged-form jobs: one launch per tap count; LDS = 16 rows x the widest span of 64 columns among the jobs
        // (span <= 63 * scale + ksize + 1 pixels; + 16 bytes for the alignment of its start, rounded up to 16)
#define MDX_STAGED_LAUNCH(bit, KTV)                                                                              \
        if (win_used & bit) {                                                                                    \
            const int pitch = win_pitch[KTV == 13 ? 0 : (KTV == 16 ? 1 : (KTV == 26 ? 2 : (KTV == 49 ? 3 : 4)))]; \
            hipLaunchKernelGGL(resample_h_staged_kernel<KTV>, cg, dim3(256), (size_t)HS_ROWS * pitch + 3 * HS_ROWS * 64, st, a, pitch); \
        }
        MDX_STAGED_LAUNCH(1u, 13) MDX_STAGED_LAUNCH(2u, 16) MDX_STAGED_LAUNCH(4u, 26) MDX_STAGED_LAUNCH(8u, 49)
        MDX_STAGED_LAUNCH(16u, 53)
#undef MDX_STAGED_LAUNCH
        const dim3 tg((taps_out_w + 3) / 4, (max_in_h + HT_ROWS - 1) / HT_ROWS, n);
        if (taps_used & 4u) hipLaunchKernelGGL(resample_h_taps_kernel<6>, tg, dim3(256), 0, st, a);
        if (taps_used & 8u) hipLaunchKernelGGL(resample_h_taps_kernel<8>, tg, dim3(256), 0, st, a);
        if (n4)
            hipLaunchKernelGGL(resample_v_kernel<4>, dim3((max_out_w + 1023) / 1024, max_out_h, 3 * n), dim3(256), 0, st, a);
        if (n4 < n)
            hipLaunchKernelGGL(resample_v_kernel<1>, dim3((max_out_w + 255) / 256, max_out_h, 3 * n), dim3(256), 0, st, a);
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
    for (int first = 0; first < njobs; first += MDX_IMG_JOBS) {
        const int n = njobs - first < MDX_IMG_JOBS ? njobs - first : MDX_IMG_JOBS;
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
