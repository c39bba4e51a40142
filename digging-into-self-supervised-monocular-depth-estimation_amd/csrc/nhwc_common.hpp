// nhwc_common.hpp -- shared pieces of the channels-last ([B][H][W][C]) network kernels (norm_nhwc.hip, glue_nhwc.hip).
//
// A channels-last map is a row-major matrix [P pixels][C channels].  Every kernel here gives a thread one 16-byte
// channel vector (4 float32 / 8 bfloat16) of one pixel: a 256-thread block is CVB channel vectors wide and
// PL = 256 / CVB pixels deep, so a wave reads whole 128-byte lines whatever C is (C = 64 float32: 4 pixels per wave).
#pragma once
#include "mdx_common.hpp"
#include <stdint.h>

namespace mdx {
namespace nhwc {

struct bf16 { uint16_t v; };

__device__ __forceinline__ float to_float(float x) { return x; }
__device__ __forceinline__ float to_float(bf16 x) { return __uint_as_float((uint32_t)x.v << 16); }
template <typename T> __device__ __forceinline__ T from_float(float x);
template <> __device__ __forceinline__ float from_float<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_float<bf16>(float x)
{
    // round to nearest even, NaN stays NaN (torch's float -> bfloat16): gfx950's v_cvt_pk_bf16_f32 -- one instruction where the
    // integer sequence (NaN test, rounding add, shift) took six, which made the bfloat16 apply passes VALU-bound
    bf16 r;
    r.v = __builtin_bit_cast(uint16_t, (__bf16)x);
    return r;
}

constexpr int NB = 256;                                   // threads per block

template <typename T> struct VecN { static constexpr int N = 16 / (int)sizeof(T); };
// N elements of T, aligned to their own size (16 bytes for the in-memory type of a kernel, 32 for a float32 copy of 8 bf16)
template <typename T, int N> struct __attribute__((aligned(sizeof(T) * N))) Vec { T v[N]; };

template <typename T, int N> __device__ __forceinline__ Vec<T, N> load_vec(const T *p)
{
    return *reinterpret_cast<const Vec<T, N> *>(p);
}
template <typename T, int N> __device__ __forceinline__ void store_vec(T *p, const Vec<T, N> &v)
{
    *reinterpret_cast<Vec<T, N> *>(p) = v;
}

// how a block of NB threads tiles [pixels][channel vectors]
struct Tile {
    int CV;        // channel vectors of the map (C / N)
    int CVB;       // channel vectors a block is wide (<= NB)
    int PL;        // pixel lanes of a block (NB / CVB; threads beyond PL * CVB idle)
    int ny;        // blocks along the channel axis
};
static inline Tile make_tile(int C, int N)
{
    Tile t;
    t.CV = C / N;
    t.CVB = t.CV < NB ? t.CV : NB;
    t.PL = NB / t.CVB;
    t.ny = (t.CV + t.CVB - 1) / t.CVB;
    return t;
}

// ---- a map as a matrix [M rows][C]: blocks own row ranges, a thread one channel vector and every PL-th row -------------------
struct Rows {
    Tile t;
    int RB;        // rows a block owns (multiple of t.PL)
    int nblk;      // blocks along the rows
};
static inline Rows make_rows(long long M, int C, int N, int max_blocks, int min_iters)
{
    Rows g;
    g.t = make_tile(C, N);
    const long long rows_min = (long long)g.t.PL * min_iters;
    long long nb = (M + rows_min - 1) / rows_min;
    if (nb > max_blocks) nb = max_blocks;
    if (nb < 1) nb = 1;
    long long RB = (M + nb - 1) / nb;
    RB = (RB + g.t.PL - 1) / g.t.PL * g.t.PL;
    g.RB = (int)RB;
    g.nblk = (int)((M + RB - 1) / RB);
    return g;
}
struct Pos { int cv, pl, r0, r1, g; bool active; };
template <int N> __device__ __forceinline__ Pos position(int M, int C, int CVB, int PL, int RB)
{
    Pos p;
    const int t = threadIdx.x;
    const int cvl = t % CVB;
    p.pl = t / CVB;
    p.cv = blockIdx.y * CVB + cvl;
    p.g = blockIdx.z;
    p.r0 = blockIdx.x * RB;
    p.r1 = min(M, p.r0 + RB);
    p.active = p.pl < PL && p.cv * N < C;
    return p;
}

// sum the block's per-thread vectors over the pixel lanes (fixed order) and write one partial per channel
template <int N>
__device__ __forceinline__ void block_partials(const float (&a)[N], const float (&q)[N], float *lds, int CVB, int PL, int C,
                                               float *__restrict__ part_blk /* [2][C] of this (group, block) */)
{
    float *la = lds, *lq = lds + NB * N;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        la[threadIdx.x * N + j] = a[j];
        lq[threadIdx.x * N + j] = q[j];
    }
    __syncthreads();
    const int width = CVB * N;                               // channels this block covers
    for (int e = threadIdx.x; e < width; e += NB) {
        const int c = blockIdx.y * width + e;
        if (c >= C) break;
        float sa = 0.f, sq = 0.f;
        for (int l = 0; l < PL; ++l) {
            sa += la[l * width + e];
            sq += lq[l * width + e];
        }
        part_blk[c] = sa;
        part_blk[C + c] = sq;
    }
}

}  // namespace nhwc
}  // namespace mdx
