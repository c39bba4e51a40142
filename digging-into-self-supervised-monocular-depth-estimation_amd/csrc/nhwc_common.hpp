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
    // round to nearest even, NaN stays NaN (torch's float -> bfloat16)
    uint32_t u = __float_as_uint(x);
    bf16 r;
    if ((u & 0x7fffffffu) > 0x7f800000u) { r.v = (uint16_t)((u >> 16) | 0x40u); return r; }
    u += 0x7fffu + ((u >> 16) & 1u);
    r.v = (uint16_t)(u >> 16);
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

}  // namespace nhwc
}  // namespace mdx
