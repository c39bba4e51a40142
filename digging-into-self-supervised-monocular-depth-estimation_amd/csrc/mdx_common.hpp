// mdx_common.hpp -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mdx.h"

#define MDX_EXPORT extern "C" __attribute__((visibility("default")))

namespace mdx {

constexpr int TX = 64;   // tile width  = one wave64 per tile row (256 B coalesced per plane)
constexpr int TY = 8;    // tile height
constexpr int NT = 256;  // threads per block = 4 waves

static inline int check_launch()
{
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_LAUNCH;
}

static inline bool aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

static inline int validate_desc(const mdx_desc *d)
{
    if (!d) return MDX_ERR_NULL_POINTER;
    if (d->B <= 0 || d->H < 4 || d->W < 4 || d->h <= 0 || d->w <= 0) return MDX_ERR_BAD_SHAPE;
    if (d->S < 1 || d->S > MDX_MAX_SRC) return MDX_ERR_BAD_SHAPE;
    if (d->h > d->H || d->w > d->W) return MDX_ERR_BAD_SHAPE;
    if ((long long)d->B * d->H * d->W >= (1ll << 31)) return MDX_ERR_BAD_SHAPE;
    return MDX_OK;
}

#include "mdx_divtable.inc"
// is x/b == the 3-instruction constant division for every normal x?  (tools/gen_divtable.c)
static inline bool div_verified(int b)
{
    return b >= 2 && b < 4096 && ((MDX_DIV_OK[b >> 5] >> (b & 31)) & 1u);
}

static inline dim3 tile_grid(const mdx_desc *d)
{
    return dim3((d->W + TX - 1) / TX, (d->H + TY - 1) / TY, d->B);
}

}  // namespace mdx
