// photo_train.hpp -- what the training kernel (photo_train.hip) and its finishing pass (photo_train_finish.hip) share.
#pragma once
#include "photo_common.hpp"

namespace mdx {

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

// The finishing pass of mdx_photometric_train / _pre, ONE launch (photo_train_finish.hip): transposes of the bilinear
// upsample for the scales below full resolution (gup[s] [B,H,W] -> gdisp[s] [B,1,h,w]; a scale at full resolution was
// written in place by the training kernel), fixed-order sums of the items' d(P) and loss partials, and -- rng given --
// the advance of the in-kernel noise generator's offset.  grad = false: the loss sums only.  ipi = items per image.
int launch_train_finish(const mdx_train_desc *d, bool grad, int ipi, const float *partP, const double *loss_part,
                        float *const *gup, float *const *gdisp, float *gP, float *loss_sum, unsigned long long *rng,
                        hipStream_t st);

}  // namespace mdx
