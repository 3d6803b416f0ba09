// Host build of gs-sdf_amd/csrc/reach_mask.h (the product's own source, libm in place of the device intrinsics) for
// tests/test_reach_mask_conservative.py: masks of many (splat, tile) pairs in one call.
#define GSDF_REACH_MASK_HOST 1
#include <stdint.h>

#include "reach_mask.h"

// Round 6: the 2x2 mask (bit 8 by + bx) from the per-splat parameters, as the pack + mask passes of the compositing kernels compute it.
extern "C" void reach_masks2x2(int64_t n, const float *ray_transforms /* [n, 9] */, const float *means2d /* [n, 2] */, const float *opacities,
                               const float *tile_xy0 /* [n, 2] */, uint64_t *masks) {
  for (int64_t i = 0; i < n; ++i) {
    float p[8];
    gsdf::reach_params(ray_transforms + 9 * i, means2d[2 * i], means2d[2 * i + 1], opacities[i], p);
    masks[i] = gsdf::reach_mask2x2(p, tile_xy0[2 * i], tile_xy0[2 * i + 1]);
  }
}
