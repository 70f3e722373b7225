// rt_wide_gen.hip — instantiations of k_render_path_wide<false, K, W> (see rt_kernels.h).
#include <hip/hip_runtime.h>

#include "rt_wide.h"

#define RT_K(KK, WW) reinterpret_cast<const void *>(rtd::k_render_path_wide<false, KK, WW>)

// K rows per lane, `waves` waves per workgroup; W > 1 is only ever asked for K > kmax/2.
const void *rt_wide_kernel_gen(int k, int waves) {
  if (waves == 1) {
    switch (k) {
    case 1: return RT_K(1, 1);
    case 2: return RT_K(2, 1);
    case 3: return RT_K(3, 1);
    case 4: return RT_K(4, 1);
    default: return RT_K(4, 1);
    }
  }
  if (waves == 2) {
    if (k <= 3) return RT_K(3, 2);
    return RT_K(4, 2);
  }
  if (waves == 4) {
    if (k <= 3) return RT_K(3, 4);
    return RT_K(4, 4);
  }
  if (waves == 8) {
    if (k <= 3) return RT_K(3, 8);
    return RT_K(4, 8);
  }
  return nullptr;
}
