// rt_wide_iso.hip — instantiations of k_render_path_wide<true, K, W> (see rt_kernels.h).
#include <hip/hip_runtime.h>

#include "rt_wide.h"

#define RT_K(KK, WW) reinterpret_cast<const void *>(rtd::k_render_path_wide<true, KK, WW>)

// K rows per lane, `waves` waves per workgroup; W > 1 is only ever asked for K > kmax/2.
const void *rt_wide_kernel_iso(int k, int waves) {
  if (waves == 1) {
    switch (k) {
    case 1: return RT_K(1, 1);
    case 2: return RT_K(2, 1);
    case 3: return RT_K(3, 1);
    case 4: return RT_K(4, 1);
    case 5: return RT_K(5, 1);
    case 6: return RT_K(6, 1);
    case 7: return RT_K(7, 1);
    case 8: return RT_K(8, 1);
    default: return RT_K(8, 1);
    }
  }
  if (waves == 2) {
    if (k <= 5) return RT_K(5, 2);
    if (k <= 6) return RT_K(6, 2);
    if (k <= 7) return RT_K(7, 2);
    return RT_K(8, 2);
  }
  if (waves == 4) {
    if (k <= 5) return RT_K(5, 4);
    if (k <= 6) return RT_K(6, 4);
    if (k <= 7) return RT_K(7, 4);
    return RT_K(8, 4);
  }
  if (waves == 8) {
    if (k <= 5) return RT_K(5, 8);
    if (k <= 6) return RT_K(6, 8);
    if (k <= 7) return RT_K(7, 8);
    return RT_K(8, 8);
  }
  return nullptr;
}
