// rt_wide_tile.hip — instantiations of k_render_path_tile<S, W> (rt_tile.h).
#include <hip/hip_runtime.h>

#include "rt_tile.h"

#ifndef RT_TILE_S
#define RT_TILE_S 8
#endif
#define RT_T(SS, WW) reinterpret_cast<const void *>(rtd::k_render_path_tile<RT_TILE_S, WW>)

// 8 sphere tiles per wave and pass (measured against 4 and 16, see rt_tile.h), `waves` waves per workgroup
const void *rt_tile_kernel(int waves, bool csg) {
  if (waves == 1 && !csg)
    return reinterpret_cast<const void *>(rtd::k_render_path_tile<RT_TILE_S, 1, false>);
  switch (waves) {
  case 1: return RT_T(8, 1);
  case 2: return RT_T(8, 2);
  case 4: return RT_T(8, 4);
  default: return RT_T(8, 8);
  }
}
