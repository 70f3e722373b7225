// rt_wide_tile.hip — instantiations of k_render_path_tile<S, W> (rt_tile.h).
#include <hip/hip_runtime.h>

#include "rt_tile.h"
#if RT_TILE_BF16
#include "rt_tile_cull.h"
#endif

#ifndef RT_TILE_S
#define RT_TILE_S 8
#endif
#define RT_T(WW, CSG) reinterpret_cast<const void *>(rtd::k_render_path_tile<RT_TILE_S, WW, CSG>)

// S = 8 sphere tiles per wave and pass (measured against 4 and 16 in both builds, see rt_tile.h), `waves` waves per workgroup; worlds without
// CSG objects run the variants compiled without the evaluators (fewer registers spilled)
const void *rt_tile_kernel(int waves, bool csg) {
  switch (waves) {
  case 1: return csg ? RT_T(1, true) : RT_T(1, false);
  case 2: return csg ? RT_T(2, true) : RT_T(2, false);
#ifdef RT_TUNING // W = 8 is dominated at every measured size (rt_tile.h) and its 100 KB of LDS leave one workgroup per CU: A/B builds only
  case 8: return csg ? RT_T(8, true) : RT_T(8, false);
#endif
  default: return csg ? RT_T(4, true) : RT_T(4, false);
  }
}

int rt_tile_s() { return RT_TILE_S; }

// k_render_path_tile_cull<4, CSG> (rt_tile_cull.h): worlds of RT_CULL_MIN_TILES sphere tiles or more; nullptr in the binary32 build
const void *rt_tile_cull_kernel(bool csg) {
#if RT_TILE_BF16
  return csg ? reinterpret_cast<const void *>(rtd::k_render_path_tile_cull<4, true>) : reinterpret_cast<const void *>(rtd::k_render_path_tile_cull<4, false>);
#else
  (void)csg;
  return nullptr;
#endif
}
