// rt_internal.h — what the translation units of the C-ABI share: error reporting, device guards, the scene handle, and the
// launch interface between the entry points (rt_api.hip), scene construction (rt_scene.hip), the multi-GPU route
// (rt_multi.hip) and the tone-mapping entry points (rt_tonemap_api.hip).  Host code only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_amd.h"
#include "rt_path.h"

namespace rt_internal {

// sets the calling thread's rt_last_error() text and returns `st` (rt_api.hip)
rt_status fail(rt_status st, const std::string &msg);

#define RT_HIP(call)                                                                                       \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,                           \
                  std::string(#call) + ": " + hipGetErrorString(e_));                                      \
  } while (0)

struct DeviceBuffer {
  void *ptr = nullptr;
  size_t bytes = 0;
};

// Makes `device` current for the lifetime of the guard and gives the caller's device back afterwards: no entry point
// leaves the calling thread on another device than it came with.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess)
      prev = -1;
    if (prev != device)
      err = hipSetDevice(device);
  }
  ~DeviceGuard() {
    if (prev >= 0)
      (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define RT_ON_DEVICE(device)                                                                               \
  DeviceGuard device_guard_(device);                                                                       \
  RT_HIP(device_guard_.err)

// grows a grow-only device buffer (frees and re-allocates: the caller makes sure nothing in flight uses it)
rt_status ensure(DeviceBuffer &b, size_t bytes);
rt_status check_cfg(const rt_render_cfg *cfg);
rt_status device_cu_count(int device, int *n_cus);

} // namespace rt_internal

struct rt_scene {
  int device = 0;
  int n_cus = 0;
  std::vector<void *> owned;     // scene tables, freed by rt_scene_destroy
  rtd::Scene dev{};              // device pointers
  unsigned long long *counters = nullptr; // rtd::COUNTER_WORDS x u64
  int32_t tab[7] = {0, 0, 0, 0, 0, 0, 0}; // table sizes: shapes, materials, pigments, csg leaves / chain steps / nodes / program entries
  // what the world-space sphere filter's rounding error depends on (see world_filter_ok): the farthest register sphere from
  // the origin (|C| + r) and the smallest r^2 among them
  double wide_reach = 0.0, wide_rmin2 = 0.0;
  bool csg_all_tt4 = false;      // every CSG tree has a truth table and at most 4 leaves
  rt_internal::DeviceBuffer samples, stack;   // grow-only scratch reused across renders
  rt_internal::DeviceBuffer frame;            // rt_render_frame_host: the frame while it is in HBM
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> slice_events; // rt_render_frame_host: one per progress report
  hipStream_t poll_stream = nullptr;    // rt_render_frame_host: reads the sample queue's head while a path frame renders
};


namespace rt_internal {

// Where one render keeps what it needs beyond the scene and the framebuffer (tally counters, the sample buffer when
// spp > 1, the node stacks of path mode): grow-only buffers of the scene handle (rt_render), or a fixed workspace the
// caller of this function brings (rt_render_noalloc: the file-scope scratch array), in which case nothing is allocated.
struct Workspace {
  unsigned char *fixed = nullptr; // device address, 256-byte aligned; nullptr: allocate in the scene handle
  size_t fixed_bytes = 0;
  // A frame rendered as several launches on one stream (rt_render_frame_host, rt_multi_render): the tallies keep
  // accumulating across them (only the first launch zeroes them) and are read once, by read_tallies().
  bool keep_tallies = false;
  struct LaunchInfo *info = nullptr; // out: what was launched
};
struct LaunchInfo {
  unsigned grid = 0, block = 0;
  uint64_t n_samples = 0;
  unsigned long long *counters = nullptr;
  char kernel[48] = "";
};

// one launch (or, with ws.keep_tallies, one more launch of a frame) of the renderer cfg names on `stream_` (rt_api.hip)
rt_status render_impl(rt_scene *sc, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_device, void *stream_, rt_stats *stats,
                      const Workspace &ws);
// Ray / capped-sample tallies of the launches since the last zeroing (the stream they ran on must be idle).
rt_status read_tallies(const unsigned long long *counters, uint64_t *rays, uint64_t *capped);
bool whole_image(const rt_render_cfg *cfg);

} // namespace rt_internal
