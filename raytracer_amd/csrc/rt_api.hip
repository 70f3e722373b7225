// rt_api.hip — the C-ABI of include/raytracer_amd.h: scene upload, launches, parity entry points.
// Host code only orchestrates; all arithmetic of the tracing loop is in rt_device.h / rt_kernels.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_amd.h"
#include "rt_kernels.h"
#include "rt_wavepath.h"
#include "rt_bvh.h"
#include "rt_tonemap.h"

namespace {

thread_local std::string g_last_error;

rt_status fail(rt_status st, const std::string &msg) {
  g_last_error = msg;
  return st;
}

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

} // namespace

// ---------------------------------------------------------------------------------------------
// rt_render_noalloc's workspace: a file-scope device array (every GPU the library runs on holds its own copy until the
// process ends).  In host code the NAME of the array is only a stand-in for the symbol, not a device address: the
// address on each device is asked for once, with that device current (scratch_on_device), and only that pointer is
// ever handed to kernels, hipMemsetAsync or copies.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int SCRATCH_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[SCRATCH_MAX_DEVICES] = {};

// Device address of `scratch` on `device`, which must be the current device.
rt_status scratch_on_device(int device, unsigned char **out) {
  if (device < 0 || device >= SCRATCH_MAX_DEVICES)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_noalloc: device index beyond the workspace table");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void *ptr = nullptr;
    RT_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr)
      return fail(RT_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
  }
  *out = g_scratch_ptr[device];
  return RT_OK;
}

} // namespace

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
  DeviceBuffer samples, stack;   // grow-only scratch reused across renders
  DeviceBuffer frame;            // rt_render_frame_host: the frame while it is in HBM
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> slice_events; // rt_render_frame_host: one per progress report
  hipStream_t poll_stream = nullptr;    // rt_render_frame_host: reads the sample queue's head while a path frame renders
};

namespace {

template <typename T> rt_status upload(rt_scene *sc, const T *host, size_t n, const T **out) {
  *out = nullptr;
  void *d = nullptr;
  RT_HIP(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)));
  sc->owned.push_back(d);
  if (n)
    RT_HIP(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T *>(d);
  return RT_OK;
}
template <typename T> rt_status upload(rt_scene *sc, const std::vector<T> &v, const T **out) {
  return upload(sc, v.data(), v.size(), out);
}

rt_status ensure(DeviceBuffer &b, size_t bytes) {
  if (b.bytes >= bytes)
    return RT_OK;
  if (b.ptr)
    RT_HIP(hipFree(b.ptr));
  b.ptr = nullptr;
  b.bytes = 0;
  RT_HIP(hipMalloc(&b.ptr, bytes));
  b.bytes = bytes;
  return RT_OK;
}

// ---- validation + CSG preprocessing ---------------------------------------------------------------

// Bounding-volume hierarchy over the top-level spheres (RT_FLAG_ACCEL_BVH, rt_bvh.h): median split of the centroids
// along the widest axis down to leaves of at most 4, nodes in heap order (root = slot 1, children of n = 2n, 2n + 1).
// The box of a sphere under the affine map M is its centre +- the row norms of M's linear part, padded so that rounding
// in the slab test cannot lose a hit.  Spheres that would blow up every box above them (a sky sphere around the whole
// scene) stay out of the tree and are scanned linearly.
struct BvhBuild {
  static constexpr int LEAF_MAX = rtd::BVH_LEAF_MAX;
  std::vector<int32_t> order;        // sphere positions: [0, n_small) grouped by leaf, then the big ones, ascending
  std::vector<float> lo, hi, centre; // per sphere, 3 each
  std::vector<float> box;            // per heap slot: lo[3], hi[3] (before quantisation)
  std::vector<uint32_t> leaf;        // per heap slot: 0 or count << 24 | first
  std::vector<rtd::BvhNode> nodes;
  float origin[3] = {0.f, 0.f, 0.f}, scale[3] = {1.f, 1.f, 1.f};
  int n_small = 0;
  std::vector<char> is_big;          // per sphere: too big for the tree (and for the register rows of the wide kernel)

  void build(size_t slot, int first, int count) {
    if (box.size() < 6 * (slot + 1))
      box.resize(6 * (slot + 1), 0.f), leaf.resize(slot + 1, 0u);
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = first; k < first + count; ++k)
      for (int c = 0; c < 3; ++c) {
        blo[c] = std::min(blo[c], lo[3 * order[k] + c]);
        bhi[c] = std::max(bhi[c], hi[3 * order[k] + c]);
        clo[c] = std::min(clo[c], centre[3 * order[k] + c]);
        chi[c] = std::max(chi[c], centre[3 * order[k] + c]);
      }
    std::memcpy(&box[6 * slot], blo, sizeof blo);
    std::memcpy(&box[6 * slot + 3], bhi, sizeof bhi);
    if (count <= LEAF_MAX) {
      leaf[slot] = (static_cast<uint32_t>(count) << 24) | static_cast<uint32_t>(first);
      return;
    }
    int axis = 0;
    for (int c = 1; c < 3; ++c)
      if (chi[c] - clo[c] > chi[axis] - clo[axis])
        axis = c;
    const int half = (count + 1) / 2;
    std::nth_element(order.begin() + first, order.begin() + first + half, order.begin() + first + count,
                     [&](int32_t x, int32_t y) { return centre[3 * x + axis] < centre[3 * y + axis]; });
    build(2 * slot, first, half);
    build(2 * slot + 1, first + half, count - half);
  }

  // `fwd_of(i)` = Object::transformation.hom_matrix of sphere i (12 floats)
  template <typename F> void run(size_t n, F fwd_of) {
    is_big.assign(n, 0);
    if (n < 2 || n >= (1u << 24))
      return;
    lo.resize(3 * n), hi.resize(3 * n), centre.resize(3 * n);
    std::vector<float> extent(n);
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; ++i) {
      const float *fwd = fwd_of(i);
      extent[i] = 0.f;
      for (int c = 0; c < 3; ++c) {
        const double half = std::sqrt(double(fwd[3 * c]) * fwd[3 * c] + double(fwd[3 * c + 1]) * fwd[3 * c + 1] +
                                      double(fwd[3 * c + 2]) * fwd[3 * c + 2]);
        const double pad = half * 1e-4 + std::fabs(double(fwd[9 + c])) * 1e-6 + 1e-6;
        centre[3 * i + c] = fwd[9 + c];
        lo[3 * i + c] = static_cast<float>(fwd[9 + c] - half - pad);
        hi[3 * i + c] = static_cast<float>(fwd[9 + c] + half + pad);
        extent[i] = std::max(extent[i], static_cast<float>(half));
        clo[c] = std::min(clo[c], centre[3 * i + c]);
        chi[c] = std::max(chi[c], centre[3 * i + c]);
      }
    }
    const float spread = std::max(chi[0] - clo[0], std::max(chi[1] - clo[1], chi[2] - clo[2]));
    std::vector<int32_t> big;
    for (size_t i = 0; i < n; ++i) {
      const bool finite = std::isfinite(lo[3 * i]) && std::isfinite(lo[3 * i + 1]) && std::isfinite(lo[3 * i + 2]) &&
                          std::isfinite(hi[3 * i]) && std::isfinite(hi[3 * i + 1]) && std::isfinite(hi[3 * i + 2]);
      if (!finite || !(extent[i] <= 0.25f * spread)) {
        big.push_back(static_cast<int32_t>(i));
        is_big[i] = 1;
      } else
        order.push_back(static_cast<int32_t>(i));
    }
    n_small = static_cast<int>(order.size());
    if (n_small < 2) {
      order.clear();
      n_small = 0;
      return;
    }
    build(1, 0, n_small);
    order.insert(order.end(), big.begin(), big.end());
    // quantise on the lattice origin + q * scale, outwards; the device evaluates fmaf(q, scale, origin) in binary32
    for (int c = 0; c < 3; ++c) {
      origin[c] = box[6 + c];
      scale[c] = std::max((box[6 + 3 + c] - box[6 + c]) / 65000.f, 1e-30f);
    }
    nodes.assign(leaf.size(), rtd::BvhNode{0u, 0u, 0u, 0u});
    for (size_t slot = 1; slot < leaf.size(); ++slot) {
      uint32_t q[6];
      for (int c = 0; c < 3; ++c) {
        const float l = box[6 * slot + c], h = box[6 * slot + 3 + c];
        long ql = std::lround(std::floor((double(l) - origin[c]) / scale[c])) - 1;
        long qh = std::lround(std::ceil((double(h) - origin[c]) / scale[c])) + 1;
        ql = std::clamp(ql, 0l, 65535l), qh = std::clamp(qh, 0l, 65535l);
        while (ql > 0 && std::fmaf(static_cast<float>(ql), scale[c], origin[c]) > l)
          --ql;
        while (qh < 65535 && std::fmaf(static_cast<float>(qh), scale[c], origin[c]) < h)
          ++qh;
        q[c] = static_cast<uint32_t>(ql), q[3 + c] = static_cast<uint32_t>(qh);
      }
      nodes[slot] = rtd::BvhNode{q[0] | (q[3] << 16), q[1] | (q[4] << 16), q[2] | (q[5] << 16), leaf[slot]};
    }
  }
};

struct CsgBuild {
  const rt_scene_desc *d;
  std::vector<rtd::CsgTop> tops;
  std::vector<rtd::CsgLeaf> leaves;
  std::vector<rtd::CsgStep> path;
  std::vector<int32_t> prog_range; // [node][side][2]
  std::vector<rtd::CsgProgEntry> prog;
  std::vector<int32_t> prog_path;
  std::string err;

  bool ref_ok(rt_ref r, int32_t below_node) const {
    if (r >= 0)
      return static_cast<uint32_t>(r) < d->n_shapes;
    const int32_t node = ~r;
    return node >= 0 && node < below_node; // children precede their parent: no cycles
  }

  // leaves of the subtree `ref`, left to right, each with the chain root..parent
  bool collect(rt_ref ref, std::vector<rtd::CsgStep> &chain, int top_index) {
    if (ref >= 0) {
      rtd::CsgLeaf leaf{};
      leaf.shape = ref;
      leaf.path_begin = static_cast<int32_t>(path.size());
      leaf.path_len = static_cast<int32_t>(chain.size());
      path.insert(path.end(), chain.begin(), chain.end());
      leaves.push_back(leaf);
      if (static_cast<int32_t>(leaves.size()) - tops[top_index].leaf_begin > RT_MAX_CSG_LEAVES) {
        err = "CSG object has more than RT_MAX_CSG_LEAVES leaves";
        return false;
      }
      return true;
    }
    const int32_t node = ~ref;
    if (static_cast<int>(chain.size()) >= RT_MAX_CSG_DEPTH) {
      err = "CSG nesting deeper than RT_MAX_CSG_DEPTH";
      return false;
    }
    chain.push_back({node, 0});
    if (!collect(d->csg_left[node], chain, top_index))
      return false;
    chain.back().side = 1;
    if (!collect(d->csg_right[node], chain, top_index))
      return false;
    chain.pop_back();
    return true;
  }

  // RPN "inside?" program of the subtree `ref`; `chain` = CSG nodes from the subtree root down
  void program(rt_ref ref, std::vector<int32_t> &chain) {
    if (ref >= 0) {
      rtd::CsgProgEntry e{};
      e.is_op = 0;
      e.arg = ref;
      e.path_begin = static_cast<int32_t>(prog_path.size());
      e.path_len = static_cast<int32_t>(chain.size());
      prog_path.insert(prog_path.end(), chain.begin(), chain.end());
      prog.push_back(e);
      return;
    }
    const int32_t node = ~ref;
    chain.push_back(node);
    program(d->csg_left[node], chain);
    program(d->csg_right[node], chain);
    chain.pop_back();
    rtd::CsgProgEntry e{};
    e.is_op = 1;
    e.arg = d->csg_op[node];
    prog.push_back(e);
  }
};

rt_status validate_desc(const rt_scene_desc *d) {
  auto need = [](const void *p, uint32_t n) { return n == 0 || p != nullptr; };
  if (!need(d->shape_kind, d->n_shapes) || !need(d->shape_material, d->n_shapes) || !need(d->shape_fwd, d->n_shapes) ||
      !need(d->shape_inv, d->n_shapes) || !need(d->csg_op, d->n_csg_nodes) || !need(d->csg_left, d->n_csg_nodes) ||
      !need(d->csg_right, d->n_csg_nodes) || !need(d->csg_fwd, d->n_csg_nodes) || !need(d->csg_inv, d->n_csg_nodes) ||
      !need(d->objects, d->n_objects) || !need(d->material_brdf, d->n_materials) ||
      !need(d->material_brdf_pigment, d->n_materials) || !need(d->material_emission_pigment, d->n_materials) ||
      !need(d->pigment_kind, d->n_pigments) || !need(d->pigment_color1, d->n_pigments) ||
      !need(d->pigment_color2, d->n_pigments) || !need(d->pigment_n_intervals, d->n_pigments) ||
      !need(d->pigment_tex_offset, d->n_pigments) || !need(d->pigment_tex_width, d->n_pigments) ||
      !need(d->pigment_tex_height, d->n_pigments) || !need(d->texels, d->n_texels) ||
      !need(d->light_point, d->n_lights) || !need(d->light_color, d->n_lights) || !need(d->light_radius, d->n_lights))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_desc: NULL array with a non-zero count");
  for (uint32_t i = 0; i < d->n_shapes; ++i) {
    if (d->shape_kind[i] > RT_SHAPE_PLANE)
      return fail(RT_ERR_INVALID_ARGUMENT, "shape_kind out of range");
    if (d->shape_material[i] >= d->n_materials)
      return fail(RT_ERR_INVALID_ARGUMENT, "shape_material out of range");
  }
  for (uint32_t i = 0; i < d->n_materials; ++i) {
    if (d->material_brdf[i] > RT_BRDF_SPECULAR)
      return fail(RT_ERR_INVALID_ARGUMENT, "material_brdf out of range");
    if (d->material_brdf_pigment[i] >= d->n_pigments || d->material_emission_pigment[i] >= d->n_pigments)
      return fail(RT_ERR_INVALID_ARGUMENT, "material pigment index out of range");
  }
  for (uint32_t i = 0; i < d->n_pigments; ++i) {
    const uint8_t k = d->pigment_kind[i];
    if (k > RT_PIGMENT_IMAGE)
      return fail(RT_ERR_INVALID_ARGUMENT, "pigment_kind out of range");
    if (k == RT_PIGMENT_CHECKERED && d->pigment_n_intervals[i] <= 0)
      return fail(RT_ERR_INVALID_ARGUMENT, "checkered pigment needs n_intervals > 0");
    if (k == RT_PIGMENT_IMAGE) {
      const int64_t w = d->pigment_tex_width[i], h = d->pigment_tex_height[i];
      if (w <= 0 || h <= 0 || static_cast<uint64_t>(d->pigment_tex_offset[i]) + static_cast<uint64_t>(w * h) > d->n_texels)
        return fail(RT_ERR_INVALID_ARGUMENT, "image pigment texels out of range");
    }
  }
  for (uint32_t i = 0; i < d->n_csg_nodes; ++i)
    if (d->csg_op[i] > RT_CSG_FUSION)
      return fail(RT_ERR_INVALID_ARGUMENT, "csg_op out of range");
  return RT_OK;
}

rt_status check_cfg(const rt_render_cfg *cfg) {
  if (cfg->width <= 0 || cfg->height <= 0 || cfg->spp_edge < 1 || cfg->spp_edge > 1024)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_cfg: width/height must be positive, 1 <= spp_edge <= 1024");
  if (cfg->mode > RT_MODE_PATH)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_cfg: unknown mode");
  if (cfg->band_rows <= 0 || cfg->band_stride < cfg->band_rows || cfg->band_offset < 0 || cfg->n_local_rows <= 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_cfg: bad row banding");
  const int last = cfg->n_local_rows - 1;
  const int64_t last_row = static_cast<int64_t>(last / cfg->band_rows) * cfg->band_stride + cfg->band_offset + last % cfg->band_rows;
  if (last_row >= cfg->height)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_cfg: row banding reaches past the image height");
  if (cfg->mode == RT_MODE_PATH) {
    if (cfg->n_rays < 1 || cfg->n_rays > 65535 || cfg->max_depth < 0 || cfg->max_depth > 32766 || cfg->roulette < 0)
      return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_cfg: path mode needs 1 <= n_rays <= 65535, 0 <= max_depth <= 32766, roulette >= 0");
    if (cfg->n_rays > 1 && cfg->max_depth > 64)
      return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_cfg: max_depth > 64 is supported only with n_rays == 1");
    if (cfg->rng_mode != RT_RNG_PER_NODE_STREAMS)
      return fail(RT_ERR_INVALID_ARGUMENT,
                  "rt_render_cfg: the device renders path mode with per-sample streams only (RT_RNG_PER_NODE_STREAMS)");
  }
  if (cfg->mode == RT_MODE_POINTLIGHT && cfg->max_mirror_bounces <= 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_cfg: pointlight needs max_mirror_bounces > 0 (a lane must never spin)");
  return RT_OK;
}

} // namespace

// Selectors of the instantiated k_render_path_wide<ISO, K, W> shapes (rt_wide_iso.hip / rt_wide_gen.hip;
// separate translation units so that the ~30 kernels compile in parallel).
const void *rt_wide_kernel_iso(int k, int waves);
const void *rt_wide_kernel_gen(int k, int waves);
const void *rt_tile_kernel(int waves, bool csg); // rt_wide_tile.hip: k_render_path_tile<S, W, CSG>
int rt_tile_s();                                // its S: sphere tiles per wave and pass
// From how many register spheres on the large-world kernels replace the per-lane scan of the small-world kernel.  Measured
// with the tile kernel (synthetic worlds, 960x540 AA=4, frame ms tile | per-lane): 16 spheres 17.2 | 9.8, 24: 20.5 | 19.2,
// 32: 26.6 | 27.5, 48: 26.2 | 27.5, 64: 25.7 | 26.6 — one full tile of 32 is the break-even.
#ifndef RT_WIDE_MIN_SPHERES
#define RT_WIDE_MIN_SPHERES 32
#endif
#ifndef RT_WAVE_OCC_NO_CSG
#define RT_WAVE_OCC_NO_CSG 6
#endif

namespace {
} // namespace

extern "C" {

const char *rt_last_error(void) { return g_last_error.c_str(); }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

rt_status rt_scene_create(const rt_scene_desc *d, int device, rt_scene **out) {
  if (!d || !out)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: NULL argument");
  *out = nullptr;
  if (rt_status st = validate_desc(d))
    return st;

  // -- split World::objects into the scan lists, preprocess CSG trees ------------------------------
  std::vector<rtd::ScanRec> spheres, planes;
  CsgBuild cb;
  cb.d = d;
  cb.prog_range.assign(static_cast<size_t>(d->n_csg_nodes) * 4, 0);
  for (uint32_t node = 0; node < d->n_csg_nodes; ++node)
    if (!cb.ref_ok(d->csg_left[node], static_cast<int32_t>(node)) || !cb.ref_ok(d->csg_right[node], static_cast<int32_t>(node)))
      return fail(RT_ERR_INVALID_ARGUMENT, "csg child reference out of range (children must precede their parent)");
  for (uint32_t i = 0; i < d->n_objects; ++i)
    if (!cb.ref_ok(d->objects[i], static_cast<int32_t>(d->n_csg_nodes)))
      return fail(RT_ERR_INVALID_ARGUMENT, "objects[] reference out of range");
  // The World is a forest, as in the reference, where every Object is owned by exactly one unique_ptr (World::objects or
  // a CSGObject's object1 / object2, include/shapes.hpp:384-385,525): a shape or node referenced twice is refused.  (The
  // per-shape tables with the CSG ancestors folded in, shape_w2l / shape_l2w below, rely on it, and a shared node would
  // make the flattened programs grow as 2^depth.)
  {
    std::vector<uint8_t> shape_refs(d->n_shapes, 0), node_refs(d->n_csg_nodes, 0);
    auto take = [&](rt_ref r) { uint8_t &c = r >= 0 ? shape_refs[static_cast<size_t>(r)] : node_refs[static_cast<size_t>(~r)]; return ++c <= 1; };
    bool tree = true;
    for (uint32_t node = 0; node < d->n_csg_nodes && tree; ++node)
      tree = take(d->csg_left[node]) && take(d->csg_right[node]);
    for (uint32_t i = 0; i < d->n_objects && tree; ++i)
      tree = take(d->objects[i]);
    if (!tree)
      return fail(RT_ERR_INVALID_ARGUMENT, "a shape or CSG node is referenced more than once (by objects[] and / or CSG children): every "
                                           "object has exactly one owner, as in the reference's World; duplicate the shape instead");
  }
  // Limits of the fixed-capacity evaluator, checked bottom-up (children precede parents) before anything recursive runs.
  {
    std::vector<int32_t> depth(d->n_csg_nodes, 0), n_leaves(d->n_csg_nodes, 0);
    for (uint32_t node = 0; node < d->n_csg_nodes; ++node) {
      int32_t dep = 0, lv = 0;
      for (rt_ref r : {d->csg_left[node], d->csg_right[node]}) {
        dep = std::max(dep, r >= 0 ? 0 : depth[static_cast<size_t>(~r)]);
        lv += r >= 0 ? 1 : n_leaves[static_cast<size_t>(~r)];
      }
      depth[node] = dep + 1;
      n_leaves[node] = std::min(lv, RT_MAX_CSG_LEAVES + 1);
      if (depth[node] > RT_MAX_CSG_DEPTH)
        return fail(RT_ERR_INVALID_ARGUMENT, "CSG nesting deeper than RT_MAX_CSG_DEPTH");
      if (n_leaves[node] > RT_MAX_CSG_LEAVES)
        return fail(RT_ERR_INVALID_ARGUMENT, "CSG object has more than RT_MAX_CSG_LEAVES leaves");
    }
  }
  for (uint32_t node = 0; node < d->n_csg_nodes; ++node) {
    for (int side = 0; side < 2; ++side) {
      std::vector<int32_t> chain;
      cb.prog_range[4 * node + 2 * side + 0] = static_cast<int32_t>(cb.prog.size());
      cb.program(side == 0 ? d->csg_left[node] : d->csg_right[node], chain);
      cb.prog_range[4 * node + 2 * side + 1] = static_cast<int32_t>(cb.prog.size());
    }
  }
  for (uint32_t i = 0; i < d->n_objects; ++i) {
    const rt_ref ref = d->objects[i];
    if (ref >= 0) {
      rtd::ScanRec r{};
      std::memcpy(r.inv, d->shape_inv + 12 * static_cast<size_t>(ref), sizeof r.inv);
      r.object = static_cast<int32_t>(i);
      r.shape = ref;
      (d->shape_kind[ref] == RT_SHAPE_SPHERE ? spheres : planes).push_back(r);
    } else {
      rtd::CsgTop top{};
      top.object = static_cast<int32_t>(i);
      top.leaf_begin = static_cast<int32_t>(cb.leaves.size());
      cb.tops.push_back(top);
      std::vector<rtd::CsgStep> chain;
      if (!cb.collect(ref, chain, static_cast<int>(cb.tops.size()) - 1))
        return fail(RT_ERR_INVALID_ARGUMENT, cb.err);
      cb.tops.back().leaf_end = static_cast<int32_t>(cb.leaves.size());
    }
  }
  // truth tables of the validity rule for small trees (rt_types.h CsgTop::tt)
  std::vector<unsigned long long> leaf_tt(cb.leaves.size(), 0ull);
  for (rtd::CsgTop &top : cb.tops) {
    const int n_leaves = top.leaf_end - top.leaf_begin;
    if (n_leaves > rtd::CSG_TT_MAX_LEAVES)
      continue;
    auto position = [&](int32_t shape) { // place of a shape among the tree's leaves (a forest: exactly one)
      for (int j = 0; j < n_leaves; ++j)
        if (cb.leaves[static_cast<size_t>(top.leaf_begin + j)].shape == shape)
          return j;
      return 0;
    };
    auto inside = [&](int begin, int end, uint32_t mask) { // CSGObject::is_point_inside of a subtree (shapes.hpp:438-457) on a mask
      uint32_t bits = 0;
      for (int e = begin; e < end; ++e) {
        const rtd::CsgProgEntry &pe = cb.prog[static_cast<size_t>(e)];
        if (!pe.is_op) {
          bits = (bits << 1) | ((mask >> position(pe.arg)) & 1u);
        } else {
          const bool in2 = bits & 1u, in1 = (bits >> 1) & 1u;
          const bool r = pe.arg == RT_CSG_INTERSECTION ? (in1 && in2) : pe.arg == RT_CSG_DIFFERENCE ? (in1 && !in2) : (in1 || in2);
          bits = ((bits >> 2) << 1) | (r ? 1u : 0u);
        }
      }
      return (bits & 1u) != 0;
    };
    for (int j = 0; j < n_leaves; ++j) {
      const rtd::CsgLeaf &leaf = cb.leaves[static_cast<size_t>(top.leaf_begin + j)];
      unsigned long long table = 0;
      for (uint32_t mask = 0; mask < (1u << n_leaves); ++mask) {
        bool valid = true;
        for (int k = leaf.path_len - 1; k >= 0 && valid; --k) { // CSGObject::hit_on_obj{1,2}_valid (shapes.hpp:460-490), parent ... root
          const rtd::CsgStep st = cb.path[static_cast<size_t>(leaf.path_begin + k)];
          const int32_t *range = &cb.prog_range[4 * static_cast<size_t>(st.node) + 2 * static_cast<size_t>(1 - st.side)];
          const bool in = inside(range[0], range[1], mask);
          switch (d->csg_op[st.node]) {
          case RT_CSG_UNION: break;
          case RT_CSG_INTERSECTION: valid = in; break;
          case RT_CSG_DIFFERENCE: valid = st.side == 0 ? !in : in; break;
          default: valid = !in; break; // FUSION
          }
        }
        if (valid)
          table |= 1ull << mask;
      }
      leaf_tt[static_cast<size_t>(top.leaf_begin + j)] = table;
    }
    top.tt = 1;
  }
  // world <-> local transforms of every shape with its CSG ancestors folded in (computed in binary64, rounded once)
  std::vector<float> shape_w2l(d->shape_inv, d->shape_inv + 12 * static_cast<size_t>(d->n_shapes));
  std::vector<float> shape_l2w(d->shape_fwd, d->shape_fwd + 12 * static_cast<size_t>(d->n_shapes));
  {
    struct Aff { double m[12]; };
    auto load = [](const float *f) { Aff a; for (int i = 0; i < 12; ++i) a.m[i] = f[i]; return a; };
    auto mul = [](const Aff &a, const Aff &b) { // x -> a(b(x))
      Aff r;
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
          r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
        r.m[9 + i] = a.m[3 * i] * b.m[9] + a.m[3 * i + 1] * b.m[10] + a.m[3 * i + 2] * b.m[11] + a.m[9 + i];
      }
      return r;
    };
    for (const rtd::CsgLeaf &leaf : cb.leaves) {
      Aff w2l = load(d->shape_inv + 12 * static_cast<size_t>(leaf.shape)); // applied last
      Aff l2w = load(d->shape_fwd + 12 * static_cast<size_t>(leaf.shape)); // applied first
      for (int k = leaf.path_len - 1; k >= 0; --k) {                       // parent ... root
        const int32_t node = cb.path[leaf.path_begin + k].node;
        w2l = mul(w2l, load(d->csg_inv + 12 * static_cast<size_t>(node)));
        l2w = mul(load(d->csg_fwd + 12 * static_cast<size_t>(node)), l2w);
      }
      for (int i = 0; i < 12; ++i) {
        shape_w2l[12 * static_cast<size_t>(leaf.shape) + i] = static_cast<float>(w2l.m[i]);
        shape_l2w[12 * static_cast<size_t>(leaf.shape) + i] = static_cast<float>(l2w.m[i]);
      }
    }
  }
  // translation * uniform scaling spheres can be tested in world space from (centre, r^2)
  std::vector<float> sphere_iso(spheres.size() * 4);
  bool all_iso = !spheres.empty();
  for (size_t i = 0; i < spheres.size(); ++i) {
    const float *inv = spheres[i].inv;
    const float *fwd = d->shape_fwd + 12 * static_cast<size_t>(spheres[i].shape);
    const bool iso = inv[0] > 0.f && inv[4] == inv[0] && inv[8] == inv[0] && inv[1] == 0.f && inv[2] == 0.f && inv[3] == 0.f &&
                     inv[5] == 0.f && inv[6] == 0.f && inv[7] == 0.f && fwd[0] > 0.f && fwd[4] == fwd[0] && fwd[8] == fwd[0] &&
                     fwd[1] == 0.f && fwd[2] == 0.f && fwd[3] == 0.f && fwd[5] == 0.f && fwd[6] == 0.f && fwd[7] == 0.f;
    all_iso = all_iso && iso;
    sphere_iso[4 * i + 0] = fwd[9], sphere_iso[4 * i + 1] = fwd[10], sphere_iso[4 * i + 2] = fwd[11];
    sphere_iso[4 * i + 3] = fwd[0] * fwd[0];
  }
  BvhBuild bvh;
  bvh.run(spheres.size(), [&](size_t i) { return d->shape_fwd + 12 * static_cast<size_t>(spheres[i].shape); });
  std::vector<rtd::ScanRec> bvh_recs, bvh_big;
  std::vector<float> bvh_iso;
  for (size_t k = 0; k < bvh.order.size(); ++k) {
    const size_t i = static_cast<size_t>(bvh.order[k]);
    if (k < static_cast<size_t>(bvh.n_small)) {
      bvh_recs.push_back(spheres[i]);
      bvh_iso.insert(bvh_iso.end(), sphere_iso.begin() + 4 * i, sphere_iso.begin() + 4 * i + 4);
    } else {
      bvh_big.push_back(spheres[i]);
    }
  }
  // Wide path kernel (rt_wide.h): which spheres sit in the register rows and which are scanned per lane.  Per lane go
  // the big ones (the line of nearly every ray meets a sky sphere, and a hit in the transposed scan costs the whole
  // wave ~50 instructions where a per-lane test costs 33 for 64 rays) and a leftover that would fill less than a
  // quarter of a register row (a fuller leftover is cheaper as one more, partly empty, row).  Both lists ascend.
  std::vector<rtd::ScanRec> wide_recs, wide_lane;
  std::vector<float> wide_iso;
  {
    std::vector<size_t> rest;
    std::vector<char> lane(spheres.size(), 0);
    for (size_t i = 0; i < spheres.size(); ++i)
      if (bvh.is_big[i])
        lane[i] = 1;
      else
        rest.push_back(i);
    const size_t leftover = rest.size() % 64;
    if (leftover < 16)
      for (size_t k = 0; k < leftover; ++k)
        lane[rest[k]] = 1;
    for (size_t i = 0; i < spheres.size(); ++i) {
      if (lane[i]) {
        wide_lane.push_back(spheres[i]);
      } else {
        wide_recs.push_back(spheres[i]);
        // the wide kernel's filter form (rt_wide.h): centre and k = C.C - r^2, in binary64 before rounding
        const double cx = sphere_iso[4 * i], cy = sphere_iso[4 * i + 1], cz = sphere_iso[4 * i + 2], r = d->shape_fwd[12 * static_cast<size_t>(spheres[i].shape)];
        wide_iso.insert(wide_iso.end(), sphere_iso.begin() + 4 * i, sphere_iso.begin() + 4 * i + 3);
        wide_iso.push_back(static_cast<float>(cx * cx + cy * cy + cz * cz - r * r));
      }
    }
  }
  double wide_reach = 0.0, wide_rmin2 = std::numeric_limits<double>::infinity();
  if (all_iso)
    for (size_t k = 0; k < wide_recs.size(); ++k) {
      const double cx = wide_iso[4 * k], cy = wide_iso[4 * k + 1], cz = wide_iso[4 * k + 2];
      const double r = std::fabs(static_cast<double>(d->shape_fwd[12 * static_cast<size_t>(wide_recs[k].shape)]));
      wide_reach = std::max(wide_reach, std::sqrt(cx * cx + cy * cy + cz * cz) + r);
      wide_rmin2 = std::min(wide_rmin2, r * r);
    }
  std::vector<float> wide_tiles; // rt_tile.h: the MFMA's A operands
  const size_t n_wide_tiles = all_iso ? (wide_recs.size() + 31) / 32 : 0;
  wide_tiles.resize(n_wide_tiles * rtd::RT_TILE_WORDS);
#if RT_TILE_BF16
  // K slot s of the two 16-slot chunks (rt_tile.h): 6 t + j, t = x y z: part {0 0 0 1 1 2}[j] of the coordinate; 18-20: 1;
  // 21-23: the parts of k; 24-31: 0.  Lane l of chunk c holds slots 16 c + 8 (l / 32) + 0..7 of sphere l % 32 as four words.
  auto bf16_rn = [](float x) -> uint32_t { // round to nearest even (finite values)
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  };
  auto bf16_value = [](uint32_t h) {
    const uint32_t u = h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  auto split3 = [&](float x, uint32_t part[3]) { // x = part0 + part1 + part2 exactly (8 + 8 + 8 significant bits)
    part[0] = bf16_rn(x);
    float r = x - bf16_value(part[0]);
    part[1] = bf16_rn(r);
    r -= bf16_value(part[1]);
    part[2] = bf16_rn(r);
  };
  static const int part_of_slot[6] = {0, 0, 0, 1, 1, 2};
  for (size_t t = 0; t < n_wide_tiles; ++t)
    for (size_t i = 0; i < 32; ++i) {
      const size_t k = t * 32 + i;
      const bool valid = k < wide_recs.size();
      const float v[4] = {valid ? wide_iso[4 * k] : 0.f, valid ? wide_iso[4 * k + 1] : 0.f, valid ? wide_iso[4 * k + 2] : 0.f,
                          valid ? wide_iso[4 * k + 3] : 1e30f}; // a place past the last sphere: c = 1e30, never hit
      uint32_t parts[4][3];
      for (int q = 0; q < 4; ++q)
        split3(v[q], parts[q]);
      uint32_t slot[32] = {0};
      for (int q = 0; q < 3; ++q)
        for (int j = 0; j < 6; ++j)
          slot[6 * q + j] = parts[q][part_of_slot[j]];
      slot[18] = slot[19] = slot[20] = 0x3f80u; // 1
      slot[21] = parts[3][0], slot[22] = parts[3][1], slot[23] = parts[3][2];
      for (size_t c = 0; c < 2; ++c)
        for (size_t half = 0; half < 2; ++half)
          for (size_t w = 0; w < 4; ++w) {
            const size_t s0 = 16 * c + 8 * half + 2 * w;
            const uint32_t word = slot[s0] | (slot[s0 + 1] << 16);
            std::memcpy(&wide_tiles[t * rtd::RT_TILE_WORDS + (c * 64 + half * 32 + i) * 4 + w], &word, 4);
          }
    }
#else
  for (size_t t = 0; t < n_wide_tiles; ++t)
    for (size_t l = 0; l < 64; ++l) {
      const size_t k = t * 32 + l % 32;
      const bool valid = k < wide_recs.size(), upper = l >= 32;
      const float cx = valid ? wide_iso[4 * k] : 0.f, cy = valid ? wide_iso[4 * k + 1] : 0.f, cz = valid ? wide_iso[4 * k + 2] : 0.f;
      const float kk = valid ? wide_iso[4 * k + 3] : std::numeric_limits<float>::infinity();
      wide_tiles[t * 192 + l] = upper ? cy : cx;
      wide_tiles[t * 192 + 64 + l] = upper ? 1.f : cz;
      wide_tiles[t * 192 + 128 + l] = upper ? kk : cz;
    }
#endif
  std::vector<int32_t> shape_origin(4 * static_cast<size_t>(d->n_shapes), -1);
  for (uint32_t i = 0; i < d->n_shapes; ++i)
    shape_origin[4 * static_cast<size_t>(i) + 3] = 0;
  for (const rtd::ScanRec &r : spheres)
    shape_origin[4 * static_cast<size_t>(r.shape) + 0] = r.object, shape_origin[4 * static_cast<size_t>(r.shape) + 3] = 1;
  for (const rtd::ScanRec &r : planes)
    shape_origin[4 * static_cast<size_t>(r.shape) + 0] = r.object;
  for (size_t k = 0; k < wide_recs.size(); ++k)
    shape_origin[4 * static_cast<size_t>(wide_recs[k].shape) + 1] = static_cast<int32_t>(k);
  for (size_t k = 0; k < bvh_recs.size(); ++k)
    shape_origin[4 * static_cast<size_t>(bvh_recs[k].shape) + 2] = static_cast<int32_t>(k);
  std::vector<uint8_t> needs_uv(d->n_materials);
  for (uint32_t m = 0; m < d->n_materials; ++m)
    needs_uv[m] = d->pigment_kind[d->material_brdf_pigment[m]] != RT_PIGMENT_UNIFORM ||
                  d->pigment_kind[d->material_emission_pigment[m]] != RT_PIGMENT_UNIFORM;

  std::vector<rtd::ShadeRec> shade(d->n_shapes);
  for (uint32_t i = 0; i < d->n_shapes; ++i) {
    const uint32_t m = d->shape_material[i], pb = d->material_brdf_pigment[m], pe = d->material_emission_pigment[m];
    rtd::ShadeRec &r = shade[i];
    std::memcpy(r.brdf_color, d->pigment_color1 + 3 * static_cast<size_t>(pb), sizeof r.brdf_color);
    std::memcpy(r.emit_color, d->pigment_color1 + 3 * static_cast<size_t>(pe), sizeof r.emit_color);
    r.material = m;
    r.flags = (d->material_brdf[m] == RT_BRDF_SPECULAR ? 1u : 0u) | (d->pigment_kind[pb] == RT_PIGMENT_UNIFORM ? 2u : 0u) |
              (d->pigment_kind[pe] == RT_PIGMENT_UNIFORM ? 4u : 0u);
  }

  // -- device ------------------------------------------------------------------------------------
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
    return fail(RT_ERR_NO_DEVICE, "rt_scene_create: no HIP device visible");
  if (device < 0 || device >= n_dev)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: device index out of range");
  RT_ON_DEVICE(device);
  rt_scene *sc = new (std::nothrow) rt_scene();
  if (!sc)
    return fail(RT_ERR_OUT_OF_MEMORY, "rt_scene_create: host allocation failed");
  sc->device = device;
  hipDeviceProp_t prop;
  rt_status st = RT_OK;
  auto bail = [&](rt_status s) {
    rt_scene_destroy(sc);
    return s;
  };
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return bail(fail(RT_ERR_HIP, "hipGetDeviceProperties failed"));
  sc->n_cus = prop.multiProcessorCount;
  {
    const size_t sizes[7] = {d->n_shapes, d->n_materials, d->n_pigments, cb.leaves.size(), cb.path.size(), d->n_csg_nodes, cb.prog.size()};
    for (int i = 0; i < 7; ++i)
      sc->tab[i] = static_cast<int32_t>(std::min<size_t>(sizes[i], 1u << 30));
  }

  rtd::Scene &S = sc->dev;
#define UP(expr)                                                                                           \
  if ((st = (expr)) != RT_OK)                                                                              \
  return bail(st)
  UP(upload(sc, spheres, &S.spheres));
  S.spheres_all_iso = all_iso ? 1 : 0;
  UP(upload(sc, bvh.nodes, &S.bvh));
  S.n_bvh_nodes = static_cast<int32_t>(bvh.nodes.size());
  std::memcpy(S.bvh_origin, bvh.origin, sizeof S.bvh_origin);
  std::memcpy(S.bvh_scale, bvh.scale, sizeof S.bvh_scale);
  for (int c = 0; c < 3; ++c)
    S.bvh_inv_scale[c] = 1.f / bvh.scale[c];
  UP(upload(sc, bvh_recs, &S.bvh_recs));
  UP(upload(sc, bvh_iso, &S.bvh_iso));
  UP(upload(sc, bvh_big, &S.bvh_big));
  S.n_bvh_big = static_cast<int32_t>(bvh_big.size());
  S.n_bvh_small = bvh.n_small;
  UP(upload(sc, planes, &S.planes));
  UP(upload(sc, cb.tops, &S.csg_tops));
  S.n_spheres = static_cast<int32_t>(spheres.size());
  UP(upload(sc, wide_recs, &S.wide_recs));
  UP(upload(sc, wide_iso, &S.wide_iso));
  UP(upload(sc, wide_lane, &S.wide_lane));
  UP(upload(sc, wide_tiles, &S.wide_tiles));
  S.n_wide_tiles = static_cast<int32_t>(n_wide_tiles);
  sc->wide_reach = wide_reach, sc->wide_rmin2 = wide_rmin2;
  S.n_wide = static_cast<int32_t>(wide_recs.size());
  S.wide_rows = (S.n_wide + 63) / 64;
  S.n_wide_lane = static_cast<int32_t>(wide_lane.size());
  S.n_planes = static_cast<int32_t>(planes.size());
  S.n_csg_tops = static_cast<int32_t>(cb.tops.size());
  sc->csg_all_tt4 = !cb.tops.empty();
  for (const rtd::CsgTop &top : cb.tops)
    sc->csg_all_tt4 = sc->csg_all_tt4 && top.tt && top.leaf_end - top.leaf_begin <= 4;
  S.n_objects = static_cast<int32_t>(d->n_objects);
  UP(upload(sc, d->shape_fwd, 12 * static_cast<size_t>(d->n_shapes), &S.shape_fwd));
  UP(upload(sc, d->shape_inv, 12 * static_cast<size_t>(d->n_shapes), &S.shape_inv));
  UP(upload(sc, shape_w2l, &S.shape_w2l));
  UP(upload(sc, shape_l2w, &S.shape_l2w));
  UP(upload(sc, d->shape_kind, d->n_shapes, &S.shape_kind));
  UP(upload(sc, d->shape_material, d->n_shapes, &S.shape_material));
  UP(upload(sc, shade, &S.shape_shade));
  UP(upload(sc, shape_origin, &S.shape_origin));
  UP(upload(sc, cb.leaves, &S.csg_leaf));
  UP(upload(sc, leaf_tt, &S.csg_leaf_tt));
  UP(upload(sc, cb.path, &S.csg_path));
  UP(upload(sc, d->csg_fwd, 12 * static_cast<size_t>(d->n_csg_nodes), &S.csg_fwd));
  UP(upload(sc, d->csg_inv, 12 * static_cast<size_t>(d->n_csg_nodes), &S.csg_inv));
  UP(upload(sc, d->csg_op, d->n_csg_nodes, &S.csg_op));
  UP(upload(sc, cb.prog_range, &S.csg_prog_range));
  UP(upload(sc, cb.prog, &S.csg_prog));
  UP(upload(sc, cb.prog_path, &S.csg_prog_path));
  UP(upload(sc, d->material_brdf, d->n_materials, &S.material_brdf));
  UP(upload(sc, d->material_brdf_pigment, d->n_materials, &S.material_brdf_pigment));
  UP(upload(sc, d->material_emission_pigment, d->n_materials, &S.material_emission_pigment));
  UP(upload(sc, needs_uv, &S.material_needs_uv));
  UP(upload(sc, d->pigment_kind, d->n_pigments, &S.pigment_kind));
  UP(upload(sc, d->pigment_color1, 3 * static_cast<size_t>(d->n_pigments), &S.pigment_color1));
  UP(upload(sc, d->pigment_color2, 3 * static_cast<size_t>(d->n_pigments), &S.pigment_color2));
  UP(upload(sc, d->pigment_n_intervals, d->n_pigments, &S.pigment_n_intervals));
  UP(upload(sc, d->pigment_tex_offset, d->n_pigments, &S.pigment_tex_offset));
  UP(upload(sc, d->pigment_tex_width, d->n_pigments, &S.pigment_tex_width));
  UP(upload(sc, d->pigment_tex_height, d->n_pigments, &S.pigment_tex_height));
  UP(upload(sc, d->texels, 3 * static_cast<size_t>(d->n_texels), &S.texels));
  UP(upload(sc, d->light_point, 3 * static_cast<size_t>(d->n_lights), &S.light_point));
  UP(upload(sc, d->light_color, 3 * static_cast<size_t>(d->n_lights), &S.light_color));
  UP(upload(sc, d->light_radius, d->n_lights, &S.light_radius));
  S.n_lights = static_cast<int32_t>(d->n_lights);
#undef UP
  void *cnt = nullptr;
  if (hipMalloc(&cnt, rtd::COUNTER_WORDS * sizeof(unsigned long long)) != hipSuccess)
    return bail(fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc(counters) failed"));
  sc->counters = static_cast<unsigned long long *>(cnt);
  if (hipEventCreate(&sc->ev0) != hipSuccess || hipEventCreate(&sc->ev1) != hipSuccess)
    return bail(fail(RT_ERR_HIP, "hipEventCreate failed"));
  *out = sc;
  return RT_OK;
}

void rt_scene_destroy(rt_scene *sc) {
  if (!sc)
    return;
  DeviceGuard guard(sc->device);
  for (void *p : sc->owned)
    (void)hipFree(p);
  if (sc->counters) (void)hipFree(sc->counters);
  if (sc->samples.ptr) (void)hipFree(sc->samples.ptr);
  if (sc->stack.ptr) (void)hipFree(sc->stack.ptr);
  if (sc->frame.ptr) (void)hipFree(sc->frame.ptr);
  for (hipEvent_t e : sc->slice_events) (void)hipEventDestroy(e);
  if (sc->poll_stream) (void)hipStreamDestroy(sc->poll_stream);
  if (sc->ev0) (void)hipEventDestroy(sc->ev0);
  if (sc->ev1) (void)hipEventDestroy(sc->ev1);
  delete sc;
}

} // extern "C"

namespace {

// The world-space sphere filter (rt_tile.h, rt_wide.h ISO form) computes c = (C.C - r^2) - 2 O.C + O.O in binary32: terms of
// size |O|^2 + |C|^2 cancel down to something compared with r^2, leaving a rounding error of about 2^-22 (|O|^2 + |C|^2) — a
// silhouette fuzzy by that much, sign at random.  At the benchmark's scale (coordinates ~10, r >= 0.05) that is 2 % of the
// smallest r^2 and the frames agree with the reference's own renders (tests/test_synth_large.py); a world far from the origin
// or with tiny spheres would silently lose accuracy, so such worlds take the object-space kernel (k_render_path_wide<GEN>:
// the ray is mapped into each sphere's frame first, nothing cancels).  Ray origins are bounded by the camera's and by the
// spheres' own reach (rays leaving far points of an unbounded plane are not: their share of a frame is negligible).
bool world_filter_ok(const rt_scene *sc, const rt_camera *cam, float resolved_asp) {
  const float *m = cam->fwd;
  const double col0 = std::sqrt(double(m[0]) * m[0] + double(m[3]) * m[3] + double(m[6]) * m[6]);
  const double col1 = std::sqrt(double(m[1]) * m[1] + double(m[4]) * m[4] + double(m[7]) * m[7]);
  const double col2 = std::sqrt(double(m[2]) * m[2] + double(m[5]) * m[5] + double(m[8]) * m[8]);
  const double shift = std::sqrt(double(m[9]) * m[9] + double(m[10]) * m[10] + double(m[11]) * m[11]);
  const double asp = std::fabs(double(resolved_asp)); // what the render uses: the camera's own ratio or width / height
  const double cam_reach = shift + std::max(1.0, std::fabs(double(cam->distance))) * col0 + asp * col1 + col2; // both camera kinds
  const double o = std::max(cam_reach, sc->wide_reach), c = sc->wide_reach;
  const double err = std::ldexp(o * o + c * c, -22);
  return err <= 0.05 * sc->wide_rmin2;
}

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
constexpr size_t WS_COUNTER_BYTES = 1024; // >= rtd::COUNTER_WORDS * 8, keeps what follows 256-byte aligned
static_assert(WS_COUNTER_BYTES >= rtd::COUNTER_WORDS * sizeof(unsigned long long), "counter area too small");

rt_status render_impl(rt_scene *sc, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_device, void *stream_,
                      rt_stats *stats, const Workspace &ws) {
  if (!sc || !cam || !cfg || !rgb_out_device)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: NULL argument");
  if (rt_status st = check_cfg(cfg))
    return st;
  if (cam->kind > RT_CAMERA_PERSPECTIVE)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_camera: unknown kind");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  size_t ws_used = WS_COUNTER_BYTES; // fixed workspace: [counters | samples | node stacks]

  rtd::RenderParams p{};
  std::memcpy(p.cam.fwd, cam->fwd, sizeof p.cam.fwd);
  // ImageTracer ctor: asp_ratio defaults to float(width)/float(height) — cameras.hpp:184-186
  p.cam.asp = cam->has_asp_ratio ? cam->asp_ratio : static_cast<float>(cfg->width) / static_cast<float>(cfg->height);
  p.cam.distance = cam->distance;
  p.cam.kind = static_cast<int32_t>(cam->kind);
  p.width = cfg->width, p.height = cfg->height, p.spp_edge = cfg->spp_edge, p.mode = static_cast<int32_t>(cfg->mode);
  p.n_rays = cfg->n_rays, p.roulette = cfg->roulette, p.max_depth = cfg->max_depth;
  p.max_mirror_bounces = cfg->max_mirror_bounces;
  p.seq_number = cfg->seq_number;
  std::memcpy(p.background, cfg->background, sizeof p.background);
  std::memcpy(p.ambient, cfg->ambient, sizeof p.ambient);
  p.band_rows = cfg->band_rows, p.band_stride = cfg->band_stride, p.band_offset = cfg->band_offset;
  p.n_local_rows = cfg->n_local_rows;
  const uint64_t n_pixels = static_cast<uint64_t>(cfg->n_local_rows) * static_cast<uint64_t>(cfg->width);
  const int spp2 = cfg->spp_edge * cfg->spp_edge;
  p.n_samples = n_pixels * static_cast<uint64_t>(spp2);
  p.counters = ws.fixed ? reinterpret_cast<unsigned long long *>(ws.fixed) : sc->counters;
  p.sample_stride = 1;
  if (p.n_samples > 2 && p.n_samples < (1ull << 31)) { // the product q * stride must fit 64 bits
    uint64_t a = static_cast<uint64_t>(0.6180339887498949 * static_cast<double>(p.n_samples)) | 1ull;
    while (std::gcd(a, p.n_samples) != 1)
      a += 2;
    p.sample_stride = a % p.n_samples;
  }

  if (spp2 > 1) {
    const size_t bytes = p.n_samples * 3 * sizeof(float);
    if (ws.fixed) {
      if (bytes > ws.fixed_bytes - ws_used)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_noalloc: the sample buffer (width * n_local_rows * spp_edge^2 * 12 = " +
                                                 std::to_string(bytes) + " bytes) does not fit the " + std::to_string(ws.fixed_bytes) +
                                                 "-byte workspace; render fewer rows per call or use rt_render");
      p.samples_out = reinterpret_cast<float *>(ws.fixed + ws_used);
      ws_used += (bytes + 255) & ~static_cast<size_t>(255);
    } else {
      if (rt_status st = ensure(sc->samples, bytes))
        return st;
      p.samples_out = static_cast<float *>(sc->samples.ptr);
    }
  } else {
    p.samples_out = rgb_out_device;
  }

  constexpr int BLOCK = 256;
  unsigned grid, block = BLOCK;
  const void *path_kernel = nullptr;
  char kernel_name[48] = "";
  static const char *const simple_names[3] = {"k_render_simple<ONOFF>", "k_render_simple<FLAT>", "k_render_simple<POINTLIGHT>"};
  if (cfg->mode != RT_MODE_PATH)
    std::snprintf(kernel_name, sizeof kernel_name, "%s", simple_names[cfg->mode]);
  size_t dyn_lds = 0;
  if (cfg->mode == RT_MODE_PATH) {
    const bool exact = (cfg->flags & RT_FLAG_EXACT_ARITHMETIC) != 0;
    const int32_t *tab = sc->tab;
    p.tab_shapes = tab[0], p.tab_materials = tab[1], p.tab_pigments = tab[2], p.tab_csg_leaves = tab[3], p.tab_csg_path = tab[4];
    p.tab_csg_nodes = tab[5], p.tab_csg_prog = tab[6];
    using L = rtd::PathLdsLimits;
    const bool small_tables = tab[0] <= L::SHAPES && tab[1] <= L::MATERIALS && tab[2] <= L::PIGMENTS && tab[3] <= L::CSG_LEAVES &&
                              tab[4] <= L::CSG_PATH && tab[5] <= L::CSG_NODES && tab[6] <= L::CSG_PROG && sc->dev.n_csg_tops <= L::CSG_TOPS;
    const bool use_bvh = (cfg->flags & RT_FLAG_ACCEL_BVH) != 0 && sc->dev.n_bvh_nodes > 0;
#ifdef RT_TUNING // -DRT_TUNING builds only (scripts/build_variant.sh): kernel selection overridden from the environment for A/B runs
    static const int wide_min = [] { const char *e = std::getenv("RT_WIDE_MIN"); return e && *e ? std::atoi(e) : RT_WIDE_MIN_SPHERES; }();
#else
    constexpr int wide_min = RT_WIDE_MIN_SPHERES;
#endif
    const bool use_wide = !exact && !use_bvh && sc->dev.n_wide >= wide_min; // measured: see RT_WIDE_MIN_SPHERES
    // Small worlds with small lookup tables (every sample scene of the reference): the pool run wave by wave (rt_wavepath.h)
    const bool use_wave = small_tables && !use_bvh && !use_wide;
    int nodes_per_level = 64, stacks_per_block = 1, node_words = rtd::WAVE_NODE_WORDS; // every path kernel: one pool of 64 lanes per wave
    if (use_wave) {
      // Worlds without CSG objects run the variant compiled without the evaluators and their tables (102 VGPRs at 4 waves per
      // SIMD, 94 at 5, 80 at 6 with 2 spilled; 25 KB of LDS, so 6 workgroups fit a CU): measured on C3, waves per SIMD
      // 4 -> 15.0, 5 -> 14.1, 6 -> 13.8 ms (the kernel with the CSG code: 15.6 at 4, its best).
      // With CSG objects: 5 (96 VGPRs) beats 4 and 6 — measured on C4: 46.1 / 43.7 / 62.7 ms.
#ifdef RT_TUNING
      static const int occ_env = [] { const char *e = std::getenv("RT_WAVEPATH_OCC"); return e ? std::atoi(e) : 0; }();
#else
      constexpr int occ_env = 0;
#endif
      const bool csg = sc->dev.n_csg_tops > 0;
      const int occ = occ_env ? occ_env : (csg ? 5 : RT_WAVE_OCC_NO_CSG);
#define RT_WAVE_K(FAST, OCC, CSG) reinterpret_cast<const void *>(rtd::k_render_path_wave<FAST, OCC, CSG>)
      if (exact)
        path_kernel = RT_WAVE_K(false, 4, true);
      else if (csg && sc->csg_all_tt4) // every tree has a truth table and at most 4 leaves (samples/csg_example.txt): that evaluator only
        path_kernel = reinterpret_cast<const void *>(rtd::k_render_path_wave<true, 5, true, true>);
      else if (csg)
        path_kernel = occ == 4 ? RT_WAVE_K(true, 4, true) : occ == 6 ? RT_WAVE_K(true, 6, true) : RT_WAVE_K(true, 5, true);
      else
        path_kernel = occ == 4 ? RT_WAVE_K(true, 4, false) : occ == 6 ? RT_WAVE_K(true, 6, false) : RT_WAVE_K(true, 5, false);
#undef RT_WAVE_K
      std::snprintf(kernel_name, sizeof kernel_name, "k_render_path_wave<%s%s>", exact ? "EXACT" : "FMA",
                    exact ? "" : !csg ? ",NO_CSG" : sc->csg_all_tt4 ? ",CSG_TT4" : "");
      block = rtd::WAVEPATH_THREADS;
      dyn_lds = (!exact && !csg) ? rtd::wave_lds_layout_for<false>().total : rtd::wave_lds_layout_for<true>().total;
      nodes_per_level = 64, stacks_per_block = rtd::WAVEPATH_WAVES, node_words = rtd::WAVE_NODE_WORDS;
    } else {
      path_kernel = exact ? reinterpret_cast<const void *>(rtd::k_render_path<false>) : reinterpret_cast<const void *>(rtd::k_render_path<true>);
      std::snprintf(kernel_name, sizeof kernel_name, "k_render_path<%s>", exact ? "EXACT" : "FMA");
      block = rtd::PATH_THREADS;
      nodes_per_level = 64, stacks_per_block = rtd::PATH_THREADS / 64, node_words = rtd::WAVE_NODE_WORDS; // one wave pool per wave
    }
    if (use_bvh) {
      // The tree (or its top) and, if they fit too, the leaf spheres go to LDS next to the workgroup's pool state:
      // 256 lanes when all of it fits 40 KiB (4 workgroups per CU), else 512 lanes within 72 KiB (2 per CU) — 4 waves
      // per SIMD either way (measured: 256 lanes are 5-8 % faster where they fit, 20 % slower at 3 workgroups per CU).
      const size_t n_nodes = static_cast<size_t>(sc->dev.n_bvh_nodes), n_small = static_cast<size_t>(sc->dev.n_bvh_small);
      p.bvh_iso = sc->dev.spheres_all_iso && !(cfg->flags & RT_FLAG_OBJECT_SPACE_SPHERES) ? 1 : 0;
      const bool want_leaves = !exact && p.bvh_iso;
      // per workgroup besides the tree: one 8-byte key per ray and 64 donor lanes per wave (rt_bvh.h) = 12 bytes per lane
      const size_t room256 = (40 * 1024 - 4 * sizeof(rtd::WaveShared) - 512 - 12 * 256) / sizeof(rtd::BvhNode);
      const size_t room512 = (72 * 1024 - 8 * sizeof(rtd::WaveShared) - 512 - 12 * 512) / sizeof(rtd::BvhNode);
      block = n_nodes + (want_leaves ? n_small : 0) <= room256 ? 256u : 512u;
      const size_t room = block == 256 ? room256 : room512;
      p.bvh_lds_nodes = static_cast<int32_t>(std::min(n_nodes, room));
      const bool split = static_cast<size_t>(p.bvh_lds_nodes) < n_nodes;
      const bool leaf_lds = want_leaves && !split && n_nodes + n_small <= room;
      p.bvh_lds_spheres = leaf_lds ? static_cast<int32_t>(n_small) : 0;
#define RT_BVH_K(FAST, T, SPLIT, LEAF) reinterpret_cast<const void *>(rtd::k_render_path_bvh<FAST, T, SPLIT, LEAF>)
#define RT_BVH_PICK(T)                                                                                              \
  (exact ? (split ? RT_BVH_K(false, T, true, false) : RT_BVH_K(false, T, false, false))                           \
         : (split ? RT_BVH_K(true, T, true, false) : (leaf_lds ? RT_BVH_K(true, T, false, true) : RT_BVH_K(true, T, false, false))))
      path_kernel = block == 256 ? RT_BVH_PICK(256) : RT_BVH_PICK(512);
      std::snprintf(kernel_name, sizeof kernel_name, "k_render_path_bvh<%s,T=%u%s%s>", exact ? "EXACT" : "FMA", block, split ? ",SPLIT" : "",
                    leaf_lds ? ",LEAF_LDS" : "");
#undef RT_BVH_PICK
#undef RT_BVH_K
      dyn_lds = static_cast<size_t>(p.bvh_lds_nodes + p.bvh_lds_spheres) * sizeof(rtd::BvhNode) + 12 * static_cast<size_t>(block);
      nodes_per_level = 64, stacks_per_block = static_cast<int>(block / 64), node_words = rtd::WAVE_NODE_WORDS; // one wave pool per wave
      if (dyn_lds > 32 * 1024) // above what a kernel gets without asking
        RT_HIP(hipFuncSetAttribute(path_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dyn_lds)));
    }
    // Large worlds: spheres in registers, rays through LDS (see k_render_path_wide).
    const int rows = sc->dev.wide_rows;
    if (use_wide) {
      const bool iso = sc->dev.spheres_all_iso && !(cfg->flags & RT_FLAG_OBJECT_SPACE_SPHERES) && world_filter_ok(sc, cam, p.cam.asp);
      // K rows per lane within 128 VGPRs (4 waves per SIMD measured faster than more rows per lane at 2),
      // as few waves per workgroup as that allows; more rows than 8 waves hold: several register passes.
      const int kmax = iso ? 8 : 4;
      int waves = 1;
      while (waves < 8 && rows > waves * kmax)
        waves *= 2;
      const int k = std::min((rows + waves - 1) / waves, kmax);
      block = 64u * static_cast<unsigned>(waves);
      path_kernel = iso ? rt_wide_kernel_iso(k, waves) : rt_wide_kernel_gen(k, waves);
      std::snprintf(kernel_name, sizeof kernel_name, "k_render_path_wide<%s,K=%d,W=%d>", iso ? "ISO" : "GEN", k, waves);
      nodes_per_level = 64, stacks_per_block = waves, node_words = rtd::WAVE_NODE_WORDS; // one wave pool per wave (rt_wavepath.h)
      if (iso && !(cfg->flags & RT_FLAG_VALU_ONLY)) {
        // The filter on the matrix cores (rt_tile.h): 8 tiles of 32 spheres per wave and pass.  Waves per workgroup, measured
        // (Mrays/s at W = 1 / 2 / 4 / 8; N=512: 10723 / 10261 / 9839 / 8894; N=1024: 5366 / 5446 / 5585 / 5207; N=4096: 1275 /
        // 1270 / 1403 / 1373): 1 up to 16 tiles, 2 up to 32, 4 beyond — small workgroups wait less at their two barriers, until
        // the passes (sphere registers reloaded from L2) get too many.
        const int tiles = sc->dev.n_wide_tiles;
#ifdef RT_TUNING
        static const int w_env = [] { const char *e = std::getenv("RT_TILE_W"); return e && *e ? std::atoi(e) : 0; }();
#else
        constexpr int w_env = 0;
#endif
        const int tile_s = rt_tile_s();
        waves = 1;
        while (waves < 4 && tiles > waves * 2 * tile_s) // up to two passes per wave count
          waves *= 2;
        if (w_env == 1 || w_env == 2 || w_env == 4 || w_env == 8)
          waves = w_env;
        block = 64u * static_cast<unsigned>(waves);
        path_kernel = rt_tile_kernel(waves, sc->dev.n_csg_tops > 0);
        nodes_per_level = 64, stacks_per_block = waves, node_words = rtd::WAVE_NODE_WORDS; // one wave pool per wave (rt_wavepath.h)
        std::snprintf(kernel_name, sizeof kernel_name, "k_render_path_tile<S=%d,W=%d%s>", tile_s, waves, RT_TILE_BF16 ? ",BF16x3" : "");
      }
    }
    int per_cu = 0;
    RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, path_kernel, static_cast<int>(block), dyn_lds));
    per_cu = std::max(per_cu, 1);
    const uint64_t want = (p.n_samples + block - 1) / block;
    grid = static_cast<unsigned>(std::min<uint64_t>(want, static_cast<uint64_t>(sc->n_cus) * per_cu));
    // one node stack per wave: at most 64 nodes per depth level (rt_path.h: the stack discipline)
    stacks_per_block = static_cast<int>(block / 64);
    p.pool_cap = nodes_per_level * (cfg->n_rays == 1 ? 1 : std::max(cfg->max_depth, 1));
    const size_t stack_bytes = static_cast<size_t>(stacks_per_block) * node_words * static_cast<size_t>(p.pool_cap) * sizeof(float); // per workgroup
    if (ws.fixed) {
      // Fewer workgroups than the device could hold is fine for the result (sample values do not depend on which
      // workgroup traces them), but below one per CU the call would crawl: refuse instead.
      const size_t fit = (ws.fixed_bytes - ws_used) / stack_bytes;
      const unsigned floor_grid = std::min<unsigned>(grid, static_cast<unsigned>(sc->n_cus));
      if (fit < floor_grid)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_noalloc: the node stacks (" + std::to_string(stack_bytes) + " bytes per workgroup x " +
                                                 std::to_string(floor_grid) + " workgroups) do not fit what the sample buffer leaves of the " +
                                                 std::to_string(ws.fixed_bytes) + "-byte workspace; lower max_depth or the rows per call, or use rt_render");
      grid = static_cast<unsigned>(std::min<size_t>(grid, fit));
      p.pool = reinterpret_cast<float *>(ws.fixed + ws_used);
      ws_used += static_cast<size_t>(grid) * stack_bytes;
    } else {
      if (rt_status st = ensure(sc->stack, static_cast<size_t>(grid) * stack_bytes))
        return st;
      p.pool = static_cast<float *>(sc->stack.ptr);
    }
  } else {
    const uint64_t want = (p.n_samples + BLOCK - 1) / BLOCK;
    grid = static_cast<unsigned>(std::min<uint64_t>(want, static_cast<uint64_t>(sc->n_cus) * 8)); // 8 waves per SIMD, grid-stride
  }

  (void)hipGetLastError(); // an error left behind by an earlier, unrelated call must not be reported as this launch's
  // counter 0 is the call's sample queue; the rest are tallies
  RT_HIP(hipMemsetAsync(p.counters, 0, (ws.keep_tallies ? 1 : rtd::COUNTER_WORDS) * sizeof(unsigned long long), stream));
  if (ws.info) {
    ws.info->grid = grid, ws.info->block = block, ws.info->n_samples = p.n_samples, ws.info->counters = p.counters;
    std::memcpy(ws.info->kernel, kernel_name, sizeof ws.info->kernel);
  }
  if (stats)
    RT_HIP(hipEventRecord(sc->ev0, stream));
  switch (cfg->mode) {
  case RT_MODE_ONOFF: hipLaunchKernelGGL(rtd::k_render_simple<RT_MODE_ONOFF>, dim3(grid), dim3(BLOCK), 0, stream, sc->dev, p); break;
  case RT_MODE_FLAT: hipLaunchKernelGGL(rtd::k_render_simple<RT_MODE_FLAT>, dim3(grid), dim3(BLOCK), 0, stream, sc->dev, p); break;
  case RT_MODE_POINTLIGHT: hipLaunchKernelGGL(rtd::k_render_simple<RT_MODE_POINTLIGHT>, dim3(grid), dim3(BLOCK), 0, stream, sc->dev, p); break;
  default: {
    void *args[] = {&sc->dev, &p};
    RT_HIP(hipLaunchKernel(path_kernel, dim3(grid), dim3(block), args, dyn_lds, stream));
    break;
  }
  }
  RT_HIP(hipGetLastError());
  if (spp2 > 1) {
    const unsigned rgrid = static_cast<unsigned>((n_pixels + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(rtd::k_resolve, dim3(rgrid), dim3(BLOCK), 0, stream, p.samples_out, rgb_out_device, n_pixels, spp2);
    RT_HIP(hipGetLastError());
  }
  if (stats) {
    RT_HIP(hipEventRecord(sc->ev1, stream));
    RT_HIP(hipEventSynchronize(sc->ev1));
    unsigned long long c[rtd::COUNTER_WORDS];
    RT_HIP(hipMemcpy(c, p.counters, sizeof c, hipMemcpyDeviceToHost));
    for (int k = 0; k < rtd::COUNTER_SLOTS; ++k) // the tallies are spread over slots (rt_path.h)
      c[1] += c[rtd::COUNTER_BASE + 2 * k], c[2] += c[rtd::COUNTER_BASE + 2 * k + 1];
#ifdef RT_PROFILE_PHASES
    std::fprintf(stderr, "[rt launch] grid=%u block=%u pool_cap=%d\n", grid, block, p.pool_cap);
    if (c[7])
      std::fprintf(stderr, "[rt phases] blocks*iters=%llu rays/iter=%.1f cycles/iter: queue=%.0f scan=%.0f shade=%.0f (100 MHz ticks x?)\n", c[7],
                   double(c[8]) / double(c[7]), double(c[4]) / double(c[7]), double(c[5]) / double(c[7]), double(c[6]) / double(c[7]));
    if (c[7] && c[9])
      std::fprintf(stderr, "[rt phases] of shade: finish (lane spheres, planes, exact record) = %.0f\n", double(c[9]) / double(c[7]));
#endif
#ifdef RT_BVH_STATS
    if (c[8])
      std::fprintf(stderr, "[rt bvh] grid=%u block=%u lds_nodes=%d lds_spheres=%d per ray: inner=%.1f leaf=%.1f pop=%.1f walk-slots=%.1f leaf-slots=%.1f steals=%.2f\n", grid, block,
                   p.bvh_lds_nodes, p.bvh_lds_spheres, double(c[4]) / double(c[8]), double(c[5]) / double(c[8]), double(c[6]) / double(c[8]),
                   double(c[7]) / double(c[8]), double(c[9]) / double(c[8]), double(c[10]) / double(c[8]));
#endif
    float ms = 0.f;
    RT_HIP(hipEventElapsedTime(&ms, sc->ev0, sc->ev1));
    stats->rays = c[1];
    stats->samples = p.n_samples;
    stats->capped_samples = c[2];
    stats->kernel_ms = ms;
    stats->grid = grid;
    stats->block = block;
    std::memcpy(stats->kernel, kernel_name, sizeof stats->kernel);
  }
  return RT_OK;
}

} // namespace

extern "C" {

rt_status rt_render(rt_scene *sc, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_device, void *stream,
                    rt_stats *stats) {
  if (!sc)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: NULL argument");
  RT_ON_DEVICE(sc->device);
  return render_impl(sc, cam, cfg, rgb_out_device, stream, stats, Workspace{});
}

rt_status rt_render_noalloc(rt_scene *sc, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_device, void *stream,
                            rt_stats *stats) {
  if (!sc)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_noalloc: NULL argument");
  RT_ON_DEVICE(sc->device);
  Workspace ws;
  if (rt_status st = scratch_on_device(sc->device, &ws.fixed))
    return st;
  ws.fixed_bytes = RT_NOALLOC_SCRATCH_BYTES;
  return render_impl(sc, cam, cfg, rgb_out_device, stream, stats, ws);
}

rt_status rt_render_host(rt_scene *sc, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_host, rt_stats *stats) {
  if (!sc || !cfg || !rgb_out_host)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_host: NULL argument");
  if (rt_status st = check_cfg(cfg))
    return st;
  RT_ON_DEVICE(sc->device);
  const size_t bytes = static_cast<size_t>(cfg->n_local_rows) * cfg->width * 3 * sizeof(float);
  float *fb = nullptr;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&fb), bytes));
  rt_status st = rt_render(sc, cam, cfg, fb, nullptr, stats);
  if (st == RT_OK) {
    hipError_t e = hipMemcpy(rgb_out_host, fb, bytes, hipMemcpyDeviceToHost); // synchronises the default stream
    if (e != hipSuccess)
      st = fail(RT_ERR_HIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
  }
  (void)hipFree(fb);
  return st;
}

} // extern "C"

namespace {

// Ray / capped-sample tallies of the launches since the last zeroing (the stream they ran on must be idle).
rt_status read_tallies(const unsigned long long *counters, uint64_t *rays, uint64_t *capped) {
  unsigned long long c[rtd::COUNTER_WORDS];
  RT_HIP(hipMemcpy(c, counters, sizeof c, hipMemcpyDeviceToHost));
  *rays = *capped = 0;
  for (int k = 0; k < rtd::COUNTER_SLOTS; ++k)
    *rays += c[rtd::COUNTER_BASE + 2 * k], *capped += c[rtd::COUNTER_BASE + 2 * k + 1];
  return RT_OK;
}

bool whole_image(const rt_render_cfg *cfg) {
  return cfg->band_rows == cfg->height && cfg->band_offset == 0 && cfg->n_local_rows == cfg->height;
}

} // namespace

extern "C" {

rt_status rt_render_frame_host(rt_scene *sc, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_host, int n_slices,
                               rt_progress_fn progress, void *user, rt_stats *stats) {
  if (!sc || !cam || !cfg || !rgb_out_host)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_frame_host: NULL argument");
  if (rt_status st = check_cfg(cfg))
    return st;
  if (!whole_image(cfg))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_frame_host renders whole images (band_rows = n_local_rows = height, band_offset = 0)");
  RT_ON_DEVICE(sc->device);
  const size_t row_floats = static_cast<size_t>(cfg->width) * 3, bytes = static_cast<size_t>(cfg->height) * row_floats * sizeof(float);
  if (rt_status st = ensure(sc->frame, bytes))
    return st;
  float *fb = static_cast<float *>(sc->frame.ptr);
  n_slices = progress ? std::clamp(n_slices, 1, std::min(cfg->height, 256)) : 1;
  // Path mode is ONE launch whatever n_slices is (slices would each end in a tail of a few long samples: measured 29 ms
  // against 16 ms for C3 in 8 slices); progress is read off the launch's sample queue instead — the fraction of the
  // frame's samples that have been handed out — by small copies on a second stream while the kernel runs.
  const bool poll = progress && cfg->mode == RT_MODE_PATH && n_slices > 1;
  const int n_reports = n_slices;
  if (poll) {
    n_slices = 1;
    if (!sc->poll_stream)
      RT_HIP(hipStreamCreateWithFlags(&sc->poll_stream, hipStreamNonBlocking));
  }
  while (sc->slice_events.size() < static_cast<size_t>(n_slices) + 1) {
    hipEvent_t e = nullptr;
    RT_HIP(hipEventCreate(&e));
    sc->slice_events.push_back(e);
  }
  hipStream_t stream = nullptr;
  Workspace ws;
  LaunchInfo info;
  ws.info = &info;
  // The slices are queued back to back and differ by a row in height: size the sample buffer for the tallest one now, so
  // that no later slice re-allocates (hipFree would wait for the slices in flight and stall the queue).
  if (cfg->spp_edge > 1) {
    const size_t tallest = static_cast<size_t>((cfg->height + n_slices - 1) / n_slices);
    if (rt_status st = ensure(sc->samples, tallest * cfg->width * static_cast<size_t>(cfg->spp_edge) * cfg->spp_edge * 3 * sizeof(float)))
      return st;
  }
  if (poll) // the queue head the poll reads must not be the previous frame's: zero it before anything is launched
    RT_HIP(hipMemset(sc->counters, 0, sizeof(unsigned long long)));
  RT_HIP(hipEventRecord(sc->slice_events[0], stream));
  std::vector<int> row_end(static_cast<size_t>(n_slices));
  for (int i = 0; i < n_slices; ++i) { // all slices are queued before the first report: the GPU never waits for the host
    const int row0 = static_cast<int>(static_cast<long long>(cfg->height) * i / n_slices);
    const int row1 = static_cast<int>(static_cast<long long>(cfg->height) * (i + 1) / n_slices);
    row_end[static_cast<size_t>(i)] = row1;
    rt_render_cfg slice = *cfg;
    slice.band_rows = slice.n_local_rows = row1 - row0, slice.band_stride = cfg->height, slice.band_offset = row0;
    ws.keep_tallies = i > 0;
    if (rt_status st = render_impl(sc, cam, &slice, fb + static_cast<size_t>(row0) * row_floats, stream, nullptr, ws))
      return st;
    RT_HIP(hipEventRecord(sc->slice_events[static_cast<size_t>(i) + 1], stream));
  }
  if (poll) {
    int reported = 0;
    for (;;) {
      const hipError_t q = hipEventQuery(sc->slice_events[1]);
      if (q == hipSuccess)
        break;
      if (q != hipErrorNotReady)
        return fail(RT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
      unsigned long long head = 0;
      RT_HIP(hipMemcpyAsync(&head, info.counters, sizeof head, hipMemcpyDeviceToHost, sc->poll_stream));
      RT_HIP(hipStreamSynchronize(sc->poll_stream));
      const int due = static_cast<int>(std::min<unsigned long long>(head, info.n_samples) * static_cast<unsigned long long>(n_reports) / info.n_samples);
      while (reported < std::min(due, n_reports - 1)) // the last report is the finished frame's
        progress(static_cast<float>(++reported) / static_cast<float>(n_reports), user);
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    (void)hipGetLastError(); // hipErrorNotReady is not an error
    progress(1.f, user);
  }
  for (int i = 0; i < n_slices && progress && !poll; ++i) {
    RT_HIP(hipEventSynchronize(sc->slice_events[static_cast<size_t>(i) + 1]));
    progress(static_cast<float>(row_end[static_cast<size_t>(i)]) / static_cast<float>(cfg->height), user);
  }
  RT_HIP(hipMemcpy(rgb_out_host, fb, bytes, hipMemcpyDeviceToHost)); // one copy; synchronises the stream
  if (stats) {
    uint64_t rays = 0, capped = 0;
    if (rt_status st = read_tallies(info.counters, &rays, &capped))
      return st;
    float ms = 0.f;
    RT_HIP(hipEventElapsedTime(&ms, sc->slice_events[0], sc->slice_events[static_cast<size_t>(n_slices)]));
    std::memset(stats, 0, sizeof *stats);
    stats->rays = rays, stats->capped_samples = capped, stats->kernel_ms = ms;
    stats->samples = static_cast<uint64_t>(cfg->width) * cfg->height * cfg->spp_edge * cfg->spp_edge;
    stats->grid = info.grid, stats->block = info.block;
    std::memcpy(stats->kernel, info.kernel, sizeof stats->kernel);
  }
  return RT_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// One frame on several GPUs of one node: rows dealt round-robin, one scene replica, stream and band
// buffer per device, gather to the first device with RCCL point-to-point calls (xGMI is point to point: every peer
// sends over its own link, nothing is gained from a ring at <= 12 MB per peer).
// ---------------------------------------------------------------------------------------------
namespace {

// RCCL is loaded on first use (single-GPU callers never pay for it, and a process that already holds a copy — PyTorch
// bundles one under the same soname — keeps using that copy).
struct Rccl {
  using Comm = void *;
  int (*CommInitAll)(Comm *, int, const int *) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  static constexpr int Float32 = 7; // ncclFloat32 (rccl.h)
  bool ok = false;
  std::string error;
};
Rccl &rccl() {
  static Rccl r = [] {
    Rccl x;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
        break;
    if (!h) {
      x.error = std::string("librccl.so.1 not found: ") + dlerror();
      return x;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(h, n);
      if (!p)
        x.error = std::string("RCCL symbol missing: ") + n;
      return p;
    };
    x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
    x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    x.ok = x.error.empty();
    return x;
  }();
  return r;
}
#define RT_NCCL(call)                                                                                      \
  do {                                                                                                     \
    int e_ = (call);                                                                                       \
    if (e_ != 0)                                                                                           \
      return fail(RT_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(e_));                     \
  } while (0)

// rows of one rank's compact band buffer -> their places in the row-major frame (image row = local row * n_ranks + rank)
__global__ void __launch_bounds__(256) k_place_rows(const float4 *band, float4 *frame, int n_local_rows, int row_quads, int rank, int n_ranks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)n_local_rows * row_quads;
  if (i >= n)
    return;
  const size_t lr = i / row_quads, q = i - lr * row_quads;
  frame[(lr * n_ranks + rank) * row_quads + q] = band[i];
}
__global__ void __launch_bounds__(256) k_place_rows_scalar(const float *band, float *frame, int n_local_rows, int row_floats, int rank, int n_ranks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)n_local_rows * row_floats;
  if (i >= n)
    return;
  const size_t lr = i / row_floats, q = i - lr * row_floats;
  frame[(lr * n_ranks + rank) * row_floats + q] = band[i];
}

} // namespace

struct rt_multi {
  std::vector<int> devices;             // rank r renders on devices[r]; rank 0's device assembles the frame
  std::vector<rt_scene *> scenes;       // one replica of the World per rank
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> begun, done;  // per rank: around its render
  std::vector<hipStream_t> poll_streams; // per rank: small copies of the sample-queue head while its kernel runs (progress)
  hipEvent_t gather_begin = nullptr, gather_end = nullptr; // on devices[0]'s stream: around transfer + placement
  std::vector<DeviceBuffer> bands;      // per rank, on its device: its rows, compact
  std::vector<DeviceBuffer> landing;    // per rank >= 1, on devices[0]: where its band arrives
  DeviceBuffer frame;                   // on devices[0]: rt_multi_render_host's frame
  std::vector<Rccl::Comm> comms;        // empty: peer copies instead of RCCL
  bool self_send = false;               // RT_MULTI_RCCL_SELF: rank 0's own band travels through RCCL too
};

extern "C" {

void rt_multi_destroy(rt_multi *m) {
  if (!m)
    return;
  for (Rccl::Comm c : m->comms)
    if (c)
      (void)rccl().CommDestroy(c);
  for (size_t r = 0; r < m->devices.size(); ++r) {
    DeviceGuard guard(m->devices[r]);
    if (r < m->streams.size() && m->streams[r]) (void)hipStreamDestroy(m->streams[r]);
    if (r < m->begun.size() && m->begun[r]) (void)hipEventDestroy(m->begun[r]);
    if (r < m->done.size() && m->done[r]) (void)hipEventDestroy(m->done[r]);
    if (r < m->poll_streams.size() && m->poll_streams[r]) (void)hipStreamDestroy(m->poll_streams[r]);
    if (r < m->bands.size() && m->bands[r].ptr) (void)hipFree(m->bands[r].ptr);
    if (r < m->scenes.size()) rt_scene_destroy(m->scenes[r]);
  }
  if (!m->devices.empty()) {
    DeviceGuard guard(m->devices[0]);
    for (DeviceBuffer &b : m->landing)
      if (b.ptr) (void)hipFree(b.ptr);
    if (m->frame.ptr) (void)hipFree(m->frame.ptr);
    if (m->gather_begin) (void)hipEventDestroy(m->gather_begin);
    if (m->gather_end) (void)hipEventDestroy(m->gather_end);
  }
  delete m;
}

rt_status rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, uint32_t flags, rt_multi **out) {
  if (!desc || !out || n_devices < 0 || (n_devices > 0 && !devices))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: NULL argument");
  *out = nullptr;
  if (flags & ~(RT_MULTI_PEER_COPIES | RT_MULTI_RCCL_SELF))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: unknown flag");
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible == 0)
    return fail(RT_ERR_NO_DEVICE, "rt_multi_create: no HIP device visible");
  rt_multi *m = new (std::nothrow) rt_multi();
  if (!m)
    return fail(RT_ERR_OUT_OF_MEMORY, "rt_multi_create: host allocation failed");
  auto bail = [&](rt_status st) {
    rt_multi_destroy(m);
    return st;
  };
  for (int r = 0; r < n_devices; ++r)
    if (devices[r] < 0 || devices[r] >= visible)
      return bail(fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: device index out of range"));
  if (n_devices == 0)
    for (int d = 0; d < visible; ++d)
      m->devices.push_back(d); // every visible device
  else
    m->devices.assign(devices, devices + n_devices);
  const size_t n = m->devices.size();
  bool distinct = true;
  for (size_t r = 0; r < n; ++r)
    for (size_t q = 0; q < r; ++q)
      distinct = distinct && m->devices[q] != m->devices[r];
  m->scenes.assign(n, nullptr), m->streams.assign(n, nullptr), m->begun.assign(n, nullptr), m->done.assign(n, nullptr);
  m->poll_streams.assign(n, nullptr);
  m->bands.assign(n, DeviceBuffer{}), m->landing.assign(n, DeviceBuffer{});
  for (size_t r = 0; r < n; ++r) {
    if (rt_status st = rt_scene_create(desc, m->devices[r], &m->scenes[r]))
      return bail(st);
    DeviceGuard guard(m->devices[r]);
    if (guard.err != hipSuccess || hipStreamCreateWithFlags(&m->streams[r], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&m->poll_streams[r], hipStreamNonBlocking) != hipSuccess ||
        (r == 0 && (hipEventCreate(&m->gather_begin) != hipSuccess || hipEventCreate(&m->gather_end) != hipSuccess)) ||
        hipEventCreate(&m->begun[r]) != hipSuccess || hipEventCreate(&m->done[r]) != hipSuccess)
      return bail(fail(RT_ERR_HIP, "rt_multi_create: stream / event creation failed"));
  }
  // Gather transport: RCCL when every rank has its own device (a communicator cannot hold one device twice); ranks that
  // share a device (several "virtual ranks" on one GPU: tests, rehearsals) or RT_MULTI_PEER_COPIES use peer copies.
  m->self_send = (flags & RT_MULTI_RCCL_SELF) != 0;
  const bool want_rccl = distinct && !(flags & RT_MULTI_PEER_COPIES) && (n > 1 || m->self_send);
  if (want_rccl && !rccl().ok && !m->self_send) {
    // no RCCL in this process (librccl.so.1 cannot be loaded): the gather falls back to peer copies, which need nothing but
    // HIP; rt_multi_uses_rccl() tells the caller which transport a handle got
    (void)fail(RT_OK, "rt_multi_create: " + rccl().error + " — gathering with peer copies");
  } else if (want_rccl) {
    if (!rccl().ok)
      return bail(fail(RT_ERR_HIP, "rt_multi_create: " + rccl().error));
    m->comms.assign(n, nullptr);
    int e = rccl().CommInitAll(m->comms.data(), static_cast<int>(n), m->devices.data());
    if (e != 0)
      return bail(fail(RT_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(e)));
  }
  *out = m;
  return RT_OK;
}

int rt_multi_rank_count(const rt_multi *m) { return m ? static_cast<int>(m->devices.size()) : 0; }
int rt_multi_uses_rccl(const rt_multi *m) { return m && !m->comms.empty() ? 1 : 0; }

} // extern "C"

namespace {
rt_status multi_render_impl(rt_multi *m, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_device0, rt_stats *stats,
                            rt_multi_timing *timing, int n_reports, rt_progress_fn progress, void *user) {
  if (!m || !cam || !cfg || !rgb_out_device0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render: NULL argument");
  if (rt_status st = check_cfg(cfg))
    return st;
  if (!whole_image(cfg))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render renders whole images (band_rows = n_local_rows = height, band_offset = 0): the rows are dealt to the devices inside");
  const int n = static_cast<int>(m->devices.size());
  const size_t row_floats = static_cast<size_t>(cfg->width) * 3;
  std::vector<int> rows(static_cast<size_t>(n));
  std::vector<LaunchInfo> info(static_cast<size_t>(n));
  // progress between the first and the last report exists in path mode only (its kernels take samples from a queue whose
  // head is counters[0]); the other modes' frames take milliseconds and report once, at the end
  const bool poll = progress && n_reports > 1 && cfg->mode == RT_MODE_PATH;
  // 1. every rank renders its rows (row r, r + n, r + 2n, ...) into its compact band buffer; launches are asynchronous,
  //    so one host thread keeps all devices busy
  for (int r = 0; r < n; ++r) {
    rows[r] = cfg->height > r ? (cfg->height - r + n - 1) / n : 0;
    if (rows[r] == 0)
      continue;
    RT_ON_DEVICE(m->devices[r]);
    if (rt_status st = ensure(m->bands[r], static_cast<size_t>(rows[r]) * row_floats * sizeof(float)))
      return st;
    rt_render_cfg band = *cfg;
    band.band_rows = 1, band.band_stride = n, band.band_offset = r, band.n_local_rows = rows[r];
    Workspace ws;
    ws.info = &info[r];
    if (poll) // the queue head the poll reads must not be the previous frame's
      RT_HIP(hipMemset(m->scenes[r]->counters, 0, sizeof(unsigned long long)));
    RT_HIP(hipEventRecord(m->begun[r], m->streams[r]));
    if (rt_status st = render_impl(m->scenes[r], cam, &band, static_cast<float *>(m->bands[r].ptr), m->streams[r], nullptr, ws))
      return st;
    RT_HIP(hipEventRecord(m->done[r], m->streams[r]));
  }
  // 2. gather to rank 0's device and put the rows in place
  RT_ON_DEVICE(m->devices[0]);
  hipStream_t s0 = m->streams[0];
  const bool use_rccl = !m->comms.empty();
  const int first_moved = use_rccl && m->self_send ? 0 : 1;
  for (int r = first_moved; r < n; ++r)
    if (rows[r])
      if (rt_status st = ensure(m->landing[r], static_cast<size_t>(rows[r]) * row_floats * sizeof(float)))
        return st;
  // device 0's stream waits for every rank's render before the gather starts, so that the two events around the gather
  // time the transfer and the placement alone (the bands are <= 12 MB per peer: nothing is lost by not overlapping them)
  for (int r = 0; r < n; ++r)
    if (rows[r])
      RT_HIP(hipStreamWaitEvent(s0, m->done[r], 0));
  RT_HIP(hipEventRecord(m->gather_begin, s0));
  if (use_rccl) {
    RT_NCCL(rccl().GroupStart());
    int first_error = 0; // a failed call must not leave this thread's RCCL group open (later RCCL calls — PyTorch's too — would queue into it)
    const char *failed = "";
    for (int r = first_moved; r < n && !first_error; ++r) {
      if (!rows[r])
        continue;
      const size_t count = static_cast<size_t>(rows[r]) * row_floats;
      if ((first_error = rccl().Send(m->bands[r].ptr, count, Rccl::Float32, 0, m->comms[r], m->streams[r])) != 0) { // ordered after rank r's render
        failed = "ncclSend";
        break;
      }
      if ((first_error = rccl().Recv(m->landing[r].ptr, count, Rccl::Float32, r, m->comms[0], s0)) != 0)
        failed = "ncclRecv";
    }
    const int end_error = rccl().GroupEnd();
    if (first_error)
      return fail(RT_ERR_HIP, std::string(failed) + ": " + rccl().GetErrorString(first_error));
    if (end_error)
      return fail(RT_ERR_HIP, std::string("ncclGroupEnd: ") + rccl().GetErrorString(end_error));
  } else {
    for (int r = 1; r < n; ++r) {
      if (!rows[r])
        continue;
      RT_HIP(hipStreamWaitEvent(s0, m->done[r], 0));
      RT_HIP(hipMemcpyPeerAsync(m->landing[r].ptr, m->devices[0], m->bands[r].ptr, m->devices[r], static_cast<size_t>(rows[r]) * row_floats * sizeof(float), s0));
    }
  }
  for (int r = 0; r < n; ++r) {
    if (!rows[r])
      continue;
    const float *src = static_cast<const float *>(r >= first_moved ? m->landing[r].ptr : m->bands[r].ptr);
    if (row_floats % 4 == 0 && reinterpret_cast<uintptr_t>(rgb_out_device0) % 16 == 0) {
      const size_t quads = static_cast<size_t>(rows[r]) * (row_floats / 4);
      hipLaunchKernelGGL(k_place_rows, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, s0, reinterpret_cast<const float4 *>(src),
                         reinterpret_cast<float4 *>(rgb_out_device0), rows[r], static_cast<int>(row_floats / 4), r, n);
    } else {
      const size_t count = static_cast<size_t>(rows[r]) * row_floats;
      hipLaunchKernelGGL(k_place_rows_scalar, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, s0, src, rgb_out_device0, rows[r],
                         static_cast<int>(row_floats), r, n);
    }
    RT_HIP(hipGetLastError());
  }
  RT_HIP(hipEventRecord(m->gather_end, s0));
  // 3. progress while the ranks render: the share of the frame's samples their queues have handed out
  if (poll) {
    uint64_t total = 0;
    for (int r = 0; r < n; ++r)
      total += rows[r] ? info[r].n_samples : 0;
    int reported = 0;
    for (;;) {
      bool all_done = true;
      unsigned long long handed = 0;
      for (int r = 0; r < n; ++r) {
        if (!rows[r])
          continue;
        const hipError_t q = hipEventQuery(m->done[r]);
        if (q != hipSuccess && q != hipErrorNotReady)
          return fail(RT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
        if (q == hipErrorNotReady) {
          all_done = false;
          RT_ON_DEVICE(m->devices[r]);
          unsigned long long head = 0;
          RT_HIP(hipMemcpyAsync(&head, info[r].counters, sizeof head, hipMemcpyDeviceToHost, m->poll_streams[r]));
          RT_HIP(hipStreamSynchronize(m->poll_streams[r]));
          handed += std::min<unsigned long long>(head, info[r].n_samples);
        } else {
          handed += info[r].n_samples;
        }
      }
      (void)hipGetLastError(); // hipErrorNotReady is not an error
      if (all_done)
        break;
      const int due = static_cast<int>(handed * static_cast<unsigned long long>(n_reports) / std::max<uint64_t>(total, 1));
      while (reported < std::min(due, n_reports - 1)) // the last report is the finished frame's
        progress(static_cast<float>(++reported) / static_cast<float>(n_reports), user);
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  // 4. the frame is complete when rank 0's stream is idle; the other streams only have their sends left
  for (int r = n - 1; r >= 0; --r) {
    RT_ON_DEVICE(m->devices[r]);
    RT_HIP(hipStreamSynchronize(m->streams[r]));
  }
  if (progress)
    progress(1.f, user);
  // what the gather adds to the frame: transfer + placement, HIP events on device 0's stream around them (the stream waits
  // for every rank's render first)
  float gather_ms = 0.f;
  {
    RT_ON_DEVICE(m->devices[0]);
    RT_HIP(hipEventElapsedTime(&gather_ms, m->gather_begin, m->gather_end));
  }
  if (stats || timing) {
    if (stats)
      std::memset(stats, 0, sizeof *stats);
    if (timing)
      std::memset(timing, 0, sizeof *timing);
    float slowest = 0.f;
    uint64_t rays_total = 0, capped_total = 0, samples_total = 0;
    for (int r = 0; r < n; ++r) {
      if (!rows[r])
        continue;
      RT_ON_DEVICE(m->devices[r]);
      uint64_t rays = 0, capped = 0;
      if (rt_status st = read_tallies(info[r].counters, &rays, &capped))
        return st;
      float ms = 0.f;
      RT_HIP(hipEventElapsedTime(&ms, m->begun[r], m->done[r]));
      slowest = std::max(slowest, ms);
      rays_total += rays, capped_total += capped, samples_total += info[r].n_samples;
      if (timing && r < RT_MULTI_MAX_TIMED_RANKS)
        timing->render_ms[r] = ms, timing->rays[r] = rays;
    }
    if (stats) {
      stats->rays = rays_total, stats->capped_samples = capped_total, stats->samples = samples_total, stats->kernel_ms = slowest;
      stats->grid = info[0].grid, stats->block = info[0].block;
      std::memcpy(stats->kernel, info[0].kernel, sizeof stats->kernel);
    }
    if (timing)
      timing->n_ranks = n, timing->slowest_render_ms = slowest, timing->gather_ms = gather_ms;
  }
  return RT_OK;
}
} // namespace

extern "C" {

rt_status rt_multi_render(rt_multi *m, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_device0, rt_stats *stats,
                          rt_multi_timing *timing) {
  return multi_render_impl(m, cam, cfg, rgb_out_device0, stats, timing, 0, nullptr, nullptr);
}

rt_status rt_multi_render_frame_host(rt_multi *m, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_host, int n_reports,
                                     rt_progress_fn progress, void *user, rt_stats *stats, rt_multi_timing *timing) {
  if (!m || !cfg || !rgb_out_host)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_frame_host: NULL argument");
  if (rt_status st = check_cfg(cfg))
    return st;
  RT_ON_DEVICE(m->devices[0]);
  const size_t bytes = static_cast<size_t>(cfg->height) * cfg->width * 3 * sizeof(float);
  if (rt_status st = ensure(m->frame, bytes))
    return st;
  if (rt_status st = multi_render_impl(m, cam, cfg, static_cast<float *>(m->frame.ptr), stats, timing, std::clamp(n_reports, 1, 256), progress, user))
    return st;
  RT_HIP(hipMemcpy(rgb_out_host, m->frame.ptr, bytes, hipMemcpyDeviceToHost));
  return RT_OK;
}

rt_status rt_multi_render_host(rt_multi *m, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_host, rt_stats *stats,
                               rt_multi_timing *timing) {
  return rt_multi_render_frame_host(m, cam, cfg, rgb_out_host, 1, nullptr, nullptr, stats, timing);
}

rt_status rt_trace_rays(rt_scene *sc, const rt_ray *rays, size_t n, rt_hit *hits) {
  if (!sc || (n && (!rays || !hits)))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_rays: NULL argument");
  if (n == 0)
    return RT_OK;
  RT_ON_DEVICE(sc->device);
  rt_ray *d_rays = nullptr;
  rt_hit *d_hits = nullptr;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_rays), n * sizeof(rt_ray)));
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_hits), n * sizeof(rt_hit));
  if (e == hipSuccess) e = hipMemcpy(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rtd::k_trace_rays, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, sc->dev, d_rays, n, d_hits);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(hits, d_hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost);
  (void)hipFree(d_rays);
  (void)hipFree(d_hits);
  if (e != hipSuccess)
    return fail(RT_ERR_HIP, std::string("rt_trace_rays: ") + hipGetErrorString(e));
  return RT_OK;
}

rt_status rt_offset_if_visible(rt_scene *sc, const float *viewer, const float *surface, const float *normal, size_t n,
                               uint8_t *visible) {
  if (!sc || (n && (!viewer || !surface || !normal || !visible)))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_offset_if_visible: NULL argument");
  if (n == 0)
    return RT_OK;
  RT_ON_DEVICE(sc->device);
  float *d_in = nullptr;
  uint8_t *d_out = nullptr;
  const size_t fb = 3 * n * sizeof(float);
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_in), 3 * fb));
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_out), n);
  if (e == hipSuccess) e = hipMemcpy(d_in, viewer, fb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_in + 3 * n, surface, fb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_in + 6 * n, normal, fb, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rtd::k_offset_if_visible, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, sc->dev, d_in,
                       d_in + 3 * n, d_in + 6 * n, n, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(visible, d_out, n, hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  if (e != hipSuccess)
    return fail(RT_ERR_HIP, std::string("rt_offset_if_visible: ") + hipGetErrorString(e));
  return RT_OK;
}

rt_status rt_aa_jitter(int device, const uint64_t *sample_index, size_t n, float *u_out, float *v_out) {
  if (n && (!sample_index || !u_out || !v_out))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_aa_jitter: NULL argument");
  if (n == 0)
    return RT_OK;
  RT_ON_DEVICE(device);
  uint64_t *d_k = nullptr;
  float *d_uv = nullptr;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_k), n * sizeof(uint64_t)));
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_uv), 2 * n * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(d_k, sample_index, n * sizeof(uint64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rtd::k_aa_jitter, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, d_k, n, d_uv, d_uv + n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(u_out, d_uv, n * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(v_out, d_uv + n, n * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d_k);
  (void)hipFree(d_uv);
  if (e != hipSuccess)
    return fail(RT_ERR_HIP, std::string("rt_aa_jitter: ") + hipGetErrorString(e));
  return RT_OK;
}

rt_status rt_scatter_rays(int device, uint32_t brdf_kind, uint32_t flags, uint64_t *pcg, const float *in_dir, const float *normal,
                          size_t n, float *dir_out) {
  if (brdf_kind > RT_BRDF_SPECULAR || (flags & ~RT_FLAG_EXACT_ARITHMETIC) || (n && (!pcg || !in_dir || !normal || !dir_out)))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_scatter_rays: bad argument");
  if (n == 0)
    return RT_OK;
  RT_ON_DEVICE(device);
  uint64_t *d_pcg = nullptr;
  float *d_f = nullptr;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_pcg), 2 * n * sizeof(uint64_t)));
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_f), 9 * n * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(d_pcg, pcg, 2 * n * sizeof(uint64_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_f, in_dir, 3 * n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_f + 3 * n, normal, 3 * n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const dim3 grid(static_cast<unsigned>((n + 255) / 256));
    if (flags & RT_FLAG_EXACT_ARITHMETIC)
      hipLaunchKernelGGL(rtd::k_scatter_rays<false>, grid, dim3(256), 0, nullptr, brdf_kind, d_pcg, d_f, d_f + 3 * n, n, d_f + 6 * n);
    else
      hipLaunchKernelGGL(rtd::k_scatter_rays<true>, grid, dim3(256), 0, nullptr, brdf_kind, d_pcg, d_f, d_f + 3 * n, n, d_f + 6 * n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(dir_out, d_f + 6 * n, 3 * n * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(pcg, d_pcg, 2 * n * sizeof(uint64_t), hipMemcpyDeviceToHost);
  (void)hipFree(d_pcg);
  (void)hipFree(d_f);
  if (e != hipSuccess)
    return fail(RT_ERR_HIP, std::string("rt_scatter_rays: ") + hipGetErrorString(e));
  return RT_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// Tone-mapping tail (rt_tonemap.h)
// ---------------------------------------------------------------------------------------------
namespace {

rt_status device_cu_count(int device, int *n_cus) {
  hipDeviceProp_t prop;
  RT_HIP(hipGetDeviceProperties(&prop, device));
  *n_cus = prop.multiProcessorCount;
  return RT_OK;
}

// cumsum of HdrImage::average_luminosity (colors.hpp:470-473), then the reference's own last line on the host
rt_status average_luminosity(int device, const float *rgb, size_t n_pixels, float delta, uint32_t flags, hipStream_t stream, float *avg) {
  if (!rgb || !avg || n_pixels == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_average_luminosity: NULL argument or empty image");
  if (flags & ~RT_TONEMAP_SERIAL_SUM)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: unknown flag");
  if (reinterpret_cast<uintptr_t>(rgb) % 16)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: the image must be 16-byte aligned");
  RT_ON_DEVICE(device);
  float cumsum = 0.f;
  const unsigned blocks = static_cast<unsigned>((n_pixels + 255) / 256);
  if (flags & RT_TONEMAP_SERIAL_SUM) {
    float *terms = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&terms), (n_pixels + 1) * sizeof(float)));
    hipLaunchKernelGGL(rtd::k_log_luminosity_terms, dim3(blocks), dim3(256), 0, stream, rgb, n_pixels, delta, terms);
    hipLaunchKernelGGL(rtd::k_serial_sum, dim3(1), dim3(64), 0, stream, terms, n_pixels, terms + n_pixels);
    hipError_t e = hipMemcpyAsync(&cumsum, terms + n_pixels, sizeof(float), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(stream);
    (void)hipFree(terms);
    if (e != hipSuccess)
      return fail(RT_ERR_HIP, std::string("rt_average_luminosity: ") + hipGetErrorString(e));
  } else {
    int n_cus = 0;
    if (rt_status st = device_cu_count(device, &n_cus))
      return st;
    unsigned long long *sum = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&sum), sizeof(unsigned long long)));
    long long fixed = 0;
    hipError_t e = hipMemsetAsync(sum, 0, sizeof(unsigned long long), stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(rtd::k_log_luminosity_sum, dim3(std::min<unsigned>(blocks, 4u * n_cus)), dim3(256), 0, stream, rgb, n_pixels, delta, sum);
      e = hipMemcpyAsync(&fixed, sum, sizeof(fixed), hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess)
      e = hipStreamSynchronize(stream);
    (void)hipFree(sum);
    if (e != hipSuccess)
      return fail(RT_ERR_HIP, std::string("rt_average_luminosity: ") + hipGetErrorString(e));
    cumsum = static_cast<float>(static_cast<double>(fixed) * (1.0 / 4294967296.0));
  }
  *avg = std::pow(10.f, cumsum / n_pixels); // colors.hpp:475, as written (float / size_t)
  return RT_OK;
}

} // namespace

rt_status rt_average_luminosity(int device, const float *rgb_device, size_t n_pixels, float delta, uint32_t flags, void *stream,
                                float *avg_out) {
  return average_luminosity(device, rgb_device, n_pixels, delta, flags, static_cast<hipStream_t>(stream), avg_out);
}

rt_status rt_tonemap(int device, const float *rgb_device, size_t n_pixels, const rt_tonemap_cfg *cfg, uint8_t *ldr_out_device,
                     float *hdr_out_device, void *stream_, float *avg_used_out) {
  if (!rgb_device || !cfg || n_pixels == 0 || (!ldr_out_device && !hdr_out_device))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: NULL argument or empty image");
  if (!(cfg->gamma > 0.f))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: gamma must be positive");
  if (reinterpret_cast<uintptr_t>(rgb_device) % 16 || reinterpret_cast<uintptr_t>(hdr_out_device) % 16 ||
      reinterpret_cast<uintptr_t>(ldr_out_device) % 4)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: float images must be 16-byte aligned, the 8-bit image 4-byte aligned");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float avg = cfg->avg_luminosity;
  if (!cfg->has_avg_luminosity)
    if (rt_status st = average_luminosity(device, rgb_device, n_pixels, cfg->delta, cfg->flags, stream, &avg))
      return st;
  RT_ON_DEVICE(device);
  const float scale = cfg->alpha / avg; // colors.hpp:485
  const size_t n_comp = 3 * n_pixels, quads = (n_comp + 3) / 4;
  hipLaunchKernelGGL(rtd::k_tonemap, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, stream, rgb_device, n_comp, scale,
                     1.f / cfg->gamma, ldr_out_device, hdr_out_device);
  RT_HIP(hipGetLastError());
  if (avg_used_out)
    *avg_used_out = avg;
  return RT_OK;
}

rt_status rt_tonemap_host(int device, const float *rgb_host, size_t n_pixels, const rt_tonemap_cfg *cfg, uint8_t *ldr_out_host,
                          float *avg_used_out) {
  if (!rgb_host || !cfg || !ldr_out_host || n_pixels == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap_host: NULL argument or empty image");
  RT_ON_DEVICE(device);
  float *rgb = nullptr;
  uint8_t *ldr = nullptr;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&rgb), 3 * n_pixels * sizeof(float)));
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&ldr), 3 * n_pixels + 4);
  rt_status st = RT_OK;
  if (e == hipSuccess)
    e = hipMemcpy(rgb, rgb_host, 3 * n_pixels * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    st = rt_tonemap(device, rgb, n_pixels, cfg, ldr, nullptr, nullptr, avg_used_out);
    if (st == RT_OK)
      e = hipMemcpy(ldr_out_host, ldr, 3 * n_pixels, hipMemcpyDeviceToHost);
  }
  (void)hipFree(rgb);
  (void)hipFree(ldr);
  if (e != hipSuccess)
    return fail(RT_ERR_HIP, std::string("rt_tonemap_host: ") + hipGetErrorString(e));
  return st;
}
