// rt_api.hip — the C-ABI of include/raytracer_amd.h: scene upload, launches, parity entry points.
// Host code only orchestrates; all arithmetic of the tracing loop is in rt_device.h / rt_kernels.h.
#include "rt_internal.h"
#include "rt_kernels.h"
#include "rt_wavepath.h"
#include "rt_bvh.h"

using namespace rt_internal;

namespace {
thread_local std::string g_last_error;
} // namespace

namespace rt_internal {

rt_status fail(rt_status st, const std::string &msg) {
  g_last_error = msg;
  return st;
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

rt_status device_cu_count(int device, int *n_cus) {
  hipDeviceProp_t prop;
  RT_HIP(hipGetDeviceProperties(&prop, device));
  *n_cus = prop.multiProcessorCount;
  return RT_OK;
}

} // namespace rt_internal

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

// Selectors of the instantiated k_render_path_wide<ISO, K, W> shapes (rt_wide_iso.hip / rt_wide_gen.hip;
// separate translation units so that the ~30 kernels compile in parallel).
const void *rt_wide_kernel_iso(int k, int waves);
const void *rt_wide_kernel_gen(int k, int waves);
const void *rt_tile_kernel(int waves, bool csg); // rt_wide_tile.hip: k_render_path_tile<S, W, CSG>
int rt_tile_s();                                // its S: sphere tiles per wave and pass
const void *rt_tile_cull_kernel(bool csg);      // rt_wide_tile.hip: k_render_path_tile_cull<4, CSG>
// From how many register spheres on the large-world kernels replace the per-lane scan of the small-world kernel.  Measured
// with the tile kernel (synthetic worlds, 960x540 AA=4, frame ms tile | per-lane): 16 spheres 17.2 | 9.8, 24: 20.5 | 19.2,
// 32: 26.6 | 27.5, 48: 26.2 | 27.5, 64: 25.7 | 26.6 — one full tile of 32 is the break-even.
#ifndef RT_WIDE_MIN_SPHERES
#define RT_WIDE_MIN_SPHERES 32
#endif
#ifndef RT_WAVE_OCC_NO_CSG
#define RT_WAVE_OCC_NO_CSG 6
#endif

extern "C" {

const char *rt_last_error(void) { return g_last_error.c_str(); }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
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

constexpr size_t WS_COUNTER_BYTES = 1024; // >= rtd::COUNTER_WORDS * 8, keeps what follows 256-byte aligned
static_assert(WS_COUNTER_BYTES >= rtd::COUNTER_WORDS * sizeof(unsigned long long), "counter area too small");

} // namespace

namespace rt_internal {

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
#ifdef RT_TUNING // the occupancies DESIGN.md 4.2 measured as dominated are compiled into A/B builds only
      else if (csg)
        path_kernel = occ == 4 ? RT_WAVE_K(true, 4, true) : occ == 6 ? RT_WAVE_K(true, 6, true) : RT_WAVE_K(true, 5, true);
      else
        path_kernel = occ == 4 ? RT_WAVE_K(true, 4, false) : occ == 5 ? RT_WAVE_K(true, 5, false) : RT_WAVE_K(true, 6, false);
#else
      else if (csg)
        path_kernel = RT_WAVE_K(true, 5, true);
      else
        path_kernel = RT_WAVE_K(true, RT_WAVE_OCC_NO_CSG, false);
      (void)occ;
#endif
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
        // Worlds of many tiles: the spheres as spatial clusters, a tile multiplied only with the rays that meet its bound
        // (rt_tile_cull.h) — the same keys, hence the same frame; RT_FLAG_SCAN_ALL_TILES keeps every ray on every tile.
        if (sc->dev.n_cull_tiles > 0 && !(cfg->flags & RT_FLAG_SCAN_ALL_TILES) && rt_tile_cull_kernel(false)) {
          waves = 4;
          block = 64u * static_cast<unsigned>(waves);
          path_kernel = rt_tile_cull_kernel(sc->dev.n_csg_tops > 0);
          stacks_per_block = waves;
          std::snprintf(kernel_name, sizeof kernel_name, "k_render_path_tile_cull<W=%d,BF16x3>", waves);
        }
      }
    }
    int per_cu = 0;
    RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, path_kernel, static_cast<int>(block), dyn_lds));
    per_cu = std::max(per_cu, 1);
    const uint64_t want = (p.n_samples + block - 1) / block;
    grid = static_cast<unsigned>(std::min<uint64_t>(want, static_cast<uint64_t>(sc->n_cus) * per_cu));
    // one node stack per wave: at most 64 nodes per depth level (rt_path.h: the stack discipline)
    stacks_per_block = static_cast<int>(block / 64);
    // Samples a wave reserves from the call's queue per atomic (RenderParams::sample_chunk).  A wave finishes alone what it has
    // reserved.  Until round 3 a reservation was 128 samples whatever the launch; now large worlds reserve exactly what the idle
    // lanes ask for and small worlds 1/32 of a wave's share of the launch, at most 128 (one address takes about 75 M atomics a
    // second: C3 makes 28 M).  Measured (scripts/ab_chunk.sh): little — N=4096 480x136 3606 -> 3553 ms, C3 11.9 -> 11.8 ms — the
    // tail of a launch is set by fresh_max below.
    {
      const uint64_t n_waves = static_cast<uint64_t>(grid) * (block / 64);
#ifdef RT_TUNING
      static const int div_env = [] { const char *e = std::getenv("RT_CHUNK_DIV"); return e && *e ? std::atoi(e) : 32; }();
      static const int large_env = [] { const char *e = std::getenv("RT_CHUNK_LARGE"); return e && *e ? std::atoi(e) : 1; }();
#else
      constexpr int div_env = 32, large_env = 1;
#endif
      const uint64_t share = p.n_samples / std::max<uint64_t>(n_waves * static_cast<uint64_t>(div_env), 1);
      p.sample_chunk = static_cast<int32_t>(std::clamp<uint64_t>(share, 1, 128));
      if (use_wide || use_bvh)
        p.sample_chunk = std::min(p.sample_chunk, large_env);
      // Fresh samples per iteration (RenderParams::fresh_max).  A wave tops up only when its whole stack holds fewer children than
      // it has lanes; what it takes then it finishes as one batch.  With 64 at a time a batch of the N=4096 scene is 770 000 rays
      // (0.9 s), and the batch a wave is in when the queue runs dry is the launch's tail: 0.25 s per launch measured
      // (scripts/ab_slices.sh).  Large worlds take 4 at a time (N=4096 480x136: 3497 -> 3355 ms, N=512 unchanged; the few
      // under-filled iterations after a top-up vanish against 190 iterations per sample); small worlds, whose samples are a
      // dozen iterations long, lose with any cap (C4: 36.9 ms at 64, 38.5 at 16, 47 at 4) and keep 64.  Shrinking the cap only as the
      // queue runs dry (guided self-scheduling: remaining samples / (waves x 1 ... 8)) was built and measured too: C3 / C4 12.0 /
      // 37.4 ms against 11.8 / 36.9 — not kept.
      p.fresh_max = use_wide ? 4 : 64;
#ifdef RT_TUNING
      static const int fresh_env = [] { const char *e = std::getenv("RT_FRESH_MAX"); return e && *e ? std::atoi(e) : 0; }();
      if (fresh_env > 0)
        p.fresh_max = std::min(fresh_env, 64);
#endif
    }
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
    if (c[7] && c[13])
      std::fprintf(stderr, "[rt cull] per wave iteration: rays=%.1f queue=%.0f level1(+wait)=%.0f level2=%.0f finish+shade=%.0f ticks; clusters=%.1f blocks=%.2f pairs=%.1f -> clusters met per ray=%.2f\n",
                   double(c[8]) / double(c[7]), double(c[4]) / double(c[7]), double(c[5]) / double(c[7]), double(c[6]) / double(c[7]), double(c[11]) / double(c[7]),
                   double(c[13]) / double(c[7]), double(c[10]) / double(c[7]), double(c[12]) / double(c[7]), double(c[12]) / double(std::max<unsigned long long>(c[8], 1)));
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

} // namespace rt_internal

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

namespace rt_internal {

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

} // namespace rt_internal

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

extern "C" {

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

