// gpu_image_tracer.hpp — the drop-in for the reference's ImageTracer + Renderer pair, in C++23.
//
// Reference seam (src/raytracer.cpp:140-153):
//
//     ImageTracer tracer{std::make_unique<HdrImage>(w, h), scene.camera, spp_edge};
//     std::unique_ptr<Renderer> renderer = build_renderer(mode_cfg, scene.world);
//     tracer.fire_all_rays([&renderer](const Ray &ray) { return (*renderer)(ray); }, show_progress);
//
// With this header (the reference's headers must already be included by the translation unit):
//
//     raytracer_amd::GpuImageTracer tracer{std::make_unique<HdrImage>(w, h), scene.camera, spp_edge};
//     raytracer_amd::GpuRenderMode mode = raytracer_amd::GpuRenderMode::path(n_rays, roulette, max_depth, seq);
//     tracer.fire_all_rays(scene.world, mode, show_progress);
//
// Same members as ImageTracer (image, camera, samples_per_pixel_edge; include/cameras.hpp:166-187) and the same
// post-condition: tracer.image->pixels filled row-major, row 0 = top; progress reported with values in (0, 1].
// Everything before (scene parsing) and after (write_pfm, tone mapping, PNG) stays the reference's own code.
// Errors surface as std::runtime_error carrying rt_last_error(), in line with how main() reports exceptions
// (src/raytracer.cpp:132-137).
#pragma once

#include "flatten.hpp"
#include "raytracer_amd.h"

#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace raytracer_amd {

/// ModeCfg + PathCfg (src/raytracer.cpp:40-51) and the constants build_renderer passes (src/raytracer.cpp:337-352).
struct GpuRenderMode {
  uint32_t mode = RT_MODE_FLAT;
  int n_rays = 10, russian_roulette_lim = 3, max_depth = 5;
  uint64_t seq_number = 54;
  Color background = BLACK;
  Color ambient = DARK_GREY;
  int max_mirror_bounces = 64; // the reference's pointlight mirror loop has no cap and can spin forever
  uint32_t flags = 0;          // RT_FLAG_* of include/raytracer_amd.h (e.g. RT_FLAG_ACCEL_BVH)

  static GpuRenderMode onoff() { return GpuRenderMode{RT_MODE_ONOFF}; }
  static GpuRenderMode flat() { return GpuRenderMode{RT_MODE_FLAT}; }
  static GpuRenderMode pointlight() { return GpuRenderMode{RT_MODE_POINTLIGHT}; }
  static GpuRenderMode path(int n_rays, int roulette, int max_depth, uint64_t seq_number) {
    GpuRenderMode m{RT_MODE_PATH};
    m.n_rays = n_rays, m.russian_roulette_lim = roulette, m.max_depth = max_depth, m.seq_number = seq_number;
    return m;
  }
};

class GpuImageTracer {
public:
  std::unique_ptr<HdrImage> image;
  std::shared_ptr<Camera> camera;
  int samples_per_pixel_edge;
  int device = 0;            ///< the GPU of a single-device render
  /// Opt in: deal the frame's rows to every visible GPU (rt_multi_*, gather over RCCL, or peer copies where RCCL cannot be
  /// loaded).  Off by default: that transport is covered by virtual ranks and a one-rank RCCL self-send on one-GPU boxes only
  /// (DESIGN.md 8), and frames of a few milliseconds (the deterministic modes) gain nothing from a gather.
  bool all_devices = false;
  /// Explicit ranks instead of "every visible GPU" (wins over all_devices when not empty).  A device may be listed more than
  /// once (several ranks on one GPU, bands moved by device copies): that is how the one-GPU tests drive this route.
  std::vector<int> rank_devices;
  rt_stats last_stats{};
  rt_multi_timing last_timing{}; ///< multi-GPU renders: per-rank kernel time and the gather

  using ProgressCallback = std::function<void(float)>;

  GpuImageTracer(std::unique_ptr<HdrImage> image, std::shared_ptr<Camera> camera, int samples_per_pixel_edge = 1)
      : image{std::move(image)}, camera{std::move(camera)}, samples_per_pixel_edge{samples_per_pixel_edge} {
    // ImageTracer's constructor fills the aspect ratio from the image (include/cameras.hpp:184-186)
    if (!this->camera->asp_ratio.has_value())
      this->camera->asp_ratio.emplace(static_cast<float>(this->image->width) / static_cast<float>(this->image->height));
  }
  GpuImageTracer(const GpuImageTracer &) = delete;
  GpuImageTracer &operator=(const GpuImageTracer &) = delete;
  ~GpuImageTracer() { release(); }

  /// Replaces ImageTracer::fire_all_rays (include/cameras.hpp:221-251) + Renderer::operator().
  ///
  /// The flattened World stays resident in HBM between calls: it is uploaded again only when the World's flattened
  /// arrays change (an animation changes the camera, not the World), and the frame lives in a device buffer the
  /// library reuses.  One GPU: the image is rendered as `progress_slices` horizontal slices queued back to back, the
  /// callback is invoked with rows_done / height as each finishes (the reference reports once per column), and one
  /// device-to-host copy fills HdrImage::pixels.  Several GPUs (all_devices): the rows are dealt round-robin to all of them
  /// and gathered to the first; progress is the share of the frame's samples the devices' queues have handed out.
  void fire_all_rays(const World &world, const GpuRenderMode &mode, ProgressCallback report_progress = [](float) {},
                     int progress_slices = 8) {
    FlatScene flat = flatten_world(world);
    const bool want_multi = !rank_devices.empty() || (all_devices && rt_device_count() > 1);
    if (!(scene_ || multi_) || want_multi != (multi_ != nullptr) || rank_devices != resident_ranks_ || !flat.same_bits(resident_)) {
      release();
      rt_scene_desc desc = flat.desc();
      if (want_multi)
        check(rt_multi_create(&desc, rank_devices.empty() ? nullptr : rank_devices.data(), static_cast<int>(rank_devices.size()), 0, &multi_));
      else
        check(rt_scene_create(&desc, device, &scene_));
      resident_ = std::move(flat);
      resident_ranks_ = rank_devices;
    }
    const rt_camera cam = flatten_camera(*camera);
    rt_render_cfg cfg{};
    cfg.width = image->width, cfg.height = image->height, cfg.spp_edge = samples_per_pixel_edge;
    cfg.mode = mode.mode;
    cfg.n_rays = mode.n_rays, cfg.roulette = mode.russian_roulette_lim, cfg.max_depth = mode.max_depth;
    cfg.seq_number = mode.seq_number;
    cfg.background[0] = mode.background.r, cfg.background[1] = mode.background.g, cfg.background[2] = mode.background.b;
    cfg.ambient[0] = mode.ambient.r, cfg.ambient[1] = mode.ambient.g, cfg.ambient[2] = mode.ambient.b;
    cfg.max_mirror_bounces = mode.max_mirror_bounces;
    cfg.flags = mode.flags;
    cfg.rng_mode = RT_RNG_PER_NODE_STREAMS;
    cfg.band_rows = cfg.band_stride = cfg.n_local_rows = image->height, cfg.band_offset = 0; // the whole image
    static_assert(sizeof(Color) == 3 * sizeof(float), "HdrImage::pixels must be packed rgb floats");
    float *pixels = reinterpret_cast<float *>(image->pixels.data());
    last_stats = rt_stats{};
    last_timing = rt_multi_timing{};
    if (multi_) {
      auto trampoline = [](float fraction, void *user) { (*static_cast<ProgressCallback *>(user))(fraction); };
      check(rt_multi_render_frame_host(multi_, &cam, &cfg, pixels, progress_slices, trampoline, &report_progress, &last_stats, &last_timing));
    } else {
      auto trampoline = [](float fraction, void *user) { (*static_cast<ProgressCallback *>(user))(fraction); };
      check(rt_render_frame_host(scene_, &cam, &cfg, pixels, progress_slices, trampoline, &report_progress, &last_stats));
    }
  }

  /// Frees the resident World (the destructor does it too).
  void release() {
    if (scene_)
      rt_scene_destroy(scene_);
    if (multi_)
      rt_multi_destroy(multi_);
    scene_ = nullptr, multi_ = nullptr;
  }

private:
  rt_scene *scene_ = nullptr;
  rt_multi *multi_ = nullptr;
  FlatScene resident_; // what the handle above was created from
  std::vector<int> resident_ranks_;

  static void check(rt_status rc) {
    if (rc != RT_OK)
      throw std::runtime_error(std::string("raytracer_amd: ") + rt_last_error());
  }
};

} // namespace raytracer_amd
