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
  int device = 0;
  rt_stats last_stats{};

  using ProgressCallback = std::function<void(float)>;

  GpuImageTracer(std::unique_ptr<HdrImage> image, std::shared_ptr<Camera> camera, int samples_per_pixel_edge = 1)
      : image{std::move(image)}, camera{std::move(camera)}, samples_per_pixel_edge{samples_per_pixel_edge} {
    // ImageTracer's constructor fills the aspect ratio from the image (include/cameras.hpp:184-186)
    if (!this->camera->asp_ratio.has_value())
      this->camera->asp_ratio.emplace(static_cast<float>(this->image->width) / static_cast<float>(this->image->height));
  }

  /// Replaces ImageTracer::fire_all_rays (include/cameras.hpp:221-251) + Renderer::operator().
  /// The image is rendered in `progress_bands` horizontal slices so that progress can be reported.
  void fire_all_rays(const World &world, const GpuRenderMode &mode, ProgressCallback report_progress = [](float) {},
                     int progress_bands = 8) {
    FlatScene flat = flatten_world(world);
    rt_scene_desc desc = flat.desc();
    rt_scene *scene = nullptr;
    check(rt_scene_create(&desc, device, &scene));
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
    static_assert(sizeof(Color) == 3 * sizeof(float), "HdrImage::pixels must be packed rgb floats");
    float *pixels = reinterpret_cast<float *>(image->pixels.data());
    last_stats = rt_stats{};
    const int H = image->height;
    const int bands = std::max(1, std::min(progress_bands, H));
    for (int b = 0; b < bands; ++b) {
      const int row0 = static_cast<int>(static_cast<long long>(H) * b / bands);
      const int row1 = static_cast<int>(static_cast<long long>(H) * (b + 1) / bands);
      if (row1 == row0)
        continue;
      cfg.band_rows = row1 - row0, cfg.band_stride = H, cfg.band_offset = row0, cfg.n_local_rows = row1 - row0;
      rt_stats st{};
      rt_status rc = rt_render_host(scene, &cam, &cfg, pixels + static_cast<size_t>(row0) * image->width * 3, &st);
      if (rc != RT_OK) {
        std::string msg = rt_last_error();
        rt_scene_destroy(scene);
        throw std::runtime_error("raytracer_amd: " + msg);
      }
      last_stats.rays += st.rays, last_stats.samples += st.samples, last_stats.capped_samples += st.capped_samples;
      last_stats.kernel_ms += st.kernel_ms;
      report_progress(static_cast<float>(row1) / static_cast<float>(H));
    }
    rt_scene_destroy(scene);
  }

private:
  static void check(rt_status rc) {
    if (rc != RT_OK)
      throw std::runtime_error(std::string("raytracer_amd: ") + rt_last_error());
  }
};

} // namespace raytracer_amd
