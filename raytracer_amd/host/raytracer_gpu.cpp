// raytracer_gpu.cpp — the reference's `raytracer render` pipeline (src/raytracer.cpp:112-157) with the GPU
// backend at the seam: reference parser -> GpuImageTracer::fire_all_rays -> reference HdrImage::write_pfm.
// Compiled against the UNMODIFIED reference headers (oracle/Makefile, output oracle/_ref/raytracer_gpu); it is
// the worked example of INTEGRATION.md and the end-to-end test of the C++ flattener.
//
//   raytracer_gpu <scene.txt> -o out.pfm [--mode onoff|flat|pointlight|path] [--width W] [--height H]
//                 [--aa N] [--n_rays N] [--roulette N] [--max-depth N] [--seq N] [--bvh] [--define name=value]...
//                 [--repeat N]   render the frame N more times on the resident World and report the best end-to-end time
//                 [--all-gpus]   spread the frame's rows over every visible GPU (rt_multi_*; opt-in, GpuImageTracer::all_devices)
//                 [--ranks a,b,..] explicit ranks for the multi-GPU route (a device may repeat: the one-GPU test box lists 0,0,0)
//                 [--progress]   print the callback's fractions to stderr
//                 [--one-gpu]    accepted and ignored: one GPU is the default
#include "colors.hpp"
#include "geometry.hpp"
#include "renderers.hpp"
#include "scenefiles.hpp"
#include "shapes.hpp"

#include "gpu_image_tracer.hpp"

#include <chrono>
#include <fstream>
#include <iostream>
#include <sstream>

int main(int argc, char **argv) {
  if (argc < 2) {
    std::cerr << "usage: raytracer_gpu <scene.txt> -o out.pfm [--mode M] [--width W] [--height H] [--aa N] ...\n";
    return EXIT_FAILURE;
  }
  std::string scene_file = argv[1], out, mode = "flat";
  int width = 1280, height = 960, aa = 1, n_rays = 10, roulette = 3, max_depth = 5;
  uint64_t seq = 54;
  bool bvh = false, all_gpus = false;
  int repeat = 0;
  bool show_progress = false;
  std::vector<int> ranks;
  std::unordered_map<std::string, float> defines;
  for (int i = 2; i < argc; ++i) {
    std::string s = argv[i];
    auto next = [&]() -> std::string {
      if (i + 1 >= argc) {
        std::cerr << "missing value after " << s << '\n';
        std::exit(EXIT_FAILURE);
      }
      return argv[++i];
    };
    if (s == "-o") out = next();
    else if (s == "--mode") mode = next();
    else if (s == "--width") width = std::stoi(next());
    else if (s == "--height") height = std::stoi(next());
    else if (s == "--aa") aa = std::stoi(next());
    else if (s == "--n_rays") n_rays = std::stoi(next());
    else if (s == "--roulette") roulette = std::stoi(next());
    else if (s == "--max-depth") max_depth = std::stoi(next());
    else if (s == "--seq") seq = std::stoull(next());
    else if (s == "--bvh") bvh = true;
    else if (s == "--one-gpu") all_gpus = false;
    else if (s == "--all-gpus") all_gpus = true;
    else if (s == "--repeat") repeat = std::stoi(next());
    else if (s == "--progress") show_progress = true;
    else if (s == "--ranks") {
      std::stringstream list{next()};
      for (std::string item; std::getline(list, item, ',');)
        ranks.push_back(std::stoi(item));
    }
    else if (s == "--define") {
      std::string d = next();
      auto eq = d.find('=');
      if (eq == std::string::npos) {
        std::cerr << "--define wants name=value\n";
        return EXIT_FAILURE;
      }
      defines[d.substr(0, eq)] = std::stof(d.substr(eq + 1));
    } else {
      std::cerr << "unknown option " << s << '\n';
      return EXIT_FAILURE;
    }
  }
  if (out.empty()) {
    std::cerr << "-o out.pfm is required\n";
    return EXIT_FAILURE;
  }

  std::ifstream is(scene_file);
  if (!is) {
    std::cerr << "Error opening input file \"" << scene_file << "\"\n";
    return EXIT_FAILURE;
  }
  InputStream input_stream{is, scene_file};
  Scene scene;
  if (!defines.empty())
    scene.initialize_float_variables_with_priority(std::move(defines));
  try {
    scene.parse_scene(input_stream);

    raytracer_amd::GpuImageTracer tracer{std::make_unique<HdrImage>(width, height), scene.camera, aa};
    raytracer_amd::GpuRenderMode m;
    if (mode == "onoff") m = raytracer_amd::GpuRenderMode::onoff();
    else if (mode == "flat") m = raytracer_amd::GpuRenderMode::flat();
    else if (mode == "pointlight") m = raytracer_amd::GpuRenderMode::pointlight();
    else if (mode == "path") m = raytracer_amd::GpuRenderMode::path(n_rays, roulette, max_depth, seq);
    else throw std::invalid_argument("unknown mode " + mode);
    if (bvh)
      m.flags |= RT_FLAG_ACCEL_BVH;

    tracer.all_devices = all_gpus;
    tracer.rank_devices = ranks;
    auto t0 = std::chrono::steady_clock::now();
    if (show_progress)
      tracer.fire_all_rays(scene.world, m, [](float fraction) { std::cerr << "progress " << fraction << '\n'; });
    else
      tracer.fire_all_rays(scene.world, m);
    auto t1 = std::chrono::steady_clock::now();
    // the first call pays for the HIP runtime, the code object and the upload of the World; later frames (an animation,
    // a resident service) do not: --repeat times fire_all_rays alone, World already in HBM
    double best = 1e30;
    for (int k = 0; k < repeat; ++k) {
      auto r0 = std::chrono::steady_clock::now();
      tracer.fire_all_rays(scene.world, m);
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count());
    }
    std::ofstream of(out, std::ios::binary);
    tracer.image->write_pfm(of);
    std::cout << "{\"elapsed_s\": " << std::chrono::duration<double>(t1 - t0).count() << ", \"rays\": " << tracer.last_stats.rays
              << ", \"kernel_ms\": " << tracer.last_stats.kernel_ms << ", \"gpus\": " << std::max(tracer.last_timing.n_ranks, 1)
              << ", \"kernel\": \"" << tracer.last_stats.kernel << "\"";
    if (repeat > 0)
      std::cout << ", \"resident_frame_ms\": " << best * 1e3;
    std::cout << "}" << std::endl;
  } catch (const std::exception &err) {
    std::cerr << err.what() << '\n';
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
