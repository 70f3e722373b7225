// ref_harness.cpp — TEST INFRASTRUCTURE.  Drives the UNMODIFIED reference (matteoilardi/Raytracer)
// from its own headers under $(RT_REFERENCE_DIR)/include to produce golden vectors and the
// "reference" CPU baseline.  It is compiled only where /root/reference exists (this container),
// by oracle/Makefile, into oracle/_ref/rt_ref (git-ignored; the binary travels to the GPU box,
// the reference sources do not).  No reference file is modified or copied.
//
// Subcommands
//   render  <scene.txt> -o out.pfm [--mode onoff|flat|pointlight|path] [--width W] [--height H]
//           [--aa N] [--n_rays N] [--roulette N] [--max-depth N] [--seq N] [--cap K]
//           [--define name=value]...
//       The reference's own pipeline: Scene::parse_scene -> ImageTracer -> build_renderer-equivalent
//       -> fire_all_rays (src/raytracer.cpp:128-153), PFM written little-endian.
//       pointlight uses CappedPointLightTracer below when --cap K > 0 (see H1 in SURVEY.md §7:
//       the stock loop at include/renderers.hpp:129-152 never terminates on some pixels).
//   flatten <scene.txt> out.json [--define name=value]...
//       World/Camera -> flattened JSON through raytracer_amd/host/flatten.hpp.
//   trace   <scene.txt> rays.bin hits.bin
//       World::ray_intersection for every ray in rays.bin (n x 8 f32: o, d, tmin, tmax);
//       hits.bin = n x {i32 object, f32 t, point[3], normal[3], uv[2]} (object -1 = miss).
//   visible <scene.txt> in.bin out.bin
//       World::offset_if_visible for n x 9 f32 (viewer, surface, normal) -> n x u8.
//   csg_ts  <scene.txt> rays.bin out.txt
//       Object::all_ray_intersections t-lists of every top-level object, one line per (ray, object).
#include "colors.hpp"
#include "geometry.hpp"
#include "renderers.hpp"
#include "scenefiles.hpp"
#include "shapes.hpp"

#include "../raytracer_amd/host/flatten.hpp"

#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

namespace {

struct Args {
  std::string cmd, scene, out, in2;
  std::string mode = "flat";
  int width = 1280, height = 960, aa = 1, n_rays = 10, roulette = 3, max_depth = 5, cap = 0;
  uint64_t seq = 54;
  std::unordered_map<std::string, float> defines;
};

[[noreturn]] void die(const std::string &msg) {
  std::cerr << "rt_ref: " << msg << std::endl;
  std::exit(2);
}

Args parse_args(int argc, char **argv) {
  Args a;
  if (argc < 3)
    die("usage: rt_ref <render|flatten|trace|visible|csg_ts> <scene.txt> ...");
  a.cmd = argv[1];
  a.scene = argv[2];
  std::vector<std::string> pos;
  for (int i = 3; i < argc; ++i) {
    std::string s = argv[i];
    auto next = [&]() -> std::string {
      if (i + 1 >= argc)
        die("missing value after " + s);
      return argv[++i];
    };
    if (s == "-o") a.out = next();
    else if (s == "--mode") a.mode = next();
    else if (s == "--width") a.width = std::stoi(next());
    else if (s == "--height") a.height = std::stoi(next());
    else if (s == "--aa") a.aa = std::stoi(next());
    else if (s == "--n_rays") a.n_rays = std::stoi(next());
    else if (s == "--roulette") a.roulette = std::stoi(next());
    else if (s == "--max-depth") a.max_depth = std::stoi(next());
    else if (s == "--seq") a.seq = std::stoull(next());
    else if (s == "--cap") a.cap = std::stoi(next());
    else if (s == "--define") {
      std::string d = next();
      auto pos_eq = d.find('=');
      if (pos_eq == std::string::npos)
        die("--define wants name=value");
      a.defines[d.substr(0, pos_eq)] = std::stof(d.substr(pos_eq + 1));
    } else pos.push_back(s);
  }
  if (a.cmd == "flatten") {
    if (pos.size() != 1) die("flatten <scene> <out.json>");
    a.out = pos[0];
  } else if (a.cmd == "trace" || a.cmd == "visible" || a.cmd == "csg_ts") {
    if (pos.size() != 2) die(a.cmd + " <scene> <in> <out>");
    a.in2 = pos[0];
    a.out = pos[1];
  }
  return a;
}

void load_scene(const Args &a, Scene &scene) {
  std::ifstream is(a.scene);
  if (!is)
    die("cannot open scene " + a.scene);
  InputStream input_stream{is, a.scene};
  if (!a.defines.empty()) {
    auto copy = a.defines;
    scene.initialize_float_variables_with_priority(std::move(copy));
  }
  try {
    scene.parse_scene(input_stream);
  } catch (const std::exception &e) {
    die(e.what());
  }
}

/// The reference's PointLightTracer outer loop with a bounce cap, built only from the reference's
/// public World / BRDF / Pigment calls.  After `cap` specular bounces the sample yields
/// background_color * attenuation-so-far ... i.e. exactly what a miss at that point would yield
/// scaled by nothing: the reference returns background_color un-attenuated on a miss
/// (include/renderers.hpp:132-134), so a capped sample returns background_color as well.
class CappedPointLightTracer : public Renderer {
public:
  Color ambient;
  int cap;
  mutable uint64_t capped = 0;
  CappedPointLightTracer(const World &w, Color ambient, Color background, int cap)
      : Renderer{w, background}, ambient{ambient}, cap{cap} {}

  Color operator()(Ray ray) const override {
    Color attenuation = WHITE;
    std::optional<HitRecord> hit;
    const Material *material = nullptr;
    int bounces = 0;
    for (;;) {
      hit = world.ray_intersection(ray);
      if (!hit)
        return background_color;
      material = &hit->shape->material;
      if (!dynamic_cast<SpecularBRDF *>(material->brdf.get()))
        break;
      if (bounces++ >= cap) {
        ++capped;
        return background_color;
      }
      attenuation *= (*material->brdf->pigment)(hit->surface_point);
      Vec reflected = ray.direction - 2 * hit->normal.to_vector() * (hit->normal * ray.direction);
      ray = Ray(hit->world_point, reflected);
    }
    Color radiance = ambient + (*material->emitted_radiance)(hit->surface_point);
    for (const auto &light : world.light_sources) {
      std::optional<Vec> in_dir = world.offset_if_visible(light->point, hit->world_point, hit->normal);
      if (!in_dir)
        continue;
      float distance = in_dir->norm();
      float distance_factor = (light->emission_radius > 0.f) ? std::pow((light->emission_radius / distance), 2) : 1.f;
      float cos_theta = (-1.f / distance) * (*in_dir) * hit->normal.normalized();
      radiance += light->color * distance_factor * cos_theta *
                  material->brdf->eval(hit->normal, *in_dir, -hit->ray.direction, hit->surface_point);
    }
    return radiance * attenuation;
  }
};

int cmd_render(const Args &a) {
  if (a.out.empty())
    die("render needs -o out.pfm");
  Scene scene;
  load_scene(a, scene);
  ImageTracer tracer{std::make_unique<HdrImage>(a.width, a.height), scene.camera, a.aa};

  std::unique_ptr<Renderer> renderer;
  CappedPointLightTracer *capped = nullptr;
  if (a.mode == "onoff")
    renderer = std::make_unique<OnOffTracer>(scene.world);
  else if (a.mode == "flat")
    renderer = std::make_unique<FlatTracer>(scene.world, BLACK);
  else if (a.mode == "pointlight") {
    if (a.cap > 0) {
      auto p = std::make_unique<CappedPointLightTracer>(scene.world, DARK_GREY, BLACK, a.cap);
      capped = p.get();
      renderer = std::move(p);
    } else
      renderer = std::make_unique<PointLightTracer>(scene.world, DARK_GREY, BLACK);
  } else if (a.mode == "path")
    renderer = std::make_unique<PathTracer>(scene.world, std::make_unique<PCG>(42ull, a.seq), a.n_rays, a.roulette,
                                            a.max_depth, BLACK);
  else
    die("unknown mode " + a.mode);

  auto t0 = std::chrono::steady_clock::now();
  tracer.fire_all_rays([&renderer](const Ray &ray) { return (*renderer)(ray); });
  auto t1 = std::chrono::steady_clock::now();
  std::ofstream of(a.out, std::ios::binary);
  tracer.image->write_pfm(of);
  std::cout << "{\"elapsed_s\": " << std::chrono::duration<double>(t1 - t0).count()
            << ", \"capped_samples\": " << (capped ? capped->capped : 0) << "}" << std::endl;
  return 0;
}

int cmd_flatten(const Args &a) {
  Scene scene;
  load_scene(a, scene);
  if (!scene.camera)
    die("scene has no camera");
  auto flat = raytracer_amd::flatten_world(scene.world);
  raytracer_amd::write_json(flat, raytracer_amd::flatten_camera(*scene.camera), a.out);
  return 0;
}

std::vector<float> read_floats(const std::string &path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f)
    die("cannot open " + path);
  auto size = static_cast<size_t>(f.tellg());
  f.seekg(0);
  std::vector<float> v(size / sizeof(float));
  f.read(reinterpret_cast<char *>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(float)));
  return v;
}

void collect_shapes(const Object *obj, int index, std::map<const Shape *, int> &owner) {
  if (auto *csg = dynamic_cast<const CSGObject *>(obj)) {
    collect_shapes(csg->object1.get(), index, owner);
    collect_shapes(csg->object2.get(), index, owner);
  } else if (auto *shape = dynamic_cast<const Shape *>(obj)) {
    owner[shape] = index;
  }
}

int cmd_trace(const Args &a) {
  Scene scene;
  load_scene(a, scene);
  std::map<const Shape *, int> owner;
  for (size_t i = 0; i < scene.world.objects.size(); ++i)
    collect_shapes(scene.world.objects[i].get(), static_cast<int>(i), owner);
  auto rays = read_floats(a.in2);
  const size_t n = rays.size() / 8;
  std::ofstream of(a.out, std::ios::binary);
  for (size_t i = 0; i < n; ++i) {
    const float *r = &rays[i * 8];
    Ray ray{Point{r[0], r[1], r[2]}, Vec{r[3], r[4], r[5]}, r[6], r[7], 0};
    auto hit = scene.world.ray_intersection(ray);
    int32_t object = -1;
    float rec[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hit) {
      object = owner.at(hit->shape);
      rec[0] = hit->t;
      rec[1] = hit->world_point.x, rec[2] = hit->world_point.y, rec[3] = hit->world_point.z;
      rec[4] = hit->normal.x, rec[5] = hit->normal.y, rec[6] = hit->normal.z;
      rec[7] = hit->surface_point.u, rec[8] = hit->surface_point.v;
    }
    of.write(reinterpret_cast<const char *>(&object), 4);
    of.write(reinterpret_cast<const char *>(rec), sizeof rec);
  }
  return 0;
}

int cmd_visible(const Args &a) {
  Scene scene;
  load_scene(a, scene);
  auto in = read_floats(a.in2);
  const size_t n = in.size() / 9;
  std::ofstream of(a.out, std::ios::binary);
  for (size_t i = 0; i < n; ++i) {
    const float *r = &in[i * 9];
    auto v = scene.world.offset_if_visible(Point{r[0], r[1], r[2]}, Point{r[3], r[4], r[5]}, Normal{r[6], r[7], r[8]});
    char c = v.has_value() ? 1 : 0;
    of.write(&c, 1);
  }
  return 0;
}

int cmd_csg_ts(const Args &a) {
  Scene scene;
  load_scene(a, scene);
  auto rays = read_floats(a.in2);
  const size_t n = rays.size() / 8;
  FILE *f = std::fopen(a.out.c_str(), "w");
  if (!f)
    die("cannot open " + a.out);
  for (size_t i = 0; i < n; ++i) {
    const float *r = &rays[i * 8];
    Ray ray{Point{r[0], r[1], r[2]}, Vec{r[3], r[4], r[5]}, r[6], r[7], 0};
    for (size_t o = 0; o < scene.world.objects.size(); ++o) {
      auto hits = scene.world.objects[o]->all_ray_intersections(ray);
      std::fprintf(f, "%zu %zu", i, o);
      for (const auto &h : hits)
        std::fprintf(f, " %.9g", static_cast<double>(h.t));
      std::fprintf(f, "\n");
    }
  }
  std::fclose(f);
  return 0;
}

} // namespace

// tonemap <in.pfm> <out.bin> <alpha> [avg_luminosity]: the tail of main() after a render (src/raytracer.cpp:176-182):
// HdrImage::normalize_image (with average_luminosity unless a value is given) + clamp_image, by the reference's
// own methods.  out.bin = the normalised + clamped pixels (row 0 = top); stdout = the average luminosity used.
int cmd_tonemap(int argc, char **argv) {
  if (argc < 5)
    die("tonemap <in.pfm> <out.bin> <alpha> [avg_luminosity]");
  HdrImage img{std::string(argv[2])};
  const float alpha = std::stof(argv[4]);
  const float avg = argc > 5 ? std::stof(argv[5]) : img.average_luminosity();
  img.normalize_image(alpha, avg);
  img.clamp_image();
  std::ofstream of(argv[3], std::ios::binary);
  of.write(reinterpret_cast<const char *>(img.pixels.data()), static_cast<std::streamsize>(img.pixels.size() * sizeof(Color)));
  std::cout.precision(9);
  std::cout << "{\"avg_luminosity\": " << avg << ", \"width\": " << img.width << ", \"height\": " << img.height << "}" << std::endl;
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "tonemap")
    return cmd_tonemap(argc, argv);
  Args a = parse_args(argc, argv);
  if (a.cmd == "render") return cmd_render(a);
  if (a.cmd == "flatten") return cmd_flatten(a);
  if (a.cmd == "trace") return cmd_trace(a);
  if (a.cmd == "visible") return cmd_visible(a);
  if (a.cmd == "csg_ts") return cmd_csg_ts(a);
  die("unknown subcommand " + a.cmd);
}
