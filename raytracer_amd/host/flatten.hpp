// flatten.hpp — host side of the drop-in boundary, in the reference's own language (C++23).
//
// Walks the reference's parsed object graph and produces the struct-of-arrays `rt_scene_desc`
// that include/raytracer_amd.h takes.  This header is compiled TOGETHER WITH the reference's
// headers (shapes.hpp, materials.hpp, cameras.hpp must already be included by the translation
// unit that includes this file); it needs no edit of any reference file because every member it
// reads is public:
//   World::objects / light_sources              (reference include/shapes.hpp:525-527)
//   Object::transformation                      (include/shapes.hpp:107)
//   Shape::material                             (include/shapes.hpp:140)
//   CSGObject::object1/object2/operation        (include/shapes.hpp:377-381)
//   Material::brdf / emitted_radiance           (include/materials.hpp:272-273)
//   BRDF::pigment                               (include/materials.hpp:147)
//   Transformation::hom_matrix / inverse_hom_matrix (include/geometry.hpp:437-438)
// Dynamic types are discovered with dynamic_cast, as the reference's own parser test does
// (test/scenefiles_test.cpp:291-296).
#pragma once

#include "raytracer_amd.h"

#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <cstring>
#include <vector>

namespace raytracer_amd {

/// Owning storage behind an rt_scene_desc.
struct FlatScene {
  std::vector<uint8_t> shape_kind;
  std::vector<uint32_t> shape_material;
  std::vector<float> shape_fwd, shape_inv;
  std::vector<uint8_t> csg_op;
  std::vector<rt_ref> csg_left, csg_right;
  std::vector<float> csg_fwd, csg_inv;
  std::vector<rt_ref> objects;
  std::vector<uint8_t> material_brdf;
  std::vector<uint32_t> material_brdf_pigment, material_emission_pigment;
  std::vector<uint8_t> pigment_kind;
  std::vector<float> pigment_color1, pigment_color2;
  std::vector<int32_t> pigment_n_intervals;
  std::vector<uint32_t> pigment_tex_offset;
  std::vector<int32_t> pigment_tex_width, pigment_tex_height;
  std::vector<float> texels;
  std::vector<float> light_point, light_color, light_radius;

  rt_scene_desc desc() const {
    rt_scene_desc d{};
    d.n_shapes = static_cast<uint32_t>(shape_kind.size());
    d.shape_kind = shape_kind.data();
    d.shape_material = shape_material.data();
    d.shape_fwd = shape_fwd.data();
    d.shape_inv = shape_inv.data();
    d.n_csg_nodes = static_cast<uint32_t>(csg_op.size());
    d.csg_op = csg_op.data();
    d.csg_left = csg_left.data();
    d.csg_right = csg_right.data();
    d.csg_fwd = csg_fwd.data();
    d.csg_inv = csg_inv.data();
    d.n_objects = static_cast<uint32_t>(objects.size());
    d.objects = objects.data();
    d.n_materials = static_cast<uint32_t>(material_brdf.size());
    d.material_brdf = material_brdf.data();
    d.material_brdf_pigment = material_brdf_pigment.data();
    d.material_emission_pigment = material_emission_pigment.data();
    d.n_pigments = static_cast<uint32_t>(pigment_kind.size());
    d.pigment_kind = pigment_kind.data();
    d.pigment_color1 = pigment_color1.data();
    d.pigment_color2 = pigment_color2.data();
    d.pigment_n_intervals = pigment_n_intervals.data();
    d.pigment_tex_offset = pigment_tex_offset.data();
    d.pigment_tex_width = pigment_tex_width.data();
    d.pigment_tex_height = pigment_tex_height.data();
    d.n_texels = static_cast<uint32_t>(texels.size() / 3);
    d.texels = texels.data();
    d.n_lights = static_cast<uint32_t>(light_radius.size());
    d.light_point = light_point.data();
    d.light_color = light_color.data();
    d.light_radius = light_radius.data();
    return d;
  }

  /// Bit-for-bit equality of two flattened Worlds (the signs of zeros matter to copysign: values are not enough).
  bool same_bits(const FlatScene &o) const {
    auto eq = [](const auto &a, const auto &b) {
      return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) == 0);
    };
    return eq(shape_kind, o.shape_kind) && eq(shape_material, o.shape_material) && eq(shape_fwd, o.shape_fwd) && eq(shape_inv, o.shape_inv) &&
           eq(csg_op, o.csg_op) && eq(csg_left, o.csg_left) && eq(csg_right, o.csg_right) && eq(csg_fwd, o.csg_fwd) && eq(csg_inv, o.csg_inv) &&
           eq(objects, o.objects) && eq(material_brdf, o.material_brdf) && eq(material_brdf_pigment, o.material_brdf_pigment) &&
           eq(material_emission_pigment, o.material_emission_pigment) && eq(pigment_kind, o.pigment_kind) &&
           eq(pigment_color1, o.pigment_color1) && eq(pigment_color2, o.pigment_color2) && eq(pigment_n_intervals, o.pigment_n_intervals) &&
           eq(pigment_tex_offset, o.pigment_tex_offset) && eq(pigment_tex_width, o.pigment_tex_width) &&
           eq(pigment_tex_height, o.pigment_tex_height) && eq(texels, o.texels) && eq(light_point, o.light_point) &&
           eq(light_color, o.light_color) && eq(light_radius, o.light_radius);
  }
};

namespace detail {

inline void push_hom(std::vector<float> &dst, const HomMatrix &m) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      dst.push_back(m.linear_part[i][j]);
  dst.push_back(m.translation_vec.x);
  dst.push_back(m.translation_vec.y);
  dst.push_back(m.translation_vec.z);
}

inline void push_color(std::vector<float> &dst, const Color &c) {
  dst.push_back(c.r);
  dst.push_back(c.g);
  dst.push_back(c.b);
}

struct Flattener {
  FlatScene &out;
  std::map<const Material *, uint32_t> material_ids; // materials are identified by address

  uint32_t add_pigment(const Pigment *p) {
    const uint32_t id = static_cast<uint32_t>(out.pigment_kind.size());
    Color c1{}, c2{};
    int32_t n = 0, w = 0, h = 0;
    uint32_t off = 0;
    uint8_t kind;
    if (auto *u = dynamic_cast<const UniformPigment *>(p)) {
      kind = RT_PIGMENT_UNIFORM;
      c1 = u->color;
    } else if (auto *ck = dynamic_cast<const CheckeredPigment *>(p)) {
      kind = RT_PIGMENT_CHECKERED;
      c1 = ck->color1;
      c2 = ck->color2;
      n = ck->n_intervals;
    } else if (auto *im = dynamic_cast<const ImagePigment *>(p)) {
      kind = RT_PIGMENT_IMAGE;
      off = static_cast<uint32_t>(out.texels.size() / 3);
      w = im->image.width;
      h = im->image.height;
      for (const Color &px : im->image.pixels)
        push_color(out.texels, px);
    } else {
      throw std::invalid_argument("raytracer_amd: unknown Pigment subclass");
    }
    out.pigment_kind.push_back(kind);
    push_color(out.pigment_color1, c1);
    push_color(out.pigment_color2, c2);
    out.pigment_n_intervals.push_back(n);
    out.pigment_tex_offset.push_back(off);
    out.pigment_tex_width.push_back(w);
    out.pigment_tex_height.push_back(h);
    return id;
  }

  uint32_t material_id(const Material &m) {
    auto it = material_ids.find(&m);
    if (it != material_ids.end())
      return it->second;
    const uint32_t id = static_cast<uint32_t>(out.material_brdf.size());
    const BRDF *brdf = m.brdf.get();
    uint8_t kind;
    if (dynamic_cast<const SpecularBRDF *>(brdf))
      kind = RT_BRDF_SPECULAR;
    else if (dynamic_cast<const DiffusiveBRDF *>(brdf))
      kind = RT_BRDF_DIFFUSE;
    else
      throw std::invalid_argument("raytracer_amd: unknown BRDF subclass");
    out.material_brdf.push_back(kind);
    out.material_brdf_pigment.push_back(add_pigment(brdf->pigment.get()));
    out.material_emission_pigment.push_back(add_pigment(m.emitted_radiance.get()));
    material_ids.emplace(&m, id);
    return id;
  }

  rt_ref add_object(const Object *obj) {
    if (auto *csg = dynamic_cast<const CSGObject *>(obj)) {
      const rt_ref left = add_object(csg->object1.get());
      const rt_ref right = add_object(csg->object2.get());
      const int32_t id = static_cast<int32_t>(out.csg_op.size());
      uint8_t op;
      switch (csg->operation) {
      case CSGObject::Operation::UNION: op = RT_CSG_UNION; break;
      case CSGObject::Operation::INTERSECTION: op = RT_CSG_INTERSECTION; break;
      case CSGObject::Operation::DIFFERENCE: op = RT_CSG_DIFFERENCE; break;
      case CSGObject::Operation::FUSION: op = RT_CSG_FUSION; break;
      default: throw std::invalid_argument("raytracer_amd: unknown CSG operation");
      }
      out.csg_op.push_back(op);
      out.csg_left.push_back(left);
      out.csg_right.push_back(right);
      push_hom(out.csg_fwd, csg->transformation.hom_matrix);
      push_hom(out.csg_inv, csg->transformation.inverse_hom_matrix);
      return ~id;
    }
    const auto *shape = dynamic_cast<const Shape *>(obj);
    if (!shape)
      throw std::invalid_argument("raytracer_amd: Object is neither Shape nor CSGObject");
    uint8_t kind;
    if (dynamic_cast<const Sphere *>(obj))
      kind = RT_SHAPE_SPHERE;
    else if (dynamic_cast<const Plane *>(obj))
      kind = RT_SHAPE_PLANE;
    else
      throw std::invalid_argument("raytracer_amd: unknown Shape subclass");
    const int32_t id = static_cast<int32_t>(out.shape_kind.size());
    out.shape_kind.push_back(kind);
    out.shape_material.push_back(material_id(shape->material));
    push_hom(out.shape_fwd, shape->transformation.hom_matrix);
    push_hom(out.shape_inv, shape->transformation.inverse_hom_matrix);
    return id;
  }
};

} // namespace detail

/// World -> FlatScene.  Object order = World::objects insertion order.
inline FlatScene flatten_world(const World &world) {
  FlatScene flat;
  detail::Flattener f{flat, {}};
  for (const auto &obj : world.objects)
    flat.objects.push_back(f.add_object(obj.get()));
  for (const auto &light : world.light_sources) {
    flat.light_point.push_back(light->point.x);
    flat.light_point.push_back(light->point.y);
    flat.light_point.push_back(light->point.z);
    detail::push_color(flat.light_color, light->color);
    flat.light_radius.push_back(light->emission_radius);
  }
  return flat;
}

/// Camera -> rt_camera.  asp_ratio stays "unset" when the scene said exact_asp_ratio and no
/// ImageTracer has filled it yet (include/cameras.hpp:184-186).
inline rt_camera flatten_camera(const Camera &camera) {
  rt_camera c{};
  if (auto *p = dynamic_cast<const PerspectiveCamera *>(&camera)) {
    c.kind = RT_CAMERA_PERSPECTIVE;
    c.distance = p->distance;
  } else if (dynamic_cast<const OrthogonalCamera *>(&camera)) {
    c.kind = RT_CAMERA_ORTHOGONAL;
    c.distance = 1.f;
  } else {
    throw std::invalid_argument("raytracer_amd: unknown Camera subclass");
  }
  const HomMatrix &m = camera.transformation.hom_matrix;
  int k = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      c.fwd[k++] = m.linear_part[i][j];
  c.fwd[9] = m.translation_vec.x;
  c.fwd[10] = m.translation_vec.y;
  c.fwd[11] = m.translation_vec.z;
  c.has_asp_ratio = camera.asp_ratio.has_value() ? 1u : 0u;
  c.asp_ratio = camera.asp_ratio.value_or(0.f);
  return c;
}

// ---- JSON dump (fixtures for the Python tests; floats as %.9g round-trip binary32 exactly) ------

namespace detail {
template <typename T> inline void json_ints(FILE *f, const char *name, const std::vector<T> &v, bool last = false) {
  std::fprintf(f, "  \"%s\": [", name);
  for (size_t i = 0; i < v.size(); ++i)
    std::fprintf(f, "%s%lld", i ? ", " : "", static_cast<long long>(v[i]));
  std::fprintf(f, "]%s\n", last ? "" : ",");
}
inline void json_floats(FILE *f, const char *name, const float *v, size_t n, bool last = false) {
  std::fprintf(f, "  \"%s\": [", name);
  for (size_t i = 0; i < n; ++i)
    std::fprintf(f, "%s%.9g", i ? ", " : "", static_cast<double>(v[i]));
  std::fprintf(f, "]%s\n", last ? "" : ",");
}
inline void json_floats(FILE *f, const char *name, const std::vector<float> &v, bool last = false) {
  json_floats(f, name, v.data(), v.size(), last);
}
} // namespace detail

inline void write_json(const FlatScene &s, const rt_camera &cam, const std::string &path) {
  FILE *f = std::fopen(path.c_str(), "w");
  if (!f)
    throw std::runtime_error("raytracer_amd: cannot open " + path);
  using namespace detail;
  std::fprintf(f, "{\n");
  json_ints(f, "shape_kind", s.shape_kind);
  json_ints(f, "shape_material", s.shape_material);
  json_floats(f, "shape_fwd", s.shape_fwd);
  json_floats(f, "shape_inv", s.shape_inv);
  json_ints(f, "csg_op", s.csg_op);
  json_ints(f, "csg_left", s.csg_left);
  json_ints(f, "csg_right", s.csg_right);
  json_floats(f, "csg_fwd", s.csg_fwd);
  json_floats(f, "csg_inv", s.csg_inv);
  json_ints(f, "objects", s.objects);
  json_ints(f, "material_brdf", s.material_brdf);
  json_ints(f, "material_brdf_pigment", s.material_brdf_pigment);
  json_ints(f, "material_emission_pigment", s.material_emission_pigment);
  json_ints(f, "pigment_kind", s.pigment_kind);
  json_floats(f, "pigment_color1", s.pigment_color1);
  json_floats(f, "pigment_color2", s.pigment_color2);
  json_ints(f, "pigment_n_intervals", s.pigment_n_intervals);
  json_ints(f, "pigment_tex_offset", s.pigment_tex_offset);
  json_ints(f, "pigment_tex_width", s.pigment_tex_width);
  json_ints(f, "pigment_tex_height", s.pigment_tex_height);
  json_floats(f, "texels", s.texels);
  json_floats(f, "light_point", s.light_point);
  json_floats(f, "light_color", s.light_color);
  json_floats(f, "light_radius", s.light_radius);
  std::fprintf(f, "  \"camera\": {\"kind\": %u, \"has_asp_ratio\": %u, \"asp_ratio\": %.9g, \"distance\": %.9g, ",
               cam.kind, cam.has_asp_ratio, static_cast<double>(cam.asp_ratio), static_cast<double>(cam.distance));
  std::fprintf(f, "\"fwd\": [");
  for (int i = 0; i < 12; ++i)
    std::fprintf(f, "%s%.9g", i ? ", " : "", static_cast<double>(cam.fwd[i]));
  std::fprintf(f, "]}\n}\n");
  std::fclose(f);
}

} // namespace raytracer_amd
