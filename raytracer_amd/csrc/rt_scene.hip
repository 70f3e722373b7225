// rt_scene.hip — rt_scene_create / rt_scene_destroy: validation of the description, CSG preprocessing, the optional BVH, the
// register-sphere tiles, upload.  Host code only (no kernel is defined or launched here).
#include "rt_internal.h"

using namespace rt_internal;

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

} // namespace

extern "C" {

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
    // The tree is the intersection of all its leaves when a boundary hit of leaf j is valid exactly where every OTHER leaf
    // contains the point: one bit per table.  Then the evaluator needs two candidates, not 2 L (rt_device.h, CsgTop::tt == 2).
    bool all_of = n_leaves >= 2;
    const uint32_t full = (1u << n_leaves) - 1u;
    for (int j = 0; j < n_leaves && all_of; ++j)
      for (uint32_t mask = 0; mask <= full && all_of; ++mask)
        if (!(mask & (1u << j))) // the evaluator never sets a leaf's own bit
          all_of = ((leaf_tt[static_cast<size_t>(top.leaf_begin + j)] >> mask) & 1ull) == (mask == (full & ~(1u << j)) ? 1ull : 0ull);
    if (all_of)
      top.tt = 2;
    // ... and the union of all its leaves when every boundary hit is valid whatever the other leaves say (the reference keeps
    // every hit of a union, inner ones too: shapes.hpp:460-490): all tables all ones.  Then the first hit is the smallest candidate.
    bool any_of = n_leaves >= 2 && !all_of;
    for (int j = 0; j < n_leaves && any_of; ++j)
      for (uint32_t mask = 0; mask <= full && any_of; ++mask)
        if (!(mask & (1u << j)))
          any_of = ((leaf_tt[static_cast<size_t>(top.leaf_begin + j)] >> mask) & 1ull) != 0ull;
    if (any_of)
      top.tt = 3;
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
  std::vector<float> cull_tiles, cull_bounds; // rt_tile_cull.h
  std::vector<int32_t> cull_rank, cull_pos;
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
  // row i of the tile whose RT_TILE_WORDS words start at `tile`: the operand words of the sphere (centre v[0..2], k = v[3])
  auto pack_tile_row = [&](float *tile, size_t i, const float v[4]) {
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
          std::memcpy(&tile[(c * 64 + half * 32 + i) * 4 + w], &word, 4);
        }
  };
  // the same row for v_mfma_f32_16x16x32_bf16 (rt_tile_cull.h, level 2): the tile as two halves of 16 rows, lane l of half h holds the
  // K slots 8 (l / 16) ... + 7 of row 16 h + l % 16 as four words
  auto pack_tile_row16 = [&](float *tile, size_t i, const float v[4]) {
    uint32_t parts[4][3];
    for (int q = 0; q < 4; ++q)
      split3(v[q], parts[q]);
    uint32_t slot[32] = {0};
    for (int q = 0; q < 3; ++q)
      for (int j = 0; j < 6; ++j)
        slot[6 * q + j] = parts[q][part_of_slot[j]];
    slot[18] = slot[19] = slot[20] = 0x3f80u; // 1
    slot[21] = parts[3][0], slot[22] = parts[3][1], slot[23] = parts[3][2];
    const size_t h = i / 16, r = i % 16;
    for (size_t kq = 0; kq < 4; ++kq)
      for (size_t w = 0; w < 4; ++w) {
        const size_t s0 = 8 * kq + 2 * w;
        const uint32_t word = slot[s0] | (slot[s0 + 1] << 16);
        std::memcpy(&tile[(h * 64 + kq * 16 + r) * 4 + w], &word, 4);
      }
  };
  static const float never_hit[4] = {0.f, 0.f, 0.f, 1e30f}; // a place past the last sphere: c = 1e30
  for (size_t t = 0; t < n_wide_tiles; ++t)
    for (size_t i = 0; i < 32; ++i) {
      const size_t k = t * 32 + i;
      pack_tile_row(&wide_tiles[t * rtd::RT_TILE_WORDS], i, k < wide_recs.size() ? &wide_iso[4 * k] : never_hit);
    }
  // The same spheres once more as spatial CLUSTERS of CS = 32 RT_CULL_C with a bounding sphere each (rt_tile_cull.h): a k-d split of
  // the centres at multiples of CS along the widest axis, so that every cluster but the last is full.  cull_rank maps a cluster
  // position back to the position in wide_recs (object order: what a hit's key carries, so that equal distances resolve as the
  // reference's strict `<` does), cull_pos is the inverse (the sphere a ray leaves is excluded by position).
  if (n_wide_tiles >= RT_CULL_MIN_TILES) {
    constexpr size_t CC = rtd::RT_CULL_C, CS = 32 * CC;
    const size_t n = wide_recs.size();
    std::vector<int32_t> order(n);
    for (size_t k = 0; k < n; ++k)
      order[k] = static_cast<int32_t>(k);
    std::vector<std::pair<size_t, size_t>> todo{{0, n}};
    while (!todo.empty()) {
      const auto [b, e] = todo.back();
      todo.pop_back();
      if (e - b <= CS)
        continue;
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (size_t k = b; k < e; ++k)
        for (int a = 0; a < 3; ++a)
          lo[a] = std::min(lo[a], wide_iso[4 * order[k] + a]), hi[a] = std::max(hi[a], wide_iso[4 * order[k] + a]);
      int axis = 0;
      for (int a = 1; a < 3; ++a)
        if (hi[a] - lo[a] > hi[axis] - lo[axis])
          axis = a;
      const size_t parts = (e - b + CS - 1) / CS, mid = b + CS * ((parts + 1) / 2);
      std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](int32_t x, int32_t y) {
        const float fx = wide_iso[4 * x + axis], fy = wide_iso[4 * y + axis];
        return fx < fy || (fx == fy && x < y);
      });
      todo.push_back({b, mid}), todo.push_back({mid, e});
    }
    for (size_t b = 0; b < n; b += CS) // within a cluster in object order: equal distances resolve by row (rt_tile_cull.h)
      std::sort(order.begin() + b, order.begin() + std::min(b + CS, n));
    cull_rank.assign(n_wide_tiles * 32, -1);
    cull_pos.assign(n, -1);
    cull_tiles.resize(n_wide_tiles * rtd::RT_TILE_WORDS);
    constexpr size_t CW = rtd::RT_CULL_W;
    const size_t n_clusters = (n_wide_tiles + CC - 1) / CC;
    const size_t n_bound_tiles = (n_clusters + 32 * CW - 1) / (32 * CW) * CW; // one per group of 32 CW clusters and wave
    cull_bounds.resize(n_bound_tiles * rtd::RT_TILE_WORDS);
    for (size_t g = 0; g < n_bound_tiles; ++g)
      for (size_t i = 0; i < 32; ++i)
        pack_tile_row(&cull_bounds[g * rtd::RT_TILE_WORDS], i, never_hit);
    auto radius_of = [&](size_t k) { return std::fabs(static_cast<double>(d->shape_fwd[12 * static_cast<size_t>(wide_recs[k].shape)])); };
    for (size_t t = 0; t < n_wide_tiles; ++t)
      for (size_t i = 0; i < 32; ++i) {
        const size_t pos = t * 32 + i;
        if (pos >= n) {
          pack_tile_row16(&cull_tiles[t * rtd::RT_TILE_WORDS], i, never_hit);
          continue;
        }
        const size_t k = static_cast<size_t>(order[pos]);
        cull_rank[pos] = static_cast<int32_t>(k), cull_pos[k] = static_cast<int32_t>(pos);
        pack_tile_row16(&cull_tiles[t * rtd::RT_TILE_WORDS], i, &wide_iso[4 * k]);
      }
    for (size_t j = 0; j < n_clusters; ++j) {
      const size_t begin = j * CS, end = std::min(begin + CS, n);
      double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (size_t pos = begin; pos < end; ++pos) {
        const size_t k = static_cast<size_t>(order[pos]);
        for (int a = 0; a < 3; ++a)
          lo[a] = std::min(lo[a], wide_iso[4 * k + a] - radius_of(k)), hi[a] = std::max(hi[a], wide_iso[4 * k + a] + radius_of(k));
      }
      // Bounding sphere: from the middle of the cluster's box the centre walks towards the farthest sphere surface with steps
      // 1 / (k + 1) (Badoiu-Clarkson: converges to the smallest enclosing ball), the best (rounded) centre seen is kept.  Its
      // radius^2 is inflated by more than the filter's rounding can move either of the level-1 decisions (world_filter_ok: that
      // error is below 0.05 r_min^2 wherever this kernel runs).
      auto reach_from = [&](const double c[3], size_t *farthest) {
        double reach = 0.0;
        for (size_t pos = begin; pos < end; ++pos) {
          const size_t k = static_cast<size_t>(order[pos]);
          const double dx = wide_iso[4 * k] - c[0], dy = wide_iso[4 * k + 1] - c[1], dz = wide_iso[4 * k + 2] - c[2];
          const double r = std::sqrt(dx * dx + dy * dy + dz * dz) + radius_of(k);
          if (r > reach)
            reach = r, *farthest = k;
        }
        return reach;
      };
      double walk[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
      float centre[3] = {static_cast<float>(walk[0]), static_cast<float>(walk[1]), static_cast<float>(walk[2])};
      size_t far_k = 0;
      double reach = 0.0;
      {
        const double c0[3] = {centre[0], centre[1], centre[2]};
        reach = reach_from(c0, &far_k);
      }
      for (int it = 1; it <= 200; ++it) {
        reach_from(walk, &far_k);
        // the farthest surface point of sphere far_k seen from the walking centre
        double dir[3] = {wide_iso[4 * far_k] - walk[0], wide_iso[4 * far_k + 1] - walk[1], wide_iso[4 * far_k + 2] - walk[2]};
        const double len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        const double scale = len > 0.0 ? (len + radius_of(far_k)) / len : 0.0;
        for (int a = 0; a < 3; ++a)
          walk[a] += dir[a] * scale / (it + 1);
        const float cand[3] = {static_cast<float>(walk[0]), static_cast<float>(walk[1]), static_cast<float>(walk[2])};
        const double cd[3] = {cand[0], cand[1], cand[2]};
        size_t unused = 0;
        const double r = reach_from(cd, &unused);
        if (r < reach)
          reach = r, centre[0] = cand[0], centre[1] = cand[1], centre[2] = cand[2];
      }
      const double r2 = reach * reach * (1.0 + 1.0 / 1024.0) + 0.5 * wide_rmin2;
      const double cc = double(centre[0]) * centre[0] + double(centre[1]) * centre[1] + double(centre[2]) * centre[2];
      const float v[4] = {centre[0], centre[1], centre[2], static_cast<float>(cc - r2)};
      // cluster j = 32 CW g + CW i + w is row i of bound tile g CW + w: a wave's 32 clusters of a group are spread over the group's space
      pack_tile_row(&cull_bounds[(j / (32 * CW) * CW + j % CW) * rtd::RT_TILE_WORDS], j % (32 * CW) / CW, v);
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
  UP(upload(sc, cull_tiles, &S.cull_tiles));
  UP(upload(sc, cull_bounds, &S.cull_bounds));
  UP(upload(sc, cull_rank, &S.cull_rank));
  UP(upload(sc, cull_pos, &S.cull_pos));
  S.n_cull_tiles = static_cast<int32_t>(cull_tiles.size() / rtd::RT_TILE_WORDS);
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
  {
    bool diag = true;
    const float *tables[4] = {d->shape_fwd, d->shape_inv, shape_w2l.data(), shape_l2w.data()};
    for (const float *t : tables)
      for (size_t i = 0; i < static_cast<size_t>(d->n_shapes) && diag; ++i)
        for (int k : {1, 2, 3, 5, 6, 7})
          diag = diag && t[12 * i + static_cast<size_t>(k)] == 0.f;
    S.all_diag = diag ? 1 : 0;
  }
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
