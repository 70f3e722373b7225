// rt_device.h — device functions of the tracing loop (gfx950).  Compiled with -ffp-contract=off:
// every predicate that decides hit / no-hit uses the reference's fp32 operation order without
// fusing, so that the deterministic modes reproduce the reference's decisions.
// Citations are to the reference sources (paths relative to the reference root).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/raytracer_amd.h"
#include "rt_types.h"

namespace rtd {

#define RT_DEV __device__ __forceinline__

// A loop count that is a kernel argument, made opaque where it is used.  Why: hipcc (ROCm 7.2) keeps "n > 0" of such a count as
// a wave mask and forms its negation with a VECTOR compare, under whatever lanes are active at that point; PointLightTracer's
// mirror loop ends with only the lanes that still bounce, and the light loop that follows then tested that mask against ITS
// lanes: a wave whose last bouncing lanes had all left took "n_csg_tops > 0" for true in a world without CSG objects and
// walked a tree that is not there (a GPU memory fault on the first scene with a point light, a mirror and no CSG object).
// An SGPR value that comes out of an asm statement is compared where it is used, with scalar instructions.
__device__ __forceinline__ int uniform_count(int n) {
  asm volatile("" : "+s"(n));
  return n;
}

constexpr float PI_F = 3.14159265358979323846f; // std::numbers::pi_v<float>
constexpr float INF_F = __builtin_inff();

struct V3 { float x, y, z; };
struct Ray { V3 o, d; float tmin, tmax; }; // include/cameras.hpp:34-41

RT_DEV V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
RT_DEV float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // geometry.hpp:263-265
RT_DEV float sqnorm3(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }     // geometry.hpp:104
RT_DEV V3 add3(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 sub3(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 scale3(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }   // Vec * scalar
RT_DEV V3 lscale3(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }  // scalar * Vec
RT_DEV V3 mul3(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }  // Color * Color

// Vec::normalized — geometry.hpp:111-114
RT_DEV V3 normalized3(V3 a) {
  float n = sqrtf(sqnorm3(a));
  return mk3(a.x / n, a.y / n, a.z / n);
}

// Fast variants for the FMA path tracer (statistical parity bar): reciprocal square root and fused transforms.
RT_DEV V3 normalized3_fast(V3 a) {
  const float inv = __builtin_amdgcn_rsqf(fmaf(a.x, a.x, fmaf(a.y, a.y, a.z * a.z)));
  return mk3(a.x * inv, a.y * inv, a.z * inv);
}
template <typename P> RT_DEV V3 hom_vec_fast(P m, V3 v) {
  return mk3(fmaf(m[0], v.x, fmaf(m[1], v.y, m[2] * v.z)), fmaf(m[3], v.x, fmaf(m[4], v.y, m[5] * v.z)),
             fmaf(m[6], v.x, fmaf(m[7], v.y, m[8] * v.z)));
}
template <typename P> RT_DEV V3 hom_point_fast(P m, V3 p) {
  return mk3(fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[2], p.z, m[9]))), fmaf(m[3], p.x, fmaf(m[4], p.y, fmaf(m[5], p.z, m[10]))),
             fmaf(m[6], p.x, fmaf(m[7], p.y, fmaf(m[8], p.z, m[11]))));
}
template <typename P> RT_DEV V3 hom_transposed_fast(P m, V3 n) {
  return mk3(fmaf(m[0], n.x, fmaf(m[3], n.y, m[6] * n.z)), fmaf(m[1], n.x, fmaf(m[4], n.y, m[7] * n.z)),
             fmaf(m[2], n.x, fmaf(m[5], n.y, m[8] * n.z)));
}

// The same three for a transformation whose linear part is diagonal (Scene::all_diag): the terms hom_*_fast multiply by zero
// are left out — bit for bit the same values, a zero result's sign aside.
template <typename P> RT_DEV V3 diag_vec_fast(P m, V3 v) { return mk3(m[0] * v.x, m[4] * v.y, m[8] * v.z); }
template <typename P> RT_DEV V3 diag_point_fast(P m, V3 p) { return mk3(fmaf(m[0], p.x, m[9]), fmaf(m[4], p.y, m[10]), fmaf(m[8], p.z, m[11])); }

// x * 2^32 truncated toward zero — what (long long)((double)x * 4294967296.0) gives — without fp64: the integer part
// and the fraction of |x| convert exactly.  |x| is clamped below 2^31 (oracle/rt_oracle.c clamps alike).
RT_DEV long long to_fixed32(float x) {
  const float ax = fminf(fabsf(x), 2147483520.f);
  const uint32_t hi = (uint32_t)ax;                                // truncates
  const uint32_t lo = (uint32_t)((ax - (float)hi) * 4294967296.f); // exact product, truncates
  const long long v = (long long)(((unsigned long long)hi << 32) | lo);
  return x < 0.f ? -v : v;
}

// HomMatrix::apply(Vec) / apply(Point) / apply_transposed(Normal) — geometry.hpp:379-400
template <typename P> RT_DEV V3 hom_vec(P m, V3 v) {
  return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
             m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
template <typename P> RT_DEV V3 hom_point(P m, V3 p) {
  return mk3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[9], m[3] * p.x + m[4] * p.y + m[5] * p.z + m[10],
             m[6] * p.x + m[7] * p.y + m[8] * p.z + m[11]);
}
template <typename P> RT_DEV V3 hom_transposed(P m, V3 n) {
  return mk3(m[0] * n.x + m[3] * n.y + m[6] * n.z, m[1] * n.x + m[4] * n.y + m[7] * n.z,
             m[2] * n.x + m[5] * n.y + m[8] * n.z);
}
// Ray::transform — cameras.hpp:66
template <typename P> RT_DEV Ray ray_transform(Ray r, P m) {
  Ray q = r;
  q.o = hom_point(m, r.o);
  q.d = hom_vec(m, r.d);
  return q;
}

// are_close / Point::is_close — colors.hpp:81, geometry.hpp:153-156
RT_DEV bool are_close(float x, float y) { return fabsf(x - y) < 1e-5f; }
RT_DEV bool point_is_close(V3 a, V3 b) { return are_close(a.x, b.x) && are_close(a.y, b.y) && are_close(a.z, b.z); }

// ---------------------------------------------------------------------------------------------
// PCG — include/random.hpp:29-101
// ---------------------------------------------------------------------------------------------
struct Pcg { uint64_t state, inc; };
constexpr uint64_t PCG_MULT = 6364136223846793005ull;

RT_DEV uint32_t pcg_random(Pcg &p) { // random.hpp:52-64
  uint64_t old = p.state;
  p.state = old * PCG_MULT + p.inc;
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
}
RT_DEV void pcg_init(Pcg &p, uint64_t init_state, uint64_t init_seq) { // random.hpp:41-47
  p.state = 0;
  p.inc = (init_seq << 1u) | 1u;
  pcg_random(p);
  p.state += init_state;
  pcg_random(p);
}
// random.hpp:67-70: float(ran) / 2^32 (exact scaling; float(ran) rounds to nearest, may give 1.0f)
RT_DEV float pcg_random_float(Pcg &p) { return (float)pcg_random(p) * 2.3283064365386963e-10f; }

// LCG jump-ahead by n steps, O(log n) (Brown 1994).
RT_DEV void pcg_advance(Pcg &p, uint64_t n) {
  uint64_t acc_mult = 1, acc_plus = 0, cur_mult = PCG_MULT, cur_plus = p.inc;
  while (n > 0) {
    if (n & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    n >>= 1;
  }
  p.state = acc_mult * p.state + acc_plus;
}

RT_DEV uint64_t mix64(uint64_t z) { // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// The AA generator is always PCG(42, 54) (ImageTracer ctor quirk, cameras.hpp:178-183); its state
// after the constructor, advanced by 2k draws, gives sample k's jitter pair (cameras.hpp:232-233).
// The jump is the product of the affine maps x -> M_i x + P_i of the set bits i of 2k, where (M_i, P_i) is the
// generator's step applied 2^i times: powers of one map commute, so they are applied to the state in any order, and
// for this fixed stream they are compile-time constants — one 64-bit multiply-add per set bit instead of the four
// multiplies per bit of the generic pcg_advance (which was most of the flat / pointlight kernels' arithmetic at AA > 1).
struct AaJumpTable {
  uint64_t mult[64], plus[64], state0;
};
constexpr AaJumpTable make_aa_jump_table() {
  AaJumpTable t{};
  const uint64_t inc = (54ull << 1u) | 1u;
  uint64_t m = PCG_MULT, c = inc;
  for (int i = 0; i < 64; ++i) {
    t.mult[i] = m;
    t.plus[i] = c;
    c = (m + 1) * c;
    m = m * m;
  }
  uint64_t st = 0; // pcg_init(42, 54): random.hpp:41-47
  st = st * PCG_MULT + inc;
  st += 42ull;
  st = st * PCG_MULT + inc;
  t.state0 = st;
  return t;
}
RT_DEV void aa_jitter(uint64_t k, float &u, float &v) {
  static constexpr AaJumpTable T = make_aa_jump_table();
  Pcg p;
  p.inc = (54ull << 1u) | 1u;
  p.state = T.state0;
  uint64_t n = 2 * k;
  for (int i = 1; n >> i; ++i) // bit 0 of 2k is clear
    if ((n >> i) & 1ull)
      p.state = p.state * T.mult[i] + T.plus[i];
  u = pcg_random_float(p);
  v = pcg_random_float(p);
}

// ---------------------------------------------------------------------------------------------
// Camera — include/cameras.hpp:112-121, 133-137, 155-159, 200-208
// ---------------------------------------------------------------------------------------------
RT_DEV Ray camera_fire_ray(const Camera &cam, int width, int height, int col, int row, float u_pixel, float v_pixel) {
  float u = ((float)col + u_pixel) / (float)width;
  float v = 1.f - (((float)row + v_pixel) / (float)height);
  float y = (1.f - u * 2.f) * cam.asp;
  float z = -1.f + v * 2.f;
  Ray r;
  r.tmin = 1e-5f;
  r.tmax = INF_F;
  if (cam.kind == RT_CAMERA_PERSPECTIVE) {
    r.o = mk3(-cam.distance, 0.f, 0.f);
    r.d = mk3(cam.distance, y, z);
  } else {
    r.o = mk3(-1.f, y, z);
    r.d = mk3(1.f, 0.f, 0.f);
  }
  return ray_transform(r, cam.fwd);
}

// ---------------------------------------------------------------------------------------------
// Shapes — include/shapes.hpp
// ---------------------------------------------------------------------------------------------

// Sphere::solve_ray_sphere — shapes.hpp:269-303.  Returns the number of valid roots (0..2).
RT_DEV int solve_ray_sphere(const Ray &ray, float &ta, float &tb) {
  float a = sqnorm3(ray.d);
  float b = dot3(ray.o, ray.d);
  float c = sqnorm3(ray.o) - 1.f;
  float reduced_disc = b * b - a * c;
  if (reduced_disc <= 0.f)
    return 0;
  float sq = sqrtf(reduced_disc);
  float t1 = (-b - sq) / a;
  float t2 = (-b + sq) / a;
  if (t1 > t2) {
    float tmp = t1;
    t1 = t2;
    t2 = tmp;
  }
  if (t1 < ray.tmin) {
    if (t2 > ray.tmin && t2 < ray.tmax) {
      ta = t2;
      return 1;
    }
    return 0;
  }
  ta = t1;
  if (t2 < ray.tmax) {
    tb = t2;
    return 2;
  }
  return 1;
}

// Plane::ray_intersection predicate — shapes.hpp:329-338
RT_DEV bool solve_ray_plane(const Ray &ray, float &t) {
  if (are_close(ray.d.z, 0.f))
    return false;
  float t_hit = -ray.o.z / ray.d.z;
  if (t_hit < ray.tmin || t_hit > ray.tmax)
    return false;
  t = t_hit;
  return true;
}

// What the renderers read from a HitRecord (shapes.hpp:65-96), plus the leaf-frame point from which
// (u,v) is derived lazily (only pigments that read uv pay for atan2/acos).
struct Hit {
  int object; // index in World::objects, -1 = miss
  int shape;
  float t;
  V3 point;  // world_point
  V3 normal; // not normalised
  V3 local;  // hit point in the standard shape's frame
};

// Shape::make_hit without uv — shapes.hpp:163-186, 254-256, 362
RT_DEV void make_hit(const Scene &s, int shape, const Ray &ray_local, float t, V3 &point, V3 &normal, V3 &local) {
  V3 p = add3(ray_local.o, lscale3(t, ray_local.d)); // Ray::at — cameras.hpp:62
  V3 n = (s.shape_kind[shape] == RT_SHAPE_SPHERE) ? p : mk3(0.f, 0.f, 1.f);
  float sign = copysignf(1.0f, -(dot3(n, ray_local.d)));
  n = scale3(n, sign);
  local = p;
  point = hom_point(s.shape_fwd + 12 * shape, p);
  normal = hom_transposed(s.shape_inv + 12 * shape, n); // Transformation * Normal — geometry.hpp:506
}

// The same record with fused transforms (FMA path tracer).
RT_DEV void make_hit_fast(const Scene &s, int shape, const Ray &ray_world, float t, V3 &point, V3 &normal, V3 &local) {
  const float *inv = s.shape_inv + 12 * shape;
  const bool diag = s.all_diag != 0; // uniform
  V3 o, d;
  if (diag)
    o = diag_point_fast(inv, ray_world.o), d = diag_vec_fast(inv, ray_world.d);
  else
    o = hom_point_fast(inv, ray_world.o), d = hom_vec_fast(inv, ray_world.d);
  const V3 p = mk3(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z));
  V3 n = (s.shape_kind[shape] == RT_SHAPE_SPHERE) ? p : mk3(0.f, 0.f, 1.f);
  const float sign = copysignf(1.0f, -fmaf(n.x, d.x, fmaf(n.y, d.y, n.z * d.z)));
  n = scale3(n, sign);
  local = p;
  if (diag) {
    point = diag_point_fast(s.shape_fwd + 12 * shape, p);
    normal = diag_vec_fast(inv, n);
  } else {
    point = hom_point_fast(s.shape_fwd + 12 * shape, p);
    normal = hom_transposed_fast(inv, n);
  }
}

// Sphere::calculate_uv / Plane::calculate_uv — shapes.hpp:259-266, 365-367
RT_DEV void shape_uv(const Scene &s, int shape, V3 p, float &u, float &v) {
  if (s.shape_kind[shape] == RT_SHAPE_SPHERE) {
    float uu = atan2f(p.y, p.x) / (2.f * PI_F);
    if (uu < 0.f)
      uu = uu + 1.f;
    u = uu;
    v = acosf(p.z) / PI_F;
  } else {
    u = p.x - floorf(p.x);
    v = p.y - floorf(p.y);
  }
}

// Sphere::is_point_inside / Plane::is_point_inside — shapes.hpp:245-250, 353-358
RT_DEV bool shape_is_point_inside(const Scene &s, int shape, V3 point_world) {
  V3 p = hom_point(s.shape_inv + 12 * shape, point_world);
  return (s.shape_kind[shape] == RT_SHAPE_SPHERE) ? (sqnorm3(p) < 1.f) : (p.z < 0.f);
}

// ---------------------------------------------------------------------------------------------
// CSG without a heap — include/shapes.hpp:396-490
// ---------------------------------------------------------------------------------------------

// CSGObject::is_point_inside of a whole subtree, as an RPN program over its leaves
// (shapes.hpp:438-457).  `p` is in the frame the subtree's root object lives in.
RT_DEV bool csg_prog_inside(const Scene &s, int begin, int end, V3 p) {
  uint32_t bits = 0; // boolean stack, top = bit 0
  for (int e = begin; e < end; ++e) {
    const CsgProgEntry pe = s.csg_prog[e];
    if (!pe.is_op) {
      V3 q = p;
      for (int k = 0; k < pe.path_len; ++k)
        q = hom_point(s.csg_inv + 12 * s.csg_prog_path[pe.path_begin + k], q);
      bits = (bits << 1) | (shape_is_point_inside(s, pe.arg, q) ? 1u : 0u);
    } else {
      const bool in2 = bits & 1u, in1 = (bits >> 1) & 1u;
      bool r;
      switch (pe.arg) {
      case RT_CSG_UNION: r = in1 || in2; break;
      case RT_CSG_INTERSECTION: r = in1 && in2; break;
      case RT_CSG_DIFFERENCE: r = in1 && !in2; break;
      default: r = in1 || in2; break; // FUSION
      }
      bits = ((bits >> 2) << 1) | (r ? 1u : 0u);
    }
  }
  return bits & 1u;
}

// CSGObject::hit_on_obj1_valid / hit_on_obj2_valid — shapes.hpp:460-490
RT_DEV bool csg_hit_valid(int op, int side, bool inside_sibling) {
  switch (op) {
  case RT_CSG_UNION: return true;
  case RT_CSG_INTERSECTION: return inside_sibling;
  case RT_CSG_DIFFERENCE: return side == 0 ? !inside_sibling : inside_sibling;
  default: return !inside_sibling; // FUSION
  }
}

// CSGObject::ray_intersection = front of all_ray_intersections (shapes.hpp:396-435), evaluated leaf
// by leaf: a leaf hit survives iff it passes the validity rule of every CSG node on its way to the
// root, each tested with the hit point expressed in that node's frame — the same chain of
// transformations, in the same order, as the recursive reference.  Ties in t between leaves go to
// the later leaf (merge_ordered_vectors takes from the second list on ties, shapes.hpp:47-52).
RT_DEV bool csg_ray_intersection(const Scene &s, const CsgTop &top, const Ray &ray_world, Hit &best) {
  bool found = false;
  for (int li = top.leaf_begin; li < top.leaf_end; ++li) {
    const CsgLeaf leaf = s.csg_leaf[li];
    Ray ray = ray_world;
    for (int k = 0; k < leaf.path_len; ++k)
      ray = ray_transform(ray, s.csg_inv + 12 * s.csg_path[leaf.path_begin + k].node);
    const Ray ray_parent = ray;
    ray = ray_transform(ray_parent, s.shape_inv + 12 * leaf.shape);
    float ts[2];
    int nt;
    if (s.shape_kind[leaf.shape] == RT_SHAPE_SPHERE) {
      nt = solve_ray_sphere(ray, ts[0], ts[1]);
    } else {
      nt = solve_ray_plane(ray, ts[0]) ? 1 : 0;
    }
    bool first_of_leaf = true;
    for (int i = 0; i < nt; ++i) {
      const float t = ts[i];
      // Sphere::all_ray_intersections re-checks the open interval (shapes.hpp:236)
      if (s.shape_kind[leaf.shape] == RT_SHAPE_SPHERE && !(t > ray.tmin && t < ray.tmax))
        continue;
      V3 point, normal, local;
      make_hit(s, leaf.shape, ray, t, point, normal, local);
      bool valid = true;
      for (int k = leaf.path_len - 1; k >= 0 && valid; --k) {
        const CsgStep st = s.csg_path[leaf.path_begin + k];
        const int *range = s.csg_prog_range + 4 * st.node + 2 * (1 - st.side);
        const bool inside = csg_prog_inside(s, range[0], range[1], point);
        valid = csg_hit_valid(s.csg_op[st.node], st.side, inside);
        if (valid) { // HitRecord::transform_in_place — shapes.hpp:91-95
          point = hom_point(s.csg_fwd + 12 * st.node, point);
          normal = hom_transposed(s.csg_inv + 12 * st.node, normal);
        }
      }
      if (!valid)
        continue;
      const bool better = !found || (first_of_leaf ? (t <= best.t) : (t < best.t));
      if (better) {
        found = true;
        best.t = t;
        best.shape = leaf.shape;
        best.point = point;
        best.normal = normal;
        best.local = local;
        best.object = top.object;
      }
      first_of_leaf = false;
    }
  }
  return found;
}

// The same evaluation for the FMA path kernels: every leaf is reached from world space with ONE composed
// transform (Scene::shape_w2l), candidate hits are tested for "inside the sibling subtree" at their world point, and
// the winner's record is built once at the end.  Same hits, same tie rules; rounding differs (statistical bar).
RT_DEV bool shape_is_point_inside_fast(const Scene &s, int shape, V3 pw) {
  const float *m = s.shape_w2l + 12 * shape;
  if (s.shape_kind[shape] == RT_SHAPE_SPHERE) {
    const V3 q = hom_point_fast(m, pw);
    return fmaf(q.x, q.x, fmaf(q.y, q.y, q.z * q.z)) < 1.f;
  }
  return fmaf(m[6], pw.x, fmaf(m[7], pw.y, fmaf(m[8], pw.z, m[11]))) < 0.f;
}

RT_DEV bool csg_prog_inside_fast(const Scene &s, int begin, int end, V3 pw) {
  uint32_t bits = 0; // boolean stack, top = bit 0
  for (int e = begin; e < end; ++e) {
    const CsgProgEntry pe = s.csg_prog[e];
    if (!pe.is_op) {
      bits = (bits << 1) | (shape_is_point_inside_fast(s, pe.arg, pw) ? 1u : 0u);
    } else {
      const bool in2 = bits & 1u, in1 = (bits >> 1) & 1u;
      bool r;
      switch (pe.arg) {
      case RT_CSG_UNION: r = in1 || in2; break;
      case RT_CSG_INTERSECTION: r = in1 && in2; break;
      case RT_CSG_DIFFERENCE: r = in1 && !in2; break;
      default: r = in1 || in2; break; // FUSION
      }
      bits = ((bits >> 2) << 1) | (r ? 1u : 0u);
    }
  }
  return bits & 1u;
}

RT_DEV bool csg_ray_intersection_fast(const Scene &s, const CsgTop &top, const Ray &ray_world, Hit &best) {
  bool found = false;
  float best_t = 0.f;
  int best_shape = -1;
  for (int li = top.leaf_begin; li < top.leaf_end; ++li) {
    const CsgLeaf leaf = s.csg_leaf[li];
    const float *inv = s.shape_w2l + 12 * leaf.shape;
    Ray ray = ray_world;
    ray.o = hom_point_fast(inv, ray_world.o);
    ray.d = hom_vec_fast(inv, ray_world.d);
    float ts[2];
    int nt;
    const bool sphere = s.shape_kind[leaf.shape] == RT_SHAPE_SPHERE;
    if (sphere)
      nt = solve_ray_sphere(ray, ts[0], ts[1]);
    else
      nt = solve_ray_plane(ray, ts[0]) ? 1 : 0;
    bool first_of_leaf = true;
    for (int i = 0; i < nt; ++i) {
      const float t = ts[i];
      if (sphere && !(t > ray.tmin && t < ray.tmax))
        continue;
      const bool better = !found || (first_of_leaf ? (t <= best_t) : (t < best_t));
      if (!better) { // cannot win: skip the inside tests (the exact evaluator tests validity first; same outcome)
        continue;
      }
      const V3 pw = mk3(fmaf(t, ray_world.d.x, ray_world.o.x), fmaf(t, ray_world.d.y, ray_world.o.y), fmaf(t, ray_world.d.z, ray_world.o.z));
      bool valid = true;
      for (int k = leaf.path_len - 1; k >= 0 && valid; --k) {
        const CsgStep st = s.csg_path[leaf.path_begin + k];
        const int *range = s.csg_prog_range + 4 * st.node + 2 * (1 - st.side);
        valid = csg_hit_valid(s.csg_op[st.node], st.side, csg_prog_inside_fast(s, range[0], range[1], pw));
      }
      if (!valid)
        continue;
      found = true;
      best_t = t;
      best_shape = leaf.shape;
      first_of_leaf = false;
    }
  }
  if (found) {
    const float *inv = s.shape_w2l + 12 * best_shape;
    const V3 o = hom_point_fast(inv, ray_world.o), d = hom_vec_fast(inv, ray_world.d);
    const V3 p = mk3(fmaf(best_t, d.x, o.x), fmaf(best_t, d.y, o.y), fmaf(best_t, d.z, o.z));
    V3 n = (s.shape_kind[best_shape] == RT_SHAPE_SPHERE) ? p : mk3(0.f, 0.f, 1.f);
    n = scale3(n, copysignf(1.0f, -fmaf(n.x, d.x, fmaf(n.y, d.y, n.z * d.z))));
    best.t = best_t;
    best.shape = best_shape;
    best.object = top.object;
    best.local = p;
    best.point = hom_point_fast(s.shape_l2w + 12 * best_shape, p);
    best.normal = hom_transposed_fast(inv, n);
  }
  return found;
}

// The FMA evaluator for trees of at most LMAX <= 6 leaves (CsgTop::tt): no program is interpreted and no point is
// transformed per candidate.  Every leaf is intersected once in its own frame (one composed transform); that also gives
// the stretch of the ray that lies INSIDE the leaf — between the two roots of a sphere, on one side of a plane's root —
// so "is this candidate inside leaf i" is two comparisons of ray parameters, the inside mask of a candidate is L such
// pairs, and its validity is one bit of the leaf's truth table.  Same candidates, same order and tie rules as
// csg_ray_intersection_fast; a point within rounding of another leaf's boundary may fall on the other side of it than
// with the transformed-point test (statistical bar of the FMA kernels).  Everything is unrolled: the per-leaf values
// stay in registers.
template <int LMAX> RT_DEV bool csg_ray_intersection_tt(const Scene &s, const CsgTop &top, const Ray &ray_world, Hit &best) {
  const int n_leaves = top.leaf_end - top.leaf_begin;
  const bool diag = s.all_diag != 0; // uniform: every transformation is translation * axis-aligned scaling
  float t_in[LMAX], t_out[LMAX], c0[LMAX], c1[LMAX]; // inside stretch (open); its two ends as candidate hits (INF: not in the ray's range)
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    t_in[j] = INF_F, t_out[j] = -INF_F, c0[j] = INF_F, c1[j] = INF_F;
    if (j < n_leaves) {
      const int shape = s.csg_leaf[top.leaf_begin + j].shape;
      const float *inv = s.shape_w2l + 12 * shape;
      if (s.shape_kind[shape] == RT_SHAPE_SPHERE) {
        V3 o, d;
        if (diag)
          o = diag_point_fast(inv, ray_world.o), d = diag_vec_fast(inv, ray_world.d);
        else
          o = hom_point_fast(inv, ray_world.o), d = hom_vec_fast(inv, ray_world.d);
        const float a = fmaf(d.x, d.x, fmaf(d.y, d.y, d.z * d.z));
        const float b = fmaf(o.x, d.x, fmaf(o.y, d.y, o.z * d.z));
        const float c = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, -1.f)));
        const float disc = fmaf(b, b, -(a * c));
        if (disc > 0.f) {
          const float sq = __builtin_amdgcn_sqrtf(disc), inv_a = __builtin_amdgcn_rcpf(a);
          const float t1 = (-b - sq) * inv_a, t2 = (-b + sq) * inv_a; // a > 0: t1 < t2
          t_in[j] = t1, t_out[j] = t2;
          if (t1 > ray_world.tmin && t1 < ray_world.tmax) // Sphere::all_ray_intersections keeps the roots in the open range (shapes.hpp:236)
            c0[j] = t1;
          if (t2 > ray_world.tmin && t2 < ray_world.tmax)
            c1[j] = t2;
        }
      } else { // plane: inside where o.z + t d.z < 0 (shapes.hpp:353-358)
        float oz, dz;
        if (diag) {
          oz = fmaf(inv[8], ray_world.o.z, inv[11]), dz = inv[8] * ray_world.d.z;
        } else {
          oz = fmaf(inv[6], ray_world.o.x, fmaf(inv[7], ray_world.o.y, fmaf(inv[8], ray_world.o.z, inv[11])));
          dz = fmaf(inv[6], ray_world.d.x, fmaf(inv[7], ray_world.d.y, inv[8] * ray_world.d.z));
        }
        const float t = -oz * __builtin_amdgcn_rcpf(dz);
        const float cand = !are_close(dz, 0.f) && !(t < ray_world.tmin || t > ray_world.tmax) ? t : INF_F; // Plane::ray_intersection (shapes.hpp:329-338)
        if (dz > 0.f) // the ray leaves the half-space at t: c0 is a leaf's ENTRY as a candidate hit, c1 its EXIT (as for a sphere's two roots)
          t_in[j] = -INF_F, t_out[j] = t, c1[j] = cand;
        else if (dz < 0.f)
          t_in[j] = t, t_out[j] = INF_F, c0[j] = cand;
        else if (oz < 0.f)
          t_in[j] = -INF_F, t_out[j] = INF_F;
      }
    }
  }
  bool found = false;
  float best_t = 0.f;
  int best_leaf = -1;
  if (top.tt == 2) { // uniform
    // The intersection of all leaves (the host read it off the truth tables): the ray is inside the solid between the LAST
    // entry T_in = max t_in and the FIRST exit T_out = min t_out, and those two boundaries are the only hits that can be valid —
    // any other entry lies outside the leaf entered last, any other exit beyond the leaf left first.  Validity is the tables' rule,
    // evaluated without them: the point is strictly inside every other leaf, i.e. L - 1 leaves contain it (a leaf does not
    // contain its own boundary, nor does one whose boundary coincides: the strict compares of the general evaluator).
    float T_in = -INF_F, E_in = INF_F, T_out = INF_F, E_out = INF_F; // E: the boundary as a candidate hit (INF: out of the ray's range)
    int J_in = 0, J_out = 0;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
      if (j < n_leaves) {
        if (t_in[j] > T_in) // c0 / c1: the leaf's entry / exit as a candidate hit (INF: out of the ray's range)
          T_in = t_in[j], E_in = c0[j], J_in = j;
        if (t_out[j] < T_out)
          T_out = t_out[j], E_out = c1[j], J_out = j;
      }
    }
    const auto leaves_containing = [&](float t) { // t finite
      uint32_t n = 0;
#pragma unroll
      for (int i = 0; i < LMAX; ++i) // a leaf past n_leaves keeps t_in = +inf: never counted
        n += (__float_as_uint(t_in[i] - t) & __float_as_uint(t - t_out[i])) >> 31;
      return n;
    };
    const bool valid_in = E_in < INF_F && leaves_containing(E_in) == (uint32_t)(n_leaves - 1);
    const bool valid_out = E_out < INF_F && leaves_containing(E_out) == (uint32_t)(n_leaves - 1);
    found = valid_in || valid_out; // both valid: T_in lies inside the leaf left first, so E_in < E_out
    best_t = valid_in ? E_in : E_out;
    best_leaf = valid_in ? J_in : J_out;
  } else if (top.tt == 3) { // uniform
    // The union of all leaves as the reference evaluates it: every boundary hit is valid, so the first hit is the smallest
    // candidate — same visiting order and tie rules as the table loop below (a later leaf takes a tie, a leaf's second root does not).
    best_t = INF_F;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
      if (j < n_leaves) {
        const bool first = c0[j] <= best_t && c0[j] < INF_F;
        best_t = first ? c0[j] : best_t, best_leaf = first ? j : best_leaf;
        const bool second = c1[j] < INF_F && (first ? c1[j] < best_t : c1[j] <= best_t);
        best_t = second ? c1[j] : best_t, best_leaf = second ? j : best_leaf;
      }
    }
    found = best_leaf >= 0;
  } else {
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    if (j < n_leaves) {
      const unsigned long long table = s.csg_leaf_tt[top.leaf_begin + j];
      bool first_of_leaf = true;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float t = r == 0 ? c0[j] : c1[j];
        const bool better = t < INF_F && (!found || (first_of_leaf ? (t <= best_t) : (t < best_t))); // ties between leaves go to the later leaf
        if (better) {
          uint32_t mask = 0;
          // t_in < t < t_out  <=>  t_in - t and t - t_out are both negative: the AND of their sign bits.  Subtractions, shifts
          // and logic ops issue at 2.3 cycles per wave64 on gfx950, compares and selects at 4.2 (issue_bench.hip).  Equal values
          // give +0 (not inside, as the strict compares say); an infinite bound gives an infinity of the right sign.
#pragma unroll
          for (int i = 0; i < LMAX; ++i)
            if (i != j) // a leaf's own bit is 0: t is one of its two bounds
              mask |= ((__float_as_uint(t_in[i] - t) & __float_as_uint(t - t_out[i])) >> (31 - i)) & (1u << i);
          if (LMAX <= 5 ? (((uint32_t)table >> mask) & 1u) != 0u : ((table >> mask) & 1ull) != 0ull) {
            found = true;
            best_t = t;
            best_leaf = j;
            first_of_leaf = false;
          }
        }
      }
    }
  }
  }
  if (found) {
    const int best_shape = s.csg_leaf[top.leaf_begin + best_leaf].shape;
    const float *inv = s.shape_w2l + 12 * best_shape;
    V3 o, d;
    if (diag)
      o = diag_point_fast(inv, ray_world.o), d = diag_vec_fast(inv, ray_world.d);
    else
      o = hom_point_fast(inv, ray_world.o), d = hom_vec_fast(inv, ray_world.d);
    const V3 p = mk3(fmaf(best_t, d.x, o.x), fmaf(best_t, d.y, o.y), fmaf(best_t, d.z, o.z));
    V3 n = (s.shape_kind[best_shape] == RT_SHAPE_SPHERE) ? p : mk3(0.f, 0.f, 1.f);
    n = scale3(n, copysignf(1.0f, -fmaf(n.x, d.x, fmaf(n.y, d.y, n.z * d.z))));
    best.t = best_t;
    best.shape = best_shape;
    best.object = top.object;
    best.local = p;
    if (diag) {
      best.point = diag_point_fast(s.shape_l2w + 12 * best_shape, p);
      best.normal = diag_vec_fast(inv, n);
    } else {
      best.point = hom_point_fast(s.shape_l2w + 12 * best_shape, p);
      best.normal = hom_transposed_fast(inv, n);
    }
  }
  return found;
}

// ---------------------------------------------------------------------------------------------
// World — include/shapes.hpp:520-604
// ---------------------------------------------------------------------------------------------

// Wave-uniform read of a scan record: the address is the same in every lane, so the 64-byte
// record is fetched by ONE scalar load (s_load_dwordx16 from the constant address space) and its
// fields are SGPR operands of the lanes' VALU instructions.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f32x16 *ConstRecV;
struct RecRegs {
  f32x16 v; // inv[0..11], object, shape, pad, pad
  RT_DEV float operator[](int k) const { return v[k]; }
  RT_DEV int object() const { return __float_as_int(v[12]); }
  RT_DEV int shape() const { return __float_as_int(v[13]); }
};
RT_DEV RecRegs load_rec(const ScanRec *recs, int i) {
  RecRegs r;
  r.v = ((ConstRecV)(uintptr_t)recs)[i];
  return r;
}

// Front hit t of a top-level sphere / plane (Sphere::ray_intersection, shapes.hpp:206-221;
// Plane::ray_intersection, shapes.hpp:322-341), reference operation order, nothing fused.
template <typename P> RT_DEV bool sphere_front_t(const P &inv, const Ray &ray_world, float &t) {
  Ray ray = ray_transform(ray_world, inv);
  float tb;
  return solve_ray_sphere(ray, t, tb) > 0;
}
template <typename P> RT_DEV bool plane_front_t(const P &inv, const Ray &ray_world, float &t) {
  Ray ray = ray_transform(ray_world, inv);
  return solve_ray_plane(ray, t);
}

struct Best {
  float t;
  int object, shape;
};

// One sphere test of the closest-hit scan.
//  FAST == false: the reference's arithmetic, operation for operation (deterministic modes).
//  FAST == true : the same quantities with fused multiply-adds and one reciprocal (path mode, whose
//                 parity bar is statistical): 33 VALU instructions up to the discriminant test
//                 instead of 52 flops unfused.  a > 0 and sqrt > 0 give t1 < t2 without the swap.
// skip_shape (FAST only): the sphere the ray leaves is not tested here but by self_sphere_exact below.
template <bool FAST> RT_DEV void sphere_scan_test(const RecRegs &r, const Ray &ray, Best &best, int skip_shape = -2) {
  if (FAST) {
    const RecRegs &m = r;
    const float dx = fmaf(m[0], ray.d.x, fmaf(m[1], ray.d.y, m[2] * ray.d.z));
    const float dy = fmaf(m[3], ray.d.x, fmaf(m[4], ray.d.y, m[5] * ray.d.z));
    const float dz = fmaf(m[6], ray.d.x, fmaf(m[7], ray.d.y, m[8] * ray.d.z));
    const float ox = fmaf(m[0], ray.o.x, fmaf(m[1], ray.o.y, fmaf(m[2], ray.o.z, m[9])));
    const float oy = fmaf(m[3], ray.o.x, fmaf(m[4], ray.o.y, fmaf(m[5], ray.o.z, m[10])));
    const float oz = fmaf(m[6], ray.o.x, fmaf(m[7], ray.o.y, fmaf(m[8], ray.o.z, m[11])));
    const float a = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const float b = fmaf(ox, dx, fmaf(oy, dy, oz * dz));
    const float c = fmaf(ox, ox, fmaf(oy, oy, fmaf(oz, oz, -1.f)));
    const float disc = fmaf(b, b, -(a * c));
    if (disc > 0.f) { // v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the correctly rounded library forms: ~20 instructions per sphere that is hit
      const float sq = __builtin_amdgcn_sqrtf(disc);
      const float inv_a = __builtin_amdgcn_rcpf(a);
      const float t1 = (-b - sq) * inv_a;
      const float t2 = (-b + sq) * inv_a;
      float t = t1;
      bool ok = true;
      if (t1 < ray.tmin) {
        t = t2;
        ok = t2 > ray.tmin && t2 < ray.tmax;
      }
      if (ok && t < best.t && r.shape() != skip_shape) {
        best.t = t;
        best.object = r.object();
        best.shape = r.shape();
      }
    }
  } else {
    float t;
    if (sphere_front_t(r, ray, t) && t < best.t) {
      best.t = t;
      best.object = r.object();
      best.shape = r.shape();
    }
  }
}

// The scan over the top-level spheres, in World::objects order (strict `<`: the lowest index wins
// ties).  Records ping-pong between two SGPR sets so that the scalar load of the next record is in
// flight while the lanes test the current one.
template <bool FAST> RT_DEV void scan_spheres(const ScanRec *recs, int n, const Ray &ray, Best &best, int skip_shape = -2) {
  if (n <= 0)
    return;
  RecRegs a = load_rec(recs, 0);
  int i = 0;
  for (; i + 1 < n; i += 2) {
    const RecRegs b = load_rec(recs, i + 1);
    sphere_scan_test<FAST>(a, ray, best, skip_shape);
    a = load_rec(recs, min(i + 2, n - 1));
    sphere_scan_test<FAST>(b, ray, best, skip_shape);
  }
  if (i < n)
    sphere_scan_test<FAST>(a, ray, best, skip_shape);
}

// The same scan for the small-world kernels whose shape tables are staged in LDS (rt_wavepath.h): the inverse matrix comes
// from the LDS table into VECTOR registers (three broadcast ds_read_b128) instead of scalar registers.  On gfx950 a v_fma_f32 /
// v_mul_f32 with a scalar-register operand issues at 4.2 cycles per wave64, with vector operands only at 2.3
// (scripts/microbench/issue_bench.hip): the 21 matrix multiply-adds of a test were a third slower than they need be.
// Only the two indices of a record are read through the scalar cache.  Same arithmetic, same order: same bits.
typedef const __attribute__((address_space(4))) int32_t *ConstI32;
struct TabRec { // one scan record of the small-world kernels: matrix in vector registers, indices in scalar registers
  float4 m0, m1, m2; // m[0..3] m[4..7] m[8..11]
  int object, shape;
};
RT_DEV TabRec load_tab_rec(const Scene &s, const ScanRec *recs, int i) {
  ConstI32 ids = (ConstI32)(uintptr_t)&recs[i].object; // {object, shape}: one s_load_dwordx2
  TabRec r;
  r.object = ids[0], r.shape = ids[1];
  const float4 *mp = reinterpret_cast<const float4 *>(s.shape_w2l + 12 * r.shape);
  r.m0 = mp[0], r.m1 = mp[1], r.m2 = mp[2];
  return r;
}
RT_DEV void sphere_tab_test(const TabRec &r, const Ray &ray, Best &best, bool diag) {
  const float4 m0 = r.m0, m1 = r.m1, m2 = r.m2;
  float dx, dy, dz, ox, oy, oz;
  if (diag) { // uniform (Scene::all_diag): the six terms below that multiply by zero are left out
    dx = m0.x * ray.d.x, dy = m1.x * ray.d.y, dz = m2.x * ray.d.z;
    ox = fmaf(m0.x, ray.o.x, m2.y), oy = fmaf(m1.x, ray.o.y, m2.z), oz = fmaf(m2.x, ray.o.z, m2.w);
  } else {
    dx = fmaf(m0.x, ray.d.x, fmaf(m0.y, ray.d.y, m0.z * ray.d.z));
    dy = fmaf(m0.w, ray.d.x, fmaf(m1.x, ray.d.y, m1.y * ray.d.z));
    dz = fmaf(m1.z, ray.d.x, fmaf(m1.w, ray.d.y, m2.x * ray.d.z));
    ox = fmaf(m0.x, ray.o.x, fmaf(m0.y, ray.o.y, fmaf(m0.z, ray.o.z, m2.y)));
    oy = fmaf(m0.w, ray.o.x, fmaf(m1.x, ray.o.y, fmaf(m1.y, ray.o.z, m2.z)));
    oz = fmaf(m1.z, ray.o.x, fmaf(m1.w, ray.o.y, fmaf(m2.x, ray.o.z, m2.w)));
  }
  const float a = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
  const float b = fmaf(ox, dx, fmaf(oy, dy, oz * dz));
  const float c = fmaf(ox, ox, fmaf(oy, oy, fmaf(oz, oz, -1.f)));
  const float disc = fmaf(b, b, -(a * c));
  if (disc > 0.f) {
    const float sq = __builtin_amdgcn_sqrtf(disc);
    const float inv_a = __builtin_amdgcn_rcpf(a);
    const float t1 = (-b - sq) * inv_a;
    const float t2 = (-b + sq) * inv_a;
    float t = t1;
    bool ok = true;
    if (t1 < ray.tmin) {
      t = t2;
      ok = t2 > ray.tmin && t2 < ray.tmax;
    }
    if (ok && t < best.t) {
      best.t = t;
      best.object = r.object;
      best.shape = r.shape;
    }
  }
}
RT_DEV void scan_spheres_tab(const Scene &s, const Ray &ray, Best &best) {
  const int n = s.n_spheres;
  if (n <= 0)
    return;
  const bool diag = s.all_diag != 0;
  TabRec a = load_tab_rec(s, s.spheres, 0); // the next record's reads are in flight while the lanes test the current one
  int i = 0;
  for (; i + 1 < n; i += 2) {
    const TabRec b = load_tab_rec(s, s.spheres, i + 1);
    sphere_tab_test(a, ray, best, diag);
    a = load_tab_rec(s, s.spheres, min(i + 2, n - 1));
    sphere_tab_test(b, ray, best, diag);
  }
  if (i < n)
    sphere_tab_test(a, ray, best, diag);
}

// The sphere a ray LEAVES is the one sphere whose test decides something systematic: a mirror ray (tmin = 1e-5) starts a
// rounding error away from the surface, and whether it finds the exit root of its own sphere beyond tmin is, in the
// reference's arithmetic, close to a coin flip for small far spheres (the root cancels ~4 digits: the reference re-hits on
// 22 % of the reflections off the 0.05-0.15 spheres of the synthetic scenes).  Each such re-hit costs a factor rho, so an
// arithmetic with another error distribution shifts the image: the fused world-space test renders the 4096-sphere scene
// 0.7 % brighter than the reference's own render (9 standard errors), the unfused kernel does not.  So the fused scans of
// the large-world kernels leave that one sphere out, and it is tested here with the reference's operation order — once
// per ray, against thousands of fused tests.
RT_DEV void self_sphere_exact(const Scene &s, int self_shape, const Ray &ray, Best &best) {
#ifdef RT_NO_SELF_EXACT
  return;
#endif
  if (self_shape < 0)
    return;
  const int4 org = reinterpret_cast<const int4 *>(s.shape_origin)[self_shape];
  if (!(org.w & 1))
    return; // a plane or a CSG leaf: tested by world_finish_hit
  float t;
  if (sphere_front_t(s.shape_inv + 12 * self_shape, ray, t) && (t < best.t || (t == best.t && org.x < best.object))) {
    best.t = t;
    best.object = org.x;
    best.shape = self_shape;
  }
}

// Second half of World::ray_intersection (shapes.hpp:547-561) once the spheres have been scanned:
// planes and CSG objects, ties to the lowest object index, full record for the winner only.
// The winner's shading record (rt_types.h ShadeRec), fetched in the same round trip as the matrices its hit record needs
// instead of one trip later by the shading code.
struct ShadeFetch {
  float4 r0, r1;
  bool have;
};

// EXACT_RECORD (the large-world FMA kernels): the fused scan only FINDS the winner; when it is a top-level sphere its t
// and its hit record are recomputed with the reference's unfused operation order, so that the point the next ray starts
// from carries the reference's rounding-error distribution (see self_sphere_exact: the brightness of a crowded scene
// depends on it).
// CSG = false: a kernel for worlds without CSG objects, compiled without the evaluators (they cost the small-world kernel
// a quarter of its register budget: 128 -> 94 VGPRs, one more wave per SIMD).
// TT4 = true: every tree of the world has a truth table and at most 4 leaves (the host checks): only that evaluator is
// compiled in (C4: 21 -> 9 spilled registers at 5 waves per SIMD, 43.8 -> 41.6 ms).
// TAB (small-world kernels, tables in LDS): the planes' matrix rows come from the LDS table into vector registers, as in scan_spheres_tab.
template <bool FAST, bool EXACT_RECORD = false, bool CSG = true, bool TT4 = false, bool TAB = false> RT_DEV Hit world_finish_hit(const Scene &s, const Ray &ray, Best best, ShadeFetch *fetch = nullptr) {
  if (fetch)
    fetch->have = false;
  for (int i = 0, n_planes = FAST ? s.n_planes : uniform_count(s.n_planes); i < n_planes; ++i) {
    if (FAST && TAB) {
      ConstI32 ids = (ConstI32)(uintptr_t)&s.planes[i].object;
      const int object = ids[0], shape = ids[1];
      const float *m = s.shape_w2l + 12 * shape;
      float oz, dz;
      if (s.all_diag) { // uniform
        oz = fmaf(m[8], ray.o.z, m[11]), dz = m[8] * ray.d.z;
      } else {
        oz = fmaf(m[6], ray.o.x, fmaf(m[7], ray.o.y, fmaf(m[8], ray.o.z, m[11])));
        dz = fmaf(m[6], ray.d.x, fmaf(m[7], ray.d.y, m[8] * ray.d.z));
      }
      const float t = -oz * __builtin_amdgcn_rcpf(dz);
      const bool hit = !are_close(dz, 0.f) && !(t < ray.tmin || t > ray.tmax);
      if (hit && (t < best.t || (t == best.t && object < best.object))) {
        best.t = t;
        best.object = object;
        best.shape = shape;
      }
      continue;
    }
    const RecRegs r = load_rec(s.planes, i);
    float t;
    bool hit;
    if (FAST) { // only the z row of the inverse matrix matters to Plane::ray_intersection
      const float oz = fmaf(r[6], ray.o.x, fmaf(r[7], ray.o.y, fmaf(r[8], ray.o.z, r[11])));
      const float dz = fmaf(r[6], ray.d.x, fmaf(r[7], ray.d.y, r[8] * ray.d.z));
      t = -oz * __builtin_amdgcn_rcpf(dz);
      hit = !are_close(dz, 0.f) && !(t < ray.tmin || t > ray.tmax);
    } else {
      hit = plane_front_t(r, ray, t);
    }
    if (hit && (t < best.t || (t == best.t && r.object() < best.object))) {
      best.t = t;
      best.object = r.object();
      best.shape = r.shape();
    }
  }
  Hit h;
  h.object = -1;
  h.shape = -1;
  h.t = 0.f;
  h.point = h.normal = h.local = mk3(0.f, 0.f, 0.f);
  bool csg_won = false;
  for (int i = 0, n_tops = !CSG ? 0 : FAST ? s.n_csg_tops : uniform_count(s.n_csg_tops); i < n_tops; ++i) {
    const CsgTop top = s.csg_tops[i];
    Hit c;
    bool csg_hit;
    if (FAST && TT4)
      csg_hit = csg_ray_intersection_tt<4>(s, top, ray, c);
    else if (FAST)
      csg_hit = top.tt ? (top.leaf_end - top.leaf_begin <= 4 ? csg_ray_intersection_tt<4>(s, top, ray, c) : csg_ray_intersection_tt<CSG_TT_MAX_LEAVES>(s, top, ray, c))
                       : csg_ray_intersection_fast(s, top, ray, c);
    else
      csg_hit = csg_ray_intersection(s, top, ray, c);
    if (csg_hit && (c.t < best.t || (c.t == best.t && c.object < best.object))) {
      best.t = c.t;
      best.object = c.object;
      h = c;
      csg_won = true;
    }
  }
  if (best.shape < 0 && !csg_won)
    return h;
  if (!csg_won) {
    h.object = best.object;
    h.shape = best.shape;
    h.t = best.t;
    if (fetch) {
      const float4 *rec = reinterpret_cast<const float4 *>(s.shape_shade + best.shape);
      fetch->r0 = rec[0], fetch->r1 = rec[1];
      fetch->have = true;
    }
    if (FAST && EXACT_RECORD && s.shape_kind[best.shape] == RT_SHAPE_SPHERE) {
      Ray local = ray_transform(ray, s.shape_inv + 12 * best.shape);
      float t_ref, t_far;
      if (solve_ray_sphere(local, t_ref, t_far) > 0) // a grazing winner the unfused test misses keeps the fused t
        h.t = t_ref;
      make_hit(s, best.shape, local, h.t, h.point, h.normal, h.local);
    } else if (FAST) {
      make_hit_fast(s, best.shape, ray, best.t, h.point, h.normal, h.local);
    } else {
      Ray local = ray_transform(ray, s.shape_inv + 12 * best.shape);
      make_hit(s, best.shape, local, best.t, h.point, h.normal, h.local);
    }
  }
  return h;
}

// World::ray_intersection — shapes.hpp:547-561.  Closest hit; ties go to the lowest object index
// (strict `<` in insertion order).
template <bool FAST, bool CSG = true, bool TT4 = false, bool TAB = false> RT_DEV Hit world_ray_intersection(const Scene &s, const Ray &ray) {
  Best best;
  best.t = INF_F;
  best.object = 0x7fffffff;
  best.shape = -1;
  if (FAST && TAB)
    scan_spheres_tab(s, ray, best);
  else
    scan_spheres<FAST>(s.spheres, FAST ? s.n_spheres : uniform_count(s.n_spheres), ray, best);
  return world_finish_hit<FAST, false, CSG, TT4, FAST && TAB>(s, ray, best);
}

// World::on_off_ray_intersection — shapes.hpp:567-578 (only the boolean is used, renderers.hpp:62-69)
RT_DEV bool world_any_hit(const Scene &s, const Ray &ray) {
  bool hit = false;
  for (int i = 0, n = uniform_count(s.n_spheres); i < n && !hit; ++i) {
    const RecRegs r = load_rec(s.spheres, i);
    float t;
    hit = sphere_front_t(r, ray, t);
  }
  for (int i = 0, n = uniform_count(s.n_planes); i < n && !hit; ++i) {
    const RecRegs r = load_rec(s.planes, i);
    float t;
    hit = plane_front_t(r, ray, t);
  }
  for (int i = 0, n = uniform_count(s.n_csg_tops); i < n && !hit; ++i) {
    Hit c;
    hit = csg_ray_intersection(s, s.csg_tops[i], ray, c);
  }
  return hit;
}

// World::offset_if_visible — shapes.hpp:584-603
RT_DEV bool world_offset_if_visible(const Scene &s, V3 viewer, V3 surface, V3 normal, V3 &in_dir_out) {
  V3 in_dir = sub3(surface, viewer);
  Ray in_ray;
  in_ray.o = viewer;
  in_ray.d = in_dir;
  in_ray.tmin = 1e-5f;
  in_ray.tmax = INF_F;
  if (dot3(in_dir, normal) > 0.f)
    return false;
  for (int i = 0, n = uniform_count(s.n_spheres); i < n; ++i) {
    const RecRegs r = load_rec(s.spheres, i);
    float t;
    if (sphere_front_t(r, in_ray, t) && t < 1.f) {
      const int shape = r.shape();
      Ray local = ray_transform(in_ray, s.shape_inv + 12 * shape);
      V3 p = hom_point(s.shape_fwd + 12 * shape, add3(local.o, lscale3(t, local.d)));
      if (!point_is_close(p, surface))
        return false;
    }
  }
  for (int i = 0, n = uniform_count(s.n_planes); i < n; ++i) {
    const RecRegs r = load_rec(s.planes, i);
    float t;
    if (plane_front_t(r, in_ray, t) && t < 1.f) {
      const int shape = r.shape();
      Ray local = ray_transform(in_ray, s.shape_inv + 12 * shape);
      V3 p = hom_point(s.shape_fwd + 12 * shape, add3(local.o, lscale3(t, local.d)));
      if (!point_is_close(p, surface))
        return false;
    }
  }
  for (int i = 0, n = uniform_count(s.n_csg_tops); i < n; ++i) {
    Hit c;
    if (csg_ray_intersection(s, s.csg_tops[i], in_ray, c) && c.t < 1.f && !point_is_close(c.point, surface))
      return false;
  }
  in_dir_out = in_dir;
  return true;
}

// ---------------------------------------------------------------------------------------------
// Materials — include/materials.hpp
// ---------------------------------------------------------------------------------------------

// UniformPigment / CheckeredPigment / ImagePigment::operator() — materials.hpp:64, 85-94, 120-137
// FAST (FMA path kernels): the checker cell from u * n instead of u / (1.f / n) — two IEEE divisions and a reciprocal
// less; the two forms can disagree only within an ulp of a cell edge.
template <bool FAST = false> RT_DEV V3 pigment_eval(const Scene &s, uint32_t pg, float u, float v) {
  const int kind = s.pigment_kind[pg];
  if (kind == RT_PIGMENT_CHECKERED) {
    int col, row;
    if (FAST) {
      const float n = (float)s.pigment_n_intervals[pg];
      col = (int)(u * n), row = (int)(v * n);
    } else {
      float subinterval = 1.f / (float)s.pigment_n_intervals[pg]; // materials.hpp:81
      col = (int)(u / subinterval);
      row = (int)(v / subinterval);
    }
    return (((col + row) & 1) == 0) ? ld3(s.pigment_color1 + 3 * pg) : ld3(s.pigment_color2 + 3 * pg);
  }
  if (kind == RT_PIGMENT_IMAGE) {
    int w = s.pigment_tex_width[pg], h = s.pigment_tex_height[pg];
    int col = (int)(u * (float)w);
    int row = (int)(v * (float)h);
    if (col >= w) col = w - 1;
    if (row >= h) row = h - 1;
    // a NaN coordinate (acos of a local z rounded a hair beyond 1 at a sphere's pole) converts to 0 on this hardware and to
    // INT_MIN on the reference's x86, where it then reads out of bounds (materials.hpp:120-137 clamps only upwards):
    // defined as index 0 here and in the oracle
    col = max(col, 0), row = max(row, 0);
    return ld3(s.texels + 3 * ((size_t)s.pigment_tex_offset[pg] + (size_t)row * (size_t)w + (size_t)col));
  }
  return ld3(s.pigment_color1 + 3 * pg);
}

// (u,v) of a hit, computed only when one of the material's pigments reads it
RT_DEV void hit_uv(const Scene &s, const Hit &h, uint32_t material, float &u, float &v) {
  u = 0.f;
  v = 0.f;
  if (s.material_needs_uv[material])
    shape_uv(s, h.shape, h.local, u, v);
}

// ONB(Vec) (Duff et al.) — geometry.hpp:591-599
RT_DEV void onb_from_e3(V3 e3, V3 &e1, V3 &e2) {
  float sign = copysignf(1.f, e3.z);
  const float a = -1.f / (sign + e3.z);
  const float b = e3.x * e3.y * a;
  e1 = mk3(1.f + sign * e3.x * e3.x * a, sign * b, -sign * e3.x);
  e2 = mk3(b, sign + e3.y * e3.y * a, -e3.y);
}

RT_DEV void onb_from_e3_fast(V3 e3, V3 &e1, V3 &e2) { // the same basis with v_rcp_f32 and fused products
  const float sign = copysignf(1.f, e3.z);
  const float a = -__builtin_amdgcn_rcpf(sign + e3.z);
  const float b = e3.x * e3.y * a;
  e1 = mk3(fmaf(sign * e3.x, e3.x * a, 1.f), sign * b, -sign * e3.x);
  e2 = mk3(b, fmaf(e3.y, e3.y * a, sign), -e3.y);
}

// DiffusiveBRDF::scatter_ray direction given the normalised normal — materials.hpp:199-211,
// PCG::random_phong(1) — random.hpp:75-84
// FAST: the same direction without the round trip through the angles — cos(acos(sqrt x)) = sqrt x,
// sin = sqrt(1 - x) — and sin/cos(2 pi y) from the hardware's v_sin_f32 / v_cos_f32, whose argument is
// in revolutions.
template <bool FAST> RT_DEV V3 diffuse_scatter_dir(V3 e3, Pcg &pcg) {
  V3 e1, e2;
  if (FAST)
    onb_from_e3_fast(e3, e1, e2);
  else
    onb_from_e3(e3, e1, e2);
  float x = pcg_random_float(pcg);
  float y = pcg_random_float(pcg);
  float st, ct, sp, cp;
  if (FAST) {
    ct = __builtin_amdgcn_sqrtf(x);
    st = __builtin_amdgcn_sqrtf(fmaxf(1.f - x, 0.f));
    sp = __builtin_amdgcn_sinf(y);
    cp = __builtin_amdgcn_cosf(y);
    const float sc = st * cp, ss = st * sp;
    return mk3(fmaf(e1.x, sc, fmaf(e2.x, ss, e3.x * ct)), fmaf(e1.y, sc, fmaf(e2.y, ss, e3.y * ct)), fmaf(e1.z, sc, fmaf(e2.z, ss, e3.z * ct)));
  } else {
    // pow(x, 1.f/(n+1)) with n = 1 is x^(1/2): sqrtf is correctly rounded, which is what glibc's powf
    // returns for this exponent (the device powf is not), so theta tracks the reference more closely.
    float theta = acosf(sqrtf(x));
    float phi = y * 2 * PI_F;
    st = sinf(theta), ct = cosf(theta);
    sp = sinf(phi), cp = cosf(phi);
  }
  V3 a = scale3(scale3(e1, st), cp);
  V3 b = scale3(scale3(e2, st), sp);
  V3 c = scale3(e3, ct);
  return add3(add3(a, b), c);
}

// SpecularBRDF::scatter_ray direction — materials.hpp:252-261
RT_DEV V3 specular_scatter_dir(V3 in_dir, V3 normal) {
  in_dir = normalized3(in_dir);
  V3 n = normalized3(normal);
  return sub3(in_dir, scale3(scale3(n, 2.f), dot3(n, in_dir)));
}
RT_DEV V3 specular_scatter_dir_fast(V3 in_dir, V3 normal) {
  const V3 d = normalized3_fast(in_dir), n = normalized3_fast(normal);
  const float k = -2.f * fmaf(n.x, d.x, fmaf(n.y, d.y, n.z * d.z));
  return mk3(fmaf(k, n.x, d.x), fmaf(k, n.y, d.y), fmaf(k, n.z, d.z));
}

} // namespace rtd
