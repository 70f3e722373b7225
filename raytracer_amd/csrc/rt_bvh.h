// rt_bvh.h — optional mode RT_FLAG_ACCEL_BVH: the closest hit over the top-level spheres through a bounding-volume
// hierarchy instead of the reference's test of every object (World::ray_intersection, shapes.hpp:547-561).
//
// Not in the reference; reported separately from the brute-force figures because it changes the algorithmic work.  The
// result is the linear scan's: a sphere is tested with the very arithmetic the scan uses, candidates compare as
// (t, object index), and the boxes are padded, so only the number of tests changes (exact mode: bit-identical images).
//
// Built for incoherent rays on CDNA4:
//   * per-lane traversal (after the first bounce the 64 rays of a wave share nothing), but NO per-lane stack: the tree
//     is stored in heap order (children of n = 2n, 2n + 1), so the pending far children are one bit per level
//     (`trail`) and a pop is two shifts and an xor — no scratch memory traffic at all;
//   * 16-byte nodes (boxes quantised outwards to 16 bits per bound), the first `lds_nodes` heap slots — the top of the
//     tree, every level of it for up to ~8000 spheres — staged in LDS: a divergent 16-byte LDS read costs a few
//     cycles per wave where the same read through the vector memory path costs one cache-line lookup per lane;
//   * both children of a node are fetched and tested together (adjacent slots), the nearer one is entered first;
//   * leaves hold up to BVH_LEAF_MAX spheres, regrouped by leaf (bvh_recs / bvh_iso), tested as one straight-line block;
//   * lanes of a wave that run out of work take over pending subtrees of the lanes that still have some
//     (bvh_closest_sphere).
#pragma once
#include "rt_wavepath.h"

namespace rtd {

// Dynamic LDS of the kernel: [heap slots 0 .. lds_nodes) [centre + r^2 of the first lds_spheres leaf positions].
// SPLIT == false: the whole heap is staged.  SPLIT == true: slots >= lds_nodes are read from HBM/L2; the LDS read is
// issued unconditionally (clamped) and the global one only by the lanes that need it, so that the compiler keeps two
// typed loads instead of one flat load through a selected pointer (which costs both memory paths on every fetch).
template <bool SPLIT> RT_DEV uint4 bvh_fetch(const Scene &s, int lds_nodes, int n) {
  extern __shared__ uint4 bvh_lds[];
  if (!SPLIT)
    return bvh_lds[n];
  uint4 v = bvh_lds[min(n, lds_nodes - 1)];
  if (n >= lds_nodes)
    v = reinterpret_cast<const uint4 *>(s.bvh)[n];
  return v;
}

// Slab test against a quantised box, in lattice coordinates: with o' = (o - origin) / scale, id' = scale / d and
// nc = -o' * id', the ray meets the plane q of an axis at t = fmaf(q, id', nc).  Which bound is the entry plane depends
// on the sign of d only, so each axis word (lo | hi << 16) is rotated per ray (rot = 0 or 16) and the low half is the
// entry, the high half the exit plane: 3 rotates + 6 conversions + 6 fma + max3 + min3 per box.
// Rounding: nc is off by 6e-8 |o' id'|, i.e. the plane is displaced by <= 6e-8 |o'| lattice steps (0.004 for an origin
// inside the scene box, 1 step at 250 scene diameters); the host rounds every box outwards by a whole extra step.
// fminf / fmaxf drop a NaN operand (inf - inf or 0 * inf on an axis the ray does not move along): conservative.
struct BvhRay {
  V3 id, nc;
  int rx, ry, rz;
};
RT_DEV BvhRay bvh_ray(const Scene &s, const Ray &ray) {
  BvhRay r;
  r.id = mk3(s.bvh_scale[0] / ray.d.x, s.bvh_scale[1] / ray.d.y, s.bvh_scale[2] / ray.d.z); // +-inf for a zero component
  r.nc = mk3(-((ray.o.x - s.bvh_origin[0]) * s.bvh_inv_scale[0]) * r.id.x, -((ray.o.y - s.bvh_origin[1]) * s.bvh_inv_scale[1]) * r.id.y,
             -((ray.o.z - s.bvh_origin[2]) * s.bvh_inv_scale[2]) * r.id.z);
  r.rx = (int)((__float_as_uint(r.id.x) >> 31) << 4);
  r.ry = (int)((__float_as_uint(r.id.y) >> 31) << 4);
  r.rz = (int)((__float_as_uint(r.id.z) >> 31) << 4);
  return r;
}
RT_DEV bool bvh_box_hit(const uint4 q, const BvhRay &r, float best_t, float &tnear) {
  const uint32_t x = __builtin_amdgcn_alignbit(q.x, q.x, r.rx), y = __builtin_amdgcn_alignbit(q.y, q.y, r.ry),
                 z = __builtin_amdgcn_alignbit(q.z, q.z, r.rz);
  const float nx = fmaf((float)(x & 0xffffu), r.id.x, r.nc.x), fx = fmaf((float)(x >> 16), r.id.x, r.nc.x);
  const float ny = fmaf((float)(y & 0xffffu), r.id.y, r.nc.y), fy = fmaf((float)(y >> 16), r.id.y, r.nc.y);
  const float nz = fmaf((float)(z & 0xffffu), r.id.z, r.nc.z), fz = fmaf((float)(z >> 16), r.id.z, r.nc.z);
  tnear = fmaxf(fmaxf(nx, ny), nz);
  const float tfar = fminf(fminf(fx, fy), fz);
  return tfar >= fmaxf(tnear, 0.f) && tnear <= best_t;
}

// The best candidate of a ray lives in LDS as one 64-bit key, t bits << 32 | leaf position (t > 0: the bits order like
// the numbers), because several lanes of the wave may be working on the same ray (bvh_closest_sphere).  Candidates
// compare as (t, object index) — World::ray_intersection's strict `<` in insertion order; the object index is only
// fetched to break a tie.  Whatever the order in which lanes offer candidates, the key ends at their minimum.
constexpr unsigned long long BVH_NO_HIT = 0x7f800000ffffffffull; // t = inf, position -1
RT_DEV void bvh_accept(const Scene &s, unsigned long long *slot, int pos, float t, float &bt) {
  unsigned long long old = *reinterpret_cast<volatile unsigned long long *>(slot);
  for (;;) {
    const float ot = __uint_as_float((uint32_t)(old >> 32));
    const int opos = (int)(uint32_t)old;
    const bool better = t < ot || (t == ot && opos != pos && s.bvh_recs[pos].object < s.bvh_recs[opos].object); // t == ot: opos >= 0
    if (!better) {
      bt = fminf(bt, ot);
      return;
    }
    const unsigned long long mine = ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)pos;
    const unsigned long long prev = atomicCAS(slot, old, mine);
    if (prev == old) {
      bt = t;
      return;
    }
    old = prev;
  }
}

// Front hit of the ISO form (rt_wide.h): sphere given by world centre + r^2; a = D.D
RT_DEV bool bvh_iso_front_t(const float4 c, const Ray &ray, float a, float inv_a, float &t) {
  const float ocx = ray.o.x - c.x, ocy = ray.o.y - c.y, ocz = ray.o.z - c.z;
  const float b = fmaf(ocx, ray.d.x, fmaf(ocy, ray.d.y, ocz * ray.d.z));
  const float cc = fmaf(ocx, ocx, fmaf(ocy, ocy, fmaf(ocz, ocz, -c.w)));
  const float disc = fmaf(b, b, -(a * cc));
  if (!(disc > 0.f))
    return false;
  const float sq = sqrtf(disc);
  const float t1 = (-b - sq) * inv_a, t2 = (-b + sq) * inv_a;
  t = t1;
  if (t1 < ray.tmin) {
    t = t2;
    return t2 > ray.tmin && t2 < ray.tmax;
  }
  return true;
}

// Front hit of the general fused form (sphere_scan_test<true>)
RT_DEV bool bvh_gen_front_t(const float *m, const Ray &ray, float &t) {
  const float dx = fmaf(m[0], ray.d.x, fmaf(m[1], ray.d.y, m[2] * ray.d.z));
  const float dy = fmaf(m[3], ray.d.x, fmaf(m[4], ray.d.y, m[5] * ray.d.z));
  const float dz = fmaf(m[6], ray.d.x, fmaf(m[7], ray.d.y, m[8] * ray.d.z));
  const float ox = fmaf(m[0], ray.o.x, fmaf(m[1], ray.o.y, fmaf(m[2], ray.o.z, m[9])));
  const float oy = fmaf(m[3], ray.o.x, fmaf(m[4], ray.o.y, fmaf(m[5], ray.o.z, m[10])));
  const float oz = fmaf(m[6], ray.o.x, fmaf(m[7], ray.o.y, fmaf(m[8], ray.o.z, m[11])));
  const float qa = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
  const float qb = fmaf(ox, dx, fmaf(oy, dy, oz * dz));
  const float qc = fmaf(ox, ox, fmaf(oy, oy, fmaf(oz, oz, -1.f)));
  const float disc = fmaf(qb, qb, -(qa * qc));
  if (!(disc > 0.f))
    return false;
  const float sq = sqrtf(disc), inv_a = 1.f / qa;
  const float t1 = (-qb - sq) * inv_a, t2 = (-qb + sq) * inv_a;
  t = t1;
  if (t1 < ray.tmin) {
    t = t2;
    return t2 > ray.tmin && t2 < ray.tmax;
  }
  return true;
}

#ifdef RT_BVH_STATS
struct BvhStats { unsigned inner, leaf, pop, wave_iters, wave_leaf_iters, rays, steals; };
#define RT_BVH_COUNT(x) (++(x))
#else
struct BvhStats {};
#define RT_BVH_COUNT(x)
#endif

// Closest sphere of the tree for the rays of one WAVE: lane l enters with its own ray (or none) and leaves with the key
// of its own ray in res[l].  All 64 lanes must call it (wave-uniform control flow around the call).
//
// "While-while": every lane first walks (down, or back up through `trail`) until it sits on a leaf or is done — LDS
// reads only —, then the lanes that reached a leaf test its spheres together, so the long-latency sphere loads are paid
// once per leaf visit of the slowest lane, not once per step.
//
// Subtree stealing inside the wave: the slowest of 64 incoherent rays takes 2-4x the mean number of steps, and a wave
// runs until its slowest lane is done.  So each time the lanes meet (before a walk), lanes without work take over
// pending subtrees of lanes that have some: the donor gives away the SHALLOWEST sibling still on its trail (the
// biggest piece of work, two bit operations to detach), the thief gets the ray and the subtree root through wave
// shuffles and traverses it with an empty trail, which keeps it inside that subtree.  Both feed the ray's key in LDS
// and refresh their pruning bound from it.  No counters are needed: the wave leaves when none of its lanes has work,
// and then every key is final.
template <bool FAST, bool SPLIT, bool LEAF_LDS>
RT_DEV void bvh_closest_sphere(const Scene &s, const RenderParams &p, const Ray &own_ray, int own_self_pos, bool active, unsigned long long *res,
                               int *donor_lane, BvhStats &st) {
  extern __shared__ uint4 bvh_lds[];
  const int lds_nodes = p.bvh_lds_nodes;
  const bool iso = FAST && p.bvh_iso != 0;
  const int lane = (int)(threadIdx.x & 63);
  const unsigned long long lanes_before = (1ull << lane) - 1ull;

  // the traversal this lane is running: ray, its lattice form, position in the tree, owner's key
  V3 o = own_ray.o, d = own_ray.d;
  float tmin = own_ray.tmin;
  BvhRay br = bvh_ray(s, own_ray);
  float a = fmaf(d.x, d.x, fmaf(d.y, d.y, d.z * d.z));
  float inv_a = 1.f / a;
  int owner = lane;
  int self_pos = own_self_pos; // leaf position of the sphere the ray leaves (FAST: not tested here, see self_sphere_exact); -1: none
  float bt = INF_F;
  int n = 1, level = 0;
  uint32_t trail = 0; // bit L: the sibling of this path's node at level L is still to be visited
  uint4 cur = bvh_fetch<SPLIT>(s, lds_nodes, 1);
  float tn;
  bool valid = active && bvh_box_hit(cur, br, bt, tn);
  bool alive = valid;  // has a node to process or a trail to follow
  bool shared = false; // other lanes work on the same ray: bt must be refreshed from the key
  res[lane] = BVH_NO_HIT;
  if (active)
    RT_BVH_COUNT(st.rays);

  while (__any(alive)) {
    // ---- lanes without work take over pending subtrees ----------------------------------------------------------
    {
      const unsigned long long idle_m = __ballot(!alive), donor_m = __ballot(alive && trail != 0u);
      const int n_pairs = min(__popcll(idle_m), __popcll(donor_m));
      if (n_pairs > 0) {
        const int idle_rank = __popcll(idle_m & lanes_before), donor_rank = __popcll(donor_m & lanes_before);
        const bool is_donor = alive && trail != 0u && donor_rank < n_pairs;
        const bool is_thief = !alive && idle_rank < n_pairs;
        int give_node = 0, give_level = 0;
        if (is_donor) {
          give_level = __ffs((int)trail) - 1; // the shallowest pending level
          trail &= trail - 1u;
          give_node = (n >> (level - give_level)) ^ 1;
          shared = true;
          donor_lane[donor_rank] = lane;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // LDS is in order within a wave; this pins the compiler
        __builtin_amdgcn_wave_barrier();
        const int src = is_thief ? *reinterpret_cast<volatile int *>(donor_lane + idle_rank) : lane;
        const int g_node = __shfl(give_node, src, 64), g_level = __shfl(give_level, src, 64), g_owner = __shfl(owner, src, 64);
        const int g_self = __shfl(self_pos, src, 64);
        const float g_bt = __shfl(bt, src, 64), g_tmin = __shfl(tmin, src, 64), g_a = __shfl(a, src, 64), g_inv_a = __shfl(inv_a, src, 64);
        const V3 g_o = mk3(__shfl(o.x, src, 64), __shfl(o.y, src, 64), __shfl(o.z, src, 64));
        const V3 g_d = mk3(__shfl(d.x, src, 64), __shfl(d.y, src, 64), __shfl(d.z, src, 64));
        const V3 g_id = mk3(__shfl(br.id.x, src, 64), __shfl(br.id.y, src, 64), __shfl(br.id.z, src, 64));
        const V3 g_nc = mk3(__shfl(br.nc.x, src, 64), __shfl(br.nc.y, src, 64), __shfl(br.nc.z, src, 64));
        if (is_thief) {
          RT_BVH_COUNT(st.steals);
          o = g_o, d = g_d, tmin = g_tmin, a = g_a, inv_a = g_inv_a;
          br.id = g_id, br.nc = g_nc;
          br.rx = (int)((__float_as_uint(g_id.x) >> 31) << 4);
          br.ry = (int)((__float_as_uint(g_id.y) >> 31) << 4);
          br.rz = (int)((__float_as_uint(g_id.z) >> 31) << 4);
          owner = g_owner;
          self_pos = g_self;
          bt = g_bt;
          n = g_node, level = g_level, trail = 0u;
          cur = bvh_fetch<SPLIT>(s, lds_nodes, n);
          valid = bvh_box_hit(cur, br, bt, tn);
          alive = valid;
          shared = true;
        }
      }
    }
    if (alive && shared)
      bt = fminf(bt, __uint_as_float((uint32_t)(*reinterpret_cast<volatile unsigned long long *>(res + owner) >> 32)));

    // ---- walk ----------------------------------------------------------------------------------------------------
    while (alive && !(valid && cur.w)) {
#ifdef RT_BVH_STATS
      if (lane == __ffsll((long long)__ballot(1)) - 1) // first active lane
        st.wave_iters += 64;
#endif
      if (valid) { // inner node: both children, nearer one first
        RT_BVH_COUNT(st.inner);
        const uint4 c0 = bvh_fetch<SPLIT>(s, lds_nodes, 2 * n), c1 = bvh_fetch<SPLIT>(s, lds_nodes, 2 * n + 1);
        float t0, t1;
        const bool h0 = bvh_box_hit(c0, br, bt, t0);
        const bool h1 = bvh_box_hit(c1, br, bt, t1);
        if (h0 || h1) {
          const bool go1 = h1 && (!h0 || t1 < t0);
          n = 2 * n + (go1 ? 1 : 0);
          cur = go1 ? c1 : c0;
          ++level;
          if (h0 && h1)
            trail |= 1u << level;
        } else {
          valid = false;
        }
      }
      if (!valid) {
        if (!trail) {
          alive = false;
          break;
        }
        RT_BVH_COUNT(st.pop);
        const int L = 31 - __clz((int)trail); // the deepest pending level
        trail ^= 1u << L;
        n = (n >> (level - L)) ^ 1;
        level = L;
        cur = bvh_fetch<SPLIT>(s, lds_nodes, n);
        valid = bvh_box_hit(cur, br, bt, tn); // bt may have shrunk since the bit was set
      }
    }

    // ---- leaves --------------------------------------------------------------------------------------------------
    if (alive) { // valid && cur.w
      RT_BVH_COUNT(st.leaf);
#ifdef RT_BVH_STATS
      if (lane == __ffsll((long long)__ballot(1)) - 1)
        st.wave_leaf_iters += 64;
#endif
      const int first = (int)(cur.w & 0xffffffu), count = (int)(cur.w >> 24);
      unsigned long long *slot = res + owner;
      if (iso) {
        // all discriminants as one straight-line block, then one trip per sphere actually hit (mostly one)
        float disc[BVH_LEAF_MAX], b[BVH_LEAF_MAX];
        float4 c[BVH_LEAF_MAX];
#pragma unroll
        for (int k = 0; k < BVH_LEAF_MAX; ++k) { // a short leaf tests its last sphere again: harmless
          const int pos = first + min(k, count - 1);
          c[k] = LEAF_LDS ? reinterpret_cast<const float4 *>(bvh_lds + lds_nodes)[pos] : reinterpret_cast<const float4 *>(s.bvh_iso)[pos];
        }
        uint32_t hits = 0;
#pragma unroll
        for (int k = 0; k < BVH_LEAF_MAX; ++k) {
          const float ocx = o.x - c[k].x, ocy = o.y - c[k].y, ocz = o.z - c[k].z;
          b[k] = fmaf(ocx, d.x, fmaf(ocy, d.y, ocz * d.z));
          const float cc = fmaf(ocx, ocx, fmaf(ocy, ocy, fmaf(ocz, ocz, -c[k].w)));
          disc[k] = fmaf(b[k], b[k], -(a * cc));
          hits |= (disc[k] > 0.f ? 1u : 0u) << k;
        }
        while (hits) {
          const int k = __ffs((int)hits) - 1;
          hits &= hits - 1;
          float dk = disc[0], bk = b[0];
#pragma unroll
          for (int j = 1; j < BVH_LEAF_MAX; ++j)
            if (k == j)
              dk = disc[j], bk = b[j];
          const float sq = sqrtf(dk);
          const float t1 = (-bk - sq) * inv_a, t2 = (-bk + sq) * inv_a;
          float t = t1;
          bool ok = true;
          if (t1 < tmin) {
            t = t2;
            ok = t2 > tmin; // tmax = inf for every ray of the path kernels
          }
          if (ok && t <= bt && first + min(k, count - 1) != self_pos)
            bvh_accept(s, slot, first + min(k, count - 1), t, bt);
        }
      } else {
        Ray ray;
        ray.o = o, ray.d = d, ray.tmin = tmin, ray.tmax = INF_F;
        for (int k = 0; k < count; ++k) {
          float t;
          const bool hit = FAST ? bvh_gen_front_t(s.bvh_recs[first + k].inv, ray, t) : sphere_front_t(s.bvh_recs[first + k].inv, ray, t);
          if (hit && t <= bt && first + k != self_pos)
            bvh_accept(s, slot, first + k, t, bt);
        }
      }
      valid = false;
      alive = trail != 0u;
    }
  }
}

// The small-world path kernel (rt_kernels.h) with the sphere scan replaced; T lanes share one staged copy of the tree
// (or of its top levels).  Dynamic LDS, in uint4 units:
//   [lds_nodes heap slots] [lds_spheres leaf spheres] [T / 2: one key per ray] [T / 4: donor lanes, 64 per wave]
template <bool FAST, int T, bool SPLIT, bool LEAF_LDS>
__global__ void __launch_bounds__(T, 4) k_render_path_bvh(Scene s, RenderParams p) {
  extern __shared__ uint4 bvh_lds[];
  __shared__ WaveShared sh_all[T / 64]; // one wave-run pool per wave (rt_wavepath.h): after the staging below the waves never meet
  for (int i = threadIdx.x; i < p.bvh_lds_nodes; i += T)
    bvh_lds[i] = reinterpret_cast<const uint4 *>(s.bvh)[i];
  if (LEAF_LDS)
    for (int i = threadIdx.x; i < p.bvh_lds_spheres; i += T)
      bvh_lds[p.bvh_lds_nodes + i] = reinterpret_cast<const uint4 *>(s.bvh_iso)[i];
  const int wave = (int)(threadIdx.x >> 6);
  unsigned long long *const res = reinterpret_cast<unsigned long long *>(bvh_lds + p.bvh_lds_nodes + p.bvh_lds_spheres) + 64 * wave;
  int *const donor_lane = reinterpret_cast<int *>(bvh_lds + p.bvh_lds_nodes + p.bvh_lds_spheres + T / 2) + 64 * wave;
  WaveShared &sh = sh_all[wave];
  WavePath<FAST> B;
  B.init(p, sh, (int)blockIdx.x * (T / 64) + wave);
  __syncthreads(); // the tree is in LDS
  BvhStats st{};
  while (B.deal(s, p, sh)) {
    // FAST: the sphere a ray leaves is kept out of the fused tests and tested with the reference's arithmetic below
    const int self_pos = FAST && B.active && B.origin_shape >= 0 ? s.shape_origin[4 * B.origin_shape + 2] : -1;
    bvh_closest_sphere<FAST, SPLIT, LEAF_LDS>(s, p, B.ray, self_pos, B.active, res, donor_lane, st);
    Hit h;
    h.object = h.shape = -1;
    h.t = 0.f;
    h.point = h.normal = h.local = mk3(0.f, 0.f, 0.f);
    ShadeFetch fetch;
    fetch.have = false;
    if (B.active) {
      Best best;
      best.t = INF_F;
      best.object = 0x7fffffff;
      best.shape = -1;
      const unsigned long long key = *reinterpret_cast<volatile unsigned long long *>(res + (threadIdx.x & 63));
      if ((int)(uint32_t)key >= 0) {
        const int2 ids = *reinterpret_cast<const int2 *>(&s.bvh_recs[(int)(uint32_t)key].object); // {object, shape}
        best.t = __uint_as_float((uint32_t)(key >> 32));
        best.object = ids.x;
        best.shape = ids.y;
      }
      // the spheres kept out of the tree, in object order
      for (int k = 0; k < s.n_bvh_big; ++k) {
        const RecRegs rec = load_rec(s.bvh_big, k);
        Best one;
        one.t = INF_F;
        one.object = 0x7fffffff;
        one.shape = -1;
        sphere_scan_test<FAST>(rec, B.ray, one, FAST ? B.origin_shape : -2);
        if (one.shape >= 0 && (one.t < best.t || (one.t == best.t && one.object < best.object)))
          best = one;
      }
      if (FAST)
        self_sphere_exact(s, B.origin_shape, B.ray, best);
      h = world_finish_hit<FAST, true>(s, B.ray, best, &fetch);
    }
    B.shade(s, p, sh, h, &fetch);
  }
  wave_add_tally(p.counters, 0, B.rays);
#ifdef RT_BVH_STATS
  wave_add_counter(p.counters + 4, st.inner);
  wave_add_counter(p.counters + 5, st.leaf);
  wave_add_counter(p.counters + 6, st.pop);
  wave_add_counter(p.counters + 7, st.wave_iters);
  wave_add_counter(p.counters + 8, st.rays);
  wave_add_counter(p.counters + 9, st.wave_leaf_iters);
  wave_add_counter(p.counters + 10, st.steals);
#endif
}

} // namespace rtd
