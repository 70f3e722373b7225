// rt_wide.h — the path kernel for large worlds: spheres in registers, rays through an LDS queue.
#pragma once
#include "rt_wavepath.h"

namespace rtd {

// ---------------------------------------------------------------------------------------------
// Path kernel for large worlds ("wide"): the closest-hit scan is transposed.
//
//   * The top-level spheres live in REGISTERS, spread over the lanes of the workgroup: lane l of
//     wave w holds K spheres per pass (64 * WAVES * K spheres per pass; more passes for bigger worlds).
//     Nothing is streamed per test: the scalar cache cannot feed 64-byte records to every wave at the
//     rate the VALU consumes them (measured: the record stream saturates near 1.4 TB/s chip-wide).
//   * The rays of the workgroup's samples are compacted into an LDS queue (ballot + prefix count);
//     every wave reads each queued ray as an LDS broadcast and tests it against its own spheres.
//     A lane that finds a hit merges {t, object index} into the ray's 64-bit LDS key with one
//     ds_min_u64: t > 0, so the float bits order like the numbers, and equal t falls to the lowest
//     object index exactly as World::ray_intersection's strict `<` does.
//   * Then every lane is a ray again: it completes the query (planes, CSG, hit record) and shades the
//     node; scattering hits push their node onto the workgroup's pool (rt_path.h), which deals the
//     next rays.
//
//   ISO: all spheres are translation * uniform scaling, tested in world space from (centre, r^2):
//        oc = O - C; b = oc.D; c = oc.oc - r^2; disc = b^2 - (D.D) c      (12 VALU per test)
//   otherwise the general affine form of sphere_scan_test<true> (inverse matrix, 28 VALU per test).
// ---------------------------------------------------------------------------------------------
// Workgroup = W waves; lane l of wave w holds K rows of 64 spheres.  One ray costs W * (12 K + c) wave
// instructions (c ~ 25: LDS ray read, loop, max, branch), so W is kept as small as the register budget
// allows: W * K >= number of 64-sphere rows, K <= 8 (ISO) / 4 (general) so that the kernel fits 128 VGPRs.

// LDS image of one queued ray: 3 x 16 bytes; the scan reads the first two.
//
// GEN form: the ray as it is.  ISO form: what the world-space FILTER needs.  Since the winner's t and hit record are
// recomputed in the reference's arithmetic afterwards (world_finish_hit<true, true>) and the sphere a ray leaves is tested
// apart (self_sphere_exact), the scan only has to FIND the closest sphere, so it may use the cheapest form that ranks
// candidates correctly: with a unit direction u = D / |D| and everything that depends on the ray alone or on the sphere
// alone hoisted out,
//     b = O.u - C.u                 3 fma      (O.u per ray)
//     c = (O.O + k) - 2 O.C         1 add + 3 fma      (O.O and -2 O per ray, k = C.C - r^2 per sphere)
//     disc = b^2 - c                1 fma
// 8 instructions per test instead of 11.  The price is cancellation in c (|O|^2 + |C|^2 against r^2: an absolute error of
// a few 1e-6, i.e. the silhouette of a sphere of radius 0.05 is fuzzy by ~0.1 % of its radius, sign at random) — harmless
// to a filter, and the reason this form was NOT usable while the scan's own t placed the next ray's origin.
template <bool ISO> struct WideRayT;
template <> struct alignas(16) WideRayT<false> {
  float ox, oy, oz, tmin;
  float dx, dy, dz, tmax;
  float a, inv_a;             // D.D and its reciprocal
  int self;                   // position in the register rows of the sphere the ray leaves (-1: none): left to self_sphere_exact
  float pad1;
  RT_DEV static WideRayT make(const Ray &ray, int self) {
    WideRayT r;
    r.ox = ray.o.x, r.oy = ray.o.y, r.oz = ray.o.z, r.tmin = ray.tmin;
    r.dx = ray.d.x, r.dy = ray.d.y, r.dz = ray.d.z, r.tmax = ray.tmax;
    r.a = fmaf(r.dx, r.dx, fmaf(r.dy, r.dy, r.dz * r.dz));
    r.inv_a = 1.f / r.a;
    r.self = self;
    r.pad1 = 0.f;
    return r;
  }
};
template <> struct alignas(16) WideRayT<true> {
  float ux, uy, uz, ou;       // unit direction, O.u
  float m2ox, m2oy, m2oz, oo; // -2 O, O.O
  float tmin_u, inv_len;      // tmin as a distance; 1 / |D| (ray parameter = distance / |D|).  tmax is infinite for every ray of the path kernels.
  int self;
  float tmin;                 // the ray's own tmin (not read by the scan: the lane takes its ray back from here afterwards)
  RT_DEV static WideRayT make(const Ray &ray, int self) {
    WideRayT r;
    const float a = fmaf(ray.d.x, ray.d.x, fmaf(ray.d.y, ray.d.y, ray.d.z * ray.d.z));
    r.inv_len = __builtin_amdgcn_rsqf(a);
    r.ux = ray.d.x * r.inv_len, r.uy = ray.d.y * r.inv_len, r.uz = ray.d.z * r.inv_len;
    r.ou = fmaf(ray.o.x, r.ux, fmaf(ray.o.y, r.uy, ray.o.z * r.uz));
    r.m2ox = -2.f * ray.o.x, r.m2oy = -2.f * ray.o.y, r.m2oz = -2.f * ray.o.z;
    r.oo = fmaf(ray.o.x, ray.o.x, fmaf(ray.o.y, ray.o.y, ray.o.z * ray.o.z));
    r.tmin_u = ray.tmin * a * r.inv_len; // tmin * |D|
    r.self = self;
    r.tmin = ray.tmin;
    return r;
  }
  // The ray's origin and tmin, bit for bit (x -> -2x -> -0.5 * that is exact): the lane does not carry them across the scan.
  RT_DEV void restore(Ray &ray) const {
    ray.o = mk3(-0.5f * m2ox, -0.5f * m2oy, -0.5f * m2oz);
    ray.tmin = tmin;
  }
};

template <bool ISO, int K> struct WideSlots;
template <int K> struct WideSlots<true, K> {
  float cx[K], cy[K], cz[K], kk[K]; // centre, C.C - r^2
  RT_DEV void load(const Scene &s, int k, int sphere, bool valid) {
    if (valid) {
      const float4 v = reinterpret_cast<const float4 *>(s.wide_iso)[sphere];
      cx[k] = v.x, cy[k] = v.y, cz[k] = v.z, kk[k] = v.w;
    } else { // empty slot: k = +inf makes c = +inf and disc = -inf: never > 0, whatever the ray
      cx[k] = cy[k] = cz[k] = 0.f;
      kk[k] = INF_F;
    }
  }
  RT_DEV float disc(int k, const WideRayT<true> &r, float &b) const {
    b = fmaf(-cx[k], r.ux, fmaf(-cy[k], r.uy, fmaf(-cz[k], r.uz, r.ou)));
    const float c = fmaf(cx[k], r.m2ox, fmaf(cy[k], r.m2oy, fmaf(cz[k], r.m2oz, r.oo + kk[k])));
    return fmaf(b, b, -c);
  }
  // front root as a ray parameter (the reference's rule: t1 unless it is below tmin, then t2 if beyond tmin).  b is
  // recomputed (3 fma) rather than kept from the discriminant block: K registers less across the block, which is what
  // lets the one-wave kernel fit 5 waves per SIMD without spilling inside the scan loop.
  RT_DEV bool front_t(int k, const WideRayT<true> &r, float, float disc, float &t) const {
    float ou = r.ou;
    asm volatile("" : "+v"(ou)); // a fresh value: keeps the compiler from carrying the block's b[k] here instead
    const float b = fmaf(-cx[k], r.ux, fmaf(-cy[k], r.uy, fmaf(-cz[k], r.uz, ou)));
    const float sq = __builtin_amdgcn_sqrtf(disc);
    const float u1 = -b - sq, u2 = -b + sq;
    float u = u1;
    bool ok = true;
    if (u1 < r.tmin_u) {
      u = u2;
      ok = u2 > r.tmin_u;
    }
    t = u * r.inv_len;
    return ok;
  }
};
template <int K> struct WideSlots<false, K> {
  float m[K][12];
  float a_[K];
  RT_DEV void load(const Scene &s, int k, int sphere, bool valid) {
    if (valid) {
      const float4 *v = reinterpret_cast<const float4 *>(s.wide_recs[sphere].inv);
      const float4 v0 = v[0], v1 = v[1], v2 = v[2];
      m[k][0] = v0.x, m[k][1] = v0.y, m[k][2] = v0.z, m[k][3] = v0.w;
      m[k][4] = v1.x, m[k][5] = v1.y, m[k][6] = v1.z, m[k][7] = v1.w;
      m[k][8] = v2.x, m[k][9] = v2.y, m[k][10] = v2.z, m[k][11] = v2.w;
    } else { // empty slot: zero matrix, origin mapped to (2,0,0): a = b = 0, c = 3, disc = 0
#pragma unroll
      for (int j = 0; j < 12; ++j)
        m[k][j] = 0.f;
      m[k][9] = 2.f;
    }
  }
  RT_DEV float disc(int k, const WideRayT<false> &r, float &b) {
    const float *q = m[k];
    const float dx = fmaf(q[0], r.dx, fmaf(q[1], r.dy, q[2] * r.dz));
    const float dy = fmaf(q[3], r.dx, fmaf(q[4], r.dy, q[5] * r.dz));
    const float dz = fmaf(q[6], r.dx, fmaf(q[7], r.dy, q[8] * r.dz));
    const float ox = fmaf(q[0], r.ox, fmaf(q[1], r.oy, fmaf(q[2], r.oz, q[9])));
    const float oy = fmaf(q[3], r.ox, fmaf(q[4], r.oy, fmaf(q[5], r.oz, q[10])));
    const float oz = fmaf(q[6], r.ox, fmaf(q[7], r.oy, fmaf(q[8], r.oz, q[11])));
    const float a = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    b = fmaf(ox, dx, fmaf(oy, dy, oz * dz));
    const float c = fmaf(ox, ox, fmaf(oy, oy, fmaf(oz, oz, -1.f)));
    a_[k] = a;
    return fmaf(b, b, -(a * c));
  }
  RT_DEV bool front_t(int k, const WideRayT<false> &r, float b, float disc, float &t) const {
    const float sq = __builtin_amdgcn_sqrtf(disc), inv_a = __builtin_amdgcn_rcpf(a_[k]);
    const float t1 = (-b - sq) * inv_a, t2 = (-b + sq) * inv_a;
    t = t1;
    bool ok = true;
    if (t1 < r.tmin) {
      t = t2;
      ok = t2 > r.tmin && t2 < r.tmax;
    }
    return ok;
  }
};

// One queued ray against a lane's K spheres.  The K discriminants are one straight-line block (no branch
// per sphere); the rare hit work sits behind a single wave-level branch.  A hit merges into the ray's key.
template <bool ISO, int K, int W>
RT_DEV void wide_test_ray(WideSlots<ISO, K> &slots, const WideRayT<ISO> &r, unsigned long long *key_slot, int first, int wave, int lane) {
  float disc[K], b[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
    disc[k] = slots.disc(k, r, b[k]);
  float dmax = disc[0];
#pragma unroll
  for (int k = 1; k < K; ++k)
    dmax = fmaxf(dmax, disc[k]);
  if (dmax > 0.f) {
    float best_t = INF_F;
    int best_k = -1;
    const int self_rel = r.self - first - wave * 64 - lane; // == k * W * 64 for the row that holds the ray's own sphere
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (disc[k] > 0.f && self_rel != k * W * 64) {
        float t;
        if (slots.front_t(k, r, b[k], disc[k], t) && t < best_t) { // ascending k = ascending object index within a lane
          best_t = t;
          best_k = k;
        }
      }
    }
    if (best_k >= 0) {
      const int sphere = first + (best_k * W + wave) * 64 + lane;
      const unsigned long long key = ((unsigned long long)__float_as_uint(best_t) << 32) | (unsigned)sphere;
      atomicMin(key_slot, key);
    }
  }
}

// Register budget: ISO rows cost 4 VGPRs, general rows 13.  Where the rows leave room, ask for 4 waves per
// SIMD (<= 128 VGPRs), so that several workgroups share a CU and one's shading overlaps another's scan.
// At 5 waves per SIMD (96 VGPRs) the one-wave kernel parks part of its lane state in scratch around the scan (the compiler's
// choice of what: throughput, random state, slot, depth).  What the scan itself does not need was taken out of its way:
// b[k] is recomputed in the hit path (-8 registers), origin and tmin come back from the ray's queue record (-4).  Scratch
// 240 -> 168 bytes per lane, L2<->fabric traffic per N=512 frame 5.7 -> 2.9 TB (0.65 TB/s, 8 % of the HBM rate; 0.5 TB at 4 waves per
// SIMD, where nothing is parked, but that is 6 % slower), 7.44 -> 7.58 Grays/s.  LDS has no room for the rest: 7680 bytes x 20
// workgroups per CU is 150 of the 160 KB.
// Waves per SIMD the register budget is set for.  Measured (N=512, W=1): 3 -> 5.98, 4 -> 7.11, 5 -> 7.43 Grays/s; (N=4096, W=8: two
// 8-wave workgroups per CU whatever the budget, so fewer registers only add spills): 3 -> 0.67, 4 -> 0.98, 5 -> 0.65.
#ifndef RT_WIDE_OCC
#define RT_WIDE_OCC (W == 1 ? RT_WIDE_OCC_W1 : 4)
#endif
#ifndef RT_WIDE_OCC_W1
#define RT_WIDE_OCC_W1 5
#endif
template <bool ISO, int K, int W>
__global__ void __launch_bounds__(64 * W, RT_WIDE_OCC) k_render_path_wide(Scene s, RenderParams p) {
  constexpr int WIDE_WAVES = W;
  constexpr int WIDE_THREADS = 64 * W;
  __shared__ WideRayT<ISO> q_ray[WIDE_THREADS];
  __shared__ unsigned long long q_key[WIDE_THREADS];
  __shared__ int q_cnt[W], q_live[W]; // per wave: rays queued this iteration, pool not yet dry
  // One wave-run pool per wave (rt_wavepath.h); the waves share the ray queue — each queues into its own 64 slots — and the
  // sphere rows, and meet twice per iteration, around the scan (see rt_tile.h, which has the same structure).
  __shared__ WaveShared sh_all[W];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // The spheres of s.wide_recs / s.wide_iso are held in registers as n_rows rows of 64 (the last one may be partly
  // empty); the spheres of s.wide_lane (big ones, a short leftover) are tested per lane with the other objects.  Row r
  // of a pass belongs to wave r % W, register row r / W; rows past n_rows are never-hit dummies.
  const int n_rows = s.wide_rows;
  const int n_pass = (n_rows + K * WIDE_WAVES - 1) / (K * WIDE_WAVES);
  WaveShared &sh = sh_all[wave];
  WavePath<true> L;
  L.init(p, sh, (int)blockIdx.x * W + wave);
  WideSlots<ISO, K> slots;

#ifdef RT_PROFILE_PHASES
  unsigned long long t_queue = 0, t_scan = 0, t_shade = 0, n_iter = 0, n_rays_q = 0;
#define RT_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define RT_STAMP(var)
#endif
  for (;;) {
    RT_STAMP(ts0);
    // -- every lane gets a ray from its wave's node pool; the live rays are queued in the wave's slots ----
    const bool wave_live = L.deal(s, p, sh);
    const unsigned long long live = __ballot(L.active);
    if (lane == 0)
      q_cnt[wave] = __popcll(live), q_live[wave] = wave_live ? 1 : 0;
    int my_q = -1;
    if (L.active) {
      my_q = 64 * wave + __popcll(live & ((1ull << lane) - 1ull));
      int self = L.origin_shape >= 0 ? s.shape_origin[4 * L.origin_shape + 1] : -1;
#ifdef RT_NO_SELF_EXACT
      self = -1;
#endif
      const WideRayT<ISO> r = WideRayT<ISO>::make(L.ray, self);
      q_ray[my_q] = r;
      q_key[my_q] = ~0ull;
    }
    __syncthreads();
    int n_q = 0, any_live = 0; // uniform
#pragma unroll
    for (int w = 0; w < W; ++w) {
      n_q += __builtin_amdgcn_readfirstlane(q_cnt[w]);
      any_live |= __builtin_amdgcn_readfirstlane(q_live[w]);
    }
    if (!any_live) // a wave whose pool has run dry keeps coming along (its spheres are part of every ray's scan) until all have
      break;
    RT_STAMP(ts1);

    // -- every lane is K spheres: all waves test every queued ray ----------------------------------
    for (int pass = 0; pass < n_pass; ++pass) {
      const int first = pass * K * WIDE_WAVES * 64;
      // (Re)load this pass's spheres: 16 bytes per sphere from L2, once per iteration — cheaper than keeping
      // K rows alive across the shading phase, where they would push its working set out of the register file.
#pragma unroll
      for (int k = 0; k < K; ++k)
        slots.load(s, k, first + (k * WIDE_WAVES + wave) * 64 + lane, first + (k * WIDE_WAVES + wave) * 64 + lane < s.n_wide);
      auto test_ray = [&](const WideRayT<ISO> &r, int qi) { wide_test_ray<ISO, K, WIDE_WAVES>(slots, r, q_key + qi, first, wave, lane); };
      // Rays ping-pong between two register sets: the LDS broadcast of the next ray is in flight while
      // the lanes test the current one.
      // (a third register set, two rays in flight, was measured and lost: N=512 7.11 -> 6.71 Grays/s — the registers cost more
      // in spills than the deeper prefetch hides)
      for (int w = 0; w < W; ++w) { // the slots of wave w
        const int q0 = 64 * w, q_end = q0 + __builtin_amdgcn_readfirstlane(q_cnt[w]);
        if (q_end == q0)
          continue;
        WideRayT<ISO> ra = q_ray[q0];
        int qi = q0;
        for (; qi + 1 < q_end; qi += 2) {
          const WideRayT<ISO> rb = q_ray[qi + 1];
          test_ray(ra, qi);
          ra = q_ray[min(qi + 2, q_end - 1)];
          test_ray(rb, qi + 1);
        }
        if (qi < q_end)
          test_ray(ra, qi);
      }
    }
    __syncthreads();
    RT_STAMP(ts2);

    // -- every lane is a ray again: finish the query, shade, produce the next ray -------------------
    Hit h;
    h.object = h.shape = -1;
    h.t = 0.f;
    h.point = h.normal = h.local = mk3(0.f, 0.f, 0.f);
    if constexpr (ISO)
      q_ray[max(my_q, 0)].restore(L.ray); // every lane, so that no lane's old values stay live across the scan
    if (L.active) {
      const unsigned long long key = q_key[my_q];
      Best best;
      best.t = INF_F;
      best.object = 0x7fffffff;
      best.shape = -1;
#ifdef RT_NO_SELF_EXACT
      scan_spheres<true>(s.wide_lane, s.n_wide_lane, L.ray, best);
#else
      scan_spheres<true>(s.wide_lane, s.n_wide_lane, L.ray, best, L.origin_shape); // in object order: strict `<` among them
#endif
      self_sphere_exact(s, L.origin_shape, L.ray, best);
      if (key != ~0ull && (unsigned)(key & 0xffffffffull) < (unsigned)s.n_wide) { // positions in the register rows ascend with the object index: the key's tie-break is the reference's
        const int2 ids = *reinterpret_cast<const int2 *>(&s.wide_recs[(int)(unsigned)(key & 0xffffffffull)].object); // {object, shape}
        const float t = __uint_as_float((unsigned)(key >> 32));
        if (t < best.t || (t == best.t && ids.x < best.object)) {
          best.t = t;
          best.object = ids.x;
          best.shape = ids.y;
        }
      }
      h = world_finish_hit<true, true>(s, L.ray, best);
    }
    L.shade(s, p, sh, h); // the wave's own pool: no barrier; a wave's queue slots are rewritten by itself only
#ifdef RT_PROFILE_PHASES
    RT_STAMP(ts3);
    t_queue += ts1 - ts0, t_scan += ts2 - ts1, t_shade += ts3 - ts2, n_iter += 1, n_rays_q += n_q;
#endif
  }
#ifdef RT_PROFILE_PHASES
  if (threadIdx.x == 0) {
    atomicAdd(p.counters + 4, t_queue);
    atomicAdd(p.counters + 5, t_scan);
    atomicAdd(p.counters + 6, t_shade);
    atomicAdd(p.counters + 7, n_iter);
    atomicAdd(p.counters + 8, n_rays_q);
  }
#endif
  wave_add_tally(p.counters, 0, L.rays);
}

} // namespace rtd
