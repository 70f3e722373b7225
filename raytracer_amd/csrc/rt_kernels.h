// rt_kernels.h — the gfx950 kernels of the tracing loop.  One pixel-sample per lane.
//
//  k_render_simple<MODE>  onoff / flat / pointlight: grid-stride over the samples of this call's rows
//  k_render_path          PathTracer as a per-lane explicit DFS with persistent lanes that refill
//                         from a global sample counter (wave-aggregated atomic) when a sample ends
//  k_resolve              box filter: sum of a pixel's spp^2 samples in the reference's order, * 1/spp^2
//  k_trace_rays / k_offset_if_visible / k_aa_jitter   batch parity entry points
#pragma once
#include "rt_device.h"

namespace rtd {

// Sample g of this call -> image pixel (col,row) and intra-pixel cell (i,j); also the index the
// sample has in the reference's serial visit order (columns outer, rows inner: cameras.hpp:222-233).
struct SampleId {
  int col, row, i, j;
  uint64_t serial; // (col*H + row)*spp^2 + i*spp + j
  uint64_t global; // (row*W + col)*spp^2 + i*spp + j   (stream id, independent of sharding)
};

RT_DEV SampleId decode_sample(const RenderParams &p, uint64_t g) {
  const uint32_t spp2 = (uint32_t)(p.spp_edge * p.spp_edge);
  const uint64_t pixel = g / spp2;
  const uint32_t k = (uint32_t)(g - pixel * spp2);
  const int lr = (int)(pixel / (uint32_t)p.width);
  SampleId s;
  s.col = (int)(pixel - (uint64_t)lr * (uint32_t)p.width);
  s.row = (lr / p.band_rows) * p.band_stride + p.band_offset + lr % p.band_rows;
  s.i = (int)(k / (uint32_t)p.spp_edge);
  s.j = (int)(k - (uint32_t)s.i * (uint32_t)p.spp_edge);
  s.serial = ((uint64_t)s.col * (uint64_t)p.height + (uint64_t)s.row) * spp2 + k;
  s.global = ((uint64_t)s.row * (uint64_t)p.width + (uint64_t)s.col) * spp2 + k;
  return s;
}

// ImageTracer::fire_all_rays body for one sample — cameras.hpp:229-243
RT_DEV Ray primary_ray(const RenderParams &p, const SampleId &s) {
  float u_pixel = 0.5f, v_pixel = 0.5f;
  if (p.spp_edge > 1) {
    float ru, rv;
    aa_jitter(s.serial, ru, rv);
    u_pixel = ((float)s.i + ru) / (float)p.spp_edge;
    v_pixel = ((float)s.j + rv) / (float)p.spp_edge;
  }
  return camera_fire_ray(p.cam, p.width, p.height, s.col, s.row, u_pixel, v_pixel);
}

RT_DEV void wave_add_counter(unsigned long long *counter, unsigned long long v) {
  // wave-level reduction, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v)
    atomicAdd(counter, v);
}

// ---------------------------------------------------------------------------------------------
// OnOffTracer / FlatTracer / PointLightTracer::operator() — renderers.hpp:62-69, 88-96, 121-172
// ---------------------------------------------------------------------------------------------
template <int MODE> RT_DEV V3 shade_simple(const Scene &s, const RenderParams &p, Ray ray, uint32_t &rays, uint32_t &capped) {
  const V3 background = mk3(p.background[0], p.background[1], p.background[2]);
  if (MODE == RT_MODE_ONOFF) {
    ++rays;
    return world_any_hit(s, ray) ? mk3(1.f, 1.f, 1.f) : mk3(0.f, 0.f, 0.f);
  }
  if (MODE == RT_MODE_FLAT) {
    ++rays;
    Hit h = world_ray_intersection<false>(s, ray);
    if (h.object < 0)
      return background;
    const uint32_t m = s.shape_material[h.shape];
    float u, v;
    hit_uv(s, h, m, u, v);
    return add3(pigment_eval(s, s.material_brdf_pigment[m], u, v), pigment_eval(s, s.material_emission_pigment[m], u, v));
  }
  // POINTLIGHT
  V3 attenuation = mk3(1.f, 1.f, 1.f);
  Hit h;
  uint32_t m;
  float u, v;
  int bounces = 0;
  for (;;) {
    ++rays;
    h = world_ray_intersection<false>(s, ray);
    if (h.object < 0)
      return background;
    m = s.shape_material[h.shape];
    hit_uv(s, h, m, u, v);
    if (s.material_brdf[m] != RT_BRDF_SPECULAR)
      break;
    if (bounces++ >= p.max_mirror_bounces) { // the reference has no cap and may spin forever (H1)
      ++capped;
      return background;
    }
    attenuation = mul3(attenuation, pigment_eval(s, s.material_brdf_pigment[m], u, v));
    V3 new_dir = sub3(ray.d, scale3(lscale3(2.f, h.normal), dot3(h.normal, ray.d)));
    ray.o = h.point;
    ray.d = new_dir;
    ray.tmin = 1e-5f;
    ray.tmax = INF_F;
  }
  V3 radiance = add3(mk3(p.ambient[0], p.ambient[1], p.ambient[2]), pigment_eval(s, s.material_emission_pigment[m], u, v));
  for (int l = 0; l < s.n_lights; ++l) {
    V3 in_dir;
    ++rays;
    if (!world_offset_if_visible(s, ld3(s.light_point + 3 * l), h.point, h.normal, in_dir))
      continue;
    float distance = sqrtf(sqnorm3(in_dir));
    float radius = s.light_radius[l];
    float ratio = radius / distance;
    float distance_factor = (radius > 0.f) ? ratio * ratio : 1.f; // float(double(x)^2) == fl(x*x)
    float cos_theta = dot3(lscale3(-1.f / distance, in_dir), normalized3(h.normal));
    V3 brdf = scale3(pigment_eval(s, s.material_brdf_pigment[m], u, v), 1.f / PI_F); // DiffusiveBRDF::eval
    radiance = add3(radiance, mul3(scale3(scale3(ld3(s.light_color + 3 * l), distance_factor), cos_theta), brdf));
  }
  return mul3(radiance, attenuation);
}

template <int MODE> __global__ void __launch_bounds__(256) k_render_simple(Scene s, RenderParams p) {
  uint32_t rays = 0, capped = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < p.n_samples; g += stride) {
    SampleId sid = decode_sample(p, g);
    V3 c = shade_simple<MODE>(s, p, primary_ray(p, sid), rays, capped);
    float *out = p.samples_out + 3 * g;
    out[0] = c.x;
    out[1] = c.y;
    out[2] = c.z;
  }
  wave_add_counter(p.counters + 1, rays);
  wave_add_counter(p.counters + 2, capped);
}

// ---------------------------------------------------------------------------------------------
// PathTracer::operator() — renderers.hpp:208-262 — as an explicit depth-first traversal.
//
// The estimator is linear in the emitted radiance, so L = Le + (rho/n) * sum L(child) is evaluated
// by accumulating throughput * Le over the nodes of the recursion tree in the reference's DFS order
// (RR draw at the node, then per child: the 2 scatter draws followed by the child's whole subtree).
// A stack frame holds what scatter_ray needs to emit the node's remaining children.
// ---------------------------------------------------------------------------------------------
constexpr int STACK_WORDS = 10; // point(3) vec(3) weight(3) meta(1)

RT_DEV float *stack_word(const RenderParams &p, int level, int word, uint32_t lane) {
  return p.stack + ((size_t)(level * STACK_WORDS + word) * p.n_lanes + lane);
}

// One pixel sample in flight on a lane.
struct PathLane {
  bool active, exhausted;
  uint64_t g; // sample index of this call
  Pcg pcg;
  Ray ray; // the ray to trace next
  int depth, level;
  V3 throughput, radiance;
  uint32_t rays;

  RT_DEV void init() {
    active = exhausted = false;
    g = 0;
    pcg.state = pcg.inc = 0;
    ray.o = ray.d = mk3(0.f, 0.f, 0.f);
    ray.tmin = 0.f;
    ray.tmax = INF_F;
    depth = 0;
    level = -1;
    throughput = mk3(1.f, 1.f, 1.f);
    radiance = mk3(0.f, 0.f, 0.f);
    rays = 0;
  }

  // Idle lanes take the next samples of the call's queue: one atomic per wave (ballot + prefix count).
  // Must be called with the idle lanes of a wave converged.
  RT_DEV void refill(const RenderParams &p) {
    if (!active && !exhausted) {
      const unsigned long long idle = __ballot(1);
      const int n_idle = __popcll(idle);
      const int rank = __popcll(idle & ((1ull << (threadIdx.x & 63)) - 1ull));
      unsigned long long base = 0;
      if (rank == 0)
        base = atomicAdd(p.counters + 0, (unsigned long long)n_idle);
      base = __shfl(base, __ffsll((long long)idle) - 1, 64);
      g = base + rank;
      if (g < p.n_samples) {
        SampleId sid = decode_sample(p, g);
        ray = primary_ray(p, sid);
        sample_stream(pcg, p.seq_number, sid.global);
        depth = 0;
        level = -1;
        throughput = mk3(1.f, 1.f, 1.f);
        radiance = mk3(0.f, 0.f, 0.f);
        active = true;
      } else {
        exhausted = true; // the queue is drained: stop asking
      }
    }
  }

  // PathTracer::operator() steps 2-6 for the node whose closest hit is `h`, then the next ray of the
  // traversal (or the end of the sample).
  RT_DEV void shade_and_advance(const Scene &s, const RenderParams &p, const Hit &h, uint32_t lane_global) {
    ++rays;
    if (h.object < 0) {
      radiance = add3(radiance, mul3(throughput, mk3(p.background[0], p.background[1], p.background[2])));
    } else {
      const uint32_t m = s.shape_material[h.shape];
      float u, v;
      hit_uv(s, h, m, u, v);
      V3 reflected = pigment_eval(s, s.material_brdf_pigment[m], u, v);
      const V3 emitted = pigment_eval(s, s.material_emission_pigment[m], u, v);
      radiance = add3(radiance, mul3(throughput, emitted));
      const float hit_lum = fmaxf(fmaxf(reflected.x, reflected.y), reflected.z);
      bool scatter = hit_lum > 0.f;
      if (depth > p.roulette) {
        const float q = fmaxf(1.f - hit_lum, 0.05f);
        if (pcg_random_float(pcg) > q)
          reflected = scale3(reflected, 1.f / (1.f - q));
        else
          scatter = false;
      }
      if (scatter) {
        const int brdf = s.material_brdf[m];
        if (depth >= p.max_depth) {
          // children would be at depth > max_depth: BLACK without a query, but DiffusiveBRDF::
          // scatter_ray has already drawn its two numbers per child (renderers.hpp:210, 250)
          if (brdf == RT_BRDF_DIFFUSE)
            pcg_advance(pcg, 2ull * (uint64_t)p.n_rays);
        } else {
          ++level;
          const V3 vec = (brdf == RT_BRDF_SPECULAR) ? specular_scatter_dir(ray.d, h.normal) : normalized3(h.normal);
          const V3 weight = mul3(scale3(throughput, 1.f / (float)p.n_rays), reflected);
          *stack_word(p, level, 0, lane_global) = h.point.x;
          *stack_word(p, level, 1, lane_global) = h.point.y;
          *stack_word(p, level, 2, lane_global) = h.point.z;
          *stack_word(p, level, 3, lane_global) = vec.x;
          *stack_word(p, level, 4, lane_global) = vec.y;
          *stack_word(p, level, 5, lane_global) = vec.z;
          *stack_word(p, level, 6, lane_global) = weight.x;
          *stack_word(p, level, 7, lane_global) = weight.y;
          *stack_word(p, level, 8, lane_global) = weight.z;
          // meta: bit 0 = brdf, bits 1..15 = depth of the children, bits 16..31 = children emitted so far
          *stack_word(p, level, 9, lane_global) = __uint_as_float((uint32_t)brdf | ((uint32_t)(depth + 1) << 1));
        }
      }
    }
    // next ray: the next child of the innermost frame that still has children
    if (level >= 0) {
      const uint32_t meta = __float_as_uint(*stack_word(p, level, 9, lane_global));
      const int brdf = meta & 1u;
      const uint32_t emitted_children = (meta >> 16) + 1u;
      depth = (int)((meta >> 1) & 32767u);
      ray.o = mk3(*stack_word(p, level, 0, lane_global), *stack_word(p, level, 1, lane_global), *stack_word(p, level, 2, lane_global));
      const V3 vec = mk3(*stack_word(p, level, 3, lane_global), *stack_word(p, level, 4, lane_global), *stack_word(p, level, 5, lane_global));
      throughput = mk3(*stack_word(p, level, 6, lane_global), *stack_word(p, level, 7, lane_global), *stack_word(p, level, 8, lane_global));
      if (brdf == RT_BRDF_SPECULAR) {
        ray.d = vec;
        ray.tmin = 1.e-5f;
      } else {
        ray.d = diffuse_scatter_dir(vec, pcg);
        ray.tmin = 1.e-3f;
      }
      ray.tmax = INF_F;
      if (emitted_children >= (uint32_t)p.n_rays)
        --level; // last child: the frame is finished
      else
        *stack_word(p, level, 9, lane_global) = __uint_as_float((meta & 65535u) | (emitted_children << 16));
    } else {
      float *out = p.samples_out + 3 * g;
      out[0] = radiance.x;
      out[1] = radiance.y;
      out[2] = radiance.z;
      active = false;
    }
  }
};

// Path kernel for small worlds: one ray per lane, every lane scans the (few) objects itself with
// wave-uniform scalar record loads.  Persistent lanes refill from the call's sample queue.
template <bool FAST> __global__ void __launch_bounds__(256) k_render_path(Scene s, RenderParams p) {
  const uint32_t lane_global = blockIdx.x * blockDim.x + threadIdx.x;
  PathLane L;
  L.init();
  for (;;) {
    L.refill(p);
    if (!__any(L.active))
      break;
    if (L.active) {
      const Hit h = world_ray_intersection<FAST>(s, L.ray);
      L.shade_and_advance(s, p, h, lane_global);
    }
  }
  wave_add_counter(p.counters + 1, L.rays);
}

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
//   * Then every lane is a ray again: it completes the query (planes, CSG, hit record), shades the
//     node and produces the next ray of its sample.
//
//   ISO: all spheres are translation * uniform scaling, tested in world space from (centre, r^2):
//        oc = O - C; b = oc.D; c = oc.oc - r^2; disc = b^2 - (D.D) c      (12 VALU per test)
//   otherwise the general affine form of sphere_scan_test<true> (inverse matrix, 28 VALU per test).
// ---------------------------------------------------------------------------------------------
constexpr int WIDE_WAVES = 8;
constexpr int WIDE_THREADS = 64 * WIDE_WAVES;

struct alignas(16) WideRay { // LDS image of one queued ray: 3 x 16 bytes
  float ox, oy, oz, tmin;
  float dx, dy, dz, tmax;
  float a, inv_a, pad0, pad1; // D.D and its reciprocal
};

template <bool ISO, int K> struct WideSlots;
template <int K> struct WideSlots<true, K> {
  float cx[K], cy[K], cz[K], r2[K];
  RT_DEV void load(const Scene &s, int k, int sphere, bool valid) {
    if (valid) {
      const float4 v = reinterpret_cast<const float4 *>(s.sphere_iso)[sphere];
      cx[k] = v.x, cy[k] = v.y, cz[k] = v.z, r2[k] = v.w;
    } else { // empty slot: c = |oc|^2 + 1 makes disc <= -a < 0 (Cauchy-Schwarz), never a hit
      cx[k] = cy[k] = cz[k] = 0.f;
      r2[k] = -1.f;
    }
  }
  RT_DEV float disc(int k, const WideRay &r, float &b) const {
    const float ocx = r.ox - cx[k], ocy = r.oy - cy[k], ocz = r.oz - cz[k];
    b = fmaf(ocx, r.dx, fmaf(ocy, r.dy, ocz * r.dz));
    const float c = fmaf(ocx, ocx, fmaf(ocy, ocy, fmaf(ocz, ocz, -r2[k])));
    return fmaf(b, b, -(r.a * c));
  }
  RT_DEV float a_of(int, const WideRay &r, float &inv_a) const {
    inv_a = r.inv_a;
    return r.a;
  }
};
template <int K> struct WideSlots<false, K> {
  float m[K][12];
  float a_[K];
  RT_DEV void load(const Scene &s, int k, int sphere, bool valid) {
    if (valid) {
      const float4 *v = reinterpret_cast<const float4 *>(s.spheres[sphere].inv);
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
  RT_DEV float disc(int k, const WideRay &r, float &b) {
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
  RT_DEV float a_of(int k, const WideRay &, float &inv_a) const {
    inv_a = 1.f / a_[k];
    return a_[k];
  }
};

template <bool ISO, int K> __global__ void __launch_bounds__(WIDE_THREADS) k_render_path_wide(Scene s, RenderParams p) {
  __shared__ WideRay q_ray[WIDE_THREADS];
  __shared__ unsigned long long q_key[WIDE_THREADS];
  __shared__ int q_count;

  const uint32_t lane_global = blockIdx.x * blockDim.x + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Spheres [wide_first, n_spheres) are held in registers, wide_slots full rows of 64*WIDE_WAVES; the
  // first wide_first (< 64*WIDE_WAVES) spheres of the list are scanned per lane with the other objects.
  const int wide_first = s.n_spheres % (64 * WIDE_WAVES);
  const int wide_slots = s.n_spheres / (64 * WIDE_WAVES);
  const int n_pass = (wide_slots + K - 1) / K;
  PathLane L;
  L.init();
  WideSlots<ISO, K> slots;
  if (n_pass == 1) { // the whole world fits: load the registers once
#pragma unroll
    for (int k = 0; k < K; ++k)
      slots.load(s, k, wide_first + (k * WIDE_WAVES + wave) * 64 + lane, k < wide_slots);
  }

  for (;;) {
    // -- every lane is a sample: refill, then queue the live rays (compacted) ----------------------
    if (threadIdx.x == 0)
      q_count = 0;
    L.refill(p);
    __syncthreads();
    int my_q = -1;
    if (L.active) {
      const unsigned long long live = __ballot(1);
      const int rank = __popcll(live & ((1ull << lane) - 1ull));
      int base = 0;
      if (rank == 0)
        base = atomicAdd(&q_count, __popcll(live));
      base = __shfl(base, __ffsll((long long)live) - 1, 64);
      my_q = base + rank;
      WideRay r;
      r.ox = L.ray.o.x, r.oy = L.ray.o.y, r.oz = L.ray.o.z, r.tmin = L.ray.tmin;
      r.dx = L.ray.d.x, r.dy = L.ray.d.y, r.dz = L.ray.d.z, r.tmax = L.ray.tmax;
      r.a = fmaf(r.dx, r.dx, fmaf(r.dy, r.dy, r.dz * r.dz));
      r.inv_a = 1.f / r.a;
      r.pad0 = r.pad1 = 0.f;
      q_ray[my_q] = r;
      q_key[my_q] = ~0ull;
    }
    __syncthreads();
    const int n_q = q_count;
    if (n_q == 0)
      break; // no live sample in the workgroup and the call's queue is drained

    // -- every lane is K spheres: all waves test every queued ray ----------------------------------
    for (int pass = 0; pass < n_pass; ++pass) {
      const int k_used = min(K, wide_slots - pass * K);
      const int first = wide_first + pass * K * WIDE_WAVES * 64;
      if (n_pass > 1) {
#pragma unroll
        for (int k = 0; k < K; ++k)
          slots.load(s, k, first + (k * WIDE_WAVES + wave) * 64 + lane, k < k_used);
      }
      for (int qi = 0; qi < n_q; ++qi) {
        const WideRay r = q_ray[qi]; // same address in every lane: LDS broadcast
        float best_t = INF_F;
        int best_k = -1;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (k < k_used) {
            float b;
            const float disc = slots.disc(k, r, b);
            if (disc > 0.f) {
              float inv_a;
              slots.a_of(k, r, inv_a);
              const float sq = sqrtf(disc);
              const float t1 = (-b - sq) * inv_a;
              const float t2 = (-b + sq) * inv_a;
              float t = t1;
              bool ok = true;
              if (t1 < r.tmin) {
                t = t2;
                ok = t2 > r.tmin && t2 < r.tmax;
              }
              if (ok && t < best_t) { // ascending k = ascending object index within a lane
                best_t = t;
                best_k = k;
              }
            }
          }
        }
        if (best_k >= 0) {
          const int sphere = first + (best_k * WIDE_WAVES + wave) * 64 + lane;
          const unsigned long long key = ((unsigned long long)__float_as_uint(best_t) << 32) | (unsigned)sphere;
          atomicMin(&q_key[qi], key);
        }
      }
    }
    __syncthreads();

    // -- every lane is a ray again: finish the query, shade, produce the next ray -------------------
    if (L.active) {
      const unsigned long long key = q_key[my_q];
      Best best;
      best.t = INF_F;
      best.object = 0x7fffffff;
      best.shape = -1;
      scan_spheres<true>(s, wide_first, L.ray, best); // the few spheres that did not fill a register row
      if (key != ~0ull) { // sphere list order == object order, so the key's tie-break is the reference's
        const int sphere = (int)(unsigned)(key & 0xffffffffull);
        const float t = __uint_as_float((unsigned)(key >> 32));
        if (t < best.t) { // a register-held sphere has a higher index than any scanned one: strict <
          best.t = t;
          best.object = s.spheres[sphere].object;
          best.shape = s.spheres[sphere].shape;
        }
      }
      const Hit h = world_finish_hit(s, L.ray, best);
      L.shade_and_advance(s, p, h, lane_global);
    }
    __syncthreads(); // q_count / q_ray are rewritten at the top
  }
  wave_add_counter(p.counters + 1, L.rays);
}

// ---------------------------------------------------------------------------------------------
// Box filter over a pixel's samples — cameras.hpp:236-239 (`cum += c` in (i,j) order, then
// Color::operator/ = * (1.f / scalar), colors.hpp:145)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_resolve(const float *samples, float *rgb, uint64_t n_pixels, int spp2) {
  const uint64_t px = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= n_pixels)
    return;
  const float *sp = samples + 3 * px * (uint64_t)spp2;
  float r = 0.f, g = 0.f, b = 0.f;
  for (int k = 0; k < spp2; ++k) {
    r = r + sp[3 * k + 0];
    g = g + sp[3 * k + 1];
    b = b + sp[3 * k + 2];
  }
  const float inv = 1.f / (float)spp2;
  rgb[3 * px + 0] = r * inv;
  rgb[3 * px + 1] = g * inv;
  rgb[3 * px + 2] = b * inv;
}

// ---------------------------------------------------------------------------------------------
// Parity entry points
// ---------------------------------------------------------------------------------------------
__global__ void k_trace_rays(Scene s, const rt_ray *rays, size_t n, rt_hit *hits) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  Ray r;
  r.o = ld3(rays[i].origin);
  r.d = ld3(rays[i].direction);
  r.tmin = rays[i].tmin;
  r.tmax = rays[i].tmax;
  Hit h = world_ray_intersection<false>(s, r);
  rt_hit out;
  out.object = h.object;
  out.shape = h.shape;
  out.t = h.t;
  out.point[0] = h.point.x, out.point[1] = h.point.y, out.point[2] = h.point.z;
  out.normal[0] = h.normal.x, out.normal[1] = h.normal.y, out.normal[2] = h.normal.z;
  out.uv[0] = out.uv[1] = 0.f;
  if (h.object >= 0)
    shape_uv(s, h.shape, h.local, out.uv[0], out.uv[1]);
  hits[i] = out;
}

__global__ void k_offset_if_visible(Scene s, const float *viewer, const float *surface, const float *normal, size_t n,
                                    uint8_t *visible) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  V3 d;
  visible[i] = world_offset_if_visible(s, ld3(viewer + 3 * i), ld3(surface + 3 * i), ld3(normal + 3 * i), d) ? 1 : 0;
}

__global__ void k_aa_jitter(const uint64_t *k, size_t n, float *u, float *v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  aa_jitter(k[i], u[i], v[i]);
}

__global__ void k_scatter_rays(uint32_t brdf_kind, uint64_t *pcg, const float *in_dir, const float *normal, size_t n,
                               float *dir_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  V3 d;
  if (brdf_kind == RT_BRDF_SPECULAR) {
    d = specular_scatter_dir(ld3(in_dir + 3 * i), ld3(normal + 3 * i));
  } else {
    Pcg p;
    p.state = pcg[2 * i];
    p.inc = pcg[2 * i + 1];
    d = diffuse_scatter_dir(normalized3(ld3(normal + 3 * i)), p);
    pcg[2 * i] = p.state;
  }
  dir_out[3 * i + 0] = d.x;
  dir_out[3 * i + 1] = d.y;
  dir_out[3 * i + 2] = d.z;
}

} // namespace rtd
