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

template <bool FAST> __global__ void __launch_bounds__(256) k_render_path(Scene s, RenderParams p) {
  const uint32_t lane_global = blockIdx.x * blockDim.x + threadIdx.x;
  const V3 background = mk3(p.background[0], p.background[1], p.background[2]);
  const float inv_n_rays = 1.f / (float)p.n_rays;
  uint32_t rays = 0;

  bool active = false, exhausted = false;
  uint64_t g = 0;
  Pcg pcg;
  pcg.state = pcg.inc = 0;
  Ray ray;
  ray.o = ray.d = mk3(0.f, 0.f, 0.f);
  ray.tmin = 0.f;
  ray.tmax = INF_F;
  int depth = 0, level = -1;
  V3 throughput = mk3(1.f, 1.f, 1.f), radiance = mk3(0.f, 0.f, 0.f);

  for (;;) {
    // -- refill: idle lanes take the next samples (one atomic per wave) ---------------------------
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
    if (!__any(active))
      break;
    if (active) {
      // -- one closest-hit query ------------------------------------------------------------------
      ++rays;
      Hit h = world_ray_intersection<FAST>(s, ray);
      // -- PathTracer::operator() steps 2-4 for this node -------------------------------------------
      if (h.object < 0) {
        radiance = add3(radiance, mul3(throughput, background));
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
            const V3 weight = mul3(scale3(throughput, inv_n_rays), reflected);
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
      // -- next ray: the next child of the innermost frame that still has children ------------------
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
  }
  wave_add_counter(p.counters + 1, rays);
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
