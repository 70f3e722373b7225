// rt_kernels.h — the gfx950 kernels of the tracing loop.  One pixel-sample per lane.
//
//  k_render_simple<MODE>  onoff / flat / pointlight: grid-stride over the samples of this call's rows
//  k_render_path          PathTracer over a per-workgroup pool of recursion-tree nodes (rt_path.h)
//  k_resolve              box filter: sum of a pixel's spp^2 samples in the reference's order, * 1/spp^2
//  k_trace_rays / k_offset_if_visible / k_aa_jitter   batch parity entry points
#pragma once
#include "rt_wavepath.h"

namespace rtd {

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
  for (int l = 0, n_lights = uniform_count(s.n_lights); l < n_lights; ++l) {
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
  wave_add_tally(p.counters, 0, rays);
  wave_add_tally(p.counters, 1, capped);
}

// The plain path kernel: one ray per lane from the wave's node pool (rt_wavepath.h), every lane scans the objects itself with
// wave-uniform scalar record loads.  It serves RT_FLAG_EXACT_ARITHMETIC (FAST = false: the reference's unfused operation
// order — the kernel the parity tests compare with the oracle) and worlds whose lookup tables exceed what k_render_path_wave
// stages in LDS.
constexpr int PATH_THREADS = 256;
template <bool FAST> __global__ void __launch_bounds__(PATH_THREADS, 5) k_render_path(Scene s, RenderParams p) {
  __shared__ WaveShared sh_all[PATH_THREADS / 64]; // one wave-run pool per wave (rt_wavepath.h): the waves of a workgroup never meet
  WaveShared &sh = sh_all[threadIdx.x >> 6];
  WavePath<FAST> B;
  B.init(p, sh, (int)blockIdx.x * (PATH_THREADS / 64) + (int)(threadIdx.x >> 6));
  while (B.deal(s, p, sh)) {
    Hit h;
    h.object = h.shape = -1;
    h.t = 0.f;
    h.point = h.normal = h.local = mk3(0.f, 0.f, 0.f);
    if (B.active)
      h = world_ray_intersection<FAST>(s, B.ray);
    B.shade(s, p, sh, h);
  }
  wave_add_tally(p.counters, 0, B.rays);
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

// FAST = the forms the default (FMA) path kernels use: rsq normalisation, cos/sin theta from square roots, v_sin / v_cos
template <bool FAST>
__global__ void k_scatter_rays(uint32_t brdf_kind, uint64_t *pcg, const float *in_dir, const float *normal, size_t n,
                               float *dir_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  V3 d;
  if (brdf_kind == RT_BRDF_SPECULAR) {
    d = FAST ? specular_scatter_dir_fast(ld3(in_dir + 3 * i), ld3(normal + 3 * i)) : specular_scatter_dir(ld3(in_dir + 3 * i), ld3(normal + 3 * i));
  } else {
    Pcg p;
    p.state = pcg[2 * i];
    p.inc = pcg[2 * i + 1];
    d = diffuse_scatter_dir<FAST>(FAST ? normalized3_fast(ld3(normal + 3 * i)) : normalized3(ld3(normal + 3 * i)), p);
    pcg[2 * i] = p.state;
  }
  dir_out[3 * i + 0] = d.x;
  dir_out[3 * i + 1] = d.y;
  dir_out[3 * i + 2] = d.z;
}

} // namespace rtd
