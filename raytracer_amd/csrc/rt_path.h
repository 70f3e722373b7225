// rt_path.h — what the render kernels share: sample decoding, primary rays, and the per-lane state machine
// of the path tracer (PathLane).
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

} // namespace rtd
