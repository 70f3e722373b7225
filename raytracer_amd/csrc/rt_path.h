// rt_path.h — what the render kernels share: sample decoding, primary rays, and the workgroup-level
// machinery of the path tracer (PathBlock): a pool of recursion-tree nodes whose children are dealt to
// the lanes of the workgroup.
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

// Tallies (rays, capped samples) are spread over COUNTER_SLOTS addresses: thousands of atomics on ONE address serialise
// at ~13 ns each (measured), which was most of the 0.24 ms of a 1280x960 onoff frame.  The host sums the slots.
constexpr int COUNTER_BASE = 16, COUNTER_SLOTS = 32, COUNTER_WORDS = COUNTER_BASE + 2 * COUNTER_SLOTS;
RT_DEV void wave_add_tally(unsigned long long *counters, int which, unsigned long long v) { // which: 0 rays, 1 capped
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v)
    atomicAdd(counters + COUNTER_BASE + 2 * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % COUNTER_SLOTS) + which, v);
}

// ---------------------------------------------------------------------------------------------
// PathTracer::operator() — renderers.hpp:208-262 — evaluated by a POOL of lanes (one wave: rt_wavepath.h), not by a lane.
//
// The estimator is linear in the emitted radiance: L = Le + (rho/n) * sum L(child), so a sample's value
// is the sum over the nodes of its recursion tree of throughput(node) * Le(node), with
// throughput(child) = throughput(parent) * (1/n_rays) * rho(parent).  Nodes are independent once every
// ray owns its random stream (RT_RNG_PER_NODE_STREAMS, see node_child_key), so the tree is not walked
// by one lane: a hit that scatters pushes ONE node record {point, normal-or-mirror direction, weight,
// stream key, slot, depth, brdf} with n_rays children still to emit onto the pool's node stack (HBM),
// and every iteration deals the children on top of the stack to the T = 64 lanes of the pool, topping up
// with fresh pixel samples when the stack runs short.  A sample whose tree has 10^5 rays (mirrors facing
// mirrors) is therefore spread over the whole pool instead of serialising one lane.  (Rounds 1-2 ran one
// pool per workgroup — PathBlock, in this file's history — with barriers and strided node words; one pool
// per wave replaced it in every kernel: C3 20.7 -> 13.6 ms, N=512 9.9 -> 10.7 Grays/s, BVH N=4096 6.0 -> 4.8 s.)
//
// Sample values are accumulated per sample slot in LDS as 64-bit fixed point (2^-32), with integer
// atomics: the sum does not depend on the order in which lanes add, so a frame is bit-reproducible for
// any grid, lane assignment or number of GPUs.  oracle/rt_oracle.c accumulates the same way.
//
// Stack discipline (bound on its size): children are taken from the top; the nodes pushed in an
// iteration are stacked in reverse lane order (the lane that took the topmost child pushes topmost).
// Then child-depth never decreases from bottom to top, a child of depth d is only taken once every
// deeper node is used up, hence at most T nodes per depth level: capacity T * max_depth (T if n_rays
// is 1, where a node is used up by the one lane that takes it).
// ---------------------------------------------------------------------------------------------
// Table sizes up to which the small-world kernels stage a scene's lookup tables in LDS (rt_kernels.h, rt_wavepath.h).
struct PathLdsLimits {
  static constexpr int SHAPES = 64, MATERIALS = 32, PIGMENTS = 64, CSG_LEAVES = 32, CSG_PATH = 128, CSG_NODES = 32, CSG_PROG = 128, CSG_TOPS = 8;
};

RT_DEV uint64_t node_child_key(uint64_t parent_state, uint32_t k) { // same in oracle/rt_oracle.c
  return mix64(parent_state + (uint64_t)k * 0x9E3779B97F4A7C15ull);
}
RT_DEV uint64_t node_root_key(uint64_t seq_number, uint64_t global_sample) {
  return mix64(global_sample ^ mix64(seq_number));
}
RT_DEV long long to_fixed(float x) { return to_fixed32(x); }
// |to_fixed32(x)| in five instructions: v_fract_f32 gives |x| - floor(|x|) exactly (the same value as ax - (float)hi in
// to_fixed32), so the two words are the same bits.
RT_DEV unsigned long long to_fixed_abs(float x) {
  const float ax = fminf(fabsf(x), 2147483520.f);
  const uint32_t hi = (uint32_t)ax;
  const uint32_t lo = (uint32_t)(__builtin_amdgcn_fractf(ax) * 4294967296.f);
  return ((unsigned long long)hi << 32) | lo;
}
RT_DEV float from_fixed(unsigned long long a) { return (float)((double)(long long)a * (1.0 / 4294967296.0)); }


} // namespace rtd
