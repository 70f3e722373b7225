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
// PathTracer::operator() — renderers.hpp:208-262 — evaluated by a WORKGROUP, not by a lane.
//
// The estimator is linear in the emitted radiance: L = Le + (rho/n) * sum L(child), so a sample's value
// is the sum over the nodes of its recursion tree of throughput(node) * Le(node), with
// throughput(child) = throughput(parent) * (1/n_rays) * rho(parent).  Nodes are independent once every
// ray owns its random stream (RT_RNG_PER_NODE_STREAMS, see node_child_key), so the tree is not walked
// by one lane: a hit that scatters pushes ONE node record {point, normal-or-mirror direction, weight,
// stream key, slot, depth, brdf} with n_rays children still to emit onto the workgroup's node stack
// (HBM), and every iteration deals the children on top of the stack to the T lanes of the workgroup,
// topping up with fresh pixel samples when the stack runs short.  A sample whose tree has 10^5 rays
// (mirrors facing mirrors) is therefore spread over the whole workgroup instead of serialising one lane.
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
constexpr int NODE_WORDS = 14; // point(3) vec(3) weight(3) key(2) meta(1) remaining(1) shape-hit(1)
constexpr int SAMPLE_CHUNK = 2; // pixel samples fetched from the call's queue per top-up, in units of T

RT_DEV uint64_t node_child_key(uint64_t parent_state, uint32_t k) { // same in oracle/rt_oracle.c
  return mix64(parent_state + (uint64_t)k * 0x9E3779B97F4A7C15ull);
}
RT_DEV uint64_t node_root_key(uint64_t seq_number, uint64_t global_sample) {
  return mix64(global_sample ^ mix64(seq_number));
}
RT_DEV long long to_fixed(float x) { return to_fixed32(x); }
RT_DEV float from_fixed(unsigned long long a) { return (float)((double)(long long)a * (1.0 / 4294967296.0)); }

template <int T> struct PathShared { // LDS state of one workgroup
  unsigned long long acc[T][3]; // per sample slot: fixed-point radiance
  unsigned long long slot_g[T]; // per sample slot: sample index of this call
  int pending[T];               // per sample slot: rays not yet shaded (-1: slot free)
  int owner[T];                 // per dealt child: node holder lane | child index << 16
  short free_list[T];
  unsigned short rem[8 * T];    // children left per stack node, when the stack holds at most 8 T nodes (else node word 12)
  int wave_tot[T / 64];
  int n_free, n_pop, n_new;
  int res_count, exhausted;
  unsigned long long res_base, new_base;
};

// Inclusive prefix sum over the T lanes of the workgroup; `total` = the sum.  All lanes must call it.
template <int T> RT_DEV int block_scan_inclusive(int v, int *wave_tot, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off)
      x += y;
  }
  if (T > 64) {
    if (lane == 63)
      wave_tot[wave] = x;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
      const int t = wave_tot[w];
      if (w < wave)
        base += t;
      total += t;
    }
    x += base;
    __syncthreads();
  } else {
    total = __shfl(x, 63, 64);
  }
  return x;
}

// The per-lane view of one iteration: the ray this lane traces and what it needs to shade its node.
template <int T, bool FAST> struct PathBlock {
  // lane state of the current iteration
  bool active;
  Ray ray;
  Pcg pcg; // this ray's own stream (already past its scatter draws)
  V3 throughput;
  int depth, slot;
  int origin_shape; // the shape this ray leaves (-1: a camera ray)
  // workgroup state mirrored in every lane (uniform)
  int top;
  uint32_t rays;
  float *pool; // this workgroup's node stack: word w of node i at pool[w * cap + i]
  int cap;

  RT_DEV void init(const RenderParams &p, PathShared<T> &sh) {
    active = false;
    ray.o = ray.d = mk3(0.f, 0.f, 0.f);
    ray.tmin = 0.f;
    ray.tmax = INF_F;
    pcg.state = 0;
    pcg.inc = (p.seq_number << 1u) | 1u;
    throughput = mk3(1.f, 1.f, 1.f);
    depth = slot = 0;
    origin_shape = -1;
    top = 0;
    rays = 0;
    cap = p.pool_cap;
    pool = p.pool + (size_t)blockIdx.x * NODE_WORDS * (size_t)cap;
    const int tid = threadIdx.x;
    sh.pending[tid] = -1;
    sh.free_list[tid] = (short)tid;
    if (tid == 0) {
      sh.n_free = T;
      sh.n_pop = 0;
      sh.n_new = 0;
      sh.res_count = 0;
      sh.exhausted = 0;
      sh.res_base = 0;
      sh.new_base = 0;
    }
    __syncthreads();
  }

  RT_DEV float &node_word(int word, int node) const { return pool[(size_t)word * cap + node]; }
  // children left of a stack node: in LDS when the whole stack's counters fit there (saves a dependent trip to HBM
  // at the head of every iteration), else word 12 of the node
  RT_DEV int get_rem(const PathShared<T> &sh, int node) const {
    return cap <= 8 * T ? (int)sh.rem[node] : __float_as_int(node_word(12, node));
  }
  RT_DEV void set_rem(PathShared<T> &sh, int node, int v) const {
    if (cap <= 8 * T)
      sh.rem[node] = (unsigned short)v;
    else
      node_word(12, node) = __int_as_float(v);
  }

  // Deals the children on top of the node stack, then fresh samples, to the lanes.  Returns false when
  // the workgroup has nothing left to do (uniform).
  RT_DEV bool deal(const Scene &s, const RenderParams &p, PathShared<T> &sh) {
    const int tid = threadIdx.x;
    const int nfree0 = sh.n_free;
    int rem = 0;
    if (tid < top)
      rem = get_rem(sh, top - 1 - tid);
    int total;
    const int incl = block_scan_inclusive<T>(rem, sh.wave_tot, total);
    const int take = min(total, T);
    if (rem > 0) {
      const int lo = incl - rem, hi = min(incl, T);
      const int first_child = p.n_rays - rem;
      for (int c = lo; c < hi; ++c)
        sh.owner[c] = tid | ((first_child + (c - lo)) << 16);
      if (incl > T && lo < T)
        set_rem(sh, top - 1 - tid, incl - T); // partly used: stays on the stack
    }
    const unsigned long long used_up = __ballot(rem > 0 && incl <= T);
    if ((tid & 63) == 0 && used_up)
      atomicAdd(&sh.n_pop, __popcll(used_up));
    if (tid == 0) {
      int want = min(T - take, nfree0);
      if (want > 0 && sh.res_count == 0 && !sh.exhausted) {
        const unsigned long long chunk = (unsigned long long)SAMPLE_CHUNK * T;
        const unsigned long long base = atomicAdd(p.counters + 0, chunk);
        const unsigned long long avail = base < p.n_samples ? min(chunk, (unsigned long long)p.n_samples - base) : 0ull;
        if (avail < chunk)
          sh.exhausted = 1;
        sh.res_base = base;
        sh.res_count = (int)avail;
      }
      want = min(want, sh.res_count);
      sh.new_base = sh.res_base;
      sh.n_new = want;
      sh.res_base += (unsigned long long)want;
      sh.res_count -= want;
    }
    __syncthreads();
    const int n_new = sh.n_new;
    const int n_pop = sh.n_pop;
    active = false;
    if (tid < take) {
      const int o = sh.owner[tid];
      const int node = top - 1 - (o & 0xffff);
      const uint32_t k = (uint32_t)o >> 16;
      const uint32_t meta = __float_as_uint(node_word(11, node));
      slot = (int)(meta & 0xffffu);
      depth = (int)((meta >> 16) & 0x7fffu);
      const uint64_t key = (uint64_t)__float_as_uint(node_word(9, node)) | ((uint64_t)__float_as_uint(node_word(10, node)) << 32);
      pcg.state = node_child_key(key, k);
      ray.o = mk3(node_word(0, node), node_word(1, node), node_word(2, node));
      const V3 vec = mk3(node_word(3, node), node_word(4, node), node_word(5, node));
      throughput = mk3(node_word(6, node), node_word(7, node), node_word(8, node));
      origin_shape = __float_as_int(node_word(13, node));
      if (meta >> 31) { // SpecularBRDF: every child is the mirror ray, no draw
        ray.d = vec;
        ray.tmin = 1.e-5f;
      } else {
        ray.d = diffuse_scatter_dir<FAST>(vec, pcg);
        ray.tmin = 1.e-3f;
      }
      ray.tmax = INF_F;
      active = true;
    } else if (tid < take + n_new) {
      const int j = tid - take;
      // The q-th sample taken from the queue is sample (q * stride) mod n: neighbouring pixels cost alike
      // (a mirror's image is thousands of expensive samples), so a workgroup must not take a run of them.
      const uint64_t g = ((sh.new_base + (uint64_t)j) * p.sample_stride) % p.n_samples;
      slot = sh.free_list[nfree0 - 1 - j];
      sh.acc[slot][0] = 0ull;
      sh.acc[slot][1] = 0ull;
      sh.acc[slot][2] = 0ull;
      sh.slot_g[slot] = g;
      sh.pending[slot] = 1;
      const SampleId sid = decode_sample(p, g);
      ray = primary_ray(p, sid);
      pcg.state = node_root_key(p.seq_number, sid.global);
      depth = 0;
      origin_shape = -1;
      throughput = mk3(1.f, 1.f, 1.f);
      active = true;
    }
    top -= n_pop;
    __syncthreads(); // every lane has read n_pop / n_new / free_list / owner
    if (tid == 0) {
      sh.n_pop = 0;
      sh.n_free = nfree0 - n_new;
    }
    return take + n_new > 0;
  }

  // PathTracer::operator() steps 2-6 for the node whose closest hit is `h`; pushes the node when it
  // scatters; retires finished samples.  All lanes must call it (it synchronises the workgroup).
  RT_DEV void shade(const Scene &s, const RenderParams &p, PathShared<T> &sh, const Hit &h, const ShadeFetch *fetch = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool push = false;
    V3 vec = mk3(0.f, 0.f, 0.f), weight = vec;
    int brdf = 0;
    if (active) {
      ++rays;
      V3 emitted;
      if (h.object < 0) {
        emitted = mk3(p.background[0], p.background[1], p.background[2]);
      } else {
        float4 r0, r1;
        if (fetch && fetch->have) {
          r0 = fetch->r0, r1 = fetch->r1;
        } else {
          const float4 *rec = reinterpret_cast<const float4 *>(s.shape_shade + h.shape);
          r0 = rec[0], r1 = rec[1];
        }
        const uint32_t mflags = __float_as_uint(r0.w);
        V3 reflected = mk3(r0.x, r0.y, r0.z);
        emitted = mk3(r1.x, r1.y, r1.z);
        if ((mflags & 6u) != 6u) { // a pigment that reads (u, v): the general route
          const uint32_t m = __float_as_uint(r1.w);
          float u, v;
          hit_uv(s, h, m, u, v);
          if (!(mflags & 2u))
            reflected = pigment_eval<FAST>(s, s.material_brdf_pigment[m], u, v);
          if (!(mflags & 4u))
            emitted = pigment_eval<FAST>(s, s.material_emission_pigment[m], u, v);
        }
        const float hit_lum = fmaxf(fmaxf(reflected.x, reflected.y), reflected.z);
        push = hit_lum > 0.f;
        if (depth > p.roulette) {
          const float q = fmaxf(1.f - hit_lum, 0.05f);
          if (pcg_random_float(pcg) > q)
            reflected = scale3(reflected, FAST ? __builtin_amdgcn_rcpf(1.f - q) : 1.f / (1.f - q));
          else
            push = false;
        }
        push = push && depth < p.max_depth; // children deeper than max_depth are BLACK (renderers.hpp:210)
        if (push) {
          brdf = (int)(mflags & 1u);
          if (FAST)
            vec = (brdf == RT_BRDF_SPECULAR) ? specular_scatter_dir_fast(ray.d, h.normal) : normalized3_fast(h.normal);
          else
            vec = (brdf == RT_BRDF_SPECULAR) ? specular_scatter_dir(ray.d, h.normal) : normalized3(h.normal);
          weight = mul3(scale3(throughput, 1.f / (float)p.n_rays), reflected);
        }
      }
      const V3 c = mul3(throughput, emitted);
      if (c.x != 0.f)
        atomicAdd(&sh.acc[slot][0], (unsigned long long)to_fixed(c.x));
      if (c.y != 0.f)
        atomicAdd(&sh.acc[slot][1], (unsigned long long)to_fixed(c.y));
      if (c.z != 0.f)
        atomicAdd(&sh.acc[slot][2], (unsigned long long)to_fixed(c.z));
      const int delta = (push ? p.n_rays : 0) - 1;
      if (delta != 0)
        atomicAdd(&sh.pending[slot], delta);
    }
    // push in reverse lane order
    const unsigned long long pushers = __ballot(push);
    int below = __popcll(pushers & ((1ull << lane) - 1ull)); // pushing lanes before this one
    int n_push = __popcll(pushers);
    if (T > 64) {
      if (lane == 0)
        sh.wave_tot[wave] = n_push;
      __syncthreads();
      n_push = 0;
#pragma unroll
      for (int w = 0; w < T / 64; ++w) {
        const int t = sh.wave_tot[w];
        if (w < wave)
          below += t;
        n_push += t;
      }
    }
    if (push) {
      const int node = top + (n_push - 1 - below);
      node_word(0, node) = h.point.x;
      node_word(1, node) = h.point.y;
      node_word(2, node) = h.point.z;
      node_word(3, node) = vec.x;
      node_word(4, node) = vec.y;
      node_word(5, node) = vec.z;
      node_word(6, node) = weight.x;
      node_word(7, node) = weight.y;
      node_word(8, node) = weight.z;
      node_word(9, node) = __uint_as_float((uint32_t)pcg.state);
      node_word(10, node) = __uint_as_float((uint32_t)(pcg.state >> 32));
      node_word(11, node) = __uint_as_float((uint32_t)slot | ((uint32_t)(depth + 1) << 16) | ((uint32_t)brdf << 31));
      node_word(13, node) = __int_as_float(h.shape);
      set_rem(sh, node, p.n_rays);
    }
    top += n_push;
    __syncthreads(); // accumulators, pending counts and node words are in place
    // retire: slot `tid` is finished when none of its rays is outstanding
    if (sh.pending[tid] == 0) {
      float *out = p.samples_out + 3 * sh.slot_g[tid];
      out[0] = from_fixed(sh.acc[tid][0]);
      out[1] = from_fixed(sh.acc[tid][1]);
      out[2] = from_fixed(sh.acc[tid][2]);
      sh.pending[tid] = -1;
      sh.free_list[atomicAdd(&sh.n_free, 1)] = (short)tid;
    }
    __syncthreads();
  }
};

} // namespace rtd
