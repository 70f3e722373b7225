// rt_wavepath.h — the path kernel for SMALL worlds (fewer than 32 register spheres: the reference's sample scenes,
// BASELINE configs 3 and 4): the node pool of rt_path.h run by ONE WAVE instead of a workgroup.
//
// Why: with a few objects a ray is ~500 instructions, and the workgroup pool spent more time meeting at barriers
// (seven per iteration) and waiting for its node records to come back from L2 than computing (PMC of the workgroup
// kernel on C3: 54 % of wave cycles waiting, the VALU issuing 44 % of the time at 5 waves per SIMD).  A wave needs no
// barrier to agree with itself:
//   * the pool state that is uniform (stack height, free slots, the reserved run of queue samples) lives in scalar
//     registers; prefix sums are DPP row operations, counts are ballots;
//   * the per-slot accumulators, the free list and the children counters are in LDS, private to the wave; LDS
//     operations of one wave execute in program order, so a wavefront-scope fence (no instruction) is all the
//     synchronisation there is;
//   * the node stack (HBM / L2, private to the wave) holds one 64-byte record per node — four 16-byte loads to take a
//     child, four 16-byte stores to push — instead of thirteen strided words;
//   * the waves of a workgroup share only the lookup tables staged in LDS at start-up (one barrier per kernel), each
//     wave leaves when the queue is empty and its own stack is drained.
// The 4 x 5 waves of a CU are then at different points of their iterations at any time, which hides the trip to the
// node records without any software pipelining.
//
// Same estimator, same per-ray streams, same 64-bit fixed-point sums as PathBlock (rt_path.h): frames are bit-identical
// to the workgroup kernel's, to any grid and to any number of GPUs.
#pragma once
#include "rt_path.h"

namespace rtd {

constexpr int WAVE_NODE_WORDS = 16; // point(3) depth-and-slot(1) | vec(3) children-left(1) | weight(3) shape-hit(1) | key(2) pad(2)

struct WaveShared {              // LDS state of one wave: 3712 bytes
  unsigned long long acc[3][64]; // per colour channel and sample slot: fixed-point radiance (slots 8 bytes apart: two slots never share a bank pair)
  unsigned long long slot_g[64]; // per sample slot: sample index of this call
  int pending[64];               // per sample slot: rays not yet shaded (-1: slot free)
  int mark[64];                  // dealing: position c is nonzero if a node's run of children starts at c
  short free_list[64];
  unsigned short rem[8 * 64];    // children left per stack node, when the stack holds at most 512 nodes (else node word 7)
};

// dpp_ctrl encodings (gfx9): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK, int BANK_MASK> RT_DEV int dpp_or_zero(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, BANK_MASK, false); // lanes without a source get 0
}
// Inclusive prefix sum over the 64 lanes of the wave, all lanes active.
RT_DEV int wave_scan_add(int x) {
  int r = x;
  r += dpp_or_zero<0x111, 0xf, 0xf>(x);
  r += dpp_or_zero<0x112, 0xf, 0xf>(x);
  r += dpp_or_zero<0x113, 0xf, 0xf>(x);
  r += dpp_or_zero<0x114, 0xf, 0xe>(r);
  r += dpp_or_zero<0x118, 0xf, 0xc>(r);
  r += dpp_or_zero<0x142, 0xa, 0xf>(r);
  r += dpp_or_zero<0x143, 0xc, 0xf>(r);
  return r;
}
// Orders this wave's LDS / global accesses as written (other lanes of the wave read what a lane wrote): the hardware
// executes a wave's memory instructions in order; this keeps the compiler from moving them across.
RT_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
RT_DEV int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// (Measured and not kept: taking fresh samples only in batches of 4-32 lanes, so that the camera branch runs less often:
// C3 15.7 -> 15.5 ms at best, within noise.)
// RUNS: one LDS atomic per run of lanes instead of one per lane (see shade()).  On for the small-world kernel, where the
// atomics were a fifth of the time; off for the large-world kernels, whose VALU shares its lanes with the matrix
// instructions of the filter and pays more for the extra ~35 instructions than the atomics cost (N=512: -1.7 % with it).
template <bool FAST, bool RUNS = false> struct WavePath {
  // lane state of the current iteration
  bool active;
  Ray ray;
  Pcg pcg;
  V3 throughput;
  int depth, slot;
  int origin_shape; // the shape this ray leaves (-1: a camera ray); read by the large-world kernel (rt_tile.h) only
  int run; // first lane | (one past the last lane) << 8 of this lane's RUN: the lanes dealt children of the same node (a fresh sample: itself)
  uint32_t rays;
  // wave state (uniform: scalar registers)
  int top, n_free, res_count, exhausted;
  unsigned long long res_base;
  float4 *pool; // this wave's node stack, 4 x float4 per node
  int cap;

  RT_DEV void init(const RenderParams &p, WaveShared &sh, int wave_id) {
    const int lane = threadIdx.x & 63;
    active = false;
    ray.o = ray.d = mk3(0.f, 0.f, 0.f);
    ray.tmin = 0.f;
    ray.tmax = INF_F;
    pcg.state = 0;
    pcg.inc = (p.seq_number << 1u) | 1u;
    throughput = mk3(1.f, 1.f, 1.f);
    depth = slot = 0;
    origin_shape = -1;
    run = 0;
    rays = 0;
    top = 0, n_free = 64, res_count = 0, exhausted = 0, res_base = 0;
    cap = p.pool_cap;
    pool = reinterpret_cast<float4 *>(p.pool) + (size_t)wave_id * (WAVE_NODE_WORDS / 4) * (size_t)cap;
    sh.pending[lane] = -1;
    sh.free_list[lane] = (short)lane;
    wave_sync();
  }

  RT_DEV int get_rem(const WaveShared &sh, int node) const {
    return cap <= 8 * 64 ? (int)sh.rem[node] : __float_as_int(reinterpret_cast<const float *>(pool + 4 * (size_t)node)[7]);
  }
  RT_DEV void set_rem(WaveShared &sh, int node, int v) const {
    if (cap <= 8 * 64)
      sh.rem[node] = (unsigned short)v;
    else
      reinterpret_cast<float *>(pool + 4 * (size_t)node)[7] = __int_as_float(v);
  }

  // Deals the children on top of the node stack, then fresh samples, to the lanes.  False when the wave has nothing
  // left to do (uniform).  Same order of dealing as PathBlock::deal: position c of the wave takes the c-th child counted
  // from the top of the stack.
  RT_DEV bool deal(const Scene &s, const RenderParams &p, WaveShared &sh) {
    const int lane = threadIdx.x & 63;
    int rem = 0;
    if (lane < top)
      rem = get_rem(sh, top - 1 - lane);
    const int incl = wave_scan_add(rem);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const int take = min(total, 64);
    const int lo = incl - rem;
    sh.mark[lane] = 0;
    wave_sync();
    if (rem > 0 && lo < 64) {
      sh.mark[lo] = lane + 1;
      if (incl > 64)
        set_rem(sh, top - 1 - lane, incl - 64); // partly used: stays on the stack
    }
    const int n_pop = __popcll(__ballot(rem > 0 && incl <= 64));
    // fresh samples for the lanes the stack does not feed
    int want = min(min(64 - take, n_free), p.fresh_max);
    if (want > 0 && res_count == 0 && !exhausted) {
      const unsigned long long chunk = (unsigned long long)max(p.sample_chunk, want);
      unsigned long long base = 0;
      if (lane == 0)
        base = atomicAdd(p.counters + 0, chunk);
      base = ((unsigned long long)(uint32_t)uniform((int)(base >> 32)) << 32) | (uint32_t)uniform((int)(uint32_t)base);
      const unsigned long long avail = base < p.n_samples ? min(chunk, (unsigned long long)p.n_samples - base) : 0ull;
      if (avail < chunk)
        exhausted = 1;
      res_base = base;
      res_count = (int)avail;
    }
    want = min(want, res_count);
    const unsigned long long new_base = res_base;
    const int n_new = want;
    res_base += (unsigned long long)want;
    res_count -= want;
    wave_sync();
    active = false;
    // owner of position `lane`: the last run start at or before it
    // every node on the stack has children left, so the runs start at the marked positions in lane order: the owner of
    // position c is the (number of marks at or before c)-th node lane
    const unsigned long long starts = __ballot(sh.mark[lane] != 0);
    const int owner = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u)) + (int)((starts >> lane) & 1ull) - 1;
    const int src = max(owner, 0);
    const int owner_lo = __shfl(lo, src, 64), owner_rem = __shfl(rem, src, 64);
    if (lane < take) {
      const int node = top - 1 - owner;
      const uint32_t k = (uint32_t)(p.n_rays - owner_rem + (lane - owner_lo));
      const float4 *rec = pool + 4 * (size_t)node;
      const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
      const uint32_t meta = __float_as_uint(a.w);
      slot = (int)(meta & 0xffffu);
      depth = (int)((meta >> 16) & 0x7fffu);
      const uint64_t key = (uint64_t)__float_as_uint(d.x) | ((uint64_t)__float_as_uint(d.y) << 32);
      pcg.state = node_child_key(key, k);
      ray.o = mk3(a.x, a.y, a.z);
      const V3 vec = mk3(b.x, b.y, b.z);
      throughput = mk3(c.x, c.y, c.z);
      origin_shape = __float_as_int(c.w);
      run = owner_lo | (min(owner_lo + owner_rem, 64) << 8);
      if (meta >> 31) { // SpecularBRDF: every child is the mirror ray, no draw
        ray.d = vec;
        ray.tmin = 1.e-5f;
      } else {
        ray.d = diffuse_scatter_dir<FAST>(vec, pcg);
        ray.tmin = 1.e-3f;
      }
      ray.tmax = INF_F;
      active = true;
    } else if (lane < take + n_new) {
      const int j = lane - take;
      // the q-th sample taken from the queue is sample (q * stride) mod n (see PathBlock::deal)
      const uint64_t g = ((new_base + (uint64_t)j) * p.sample_stride) % p.n_samples;
      slot = sh.free_list[n_free - 1 - j];
      sh.acc[0][slot] = 0ull;
      sh.acc[1][slot] = 0ull;
      sh.acc[2][slot] = 0ull;
      sh.slot_g[slot] = g;
      sh.pending[slot] = 1;
      const SampleId sid = decode_sample(p, g);
      ray = primary_ray(p, sid);
      pcg.state = node_root_key(p.seq_number, sid.global);
      depth = 0;
      origin_shape = -1;
      throughput = mk3(1.f, 1.f, 1.f);
      run = lane | ((lane + 1) << 8);
      active = true;
    }
    top -= n_pop;
    n_free -= n_new;
    wave_sync();
    return take + n_new > 0;
  }

  // PathTracer::operator() steps 2-6 for the node whose closest hit is `h` (see PathBlock::shade): accumulates the
  // emitted radiance, pushes the node when it scatters, retires finished samples.
  RT_DEV void shade(const Scene &s, const RenderParams &p, WaveShared &sh, const Hit &h, const ShadeFetch *fetch = nullptr) {
    const int lane = threadIdx.x & 63;
    bool push = false;
    V3 vec = mk3(0.f, 0.f, 0.f), weight = vec;
    int brdf = 0;
    V3 c = vec; // this ray's contribution to its sample: throughput * emitted radiance
    if (active) {
      ++rays;
      V3 emitted;
      if (h.object < 0) {
        emitted = mk3(p.background[0], p.background[1], p.background[2]);
      } else {
        float4 r0, r1;
        if (fetch && fetch->have) { // the shading record came with the winner's matrices (rt_bvh.h)
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
        // At depth == max_depth nothing is pushed whatever the roulette says, and the rescaled colour and the generator
        // are used by children only: the draw is skipped (most rays are at the deepest level, and the stack order brings
        // them to a wave together, so the whole wave skips it).
        if (depth > p.roulette && depth < p.max_depth) {
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
      c = mul3(throughput, emitted);
    }
    // An LDS atomic costs about one cycle per active lane, and the lanes of a run hit the same address: a 64-lane ds_add_u64
    // was ~50 LDS cycles, three of them plus the pending counter a fifth of the small-world kernels' time.  So each run adds
    // ONCE.  pending: the run's first lane adds n_rays x (pushers of the run) - (lanes of the run).  Radiance: the lanes of a
    // run share their throughput, and rays that reach the same uniform emitter (the sky) then contribute the very same
    // number: the first contributing lane of the run (its leader) adds  count x value  for all lanes whose contribution
    // equals its own, bit for bit; a lane with another value adds for itself.  Integer sums: the same total, whoever adds.
    const unsigned long long pushers = __ballot(push);
    if (RUNS && p.n_rays <= 31) { // a run fits a 32-bit mask (uniform branch)
      const int run_lo = run & 0xff, run_len = (run >> 8) - run_lo;
      const uint32_t len_mask = (1u << run_len) - 1u;
      const bool nz = active && (c.x != 0.f || c.y != 0.f || c.z != 0.f);
      const uint32_t contributors = (uint32_t)(__ballot(nz) >> run_lo) & len_mask; // bit k: lane run_lo + k contributes
      const int leader = contributors ? run_lo + __builtin_ctz(contributors) : lane;
      const float lx = __shfl(c.x, leader, 64), ly = __shfl(c.y, leader, 64), lz = __shfl(c.z, leader, 64);
      const bool same = nz && c.x == lx && c.y == ly && c.z == lz;
      const uint32_t same_mask = (uint32_t)(__ballot(same) >> run_lo) & len_mask;
      if (active && lane == run_lo) {
        const int delta = p.n_rays * __popc((uint32_t)(pushers >> run_lo) & len_mask) - run_len;
        if (delta != 0)
          atomicAdd(&sh.pending[slot], delta);
      }
      if (nz && (lane == leader || !same)) {
        const unsigned long long count = lane == leader ? (unsigned long long)__popc(same_mask) : 1ull;
        unsigned long long fx = to_fixed_abs(c.x), fy = to_fixed_abs(c.y), fz = to_fixed_abs(c.z);
        if (__ballot(fminf(fminf(c.x, c.y), c.z) < 0.f)) { // negative radiance: in no scene of the reference; a uniform branch
          fx = c.x < 0.f ? 0ull - fx : fx, fy = c.y < 0.f ? 0ull - fy : fy, fz = c.z < 0.f ? 0ull - fz : fz;
        }
        atomicAdd(&sh.acc[0][slot], fx * count);
        atomicAdd(&sh.acc[1][slot], fy * count);
        atomicAdd(&sh.acc[2][slot], fz * count);
      }
    } else if (active) { // long runs: every lane adds for itself
      if (c.x != 0.f)
        atomicAdd(&sh.acc[0][slot], (unsigned long long)to_fixed(c.x));
      if (c.y != 0.f)
        atomicAdd(&sh.acc[1][slot], (unsigned long long)to_fixed(c.y));
      if (c.z != 0.f)
        atomicAdd(&sh.acc[2][slot], (unsigned long long)to_fixed(c.z));
      const int delta = (push ? p.n_rays : 0) - 1;
      if (delta != 0)
        atomicAdd(&sh.pending[slot], delta);
    }
    // push in reverse lane order (the lane that took the topmost child pushes topmost: see rt_path.h on the stack bound)
    const int n_push = __popcll(pushers);
    if (push) {
      const int below = __popcll(pushers & ((1ull << lane) - 1ull));
      const int node = top + (n_push - 1 - below);
      float4 *rec = pool + 4 * (size_t)node;
      rec[0] = make_float4(h.point.x, h.point.y, h.point.z, __uint_as_float((uint32_t)slot | ((uint32_t)(depth + 1) << 16) | ((uint32_t)brdf << 31)));
      rec[1] = make_float4(vec.x, vec.y, vec.z, __int_as_float(p.n_rays));
      rec[2] = make_float4(weight.x, weight.y, weight.z, __int_as_float(h.shape));
      rec[3] = make_float4(__uint_as_float((uint32_t)pcg.state), __uint_as_float((uint32_t)(pcg.state >> 32)), 0.f, 0.f);
      if (cap <= 8 * 64)
        sh.rem[node] = (unsigned short)p.n_rays;
    }
    top += n_push;
    wave_sync(); // accumulators, pending counts and node records are in place
    // retire: slot `lane` is finished when none of its rays is outstanding
    const bool done = sh.pending[lane] == 0;
    const unsigned long long retiring = __ballot(done);
    if (done) {
      float *out = p.samples_out + 3 * sh.slot_g[lane];
      out[0] = from_fixed(sh.acc[0][lane]);
      out[1] = from_fixed(sh.acc[1][lane]);
      out[2] = from_fixed(sh.acc[2][lane]);
      sh.pending[lane] = -1;
      sh.free_list[n_free + __popcll(retiring & ((1ull << lane) - 1ull))] = (short)lane;
    }
    n_free += __popcll(retiring);
    wave_sync();
  }
};

// ---------------------------------------------------------------------------------------------
// The kernel: 4 waves per workgroup share the lookup tables (dynamic LDS), each wave runs its own pool.
// ---------------------------------------------------------------------------------------------
constexpr int WAVEPATH_THREADS = 256, WAVEPATH_WAVES = WAVEPATH_THREADS / 64;

struct WaveLdsLayout { // byte offsets into the dynamic LDS block; the same function lays it out on host and device
  uint32_t w2l, l2w, shade, kind, mat_brdf_pig, mat_emit_pig, mat_needs_uv, pig_kind, pig_c1, pig_c2, pig_n, csg_leaf, csg_leaf_tt, csg_path,
      csg_range, csg_op, csg_prog, csg_tops, total;
};
__host__ __device__ constexpr WaveLdsLayout wave_lds_layout(int n_shapes, int n_materials, int n_pigments, int n_leaves, int n_path, int n_nodes,
                                                         int n_prog, int n_tops) {
  WaveLdsLayout L{};
  uint32_t off = WAVEPATH_WAVES * (uint32_t)sizeof(WaveShared);
  auto take = [&off](uint32_t bytes) {
    const uint32_t at = off;
    off += (bytes + 15u) & ~15u;
    return at;
  };
  L.w2l = take(48u * n_shapes), L.l2w = take(48u * n_shapes), L.shade = take((uint32_t)sizeof(ShadeRec) * n_shapes), L.kind = take(n_shapes);
  L.mat_brdf_pig = take(4u * n_materials), L.mat_emit_pig = take(4u * n_materials), L.mat_needs_uv = take(n_materials);
  L.pig_kind = take(n_pigments), L.pig_c1 = take(12u * n_pigments), L.pig_c2 = take(12u * n_pigments), L.pig_n = take(4u * n_pigments);
  L.csg_leaf = take((uint32_t)sizeof(CsgLeaf) * n_leaves), L.csg_leaf_tt = take(8u * n_leaves), L.csg_path = take((uint32_t)sizeof(CsgStep) * n_path);
  L.csg_range = take(16u * n_nodes), L.csg_op = take(n_nodes), L.csg_prog = take((uint32_t)sizeof(CsgProgEntry) * n_prog);
  L.csg_tops = take((uint32_t)sizeof(CsgTop) * n_tops);
  L.total = off;
  return L;
}

template <typename T> RT_DEV void stage_table_at(const T *&table, unsigned char *lds, uint32_t offset, int n) {
  T *dst = reinterpret_cast<T *>(lds + offset);
  for (int i = threadIdx.x; i < n; i += WAVEPATH_THREADS)
    dst[i] = table[i];
  table = dst;
}

// The layout the kernel variant <.., CSG> uses, for the host's launch and the kernel itself: laid out for the LIMITS, not for
// a scene's table sizes, so that the offsets are compile-time constants (immediate offsets of the ds_read instructions
// instead of ~20 scalar registers of addresses; measured 2-3 %).  31 KB per workgroup with the CSG tables (5 workgroups per
// CU), 25 KB without (6).
template <bool CSG> constexpr WaveLdsLayout wave_lds_layout_for() {
  using LIM = PathLdsLimits;
  return wave_lds_layout(LIM::SHAPES, LIM::MATERIALS, LIM::PIGMENTS, CSG ? LIM::CSG_LEAVES : 0, CSG ? LIM::CSG_PATH : 0, CSG ? LIM::CSG_NODES : 0,
                         CSG ? LIM::CSG_PROG : 0, CSG ? LIM::CSG_TOPS : 0);
}

// CSG = false: the variant for worlds without CSG objects (no evaluator code, no CSG tables: fewer registers, less LDS).
// TT4 = true: every CSG tree has a truth table and at most 4 leaves: only that evaluator is compiled in (fewer registers).
template <bool FAST, int WAVES_PER_SIMD, bool CSG = true, bool TT4 = false>
__global__ void __launch_bounds__(WAVEPATH_THREADS, WAVES_PER_SIMD) k_render_path_wave(Scene s, RenderParams p) {
  extern __shared__ __align__(16) unsigned char wave_lds[];
  constexpr WaveLdsLayout L = wave_lds_layout_for<CSG>();
  {
    const int ns = p.tab_shapes, nm = p.tab_materials, np = p.tab_pigments;
    stage_table_at(s.shape_w2l, wave_lds, L.w2l, 12 * ns);
    stage_table_at(s.shape_l2w, wave_lds, L.l2w, 12 * ns);
    stage_table_at(s.shape_shade, wave_lds, L.shade, ns);
    stage_table_at(s.shape_kind, wave_lds, L.kind, ns);
    stage_table_at(s.material_brdf_pigment, wave_lds, L.mat_brdf_pig, nm);
    stage_table_at(s.material_emission_pigment, wave_lds, L.mat_emit_pig, nm);
    stage_table_at(s.material_needs_uv, wave_lds, L.mat_needs_uv, nm);
    stage_table_at(s.pigment_kind, wave_lds, L.pig_kind, np);
    stage_table_at(s.pigment_color1, wave_lds, L.pig_c1, 3 * np);
    stage_table_at(s.pigment_color2, wave_lds, L.pig_c2, 3 * np);
    stage_table_at(s.pigment_n_intervals, wave_lds, L.pig_n, np);
    if constexpr (CSG) {
      stage_table_at(s.csg_leaf, wave_lds, L.csg_leaf, p.tab_csg_leaves);
      stage_table_at(s.csg_leaf_tt, wave_lds, L.csg_leaf_tt, p.tab_csg_leaves);
      stage_table_at(s.csg_path, wave_lds, L.csg_path, p.tab_csg_path);
      stage_table_at(s.csg_prog_range, wave_lds, L.csg_range, 4 * p.tab_csg_nodes);
      stage_table_at(s.csg_op, wave_lds, L.csg_op, p.tab_csg_nodes);
      stage_table_at(s.csg_prog, wave_lds, L.csg_prog, p.tab_csg_prog);
      stage_table_at(s.csg_tops, wave_lds, L.csg_tops, s.n_csg_tops);
    }
    if (FAST) { // the FMA code reads these two only for top-level shapes (make_hit_fast), where they equal the composed tables
      s.shape_inv = s.shape_w2l;
      s.shape_fwd = s.shape_l2w;
    }
  }
  const int wave = threadIdx.x >> 6;
  WaveShared &sh = reinterpret_cast<WaveShared *>(wave_lds)[wave];
  WavePath<FAST, true> B;
  B.init(p, sh, blockIdx.x * WAVEPATH_WAVES + wave);
  __syncthreads(); // the tables are in place; from here on the waves of the workgroup never meet again
#ifdef RT_PROFILE_PHASES
  unsigned long long t_deal = 0, t_trace = 0, t_shade = 0, n_iter = 0, n_live = 0;
#define RT_WSTAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define RT_WSTAMP(var)
#endif
  for (;;) {
    RT_WSTAMP(ts0);
    if (!B.deal(s, p, sh))
      break;
    RT_WSTAMP(ts1);
    Hit h;
    h.object = h.shape = -1;
    h.t = 0.f;
    h.point = h.normal = h.local = mk3(0.f, 0.f, 0.f);
    if (B.active)
            h = world_ray_intersection<FAST, CSG, TT4, true>(s, B.ray); // TAB: the shape tables are in LDS
    RT_WSTAMP(ts2);
    B.shade(s, p, sh, h);
#ifdef RT_PROFILE_PHASES
    RT_WSTAMP(ts3);
    t_deal += ts1 - ts0, t_trace += ts2 - ts1, t_shade += ts3 - ts2, n_iter += 1, n_live += __popcll(__ballot(B.active));
#endif
  }
#ifdef RT_PROFILE_PHASES
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(p.counters + 4, t_deal);
    atomicAdd(p.counters + 5, t_trace);
    atomicAdd(p.counters + 6, t_shade);
    atomicAdd(p.counters + 7, n_iter);
    atomicAdd(p.counters + 8, n_live);
  }
#endif
  wave_add_tally(p.counters, 0, B.rays);
}

} // namespace rtd
