// rt_tile_cull.h — the tile kernel (rt_tile.h, bf16 build) for worlds of many tiles: a sphere tile is multiplied only with the
// rays whose line meets the tile's bounding sphere.
//
// k_render_path_tile tests every queued ray against every sphere: 128 tile products per 32 rays at N = 4096, of which a ray's
// line meets a sphere in a handful.  Here the host also keeps the register spheres as spatial CLUSTERS of 32 (a k-d split of
// the centres, rt_scene.hip) with one bounding sphere per cluster, and an iteration of a workgroup runs per GROUP of 32 clusters per wave:
//   level 1  every wave multiplies ALL the workgroup's rays with the bounds of ITS OWN 32 clusters of the group (the clusters are dealt
//            to the waves round-robin) — the very products of the sphere filter, a bound is a sphere — and keeps per ray a 32-bit
//            word in registers: bit i set if the ray's line meets the bound of the wave's i-th cluster with the far root beyond the
//            ray's tmin (a cluster wholly behind the ray cannot hold its hit).  Nothing is exchanged between the waves.
//   level 2  every wave takes its share of the group's clusters; per cluster it collects the rays of the WHOLE workgroup whose bit
//            is set (one ballot per 64 rays, the slot numbers compacted into a byte list in LDS) and multiplies the cluster's tile
//            with those rays only, 16 at a time (v_mfma_f32_16x16x32_bf16: a cluster meets 9 rays of a workgroup on average) — the
//            same K slots, hence the same sums, candidates and keys as k_render_path_tile forms them.
// The result is the same: a ray whose line misses a cluster's bound misses its spheres, a product element does not depend on
// which other rays share its block, so every (ray, sphere) pair that could give a candidate is evaluated with the same bits and
// the keys' minimum is the same — the frames are bit-identical to the unculled kernel's (tests/test_tile_cull.py).  A key
// carries the sphere's position in OBJECT order (Scene::cull_rank), so equal distances still resolve as the reference's `<`.
// The bounds are inflated by more than the filter's rounding can move either level-1 decision (rt_scene.hip).
//
// Work per iteration of a 256-ray workgroup at N = 4096 (128 clusters, a ray meets 4.6 of them on average): 4 x 8 bound products
// + 115 lists of on average 9 rays (16-ray blocks), against 128 x 8 full 32-ray blocks before; measured 2 449 -> 5 589 Mrays/s
// (DESIGN.md 4.4b has the counters, what bounds the kernel now, and what was tried and not kept).
#pragma once
#include "rt_tile.h"

namespace rtd {

// the two products of one tile (a0 | a1: its operand words for K chunks 0 | 1) with one block of 32 rays
RT_DEV void tile_products_bf16(const float4 &a0, const float4 &a1, const float4 &qb0, const float4 &qb1, const float4 &qc0, const float4 &qc1,
                               f32x16 &b, f32x16 &c) {
  union Op {
    bf16x8 v;
    float f[4];
  };
  auto op = [](const float4 &q) {
    Op o;
    o.f[0] = q.x, o.f[1] = q.y, o.f[2] = q.z, o.f[3] = q.w;
    return o.v;
  };
  b = f32x16{0}, c = f32x16{0};
  b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(a0), op(qb0), b, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(a0), op(qc0), c, 0, 0, 0);
  b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(a1), op(qb1), b, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op(a1), op(qc1), c, 0, 0, 0);
}

template <int W, bool CSG>
__global__ void __launch_bounds__(64 * W, RT_TILE_OCC) k_render_path_tile_cull(Scene s, RenderParams p) {
  constexpr int T = 64 * W;
  static_assert(T <= 256 && W == RT_CULL_W, "ray slots are listed as bytes; the host deals the clusters to RT_CULL_W waves");
  __shared__ TileRayBf q_ray[T];
  __shared__ unsigned long long q_key[T];
  __shared__ uint8_t q_list[W][T];   // per wave: the slots of the rays it multiplies the current cluster with
  __shared__ int q_rank[W][32];      // per wave: the current cluster's positions in object order (Scene::cull_rank)
  __shared__ int q_cnt[W], q_live[W];
  __shared__ WaveShared sh_all[W];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  WaveShared &sh = sh_all[wave];
  const int n_tiles = s.n_cull_tiles;
  constexpr int C = RT_CULL_C;                                  // tiles per cluster
  const int n_clusters = (n_tiles + C - 1) / C;
  const int n_groups = (n_clusters + 32 * W - 1) / (32 * W); // a group: 32 clusters per wave, cluster 32 W g + W i + wave is this wave's i-th
  const float4 *ray_quads = reinterpret_cast<const float4 *>(q_ray);
  WavePath<true> L;
  L.init(p, sh, blockIdx.x * W + wave);
#ifdef RT_PROFILE_PHASES
  unsigned long long t_queue = 0, t_l1 = 0, t_l2 = 0, t_shade = 0, n_iter = 0, n_rays_q = 0, n_blocks = 0, n_pairs = 0, n_lists = 0;
#define RT_CSTAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define RT_CSTAMP(var)
#endif

  // the B operands and the lane's own quad of the ray in `slot`
  struct RayBlock {
    float4 qb0, qb1, qc0, qc1, qc;
  };
  auto load_block = [&](int slot) {
    const float4 *quads = ray_quads + 8 * slot;
    const int sw = (slot >> 1) & 7;
    RayBlock r;
    r.qb0 = quads[half ^ sw], r.qb1 = quads[2 ^ sw], r.qc0 = quads[(3 + half) ^ sw], r.qc1 = quads[5 ^ sw], r.qc = quads[6 ^ sw];
    return r;
  };
  auto load_tile = [&](const float *tiles, int tile, float4 &a0, float4 &a1) {
    const float4 *rec = reinterpret_cast<const float4 *>(tiles + (size_t)tile * RT_TILE_WORDS);
    a0 = rec[lane], a1 = rec[64 + lane];
  };
  auto load_cluster = [&](int tile, float4 &a0, float4 &a1, int &rk) {
    load_tile(s.cull_tiles, tile, a0, a1);
    rk = s.cull_rank[tile * 32 + col];
  };

  for (;;) {
    // -- every lane gets a ray from its wave's node pool; the live rays are queued, compacted per wave (as rt_tile.h) ----
    RT_CSTAMP(ts0);
    const bool wave_live = L.deal(s, p, sh);
    const unsigned long long live = __ballot(L.active);
    if (lane == 0)
      q_cnt[wave] = __popcll(live), q_live[wave] = wave_live ? 1 : 0;
    int my_q = -1;
    if (L.active) {
      my_q = 64 * wave + __popcll(live & ((1ull << lane) - 1ull));
      int self = L.origin_shape >= 0 ? s.shape_origin[4 * L.origin_shape + 1] : -1;
      if (self >= 0)
        self = s.cull_pos[self];
      TileRayBf::make(L.ray, self).store(reinterpret_cast<float4 *>(q_ray), my_q);
      q_key[my_q] = ~0ull;
    }
    __syncthreads();
    int any_live = 0, cnt[W]; // uniform
#pragma unroll
    for (int w = 0; w < W; ++w) {
      cnt[w] = __builtin_amdgcn_readfirstlane(q_cnt[w]);
      any_live |= __builtin_amdgcn_readfirstlane(q_live[w]);
    }
    if (!any_live)
      break;
    RT_CSTAMP(ts1);
#ifdef RT_PROFILE_PHASES
    t_queue += ts1 - ts0, n_iter += 1, n_rays_q += __builtin_amdgcn_readfirstlane(q_cnt[wave]);
#endif

    for (int g = 0; g < n_groups; ++g) {
      RT_CSTAMP(tg0);
      // -- level 1: every ray of the workgroup against the bounds of this wave's 32 clusters of the group -------------
      // lane l ends up with the word of the ray in slot 64 w + l in mw[w] (bit i: the ray's line meets the bound of the wave's i-th
      // cluster with the far root beyond tmin) — in registers, nothing is exchanged between the waves
      uint32_t mw[W];
#pragma unroll
      for (int w = 0; w < W; ++w)
        mw[w] = 0u;
      {
        float4 a0, a1;
        load_tile(s.cull_bounds, g * W + wave_u, a0, a1);
#pragma unroll 1
        for (int blk = 0; blk < 2 * W; ++blk) {
          const int w_src = blk >> 1, r0 = 32 * (blk & 1);
          int n_src = cnt[0];
#pragma unroll
          for (int w = 1; w < W; ++w)
            n_src = w_src == w ? cnt[w] : n_src;
          if (r0 >= n_src)
            continue;
          const bool valid = r0 + col < n_src;
          const int slot = 64 * w_src + (valid ? r0 + col : 0);
          const RayBlock rb = load_block(slot);
          const float oo = valid ? rb.qc.x : INF_F, tmin_u = rb.qc.y;
          f32x16 b, c;
          tile_products_bf16(a0, a1, rb.qb0, rb.qb1, rb.qc0, rb.qc1, b, c);
          // bit r (from the top, in the order the registers come): the line meets bound r (d > O.O) and the far root
          // -b + sqrt(d - O.O) is beyond tmin, i.e. NOT (b + tmin > 0 and d - O.O < (b + tmin)^2); all in sign bits
          uint32_t m = 0;
#pragma unroll
          for (int r = 15; r >= 0; --r) {
            const float d = fmaf(b[r], b[r], -c[r]);
            const float miss = oo - d;           // < 0: the line meets the bound
            const float x = b[r] + tmin_u;       // < 0: the bound's centre is ahead of tmin
            const float y = fmaf(x, x, miss);    // < 0: the far root is beyond tmin
            const uint32_t keep = __float_as_uint(miss) & (__float_as_uint(x) | __float_as_uint(y));
            m = __builtin_amdgcn_alignbit(m, keep, 31); // m << 1 | sign of keep
          }
          // bit r = register r = row 8 (r / 4) + 4 half + r % 4 of the tile
          uint32_t word = (m & 0xfu) | ((m & 0xf0u) << 4) | ((m & 0xf00u) << 8) | ((m & 0xf000u) << 12);
          word <<= 4 * half;
          word |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(word), 32)); // the other half of the wave holds the ray's other 16 rows
          const bool mine = half == (blk & 1); // lanes 0-31 keep the words of rays 0-31 of the 64, lanes 32-63 of rays 32-63
#pragma unroll
          for (int w = 0; w < W; ++w)
            mw[w] = (mine && w_src == w) ? word : mw[w];
        }
      }
      RT_CSTAMP(tg1);
      // -- level 2: this wave's clusters against the rays that meet their bounds --------------------------------
      const int first = 32 * W * g + wave_u; // this wave's first cluster of the group
      int n_tot = 0;                         // rays listed for the cluster at hand
      // the rays of the workgroup that meet the bound of the wave's i-th cluster, listed in q_list[wave]
      auto list_rays = [&](int tb) {
        wave_sync(); // the list may still be read
        // all four ballots first, then the offsets (scalar), then the stores: the four chains run side by side
        unsigned long long bal[W];
        bool meets[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          meets[w] = (mw[w] >> tb) & 1u;
          bal[w] = __ballot(meets[w]);
        }
        int off[W + 1];
        off[0] = 0;
#pragma unroll
        for (int w = 0; w < W; ++w)
          off[w + 1] = off[w] + __popcll(bal[w]);
#pragma unroll
        for (int w = 0; w < W; ++w)
          if (meets[w])
            q_list[wave][off[w] + __builtin_amdgcn_mbcnt_hi((unsigned)(bal[w] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[w], 0u))] =
                static_cast<uint8_t>(64 * w + lane);
        n_tot = __builtin_amdgcn_readfirstlane(off[W]);
#ifdef RT_PROFILE_PHASES
        n_pairs += n_tot, n_blocks += (n_tot + 15) / 16, n_lists += 1;
#endif
      };
      // One tile against the listed rays, 16 rays at a time on v_mfma_f32_16x16x32_bf16: K = 32 holds all 24 slots of a product, so a
      // block of 16 rays x 32 spheres is four MFMAs of 16 cycles (two products x two halves of the tile) and 8 values of d per lane —
      // half the matrix cycles, half the vector work and 3 instead of 5 operand reads per lane of what a 32-ray block costs, and
      // most clusters meet fewer than 16 rays of a workgroup.  a0 | a1: the operand words of the tile's rows 0-15 | 16-31 (lane l: row
      // l % 16, K slots 8 (l / 16) ... + 7, rt_scene.hip); lane l is ray l % 16 of the block and receives rows 4 (l / 16) + 0..3 of
      // each half.  rk: the rows' positions in object order.
      auto do_tile = [&](int tile, const float4 &a0, const float4 &a1, int rk) {
        union Op {
          bf16x8 v;
          float f[4];
        };
        auto op = [](const float4 &q) {
          Op o;
          o.f[0] = q.x, o.f[1] = q.y, o.f[2] = q.z, o.f[3] = q.w;
          return o.v;
        };
        if (half == 0)
          q_rank[wave][col] = rk;
        wave_sync();
        const int ray16 = lane & 15, kq = lane >> 4, kb = kq < 3 ? kq : 0; // K slots 24-31 are zero on the sphere side: any finite words do
#pragma unroll 1
        for (int r0 = 0; r0 < n_tot; r0 += 16) {
          const bool valid = r0 + ray16 < n_tot;
          const int slot = q_list[wave][valid ? r0 + ray16 : 0];
          const float4 *quads = ray_quads + 8 * slot;
          const int sw = (slot >> 1) & 7;
          const float4 qb = quads[kb ^ sw], qcq = quads[(3 + kb) ^ sw], own = quads[6 ^ sw];
          const float oo = valid ? own.x : INF_F; // a lane past the list's end asks d > +inf
          const float tmin_u = own.y, inv_len = own.z;
          const int self = __float_as_int(own.w);
          f32x4 b[2], c[2];
          b[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op(a0), op(qb), f32x4{0}, 0, 0, 0);
          c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op(a0), op(qcq), f32x4{0}, 0, 0, 0);
          b[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op(a1), op(qb), f32x4{0}, 0, 0, 0);
          c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op(a1), op(qcq), f32x4{0}, 0, 0, 0);
          float d[2][4];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              d[h][r] = fmaf(b[h][r], b[h][r], -c[h][r]);
          const float g0 = fmaxf(fmaxf(d[0][0], d[0][1]), fmaxf(d[0][2], d[0][3])), g1 = fmaxf(fmaxf(d[1][0], d[1][1]), fmaxf(d[1][2], d[1][3]));
          if (fmaxf(g0, g1) > oo) { // this ray's line meets one of the cluster's spheres
            unsigned long long best = ~0ull;
            const int base = tile * 32 + 4 * kq; // cluster position of this lane's row 0
            const int self_rel = self - base;
            auto candidate = [&](int h, int r) {
              if (d[h][r] > oo) {
                const int row = 16 * h + r; // a literal
                // front root as a distance (the reference's rule: t1 unless it is below tmin, then t2 if beyond tmin)
                const float sq = __builtin_amdgcn_sqrtf(d[h][r] - oo);
                const float u1 = -b[h][r] - sq, u2 = -b[h][r] + sq;
                const float u = u1 < tmin_u ? u2 : u1;
                if (u > tmin_u && self_rel != row) {
                  // within a cluster the rows ascend with the object index (rt_scene.hip): equal t resolve as the reference's strict `<`
                  const unsigned long long key = ((unsigned long long)__float_as_uint(u * inv_len) << 32) | (unsigned)(4 * kq + row);
                  best = min(best, key);
                }
              }
            };
            if (g0 > oo) {
              candidate(0, 0), candidate(0, 1), candidate(0, 2), candidate(0, 3);
            }
            if (g1 > oo) {
              candidate(1, 0), candidate(1, 1), candidate(1, 2), candidate(1, 3);
            }
            if (best != ~0ull) {
              // across clusters the key carries the sphere's position in OBJECT order
              const unsigned rank = static_cast<unsigned>(q_rank[wave][(unsigned)best & 31u]);
              atomicMin(&q_key[slot], (best & 0xffffffff00000000ull) | rank);
            }
          }
        }
      };
      // This wave's tiles of the group in sequence (cluster after cluster, C tiles each); step q: cluster first + W (q / C), its
      // tile q % C.  The tiles come from L2 one step ahead, into two register sets used in turn (copying from a "next" set
      // makes the compiler wait for the load where it is issued).
      auto tile_of = [&](int q) { // -1: past the end
        const int cluster = first + W * (q / C), tile = cluster * C + q % C;
        return q < 32 * C && cluster < n_clusters && tile < n_tiles ? tile : -1;
      };
      float4 e0, e1, o0, o1;
      int e_rk = 0, o_rk = 0;
      if (tile_of(0) >= 0)
        load_cluster(tile_of(0), e0, e1, e_rk);
#pragma unroll 1
      for (int q = 0; q < 32 * C; q += 2) {
        const int t_even = tile_of(q), t_odd = tile_of(q + 1), t_next = tile_of(q + 2); // uniform
        if (t_even < 0)
          break;
        if (t_odd >= 0)
          load_cluster(t_odd, o0, o1, o_rk);
        if (q % C == 0)
          list_rays(q / C);
        if (n_tot > 0)
          do_tile(t_even, e0, e1, e_rk);
        if (t_odd < 0)
          break;
        if (t_next >= 0)
          load_cluster(t_next, e0, e1, e_rk);
        if ((q + 1) % C == 0)
          list_rays((q + 1) / C);
        if (n_tot > 0)
          do_tile(t_odd, o0, o1, o_rk);
      }
#ifdef RT_PROFILE_PHASES
      RT_CSTAMP(tg2);
      t_l1 += tg1 - tg0, t_l2 += tg2 - tg1;
#endif
    }
    __syncthreads();
    RT_CSTAMP(ts2);

    // -- every lane is a ray again: finish the query, shade, produce the next ray (as rt_tile.h) -------------------
    Hit h;
    h.object = h.shape = -1;
    h.t = 0.f;
    h.point = h.normal = h.local = mk3(0.f, 0.f, 0.f);
    TileRayBf::restore(ray_quads, max(my_q, 0), L.ray); // every lane, so that no lane's old values stay live across the scan
    if (L.active) {
      const unsigned long long key = q_key[my_q];
      Best best;
      best.t = INF_F;
      best.object = 0x7fffffff;
      best.shape = -1;
      scan_spheres<true>(s.wide_lane, s.n_wide_lane, L.ray, best, L.origin_shape); // in object order: strict `<` among them
      self_sphere_exact(s, L.origin_shape, L.ray, best);
      if (key != ~0ull) {
        const int2 ids = *reinterpret_cast<const int2 *>(&s.wide_recs[(int)(unsigned)(key & 0xffffffffull)].object); // {object, shape}
        const float t = __uint_as_float((unsigned)(key >> 32));
        if (t < best.t || (t == best.t && ids.x < best.object)) {
          best.t = t;
          best.object = ids.x;
          best.shape = ids.y;
        }
      }
      h = world_finish_hit<true, true, CSG>(s, L.ray, best);
    }
    L.shade(s, p, sh, h);
#ifdef RT_PROFILE_PHASES
    RT_CSTAMP(ts3);
    t_shade += ts3 - ts2;
#endif
  }
#ifdef RT_PROFILE_PHASES
  if (lane == 0) { // per wave; counters 4-12
    atomicAdd(p.counters + 4, t_queue), atomicAdd(p.counters + 5, t_l1), atomicAdd(p.counters + 6, t_l2), atomicAdd(p.counters + 11, t_shade);
    atomicAdd(p.counters + 7, n_iter), atomicAdd(p.counters + 8, n_rays_q), atomicAdd(p.counters + 10, n_blocks), atomicAdd(p.counters + 12, n_pairs);
    atomicAdd(p.counters + 13, n_lists);
  }
#endif
  wave_add_tally(p.counters, 0, L.rays);
}

} // namespace rtd
