// rt_tile.h — the path kernel for large worlds of translation * uniform-scaling spheres: the closest-hit FILTER on the
// matrix cores, 32 rays x 32 spheres per tile.  Two builds of one kernel: the default multiplies on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, every binary32 operand split exactly into three bf16 parts: the second comment block below);
// -DRT_TILE_BF16=0 keeps round 2's binary32 MFMAs (v_mfma_f32_32x32x2_f32: the first block) for A/B runs.
#pragma once
#include <type_traits>

#include "rt_wavepath.h"
#include "rt_wide.h"

namespace rtd {

// ---------------------------------------------------------------------------------------------
// The filter as a matrix product, and its binary32 form (round 2).
// Same pool, queue and finish as k_render_path_wide (rt_wide.h); what changes is who evaluates the filter
//     b = O.u - C.u,   c = (O.O + k) - 2 O.C,   disc = b^2 - c          (u = D / |D|, k = C.C - r^2; see WideRayT<true>)
// b and c are inner products of a per-ray vector with a per-sphere vector, so a block of rays against a block of spheres
// is a small matrix product, and binary32 in / binary32 accumulate MFMAs compute exactly such fmaf chains at the full fp32
// rate of the SIMD (64 flop per cycle) with one instruction per 4096 flops:
//     b[i][j] = sum_k A_b[i][k] B_b[k][j],   A_b = ( cx  cy  cz  1 ) per sphere i,   B_b = ( -ux  -uy  -uz  O.u ) per ray j
//     c[i][j] = sum_k A_c[i][k] B_c[k][j],   A_c = ( cx  cy  cz  k ),               B_c = ( -2ox -2oy -2oz  1  )
// two v_mfma_f32_32x32x2_f32 (K = 2 each) per product, four per 1024 tests = 256 cycles per SIMD, against 8 VALU
// instructions per 64 tests = 256 cycles at best (625 measured) for the same tests before.  O.O stays out of the product (it
// is the ray's own constant): the VALU forms d = b^2 - c (16 fma per lane per tile) and asks d > O.O.
//
// Operand layout of the instruction (probed on the device, scripts/microbench/mfma_probe.hip): lane l supplies
// A[i = l % 32][k = l / 32] and B[k = l / 32][j = l % 32]; register r of lane l receives D[i = 8 (r / 4) + 4 (l / 32) + r % 4]
// [j = l % 32].  With spheres as rows and rays as columns a lane IS a ray (j) for the whole tile: what the hit path needs of
// the ray (O.O, tmin, 1 / |D|, the sphere it leaves) sits in that lane's registers, the sphere's position follows from r,
// and a lane merges the closest of its candidates over all the wave's sphere tiles into the ray's LDS key once.
//
// Spheres in registers: 3 per lane per 32-sphere tile (a1 = cx | cy for the lower | upper half of the wave, a2 = cz | 1,
// a3 = cz | k), prepared by the host (Scene::wide_tiles).  S = 8 tiles per wave and pass (256 spheres in 24 registers; bigger
// worlds take several passes, the registers reloaded from L2 per pass).  Measured, N=512, W=1: S=4 9822, S=8 9926, S=16 8900
// Mrays/s (at 16 the lane state that shading needs is spilled around the scan); 5 waves per SIMD (96 VGPRs): 8950.
// With the per-wave pools: S=4 / 8 / 16 10 451 / 10 635 / 10 583; 5 waves per SIMD 10 754 against 10 710 at 4 (31 spilled
// registers: within noise, not taken).
//
// What it buys (N=512 1080p AA=4 / a 1/16 slice of the N=4096 frame): 7.60 -> 9.94 and 1.02 -> 1.46 Grays/s over the VALU filter of
// rt_wide.h.  The binary32 MFMA and the VALU share the SIMD's fp32 lanes (scripts/microbench/mfma_probe.hip: 4 MFMAs alone
// take 278 cycles per tile, + 24 VALU instructions 345, + 72 459 — the times add, they do not overlap), so the gain is not a
// second pipe but instruction efficiency: the 4 MFMAs carry 16 of the 18 flops per test at the full rate of 64 per cycle and
// SIMD, where the VALU filter issued its 8 instructions per test at 3.9 cycles each instead of 2.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS image of one queued ray: quad 0 is read by lanes 0-31 (k = 0 of every product), quad 1 by lanes 32-63 (k = 1),
// quad 2 by both.
struct alignas(16) TileRay {
  float nux, nuz, m2ox, m2oz;  // -ux, -uz, -2 ox, -2 oz
  float nuy, ou, m2oy, tmin;   // -uy, O.u, -2 oy; the reader puts 1 in the fourth place, the slot keeps the ray's own tmin
  float oo, tmin_u, inv_len;   // O.O; tmin as a distance; 1 / |D|
  int self;                    // position among the register spheres of the sphere the ray leaves (-1: none)
  RT_DEV static TileRay make(const Ray &ray, int self) {
    TileRay r;
    const float a = fmaf(ray.d.x, ray.d.x, fmaf(ray.d.y, ray.d.y, ray.d.z * ray.d.z));
    r.inv_len = __builtin_amdgcn_rsqf(a);
    const float ux = ray.d.x * r.inv_len, uy = ray.d.y * r.inv_len, uz = ray.d.z * r.inv_len;
    r.nux = -ux, r.nuy = -uy, r.nuz = -uz;
    r.ou = fmaf(ray.o.x, ux, fmaf(ray.o.y, uy, ray.o.z * uz));
    r.m2ox = -2.f * ray.o.x, r.m2oy = -2.f * ray.o.y, r.m2oz = -2.f * ray.o.z;
    r.oo = fmaf(ray.o.x, ray.o.x, fmaf(ray.o.y, ray.o.y, ray.o.z * ray.o.z));
    r.tmin_u = ray.tmin * a * r.inv_len;
    r.tmin = ray.tmin;
    r.self = self;
    return r;
  }
  // origin and tmin bit for bit (x -> -2x -> -0.5 * that is exact): the lane does not carry them across the scan
  RT_DEV void restore(Ray &ray) const {
    ray.o = mk3(-0.5f * m2ox, -0.5f * m2oy, -0.5f * m2oz);
    ray.tmin = tmin;
  }
};

// ---------------------------------------------------------------------------------------------
// The same two products on the bf16 matrix cores (round 3; RT_TILE_BF16, the default build).
// v_mfma_f32_32x32x16_bf16 multiplies 16 K slots in 8 passes (32 cycles) where the binary32 instruction takes 16 passes for 2.
// Every binary32 operand is split into three bf16 parts, x = x0 + x1 + x2 EXACTLY (round to nearest: 8 + 8 + 8 significant
// bits), and a term x y of the inner product becomes the six partial products x0 y0 + x0 y1 + x0 y2 + x1 y0 + x1 y1 + x2 y0 —
// each exact in the binary32 accumulator; what is dropped (x1 y2, x2 y1, x2 y2) is below 2^-24 of the term.  Measured on the
// device (scripts/microbench/bf16_probe.hip, 200 k products of filter-like data): |error| <= 2^-22.4 of sum |terms| against
// 2^-22.6 for the binary32 fmaf chain — the same accuracy, which world_filter_ok (rt_api.hip) already prices.
// K slots (32 = two MFMAs per product; the A operand is the SAME for b and c):
//     slot 6 t + j, t = x y z:  A = part {0 0 0 1 1 2}[j] of c_t,   B_b = part {0 1 2 0 1 0}[j] of -u_t,   B_c = the same of -2 o_t
//     slots 18-20:              A = 1,                             B_b = the parts of O.u,               B_c = 0
//     slots 21-23:              A = the parts of k,                B_b = 0,                              B_c = 1
//     slots 24-31:              0
// Lane l supplies slots 8 (l / 32) + 0..7 of a 16-slot chunk for row / column l % 32 as four words of bf16 pairs; the result
// registers are laid out as the binary32 shape's (probed), so everything after the products is shared with it.
// Rate (same probe, 4 waves per SIMD): 163 SIMD-cycles per 32 x 32 tile with the 24 VALU instructions of the filter against
// 352 for the binary32 MFMAs, and 139 for the four bf16 MFMAs alone — unlike the binary32 MFMA, which runs on the VALU's
// fp32 lanes, the bf16 matrix pipe overlaps with the vector instructions of the other waves.
// Cost: 8 sphere registers per tile instead of 3, and a 128-byte ray record in LDS, which limits a CU to 12 waves (3 per SIMD,
// 168 VGPRs each).  Measured (Mrays/s; N=512 1080p AA=4 at W = 1 | N=4096 480x136 AA=4 at W = 4; scripts/ab_tile_bf16.sh):
//     binary32 MFMAs, S = 8, 4 waves per SIMD   10 787 |  1 352
//     bf16, S = 4, 3 waves per SIMD             13 011 |  2 030
//     bf16, S = 8, 3 waves per SIMD (17 spilled) 13 973 |  2 165      <- the build
//     bf16, S = 8 / 16, 2 waves per SIMD        10 994 / 11 381 | 1 772 / 1 871   (the pipes overlap ACROSS waves: fewer waves, less overlap)
// Phase split per wave iteration at N=4096, W=4 (-DRT_PROFILE_PHASES): queue 7.2 k, filter 147 k (308 k with binary32 MFMAs),
// finish + shade 9.7 k cycles; N=512, W=1: 4.1 k / 19.6 k / 9.1 k.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = p[0] + p[1] + p[2] exactly; parts as bf16 bit patterns
RT_DEV void bf16_split3(float x, uint32_t p[3]) {
  const auto rn = [](float v) { // round to nearest even; finite values
    const uint32_t u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  };
  p[0] = rn(x);
  float r = x - __uint_as_float(p[0] << 16);
  p[1] = rn(r);
  r -= __uint_as_float(p[1] << 16);
  p[2] = rn(r);
}

// LDS image of one queued ray, eight quads: 0-2 the 24 slots of B_b, 3-5 of B_c (lanes 0-31 read quads 0 | 3 for the first chunk
// and 2 | 5 for the second, lanes 32-63 quads 1 | 4 and — their slots 24-31 being zero on the sphere side — anything finite),
// 6 and 7 what the lane needs around the products.
struct alignas(16) TileRayBf {
  uint32_t b[12], c[12];
  float oo, tmin_u, inv_len; // O.O; tmin as a distance; 1 / |D|
  int self;                  // position among the register spheres of the sphere the ray leaves (-1: none)
  float ox, oy, oz, tmin;    // the lane does not carry them across the scan
  RT_DEV static void pack(const uint32_t (&x)[3], const uint32_t (&y)[3], const uint32_t (&z)[3], uint32_t *w) {
    w[0] = x[0] | (x[1] << 16), w[1] = x[2] | (x[0] << 16), w[2] = x[1] | (x[0] << 16);
    w[3] = y[0] | (y[1] << 16), w[4] = y[2] | (y[0] << 16), w[5] = y[1] | (y[0] << 16);
    w[6] = z[0] | (z[1] << 16), w[7] = z[2] | (z[0] << 16), w[8] = z[1] | (z[0] << 16);
  }
  RT_DEV static TileRayBf make(const Ray &ray, int self) {
    TileRayBf r;
    const float a = fmaf(ray.d.x, ray.d.x, fmaf(ray.d.y, ray.d.y, ray.d.z * ray.d.z));
    r.inv_len = __builtin_amdgcn_rsqf(a);
    const float ux = ray.d.x * r.inv_len, uy = ray.d.y * r.inv_len, uz = ray.d.z * r.inv_len;
    const float ou = fmaf(ray.o.x, ux, fmaf(ray.o.y, uy, ray.o.z * uz));
    uint32_t px[3], py[3], pz[3], pk[3];
    bf16_split3(-ux, px), bf16_split3(-uy, py), bf16_split3(-uz, pz), bf16_split3(ou, pk);
    pack(px, py, pz, r.b);
    r.b[9] = pk[0] | (pk[1] << 16), r.b[10] = pk[2], r.b[11] = 0u;
    bf16_split3(-2.f * ray.o.x, px), bf16_split3(-2.f * ray.o.y, py), bf16_split3(-2.f * ray.o.z, pz);
    pack(px, py, pz, r.c);
    r.c[9] = 0u, r.c[10] = 0x3f80u << 16, r.c[11] = 0x3f80u | (0x3f80u << 16);
    r.oo = fmaf(ray.o.x, ray.o.x, fmaf(ray.o.y, ray.o.y, ray.o.z * ray.o.z));
    r.tmin_u = ray.tmin * a * r.inv_len;
    r.self = self;
    r.ox = ray.o.x, r.oy = ray.o.y, r.oz = ray.o.z, r.tmin = ray.tmin;
    return r;
  }
  // In LDS the eight quads of record n are stored swizzled: logical quad q sits at physical quad q ^ ((n >> 1) & 7).  With plain
  // 128-byte records the 16 lanes one ds_read_b128 cycle serves would find the same quad of consecutive records in only two bank
  // groups (record stride = half the 64 banks): measured 0.85 conflict cycles per LDS cycle; swizzled they cover all 64 banks.
  RT_DEV static int quad_index(int n, int q) { return 8 * n + (q ^ ((n >> 1) & 7)); }
  RT_DEV void store(float4 *quads, int n) const {
    const float4 *src = reinterpret_cast<const float4 *>(this);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      quads[quad_index(n, q)] = src[q];
  }
  RT_DEV static void restore(const float4 *quads, int n, Ray &ray) {
    const float4 v = quads[quad_index(n, 7)];
    ray.o = mk3(v.x, v.y, v.z);
    ray.tmin = v.w;
  }
};
static_assert(sizeof(TileRayBf) == 128, "eight quads");

#ifndef RT_TILE_OCC
#define RT_TILE_OCC (RT_TILE_BF16 ? 3 : 4)
#endif
// CSG = false: for worlds without CSG objects, compiled without the evaluators (N=512: +1.3 %; nothing at W = 4, where the
// scan is 97 % of the time — only the one-wave shape is instantiated that way).
template <int S, int W, bool CSG = true, bool BF = (RT_TILE_BF16 != 0)>
__global__ void __launch_bounds__(64 * W, RT_TILE_OCC) k_render_path_tile(Scene s, RenderParams p) {
  constexpr int T = 64 * W;
  using QRay = std::conditional_t<BF, TileRayBf, TileRay>;
  constexpr int NA = BF ? 8 : 3; // sphere registers per tile
  __shared__ QRay q_ray[T];
  __shared__ unsigned long long q_key[T];
  __shared__ int q_cnt[W], q_live[W]; // per wave: rays queued this iteration, pool not yet dry
  // The pool of rt_path.h, one per WAVE, run as the small-world kernel runs it (rt_wavepath.h: no barriers, prefix sums in
  // DPP, one 64-byte record per node).  The waves of a workgroup share the ray queue and the sphere tiles, nothing else:
  // they meet twice per iteration, around the filter.  Every wave queues into its OWN 64 slots (q_ray / q_key / q_cnt /
  // q_live [wave]) and nobody else ever writes them outside the filter, so a wave that is still reading its results may be
  // overtaken by one already queueing its next rays.  Measured against one pool per workgroup (PathBlock): N=512 (W=1)
  // 9.94 -> 10.58 Grays/s — dealing and shading were 28 % of an iteration.
  __shared__ WaveShared sh_all[W];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  WaveShared &sh = sh_all[wave];
  const int n_tiles = s.n_wide_tiles;
  const int n_pass = (n_tiles + S * W - 1) / (S * W);
  WavePath<true> L;
  L.init(p, sh, blockIdx.x * W + wave);
#ifdef RT_PROFILE_PHASES
  unsigned long long t_queue = 0, t_scan = 0, t_finish = 0, t_shade = 0, n_iter = 0, n_rays_q = 0;
#define RT_TSTAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define RT_TSTAMP(var)
#endif
  for (;;) {
    RT_TSTAMP(ts0);
    // -- every lane gets a ray from its wave's node pool; the live rays of the workgroup are queued (compacted) ----
    // A wave whose pool has run dry keeps coming along (its sphere tiles are part of every ray's scan) until all have.
    const bool wave_live = L.deal(s, p, sh);
    const unsigned long long live = __ballot(L.active);
    if (lane == 0)
      q_cnt[wave] = __popcll(live), q_live[wave] = wave_live ? 1 : 0;
    int my_q = -1;
    if (L.active) {
      my_q = 64 * wave + __popcll(live & ((1ull << lane) - 1ull));
      const int self = L.origin_shape >= 0 ? s.shape_origin[4 * L.origin_shape + 1] : -1;
      if constexpr (BF)
        QRay::make(L.ray, self).store(reinterpret_cast<float4 *>(q_ray), my_q);
      else
        q_ray[my_q] = QRay::make(L.ray, self);
      q_key[my_q] = ~0ull;
    }
    __syncthreads();
    int n_q = 0, any_live = 0; // uniform
#pragma unroll
    for (int w = 0; w < W; ++w) {
      n_q += __builtin_amdgcn_readfirstlane(q_cnt[w]);
      any_live |= __builtin_amdgcn_readfirstlane(q_live[w]);
    }
    if (!any_live)
      break;
    RT_TSTAMP(ts1);

    // -- the filter: every wave takes the queued rays 32 at a time against its sphere tiles -------------
    for (int pass = 0; pass < n_pass; ++pass) {
      // tile s_ of this wave in this pass is sphere tile first + s_ * W + wave (positions ascend with the object index)
      const int first = pass * S * W;
      const int first_w = first + __builtin_amdgcn_readfirstlane(wave);
      const int n_s = min(S, max(0, (n_tiles - first_w + W - 1) / W));
      float a[S][NA];
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) {
        const int tile = min(first + s_ * W + wave, n_tiles - 1); // tiles past n_s are not multiplied
        const float *rec = s.wide_tiles + (size_t)tile * RT_TILE_WORDS;
        if constexpr (BF) {
          const float4 lo = reinterpret_cast<const float4 *>(rec)[lane], hi = reinterpret_cast<const float4 *>(rec)[64 + lane];
          a[s_][0] = lo.x, a[s_][1] = lo.y, a[s_][2] = lo.z, a[s_][3] = lo.w;
          a[s_][4] = hi.x, a[s_][5] = hi.y, a[s_][6] = hi.z, a[s_][7] = hi.w;
        } else {
          a[s_][0] = rec[lane], a[s_][1] = rec[64 + lane], a[s_][2] = rec[128 + lane];
        }
      }
      for (int rt = 0; rt < 2 * W; ++rt) { // the two 32-ray blocks of every wave's slots
        const int n_w = __builtin_amdgcn_readfirstlane(q_cnt[rt >> 1]), r0 = (rt & 1) * 32;
        if (r0 >= n_w)
          continue;
        const int n = 64 * (rt >> 1) + r0 + col;
        const bool valid = r0 + col < n_w;
        const int slot = valid ? n : 64 * (rt >> 1);
        float4 qb0, qb1, qc0, qc1, qc; // B_b and B_c (binary32 build: qb0 alone carries both), and the lane's own quad
        if constexpr (BF) {
          const float4 *quads = reinterpret_cast<const float4 *>(q_ray) + 8 * slot;
          const int sw = (slot >> 1) & 7;
          qb0 = quads[half ^ sw], qb1 = quads[2 ^ sw], qc0 = quads[(3 + half) ^ sw], qc1 = quads[5 ^ sw], qc = quads[6 ^ sw];
        } else {
          const float4 *rec = reinterpret_cast<const float4 *>(&q_ray[slot]);
          qb0 = rec[half], qc = rec[2];
        }
        const float oo = valid ? qc.x : INF_F; // a lane past the queue's end asks d > +inf
        const float tmin_u = qc.y, inv_len = qc.z;
        const int self = __float_as_int(qc.w);
        unsigned long long best = ~0ull;
        // the four (bf16: 4 x K = 16; binary32: 4 x K = 2) MFMAs of tile s_ of this pass against the ray block in qb0 .. qc1
        auto products = [&](int s_, f32x16 &b, f32x16 &c) {
          b = f32x16{0}, c = f32x16{0};
          if constexpr (BF) {
            union Op {
              bf16x8 v;
              float f[4];
            };
            Op a0, a1, b0, b1, c0, c1;
            a0.f[0] = a[s_][0], a0.f[1] = a[s_][1], a0.f[2] = a[s_][2], a0.f[3] = a[s_][3];
            a1.f[0] = a[s_][4], a1.f[1] = a[s_][5], a1.f[2] = a[s_][6], a1.f[3] = a[s_][7];
            b0.f[0] = qb0.x, b0.f[1] = qb0.y, b0.f[2] = qb0.z, b0.f[3] = qb0.w;
            b1.f[0] = qb1.x, b1.f[1] = qb1.y, b1.f[2] = qb1.z, b1.f[3] = qb1.w;
            c0.f[0] = qc0.x, c0.f[1] = qc0.y, c0.f[2] = qc0.z, c0.f[3] = qc0.w;
            c1.f[0] = qc1.x, c1.f[1] = qc1.y, c1.f[2] = qc1.z, c1.f[3] = qc1.w;
            b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, b0.v, b, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, c0.v, c, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1.v, b, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, c1.v, c, 0, 0, 0);
          } else {
            const float bq1 = qb0.x, bq2 = qb0.y, cq1 = qb0.z, cq2 = half ? 1.f : qb0.w;
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_][0], bq1, b, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_][0], cq1, c, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_][1], bq2, b, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_][2], cq2, c, 0, 0, 0);
          }
        };
        // what the vector ALU does with a tile's products: d = b^2 - c, its maximum, and — where some d beats O.O — the candidates
        auto consume = [&](int s_, const f32x16 &b, const f32x16 &c) {
          // (8 v_pk_fma_f32 on a negated c product instead of these 16 v_fma: measured, no gain — 9929 -> 9949 and 1423 -> 1388
          // Mrays/s with the binary32 MFMAs)
          float d[16];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            d[r] = fmaf(b[r], b[r], -c[r]);
          // maxima of five blocks of 3 registers (v_max3_f32 each) and the tile maximum: 8 instructions for the 16 values
          float g[5];
#pragma unroll
          for (int k = 0; k < 5; ++k)
            g[k] = fmaxf(fmaxf(d[3 * k], d[3 * k + 1]), d[3 * k + 2]);
          const float m = fmaxf(fmaxf(fmaxf(g[0], g[1]), g[2]), fmaxf(fmaxf(g[3], g[4]), d[15]));
          if (m > oo) { // this ray's line meets one of the tile's spheres
            // the tile's first position is wave-uniform; the lane's own part (the lower or upper 4 of each 8 rows) is made
            // opaque here so that the 256 positions of a pass are not all computed ahead of the loops and kept in registers
            int h4 = 4 * half;
            asm volatile("" : "+v"(h4));
            const int base = (first_w + s_ * W) * 32 + h4; // position of this lane's row 0
            const int self_rel = self - base;
            auto candidate = [&](int r) {
              if (d[r] > oo) {
                const int row = 8 * (r >> 2) + (r & 3); // a literal
                // front root as a distance (the reference's rule: t1 unless it is below tmin, then t2 if beyond tmin)
                const float sq = __builtin_amdgcn_sqrtf(d[r] - oo);
                const float u1 = -b[r] - sq, u2 = -b[r] + sq;
                const float u = u1 < tmin_u ? u2 : u1;
                if (u > tmin_u && self_rel != row) {
                  const unsigned long long key = ((unsigned long long)__float_as_uint(u * inv_len) << 32) | (unsigned)(base + row);
                  best = min(best, key); // equal t: the lowest position = the lowest object index, the reference's strict `<`
                }
              }
            };
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              if (g[k] > oo) {
                candidate(3 * k);
                candidate(3 * k + 1);
                candidate(3 * k + 2);
              }
            }
            candidate(15);
          }
        };
        // (Issuing the MFMAs of tile s_ + 1 ahead of the vector work on tile s_, into a second pair of accumulators, was built and
        // measured: the compiler interleaves them as hoped, but N=4096 gains 2 % and N=512 loses 10 % to the spills — with three
        // waves per SIMD the two pipes already overlap across waves, and counters show the SIMDs 92 % busy, vector ALU 64 %.)
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
          __builtin_amdgcn_sched_barrier(0); // one tile at a time: interleaving the tiles' accumulators spills them
          if (s_ < n_s) { // uniform over the wave
            f32x16 b, c;
            products(s_, b, c);
            consume(s_, b, c);
          }
        }
        if (best != ~0ull)
          atomicMin(&q_key[n], best);
      }
    }
    __syncthreads();
    RT_TSTAMP(ts2);

    // -- every lane is a ray again: finish the query, shade, produce the next ray -------------------
    // (Phase split per iteration of a wave at N=512, -DRT_PROFILE_PHASES: deal + queue 5.6 k cycles, filter 35.9 k, this
    // finish 10.7 k, shade 5.6 k; at N=4096, W=4: 13 k / 308 k / 14 k / 9 k.  The finish is a chain of dependent trips to
    // memory: position -> ids -> matrices -> shading record.  Fetching the winner's matrices and shading record by position
    // in one trip was built and measured twice — the registers it holds cost more than the trips: 10 765 -> 10 400 Mrays/s —
    // and was not kept.)
    Hit h;
    h.object = h.shape = -1;
    h.t = 0.f;
    h.point = h.normal = h.local = mk3(0.f, 0.f, 0.f);
    if constexpr (BF) // every lane, so that no lane's old values stay live across the scan
      QRay::restore(reinterpret_cast<const float4 *>(q_ray), max(my_q, 0), L.ray);
    else
      q_ray[max(my_q, 0)].restore(L.ray);
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
    RT_TSTAMP(ts3);
    L.shade(s, p, sh, h); // the wave's own pool: no barrier; q_ray / q_key slots are rewritten only after the next deal
#ifdef RT_PROFILE_PHASES
    RT_TSTAMP(ts4);
    t_queue += ts1 - ts0, t_scan += ts2 - ts1, t_finish += ts3 - ts2, t_shade += ts4 - ts3, n_iter += 1, n_rays_q += n_q;
#endif
  }
#ifdef RT_PROFILE_PHASES
  if (threadIdx.x == 0) { // counters 4-8 as the wide kernel reports them (finish + shade together) and the finish share apart
    atomicAdd(p.counters + 4, t_queue);
    atomicAdd(p.counters + 5, t_scan);
    atomicAdd(p.counters + 6, t_finish + t_shade);
    atomicAdd(p.counters + 7, n_iter);
    atomicAdd(p.counters + 8, n_rays_q);
    atomicAdd(p.counters + 9, t_finish);
  }
#endif
  wave_add_tally(p.counters, 0, L.rays);
}

} // namespace rtd
