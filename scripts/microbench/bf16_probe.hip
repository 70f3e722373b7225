// bf16_probe.hip — could v_mfma_f32_32x32x16_bf16 carry the sphere filter of rt_tile.h?  
//  (1) operand / result layout of the instruction on gfx950,
//  (2) accuracy of a binary32 inner product rebuilt from 3-way (and 2-way) bf16 splits of both operands,
//  (3) the rate of the filter block with the two products (b and c, K = 4 logical terms each) on bf16 MFMAs:
//      3-way split = 6 partial products per term = 24 K slots per product = 2 MFMAs (K = 16) per product, 4 per tile;
//      2-way split = 3 partial products per term = 12 K slots           = 1 MFMA per product, 2 per tile;
//      each followed by the 16 fma + 8 max3 of the filter, against the 4 binary32 MFMAs of today.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

__device__ __host__ inline uint16_t bf16_rn(float x) { // round to nearest even, finite inputs
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __host__ inline float bf16_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
__device__ __host__ inline void split3(float x, uint16_t p[3]) {
  p[0] = bf16_rn(x);
  float r = x - bf16_f(p[0]);
  p[1] = bf16_rn(r);
  r -= bf16_f(p[1]);
  p[2] = bf16_rn(r);
}

// A[i][k] = 1 + i + 64 k (exact in bf16 for these sizes?  1 + 31 + 64 * 15 = 992: needs 10 bits -> use small values)
// A[i][k] = (i == probe_i) ? 1 : 0 at k == probe_k ... simpler: A[i][k] = i + 1 if k == kk else 0, B[k][j] = j + 1 if k == kk: D = (i+1)(j+1) for every kk
__global__ void k_layout(float *out, int kk) {
  const int lane = threadIdx.x, i = lane & 31, kb = 8 * (lane >> 5);
  union { bf8v v; uint16_t h[8]; } a, b;
  for (int e = 0; e < 8; ++e) {
    a.h[e] = bf16_rn(kb + e == kk ? (float)(i + 1) : 0.f);
    b.h[e] = bf16_rn(kb + e == kk ? (float)(i + 1) : 0.f);
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r)
    out[lane * 16 + r] = c[r];
}

// accuracy: D[i][j] = sum over 4 logical terms x[i][t] * y[t][j], each rebuilt from split parts in K slots
template <int WAYS> __global__ void k_accuracy(const float *x, const float *y, float *out) { // x[32][4], y[4][32]
  const int lane = threadIdx.x, i = lane & 31, half = lane >> 5;
  // slot s of a term: 3-way (a0b0 a0b1 a0b2 a1b0 a1b1 a2b0), 2-way (a0b0 a0b1 a1b0)
  const int ia3[6] = {0, 0, 0, 1, 1, 2}, ib3[6] = {0, 1, 2, 0, 1, 0}, ia2[3] = {0, 0, 1}, ib2[3] = {0, 1, 0};
  constexpr int PER = WAYS == 3 ? 6 : 3, CHUNKS = WAYS == 3 ? 2 : 1;
  f16v c = {0};
  for (int ch = 0; ch < CHUNKS; ++ch) {
    union { bf8v v; uint16_t h[8]; } a, b;
    for (int e = 0; e < 8; ++e) {
      const int slot = 16 * ch + 8 * half + e, t = slot / PER, s = slot % PER;
      uint16_t pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
      if (t < 4) {
        split3(x[i * 4 + t], pa);
        split3(y[t * 32 + i], pb);
      }
      a.h[e] = t < 4 ? pa[WAYS == 3 ? ia3[s] : ia2[s]] : 0;
      b.h[e] = t < 4 ? pb[WAYS == 3 ? ib3[s] : ib2[s]] : 0;
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r)
    out[lane * 16 + r] = c[r];
}

// MODE 0: binary32 MFMAs (4 per tile) + filter VALU; 1: bf16 3-way (4 MFMAs of K = 16) + VALU; 2: bf16 2-way (2 MFMAs) + VALU;
// 3: bf16 3-way MFMAs alone; 4: bf16 2-way MFMAs alone; 5: one combined operand: A shared by b and c (same 8 registers), 3-way
template <int S, int MODE> __global__ void __launch_bounds__(64, 4) k_rate(const float *sph, int iters, float *out) {
  const int lane = threadIdx.x;
  constexpr int NA = MODE == 0 ? 3 : (MODE == 2 || MODE == 4) ? 4 : 8; // A registers per tile
  float a[S][NA];
  for (int s = 0; s < S; ++s)
    for (int q = 0; q < NA; ++q)
      a[s][q] = sph[(NA * s + q) * 64 + lane];
  float bq[8], cq[8];
  for (int q = 0; q < 8; ++q)
    bq[q] = sph[q * 64 + lane] + 0.25f, cq[q] = sph[(q + 8) * 64 + lane] - 0.5f;
  float oo = 1e30f, acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      f16v b = {0}, c = {0};
      if (MODE == 0) {
        b = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][0], bq[0], b, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][0], cq[0], c, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][1], bq[1], b, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][2 % NA], cq[1], c, 0, 0, 0);
      } else {
        union U { bf8v v; float f[4]; };
        U a0, a1, b0, b1, c0, c1;
        for (int q = 0; q < 4; ++q) {
          a0.f[q] = a[s][q], a1.f[q] = a[s][(4 + q) % NA];
          b0.f[q] = bq[q], b1.f[q] = bq[4 + q], c0.f[q] = cq[q], c1.f[q] = cq[4 + q];
        }
        b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, b0.v, b, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, c0.v, c, 0, 0, 0);
        if (MODE == 1 || MODE == 3 || MODE == 5) {
          b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1.v, b, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, c1.v, c, 0, 0, 0);
        }
      }
      float m;
      if (MODE == 3 || MODE == 4) {
        m = b[0] + c[0];
      } else {
        float d[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          d[r] = fmaf(b[r], b[r], -c[r]);
        float g[5];
#pragma unroll
        for (int k = 0; k < 5; ++k)
          g[k] = fmaxf(fmaxf(d[3 * k], d[3 * k + 1]), d[3 * k + 2]);
        m = fmaxf(fmaxf(fmaxf(g[0], g[1]), g[2]), fmaxf(fmaxf(g[3], g[4]), d[15]));
      }
      if (m > oo)
        acc += m;
    }
    bq[0] += 1e-7f;
  }
  out[blockIdx.x * 64 + lane] = acc;
}

template <int S, int MODE> void run(const float *sph, float *out, int waves_per_simd, const char *what) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * 4 * waves_per_simd, iters = 200000 / S;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k_rate<S, MODE><<<grid, 64>>>(sph, iters / 4, out);
  hipEventRecord(e0);
  k_rate<S, MODE><<<grid, 64>>>(sph, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double tiles = (double)grid * S * iters;
  const double cyc = ms * 1e-3 * 2.4e9 * prop.multiProcessorCount * 4 / tiles;
  printf("%-44s S=%d waves/SIMD=%d: %7.1f ms  %.0f SIMD-cycles per 1024-test tile at 2.4 GHz\n", what, S, waves_per_simd, ms, cyc);
  fflush(stdout);
}

int main() {
  float *out, *sph;
  hipMalloc(&out, 256 * 64 * 64 * sizeof(float));
  hipMalloc(&sph, 8 * 16 * 64 * sizeof(float));
  hipMemset(sph, 0, 8 * 16 * 64 * sizeof(float));
  static float h[64 * 16];
  // (1) layout: for every k the product must be (i + 1)(j + 1) with D[i = 8 (r / 4) + 4 (lane / 32) + r % 4][j = lane % 32],
  //     which also shows that lane l supplies A[l % 32][8 (l / 32) + e] and B[8 (l / 32) + e][l % 32]
  int bad = 0;
  for (int kk = 0; kk < 16; ++kk) {
    k_layout<<<1, 64>>>(out, kk);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 16; ++r) {
        const int j = lane & 31, i = 8 * (r / 4) + 4 * (lane >> 5) + (r & 3);
        if (h[lane * 16 + r] != (float)((i + 1) * (j + 1)))
          ++bad;
      }
  }
  printf("layout: lane l holds A[l %% 32][8 (l / 32) + e], B[8 (l / 32) + e][l %% 32], e = 0..7; D as the binary32 32x32 shape: %s (%d mismatches)\n",
         bad ? "NO" : "yes", bad);
  // (2) accuracy on filter-like data: x = (cx, cy, cz, 1) with |c| <= 10, y = (-ux, -uy, -uz, O.u)
  float hx[32 * 4], hy[4 * 32], *dx, *dy;
  hipMalloc(&dx, sizeof hx), hipMalloc(&dy, sizeof hy);
  double worst3 = 0, worst2 = 0, worst32 = 0;
  srand(7);
  auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
  for (int trial = 0; trial < 200; ++trial) {
    for (int i = 0; i < 32; ++i) {
      hx[i * 4 + 0] = 20.f * rnd() - 10.f, hx[i * 4 + 1] = 20.f * rnd() - 10.f, hx[i * 4 + 2] = 8.f * rnd() - 4.f, hx[i * 4 + 3] = 1.f;
      float u[3] = {rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f};
      const float n = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      hy[0 * 32 + i] = -u[0] / n, hy[1 * 32 + i] = -u[1] / n, hy[2 * 32 + i] = -u[2] / n, hy[3 * 32 + i] = 12.f * rnd() - 6.f;
    }
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice), hipMemcpy(dy, hy, sizeof hy, hipMemcpyHostToDevice);
    for (int ways = 2; ways <= 3; ++ways) {
      if (ways == 3)
        k_accuracy<3><<<1, 64>>>(dx, dy, out);
      else
        k_accuracy<2><<<1, 64>>>(dx, dy, out);
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
          const int j = lane & 31, i = 8 * (r / 4) + 4 * (lane >> 5) + (r & 3);
          double exact = 0, mag = 0;
          float chain = 0.f;
          for (int t = 0; t < 4; ++t) {
            exact += (double)hx[i * 4 + t] * (double)hy[t * 32 + j], mag += fabs((double)hx[i * 4 + t] * (double)hy[t * 32 + j]);
            chain = fmaf(hx[i * 4 + t], hy[t * 32 + j], chain);
          }
          const double e = fabs((double)h[lane * 16 + r] - exact) / mag;
          (ways == 3 ? worst3 : worst2) = fmax(ways == 3 ? worst3 : worst2, e);
          worst32 = fmax(worst32, fabs((double)chain - exact) / mag);
        }
    }
  }
  printf("accuracy, |error| / sum |terms| over 200 x 1024 products: 3-way split %.3g (2^%.1f), 2-way %.3g (2^%.1f), binary32 fma chain %.3g (2^%.1f)\n", worst3,
         log2(worst3), worst2, log2(worst2), worst32, log2(worst32));
  // (3) rates
  for (int w = 1; w <= 4; w *= 2) {
    run<8, 0>(sph, out, w, "binary32: 4 x mfma_32x32x2_f32 + 24 VALU");
    run<4, 1>(sph, out, w, "bf16 3-way: 4 x mfma_32x32x16_bf16 + 24 VALU");
    run<4, 3>(sph, out, w, "bf16 3-way: the 4 MFMAs alone");
    run<8, 2>(sph, out, w, "bf16 2-way: 2 x mfma_32x32x16_bf16 + 24 VALU");
    run<8, 4>(sph, out, w, "bf16 2-way: the 2 MFMAs alone");
  }
  return 0;
}
