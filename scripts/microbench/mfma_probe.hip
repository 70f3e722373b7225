// mfma_probe.hip — v_mfma_f32_32x32x2_f32 on gfx950: (1) where D[i][j] lands (lane, register), (2) the rate of the
// ray-tile x sphere-tile discriminant block: 4 MFMAs (b and c, K = 4 each) + 16 fma + 8 max3 per 32 rays x 32 spheres.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_layout(float *out) { // A[i][k] = 1 + i + 100 k, B[k][j] = (k == 0) ? 1 : 1000 * (j + 1): D = A[i][0] + A[i][1] * 1000 (j + 1)
  const int lane = threadIdx.x, i = lane & 31, k = lane >> 5;
  const float a = 1.f + i + 100.f * k;
  const float b = k == 0 ? 1.f : 1000.f * (i + 1);
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r)
    out[lane * 16 + r] = c[r];
}

// MODE 0: the 4 MFMAs and one use of their results; 1: + 16 fma; 2: + 16 fma + 8 max3 (the filter block); 3: + 48 more fma;
// 4: no MFMA, the 24 VALU instructions of mode 2 alone (on loop-carried registers)
template <int S, int MODE> __global__ void __launch_bounds__(64, 4) k_rate(const float *sph, int iters, float *out) {
  const int lane = threadIdx.x;
  float a1[S], a2[S], a3[S];
  for (int s = 0; s < S; ++s)
    a1[s] = sph[(3 * s) * 64 + lane], a2[s] = sph[(3 * s + 1) * 64 + lane], a3[s] = sph[(3 * s + 2) * 64 + lane];
  float b1 = 0.1f * lane, b2 = 0.3f, c1 = -0.2f, c2 = 0.5f, oo = 1e30f;
  float acc = 0.f;
  f16v vb = {0}, vc = {0};
  for (int r = 0; r < 16; ++r)
    vb[r] = sph[r] + lane, vc[r] = sph[16 + r] - lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      f16v b = vb, c = vc;
      if (MODE != 4) {
        f16v z = {0};
        b = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1, z, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], c1, z, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], b2, b, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[s], c2, c, 0, 0, 0);
      }
      float m;
      if (MODE == 0) {
        m = b[0] + c[0];
      } else if (MODE == 6) { // 24 integer instructions on the MFMA results, no fp32 VALU work
        int x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          x[r] = __float_as_int(b[r]) ^ (__float_as_int(c[r]) >> 1);
        int mi = max(max(x[0], x[1]), x[2]);
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2)
          mi = max(max(mi, x[r]), x[r + 1]);
        mi = max(mi, x[15]);
        m = __int_as_float(mi);
      } else {
        float d[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          d[r] = fmaf(b[r], b[r], -c[r]);
        if (MODE == 3) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              d[r] = fmaf(d[r], b[r], c[(r + j) & 15]);
        }
        if (MODE == 5) { // integer maxima of the same bits
          int mi = max(max(__float_as_int(d[0]), __float_as_int(d[1])), __float_as_int(d[2]));
#pragma unroll
          for (int r = 3; r + 1 < 16; r += 2)
            mi = max(max(mi, __float_as_int(d[r])), __float_as_int(d[r + 1]));
          mi = max(mi, __float_as_int(d[15]));
          m = __int_as_float(mi);
        } else if (MODE == 1) {
          m = d[0];
#pragma unroll
          for (int r = 1; r < 16; ++r)
            asm volatile("" ::"v"(d[r]));
        } else {
          m = __builtin_fmaxf(__builtin_fmaxf(d[0], d[1]), d[2]);
#pragma unroll
          for (int r = 3; r + 1 < 16; r += 2)
            m = __builtin_fmaxf(__builtin_fmaxf(m, d[r]), d[r + 1]);
          m = fmaxf(m, d[15]);
        }
        if (MODE == 4)
          vb[s & 15] = m * 1e-30f;
      }
      if (m > oo)
        acc += m;
    }
    b1 += 1e-7f;
  }
  out[blockIdx.x * 64 + lane] = acc;
}

template <int S, int MODE> void run(const float *sph, float *out, int waves_per_simd) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * 4 * waves_per_simd, iters = 400000 / S;
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
  printf("mode %d S=%d waves/SIMD=%d: %.1f ms  %.3g tests/s  %.0f SIMD-cycles per 1024-test tile at 2.4 GHz (4 MFMAs alone: 256)\n", MODE, S, waves_per_simd, ms,
         tiles * 1024 / (ms * 1e-3), cyc);
}

int main() {
  float *out, *sph;
  hipMalloc(&out, 256 * 64 * 64 * sizeof(float));
  hipMalloc(&sph, 3 * 16 * 64 * sizeof(float));
  hipMemset(sph, 0, 3 * 16 * 64 * sizeof(float));
  k_layout<<<1, 64>>>(out);
  static float h[64 * 16];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int r = 0; r < 16; ++r) {
      const int j = lane & 31, i = 8 * (r / 4) + 4 * (lane >> 5) + (r & 3); // the expected layout
      const float want = (1.f + i) + (101.f + i) * 1000.f * (j + 1);
      if (h[lane * 16 + r] != want) {
        if (bad < 8)
          printf("lane %d reg %d: got %g want %g\n", lane, r, h[lane * 16 + r], want);
        ++bad;
      }
    }
  printf("layout D[i = 8 (r / 4) + 4 (lane / 32) + r %% 4][j = lane %% 32]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  run<8, 0>(sph, out, 1), run<8, 0>(sph, out, 2), run<8, 0>(sph, out, 4);
  run<8, 1>(sph, out, 1), run<8, 1>(sph, out, 2), run<8, 1>(sph, out, 4);
  run<8, 2>(sph, out, 1), run<8, 2>(sph, out, 2), run<8, 2>(sph, out, 4);
  run<8, 3>(sph, out, 1), run<8, 3>(sph, out, 2), run<8, 3>(sph, out, 4);
  run<8, 4>(sph, out, 1), run<8, 4>(sph, out, 2), run<8, 4>(sph, out, 4);
  run<8, 5>(sph, out, 2), run<8, 5>(sph, out, 4);
  run<8, 6>(sph, out, 2), run<8, 6>(sph, out, 4);
  return 0;
}
