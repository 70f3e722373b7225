// pk_bench.hip — does packed fp32 (v_pk_fma_f32 ...) buy anything for the sphere discriminant block on gfx950?
// Same work as scan_bench's "rays in registers" case, with the K spheres of a lane processed in pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int K, bool PK> __global__ void __launch_bounds__(64, 4) k(const float4 *sph, int iters, float *out) {
  const int lane = threadIdx.x;
  float ox = 0.1f * lane, oy = 0.3f, oz = -0.2f, dx = 0.5f, dy = 0.4f, dz = 0.3f, a = dx * dx + dy * dy + dz * dz;
  float acc = 0.f;
  if (PK) {
    v2f cx[K / 2], cy[K / 2], cz[K / 2], r2[K / 2];
    for (int k = 0; k < K / 2; ++k) {
      float4 s0 = sph[(2 * k) * 64 + lane], s1 = sph[(2 * k + 1) * 64 + lane];
      cx[k] = v2f{s0.x, s1.x}, cy[k] = v2f{s0.y, s1.y}, cz[k] = v2f{s0.z, s1.z}, r2[k] = v2f{s0.w, s1.w};
    }
    for (int it = 0; it < iters; ++it) {
      v2f m = v2f{-1e30f, -1e30f};
#pragma unroll
      for (int k = 0; k < K / 2; ++k) {
        v2f ocx = v2f{ox, ox} - cx[k], ocy = v2f{oy, oy} - cy[k], ocz = v2f{oz, oz} - cz[k];
        v2f b = __builtin_elementwise_fma(ocx, v2f{dx, dx}, __builtin_elementwise_fma(ocy, v2f{dy, dy}, ocz * v2f{dz, dz}));
        v2f c = __builtin_elementwise_fma(ocx, ocx, __builtin_elementwise_fma(ocy, ocy, __builtin_elementwise_fma(ocz, ocz, -r2[k])));
        v2f d = __builtin_elementwise_fma(b, b, -(v2f{a, a} * c));
        m = __builtin_elementwise_max(m, d);
      }
      acc += fmaxf(m.x, m.y);
      ox += 1e-7f;
    }
  } else {
    float cx[K], cy[K], cz[K], r2[K];
    for (int k = 0; k < K; ++k) {
      float4 s0 = sph[k * 64 + lane];
      cx[k] = s0.x, cy[k] = s0.y, cz[k] = s0.z, r2[k] = s0.w;
    }
    for (int it = 0; it < iters; ++it) {
      float m = -1e30f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float ocx = ox - cx[k], ocy = oy - cy[k], ocz = oz - cz[k];
        float b = fmaf(ocx, dx, fmaf(ocy, dy, ocz * dz));
        float c = fmaf(ocx, ocx, fmaf(ocy, ocy, fmaf(ocz, ocz, -r2[k])));
        float d = fmaf(b, b, -(a * c));
        m = fmaxf(m, d);
      }
      acc += m;
      ox += 1e-7f;
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}

template <int K, bool PK> void run(const float4 *sph, float *out) {
  int per_cu = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k<K, PK>, 64, 0);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * per_cu, iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<K, PK><<<grid, 64>>>(sph, 100, out);
  hipEventRecord(e0);
  k<K, PK><<<grid, 64>>>(sph, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double tests = (double)grid * 64.0 * K * iters;
  printf("K=%d %s waves/CU=%d: %.2f ms  %.3g tests/s  %.1f TFLOP/s (17 flops/test)\n", K, PK ? "packed" : "scalar", per_cu, ms, tests / (ms * 1e-3),
         tests * 17 / (ms * 1e-3) / 1e12);
}

int main() {
  float4 *sph;
  float *out;
  hipMalloc(&sph, 64 * 64 * sizeof(float4));
  hipMemset(sph, 0, 64 * 64 * sizeof(float4));
  hipMalloc(&out, 256 * 64 * 64 * sizeof(float));
  run<8, false>(sph, out);
  run<8, true>(sph, out);
  run<16, false>(sph, out);
  run<16, true>(sph, out);
  return 0;
}
