// scan_bench.hip — the sphere-scan phase of k_render_path_wide alone (rt_wide.h: wide_test_ray), on synthetic
// rays, to measure how close the inner loop gets to the VALU issue rate at the kernel's own occupancy.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I../../raytracer_amd/csrc scan_bench.hip -o scan_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rt_wide.h"

using namespace rtd;

RT_DEV float sgpr(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
RT_DEV WideRay to_sgprs(const WideRay &r) {
  WideRay q;
  q.ox = sgpr(r.ox), q.oy = sgpr(r.oy), q.oz = sgpr(r.oz), q.tmin = sgpr(r.tmin);
  q.dx = sgpr(r.dx), q.dy = sgpr(r.dy), q.dz = sgpr(r.dz), q.tmax = sgpr(r.tmax);
  q.a = sgpr(r.a), q.inv_a = sgpr(r.inv_a), q.pad0 = q.pad1 = 0.f;
  return q;
}
template <int K, int W, bool SG, bool NOLDS = false> __global__ void __launch_bounds__(64 * W, 4) k_scan(const float4 *spheres, int n_rows, int iters, float spread,
                                                                            unsigned long long *out) {
  constexpr int T = 64 * W;
  __shared__ WideRay q_ray[T];
  __shared__ unsigned long long q_key[T];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // a ray per lane: origin in the cloud's bounding box, direction on a sphere (cheap hash)
  unsigned h = (blockIdx.x * T + threadIdx.x) * 2654435761u + 12345u;
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (float)(h >> 8) * (1.f / 16777216.f); };
  WideRay r;
  r.ox = -2.f + 10.f * rnd(), r.oy = -6.f + 12.f * rnd(), r.oz = -2.f + 4.f * rnd();
  r.dx = rnd() - 0.5f, r.dy = rnd() - 0.5f, r.dz = (rnd() - 0.5f) * spread;
  r.tmin = 1e-3f, r.tmax = INF_F;
  r.a = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
  r.inv_a = 1.f / r.a;
  r.pad0 = r.pad1 = 0.f;
  q_ray[threadIdx.x] = r;
  q_key[threadIdx.x] = ~0ull;
  WideSlots<true, K> slots;
  Scene s{};
  s.sphere_iso = reinterpret_cast<const float *>(spheres);
#pragma unroll
  for (int k = 0; k < K; ++k)
    slots.load(s, k, (k * W + wave) * 64 + lane, k * W + wave < n_rows);
  __syncthreads();
  const int n_q = T;
  if (NOLDS) { // the same instruction mix with no LDS traffic: two rays held in registers, nudged every round
    WideRay ra = q_ray[lane & 1], rb = q_ray[2 + (lane & 1)];
    for (int it = 0; it < iters; ++it) {
      for (int qi = 0; qi + 1 < n_q; qi += 2) {
        wide_test_ray<true, K, W>(slots, ra, q_key + (qi & 63), 0, wave, lane);
        ra.ox += 1e-7f;
        wide_test_ray<true, K, W>(slots, rb, q_key + (qi & 63) + 1, 0, wave, lane);
        rb.ox -= 1e-7f;
      }
    }
    if (q_key[threadIdx.x] != ~0ull)
      atomicAdd(out, 1ull);
    return;
  }
  for (int it = 0; it < iters; ++it) {
    WideRay ra = q_ray[0];
    int qi = 0;
    for (; qi + 1 < n_q; qi += 2) {
      const WideRay rb = q_ray[qi + 1];
      wide_test_ray<true, K, W>(slots, SG ? to_sgprs(ra) : ra, q_key + qi, 0, wave, lane);
      ra = q_ray[min(qi + 2, n_q - 1)];
      wide_test_ray<true, K, W>(slots, SG ? to_sgprs(rb) : rb, q_key + qi + 1, 0, wave, lane);
    }
    __syncthreads();
  }
  if (q_key[threadIdx.x] != ~0ull)
    atomicAdd(out, 1ull);
}

template <int K, int W, bool SG = false, bool NOLDS = false> void run(const float4 *d_sph, int n_rows, unsigned long long *d_out, float spread, const char *label) {
  int per_cu = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan<K, W, SG, NOLDS>, 64 * W, 0);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * per_cu, iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k_scan<K, W, SG, NOLDS><<<grid, 64 * W>>>(d_sph, n_rows, 10, spread, d_out);
  hipMemset(d_out, 0, 8);
  hipEventRecord(e0);
  k_scan<K, W, SG, NOLDS><<<grid, 64 * W>>>(d_sph, n_rows, iters, spread, d_out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hits;
  hipMemcpy(&hits, d_out, 8, hipMemcpyDeviceToHost);
  const double ray_wave = (double)grid * W * (64.0 * W) * iters;          // (ray, wave) pairs
  const double tests = ray_wave * 64.0 * K;
  const double valu = ray_wave * (11.0 * K + 4.0);                          // wave instructions of the discriminant block
  const double simd_cycles = 4.0 * prop.multiProcessorCount * (ms * 1e-3) * (prop.clockRate * 1e3);
  printf("%s K=%d W=%d blocks/CU=%d: %.2f ms, %.3g tests/s, %.1f TFLOP/s executed (17/test), VALU issue %.1f%% of 1 instr / 2 cycles (clock %.2f GHz), hit lanes %llu\n",
         label, K, W, per_cu, ms, tests / (ms * 1e-3), tests * 17 / (ms * 1e-3) / 1e12, 100.0 * valu * 2.0 / simd_cycles, prop.clockRate * 1e-6, hits);
}

int main() {
  const int n = 64 * 64;
  std::vector<float4> sph(n);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX; };
  for (auto &v : sph) {
    float r = 0.05f + 0.1f * rnd();
    v = make_float4(-2.f + 10.f * rnd(), -6.f + 12.f * rnd(), -2.f + 4.f * rnd(), r * r);
  }
  float4 *d_sph;
  unsigned long long *d_out;
  hipMalloc(&d_sph, n * sizeof(float4));
  hipMalloc(&d_out, 8);
  hipMemcpy(d_sph, sph.data(), n * sizeof(float4), hipMemcpyHostToDevice);
  run<8, 1>(d_sph, 8, d_out, 1.f, "cloud rays");
  run<8, 8>(d_sph, 64, d_out, 1.f, "cloud rays");
  run<4, 1>(d_sph, 4, d_out, 1.f, "cloud rays");
  run<8, 1, false, true>(d_sph, 8, d_out, 1.f, "cloud rays, rays in registers (no LDS)");
  run<8, 1, false, true>(d_sph, 8, d_out, 1e4f, "missing rays, rays in registers (no LDS)");
  run<8, 1>(d_sph, 8, d_out, 1e4f, "rays that miss (vertical)");
  run<8, 8>(d_sph, 64, d_out, 1e4f, "rays that miss (vertical)");
  return 0;
}
