// rt_tonemap.h — the tone-mapping tail on the framebuffer in HBM: HdrImage::average_luminosity,
// normalize_image, clamp_image and write_ldr_image's per-component transform
// (reference include/colors.hpp:170-173, 87, 469-516; called from src/raytracer.cpp:176-186).
// HBM-bound: 12 bytes read per pixel for the luminosity pass, 12 read + 3 written for the map.
#pragma once
#include "rt_device.h"

namespace rtd {

// Color::luminosity (Shirley & Morley) and one term of average_luminosity's sum — colors.hpp:170-173, 472
RT_DEV float log_luminosity(const float *px, float delta) {
  const float r = px[0], g = px[1], b = px[2];
  const float lum = 0.5f * (fminf(fminf(r, g), b) + fmaxf(fmaxf(r, g), b));
  return log10f(delta + lum);
}

// Exact sum of the terms in 64-bit fixed point (2^-32): independent of the order of the additions.  Four pixels
// (48 contiguous bytes, three 16-byte loads) per lane and step.
__global__ void __launch_bounds__(256) k_log_luminosity_sum(const float *rgb, size_t n_pixels, float delta, unsigned long long *sum) {
  long long acc = 0;
  const size_t n_quads = n_pixels / 4, stride = (size_t)gridDim.x * blockDim.x;
  const float4 *v = reinterpret_cast<const float4 *>(rgb);
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += stride) {
    const float4 a = v[3 * q], b = v[3 * q + 1], c = v[3 * q + 2];
    const float px[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      acc += to_fixed32(log_luminosity(px + 3 * k, delta));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n_pixels & 3)) // the last 0..3 pixels
    acc += to_fixed32(log_luminosity(rgb + 3 * (4 * n_quads + threadIdx.x), delta));
  for (int off = 32; off > 0; off >>= 1)
    acc += __shfl_down(acc, off, 64);
  // one atomic per workgroup: thousands of same-address atomics would serialise in L2 and dominate the pass
  __shared__ long long part[4];
  if ((threadIdx.x & 63) == 0)
    part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long total = part[0] + part[1] + part[2] + part[3];
    if (total != 0)
      atomicAdd(sum, (unsigned long long)total);
  }
}

// The terms, for the serial sum below.
__global__ void __launch_bounds__(256) k_log_luminosity_terms(const float *rgb, size_t n_pixels, float delta, float *terms) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pixels)
    terms[i] = log_luminosity(rgb + 3 * i, delta);
}

// HdrImage::average_luminosity's `cumsum += term` in pixel order, in binary32 — colors.hpp:470-473.  One wave:
// 64 terms are fetched with one coalesced load, then added one after the other (every lane carries the same
// running sum).  ~10 cycles per pixel; the roundings are those of the reference's loop.
__global__ void __launch_bounds__(64) k_serial_sum(const float *terms, size_t n, float *out) {
  const int lane = threadIdx.x;
  float sum = 0.f;
  size_t base = 0;
  float next = (size_t)lane < n ? terms[lane] : 0.f;
  for (; base + 64 <= n; base += 64) {
    const float cur = next;
    const size_t j = base + 64 + lane;
    next = j < n ? terms[j] : 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k)
      sum = sum + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur), k));
  }
  for (int k = 0; base + k < n; ++k)
    sum = sum + __shfl(next, k, 64);
  if (lane == 0)
    *out = sum;
}

// normalize_image + clamp_image + write_ldr_image's transform on one colour component — colors.hpp:485, 87, 514-515
RT_DEV float normalize_clamp(float x, float scale) {
  const float v = x * scale;
  return v / (1.f + v);
}
// pow through the hardware's log2 / exp2 (relative error < 1e-6: 3e-4 of an 8-bit step); only a value that lands
// within 2e-3 of a rounding boundary is recomputed with the correctly rounded-ish library powf, so the byte is the
// one the library function gives.
RT_DEV uint32_t to_ldr(float c, float inv_gamma) {
  float g = c;
  if (inv_gamma != 1.f) {
    g = (c > 0.f) ? __builtin_amdgcn_exp2f(inv_gamma * __builtin_amdgcn_logf(c)) : powf(c, inv_gamma);
    const float y = g * 255.f;
    if (!(fabsf((y - floorf(y)) - 0.5f) > 2e-3f))
      g = powf(c, inv_gamma);
  }
  return (uint32_t)fminf(fmaxf(roundf(g * 255.f), 0.f), 255.f);
}

// The map works on the flat array of 3 * n_pixels components, four per lane: one 16-byte load, one 4-byte store.
__global__ void __launch_bounds__(256) k_tonemap(const float *rgb, size_t n_comp, float scale, float inv_gamma, uint8_t *ldr, float *hdr) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = 4 * q;
  if (i + 4 <= n_comp) {
    const float4 v = reinterpret_cast<const float4 *>(rgb)[q];
    const float c0 = normalize_clamp(v.x, scale), c1 = normalize_clamp(v.y, scale), c2 = normalize_clamp(v.z, scale),
                c3 = normalize_clamp(v.w, scale);
    if (hdr)
      reinterpret_cast<float4 *>(hdr)[q] = make_float4(c0, c1, c2, c3);
    if (ldr)
      reinterpret_cast<uint32_t *>(ldr)[q] =
          to_ldr(c0, inv_gamma) | (to_ldr(c1, inv_gamma) << 8) | (to_ldr(c2, inv_gamma) << 16) | (to_ldr(c3, inv_gamma) << 24);
  } else {
    for (size_t k = i; k < n_comp; ++k) {
      const float c = normalize_clamp(rgb[k], scale);
      if (hdr)
        hdr[k] = c;
      if (ldr)
        ldr[k] = (uint8_t)to_ldr(c, inv_gamma);
    }
  }
}

} // namespace rtd
