// rt_tonemap_api.hip — the tone-mapping entry points (kernels in rt_tonemap.h).
#include "rt_internal.h"
#include "rt_tonemap.h"

using namespace rt_internal;

// ---------------------------------------------------------------------------------------------
// Tone-mapping tail (rt_tonemap.h)
// ---------------------------------------------------------------------------------------------
namespace {

// cumsum of HdrImage::average_luminosity (colors.hpp:470-473), then the reference's own last line on the host
rt_status average_luminosity(int device, const float *rgb, size_t n_pixels, float delta, uint32_t flags, hipStream_t stream, float *avg) {
  if (!rgb || !avg || n_pixels == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_average_luminosity: NULL argument or empty image");
  if (flags & ~RT_TONEMAP_SERIAL_SUM)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: unknown flag");
  if (reinterpret_cast<uintptr_t>(rgb) % 16)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: the image must be 16-byte aligned");
  RT_ON_DEVICE(device);
  float cumsum = 0.f;
  const unsigned blocks = static_cast<unsigned>((n_pixels + 255) / 256);
  if (flags & RT_TONEMAP_SERIAL_SUM) {
    float *terms = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&terms), (n_pixels + 1) * sizeof(float)));
    hipLaunchKernelGGL(rtd::k_log_luminosity_terms, dim3(blocks), dim3(256), 0, stream, rgb, n_pixels, delta, terms);
    hipLaunchKernelGGL(rtd::k_serial_sum, dim3(1), dim3(64), 0, stream, terms, n_pixels, terms + n_pixels);
    hipError_t e = hipMemcpyAsync(&cumsum, terms + n_pixels, sizeof(float), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(stream);
    (void)hipFree(terms);
    if (e != hipSuccess)
      return fail(RT_ERR_HIP, std::string("rt_average_luminosity: ") + hipGetErrorString(e));
  } else {
    int n_cus = 0;
    if (rt_status st = device_cu_count(device, &n_cus))
      return st;
    unsigned long long *sum = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&sum), sizeof(unsigned long long)));
    long long fixed = 0;
    hipError_t e = hipMemsetAsync(sum, 0, sizeof(unsigned long long), stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(rtd::k_log_luminosity_sum, dim3(std::min<unsigned>(blocks, 4u * n_cus)), dim3(256), 0, stream, rgb, n_pixels, delta, sum);
      e = hipMemcpyAsync(&fixed, sum, sizeof(fixed), hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess)
      e = hipStreamSynchronize(stream);
    (void)hipFree(sum);
    if (e != hipSuccess)
      return fail(RT_ERR_HIP, std::string("rt_average_luminosity: ") + hipGetErrorString(e));
    cumsum = static_cast<float>(static_cast<double>(fixed) * (1.0 / 4294967296.0));
  }
  *avg = std::pow(10.f, cumsum / n_pixels); // colors.hpp:475, as written (float / size_t)
  return RT_OK;
}

} // namespace

rt_status rt_average_luminosity(int device, const float *rgb_device, size_t n_pixels, float delta, uint32_t flags, void *stream,
                                float *avg_out) {
  return average_luminosity(device, rgb_device, n_pixels, delta, flags, static_cast<hipStream_t>(stream), avg_out);
}

rt_status rt_tonemap(int device, const float *rgb_device, size_t n_pixels, const rt_tonemap_cfg *cfg, uint8_t *ldr_out_device,
                     float *hdr_out_device, void *stream_, float *avg_used_out) {
  if (!rgb_device || !cfg || n_pixels == 0 || (!ldr_out_device && !hdr_out_device))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: NULL argument or empty image");
  if (!(cfg->gamma > 0.f))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: gamma must be positive");
  if (reinterpret_cast<uintptr_t>(rgb_device) % 16 || reinterpret_cast<uintptr_t>(hdr_out_device) % 16 ||
      reinterpret_cast<uintptr_t>(ldr_out_device) % 4)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap: float images must be 16-byte aligned, the 8-bit image 4-byte aligned");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float avg = cfg->avg_luminosity;
  if (!cfg->has_avg_luminosity)
    if (rt_status st = average_luminosity(device, rgb_device, n_pixels, cfg->delta, cfg->flags, stream, &avg))
      return st;
  RT_ON_DEVICE(device);
  const float scale = cfg->alpha / avg; // colors.hpp:485
  const size_t n_comp = 3 * n_pixels, quads = (n_comp + 3) / 4;
  hipLaunchKernelGGL(rtd::k_tonemap, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, stream, rgb_device, n_comp, scale,
                     1.f / cfg->gamma, ldr_out_device, hdr_out_device);
  RT_HIP(hipGetLastError());
  if (avg_used_out)
    *avg_used_out = avg;
  return RT_OK;
}

rt_status rt_tonemap_host(int device, const float *rgb_host, size_t n_pixels, const rt_tonemap_cfg *cfg, uint8_t *ldr_out_host,
                          float *avg_used_out) {
  if (!rgb_host || !cfg || !ldr_out_host || n_pixels == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_tonemap_host: NULL argument or empty image");
  RT_ON_DEVICE(device);
  float *rgb = nullptr;
  uint8_t *ldr = nullptr;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&rgb), 3 * n_pixels * sizeof(float)));
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&ldr), 3 * n_pixels + 4);
  rt_status st = RT_OK;
  if (e == hipSuccess)
    e = hipMemcpy(rgb, rgb_host, 3 * n_pixels * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    st = rt_tonemap(device, rgb, n_pixels, cfg, ldr, nullptr, nullptr, avg_used_out);
    if (st == RT_OK)
      e = hipMemcpy(ldr_out_host, ldr, 3 * n_pixels, hipMemcpyDeviceToHost);
  }
  (void)hipFree(rgb);
  (void)hipFree(ldr);
  if (e != hipSuccess)
    return fail(RT_ERR_HIP, std::string("rt_tonemap_host: ") + hipGetErrorString(e));
  return st;
}

