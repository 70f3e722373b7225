// rt_multi.hip — rt_multi_*: one frame on several GPUs of one node from one process.
#include <dlfcn.h>

#include "rt_internal.h"

using namespace rt_internal;

// ---------------------------------------------------------------------------------------------
// One frame on several GPUs of one node: rows dealt round-robin, one scene replica, stream and band
// buffer per device, gather to the first device with RCCL point-to-point calls (xGMI is point to point: every peer
// sends over its own link, nothing is gained from a ring at <= 12 MB per peer).
// ---------------------------------------------------------------------------------------------
namespace {

// RCCL is loaded on first use (single-GPU callers never pay for it, and a process that already holds a copy — PyTorch
// bundles one under the same soname — keeps using that copy).
struct Rccl {
  using Comm = void *;
  int (*CommInitAll)(Comm *, int, const int *) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  static constexpr int Float32 = 7; // ncclFloat32 (rccl.h)
  bool ok = false;
  std::string error;
};
Rccl &rccl() {
  static Rccl r = [] {
    Rccl x;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
        break;
    if (!h) {
      x.error = std::string("librccl.so.1 not found: ") + dlerror();
      return x;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(h, n);
      if (!p)
        x.error = std::string("RCCL symbol missing: ") + n;
      return p;
    };
    x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
    x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    x.ok = x.error.empty();
    return x;
  }();
  return r;
}
#define RT_NCCL(call)                                                                                      \
  do {                                                                                                     \
    int e_ = (call);                                                                                       \
    if (e_ != 0)                                                                                           \
      return fail(RT_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(e_));                     \
  } while (0)

// rows of one rank's compact band buffer -> their places in the row-major frame (image row = local row * n_ranks + rank)
__global__ void __launch_bounds__(256) k_place_rows(const float4 *band, float4 *frame, int n_local_rows, int row_quads, int rank, int n_ranks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)n_local_rows * row_quads;
  if (i >= n)
    return;
  const size_t lr = i / row_quads, q = i - lr * row_quads;
  frame[(lr * n_ranks + rank) * row_quads + q] = band[i];
}
__global__ void __launch_bounds__(256) k_place_rows_scalar(const float *band, float *frame, int n_local_rows, int row_floats, int rank, int n_ranks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)n_local_rows * row_floats;
  if (i >= n)
    return;
  const size_t lr = i / row_floats, q = i - lr * row_floats;
  frame[(lr * n_ranks + rank) * row_floats + q] = band[i];
}

} // namespace

struct rt_multi {
  std::vector<int> devices;             // rank r renders on devices[r]; rank 0's device assembles the frame
  std::vector<rt_scene *> scenes;       // one replica of the World per rank
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> begun, done;  // per rank: around its render
  std::vector<hipStream_t> poll_streams; // per rank: small copies of the sample-queue head while its kernel runs (progress)
  hipEvent_t gather_begin = nullptr, gather_end = nullptr; // on devices[0]'s stream: around transfer + placement
  std::vector<DeviceBuffer> bands;      // per rank, on its device: its rows, compact
  std::vector<DeviceBuffer> landing;    // per rank >= 1, on devices[0]: where its band arrives
  DeviceBuffer frame;                   // on devices[0]: rt_multi_render_host's frame
  std::vector<Rccl::Comm> comms;        // empty: peer copies instead of RCCL
  bool self_send = false;               // RT_MULTI_RCCL_SELF: rank 0's own band travels through RCCL too
};

extern "C" {

void rt_multi_destroy(rt_multi *m) {
  if (!m)
    return;
  for (Rccl::Comm c : m->comms)
    if (c)
      (void)rccl().CommDestroy(c);
  for (size_t r = 0; r < m->devices.size(); ++r) {
    DeviceGuard guard(m->devices[r]);
    if (r < m->streams.size() && m->streams[r]) (void)hipStreamDestroy(m->streams[r]);
    if (r < m->begun.size() && m->begun[r]) (void)hipEventDestroy(m->begun[r]);
    if (r < m->done.size() && m->done[r]) (void)hipEventDestroy(m->done[r]);
    if (r < m->poll_streams.size() && m->poll_streams[r]) (void)hipStreamDestroy(m->poll_streams[r]);
    if (r < m->bands.size() && m->bands[r].ptr) (void)hipFree(m->bands[r].ptr);
    if (r < m->scenes.size()) rt_scene_destroy(m->scenes[r]);
  }
  if (!m->devices.empty()) {
    DeviceGuard guard(m->devices[0]);
    for (DeviceBuffer &b : m->landing)
      if (b.ptr) (void)hipFree(b.ptr);
    if (m->frame.ptr) (void)hipFree(m->frame.ptr);
    if (m->gather_begin) (void)hipEventDestroy(m->gather_begin);
    if (m->gather_end) (void)hipEventDestroy(m->gather_end);
  }
  delete m;
}

rt_status rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, uint32_t flags, rt_multi **out) {
  if (!desc || !out || n_devices < 0 || (n_devices > 0 && !devices))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: NULL argument");
  *out = nullptr;
  if (flags & ~(RT_MULTI_PEER_COPIES | RT_MULTI_RCCL_SELF))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: unknown flag");
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible == 0)
    return fail(RT_ERR_NO_DEVICE, "rt_multi_create: no HIP device visible");
  rt_multi *m = new (std::nothrow) rt_multi();
  if (!m)
    return fail(RT_ERR_OUT_OF_MEMORY, "rt_multi_create: host allocation failed");
  auto bail = [&](rt_status st) {
    rt_multi_destroy(m);
    return st;
  };
  for (int r = 0; r < n_devices; ++r)
    if (devices[r] < 0 || devices[r] >= visible)
      return bail(fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_create: device index out of range"));
  if (n_devices == 0)
    for (int d = 0; d < visible; ++d)
      m->devices.push_back(d); // every visible device
  else
    m->devices.assign(devices, devices + n_devices);
  const size_t n = m->devices.size();
  bool distinct = true;
  for (size_t r = 0; r < n; ++r)
    for (size_t q = 0; q < r; ++q)
      distinct = distinct && m->devices[q] != m->devices[r];
  m->scenes.assign(n, nullptr), m->streams.assign(n, nullptr), m->begun.assign(n, nullptr), m->done.assign(n, nullptr);
  m->poll_streams.assign(n, nullptr);
  m->bands.assign(n, DeviceBuffer{}), m->landing.assign(n, DeviceBuffer{});
  for (size_t r = 0; r < n; ++r) {
    if (rt_status st = rt_scene_create(desc, m->devices[r], &m->scenes[r]))
      return bail(st);
    DeviceGuard guard(m->devices[r]);
    if (guard.err != hipSuccess || hipStreamCreateWithFlags(&m->streams[r], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&m->poll_streams[r], hipStreamNonBlocking) != hipSuccess ||
        (r == 0 && (hipEventCreate(&m->gather_begin) != hipSuccess || hipEventCreate(&m->gather_end) != hipSuccess)) ||
        hipEventCreate(&m->begun[r]) != hipSuccess || hipEventCreate(&m->done[r]) != hipSuccess)
      return bail(fail(RT_ERR_HIP, "rt_multi_create: stream / event creation failed"));
  }
  // Gather transport: RCCL when every rank has its own device (a communicator cannot hold one device twice); ranks that
  // share a device (several "virtual ranks" on one GPU: tests, rehearsals) or RT_MULTI_PEER_COPIES use peer copies.
  m->self_send = (flags & RT_MULTI_RCCL_SELF) != 0;
  const bool want_rccl = distinct && !(flags & RT_MULTI_PEER_COPIES) && (n > 1 || m->self_send);
  if (want_rccl && !rccl().ok && !m->self_send) {
    // no RCCL in this process (librccl.so.1 cannot be loaded): the gather falls back to peer copies, which need nothing but
    // HIP; rt_multi_uses_rccl() tells the caller which transport a handle got
    (void)fail(RT_OK, "rt_multi_create: " + rccl().error + " — gathering with peer copies");
  } else if (want_rccl) {
    if (!rccl().ok)
      return bail(fail(RT_ERR_HIP, "rt_multi_create: " + rccl().error));
    m->comms.assign(n, nullptr);
    int e = rccl().CommInitAll(m->comms.data(), static_cast<int>(n), m->devices.data());
    if (e != 0)
      return bail(fail(RT_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(e)));
  }
  *out = m;
  return RT_OK;
}

int rt_multi_rank_count(const rt_multi *m) { return m ? static_cast<int>(m->devices.size()) : 0; }
int rt_multi_uses_rccl(const rt_multi *m) { return m && !m->comms.empty() ? 1 : 0; }

} // extern "C"

namespace {
rt_status multi_render_impl(rt_multi *m, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_device0, rt_stats *stats,
                            rt_multi_timing *timing, int n_reports, rt_progress_fn progress, void *user) {
  if (!m || !cam || !cfg || !rgb_out_device0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render: NULL argument");
  if (rt_status st = check_cfg(cfg))
    return st;
  if (!whole_image(cfg))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render renders whole images (band_rows = n_local_rows = height, band_offset = 0): the rows are dealt to the devices inside");
  const int n = static_cast<int>(m->devices.size());
  const size_t row_floats = static_cast<size_t>(cfg->width) * 3;
  std::vector<int> rows(static_cast<size_t>(n));
  std::vector<LaunchInfo> info(static_cast<size_t>(n));
  // progress between the first and the last report exists in path mode only (its kernels take samples from a queue whose
  // head is counters[0]); the other modes' frames take milliseconds and report once, at the end
  const bool poll = progress && n_reports > 1 && cfg->mode == RT_MODE_PATH;
  // 1. every rank renders its rows (row r, r + n, r + 2n, ...) into its compact band buffer; launches are asynchronous,
  //    so one host thread keeps all devices busy
  for (int r = 0; r < n; ++r) {
    rows[r] = cfg->height > r ? (cfg->height - r + n - 1) / n : 0;
    if (rows[r] == 0)
      continue;
    RT_ON_DEVICE(m->devices[r]);
    if (rt_status st = ensure(m->bands[r], static_cast<size_t>(rows[r]) * row_floats * sizeof(float)))
      return st;
    rt_render_cfg band = *cfg;
    band.band_rows = 1, band.band_stride = n, band.band_offset = r, band.n_local_rows = rows[r];
    Workspace ws;
    ws.info = &info[r];
    if (poll) // the queue head the poll reads must not be the previous frame's
      RT_HIP(hipMemset(m->scenes[r]->counters, 0, sizeof(unsigned long long)));
    RT_HIP(hipEventRecord(m->begun[r], m->streams[r]));
    if (rt_status st = render_impl(m->scenes[r], cam, &band, static_cast<float *>(m->bands[r].ptr), m->streams[r], nullptr, ws))
      return st;
    RT_HIP(hipEventRecord(m->done[r], m->streams[r]));
  }
  // 2. gather to rank 0's device and put the rows in place
  RT_ON_DEVICE(m->devices[0]);
  hipStream_t s0 = m->streams[0];
  const bool use_rccl = !m->comms.empty();
  const int first_moved = use_rccl && m->self_send ? 0 : 1;
  for (int r = first_moved; r < n; ++r)
    if (rows[r])
      if (rt_status st = ensure(m->landing[r], static_cast<size_t>(rows[r]) * row_floats * sizeof(float)))
        return st;
  // device 0's stream waits for every rank's render before the gather starts, so that the two events around the gather
  // time the transfer and the placement alone (the bands are <= 12 MB per peer: nothing is lost by not overlapping them)
  for (int r = 0; r < n; ++r)
    if (rows[r])
      RT_HIP(hipStreamWaitEvent(s0, m->done[r], 0));
  RT_HIP(hipEventRecord(m->gather_begin, s0));
  if (use_rccl) {
    RT_NCCL(rccl().GroupStart());
    int first_error = 0; // a failed call must not leave this thread's RCCL group open (later RCCL calls — PyTorch's too — would queue into it)
    const char *failed = "";
    for (int r = first_moved; r < n && !first_error; ++r) {
      if (!rows[r])
        continue;
      const size_t count = static_cast<size_t>(rows[r]) * row_floats;
      if ((first_error = rccl().Send(m->bands[r].ptr, count, Rccl::Float32, 0, m->comms[r], m->streams[r])) != 0) { // ordered after rank r's render
        failed = "ncclSend";
        break;
      }
      if ((first_error = rccl().Recv(m->landing[r].ptr, count, Rccl::Float32, r, m->comms[0], s0)) != 0)
        failed = "ncclRecv";
    }
    const int end_error = rccl().GroupEnd();
    if (first_error)
      return fail(RT_ERR_HIP, std::string(failed) + ": " + rccl().GetErrorString(first_error));
    if (end_error)
      return fail(RT_ERR_HIP, std::string("ncclGroupEnd: ") + rccl().GetErrorString(end_error));
  } else {
    for (int r = 1; r < n; ++r) {
      if (!rows[r])
        continue;
      RT_HIP(hipStreamWaitEvent(s0, m->done[r], 0));
      RT_HIP(hipMemcpyPeerAsync(m->landing[r].ptr, m->devices[0], m->bands[r].ptr, m->devices[r], static_cast<size_t>(rows[r]) * row_floats * sizeof(float), s0));
    }
  }
  for (int r = 0; r < n; ++r) {
    if (!rows[r])
      continue;
    const float *src = static_cast<const float *>(r >= first_moved ? m->landing[r].ptr : m->bands[r].ptr);
    if (row_floats % 4 == 0 && reinterpret_cast<uintptr_t>(rgb_out_device0) % 16 == 0) {
      const size_t quads = static_cast<size_t>(rows[r]) * (row_floats / 4);
      hipLaunchKernelGGL(k_place_rows, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, s0, reinterpret_cast<const float4 *>(src),
                         reinterpret_cast<float4 *>(rgb_out_device0), rows[r], static_cast<int>(row_floats / 4), r, n);
    } else {
      const size_t count = static_cast<size_t>(rows[r]) * row_floats;
      hipLaunchKernelGGL(k_place_rows_scalar, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, s0, src, rgb_out_device0, rows[r],
                         static_cast<int>(row_floats), r, n);
    }
    RT_HIP(hipGetLastError());
  }
  RT_HIP(hipEventRecord(m->gather_end, s0));
  // 3. progress while the ranks render: the share of the frame's samples their queues have handed out
  if (poll) {
    uint64_t total = 0;
    for (int r = 0; r < n; ++r)
      total += rows[r] ? info[r].n_samples : 0;
    int reported = 0;
    for (;;) {
      bool all_done = true;
      unsigned long long handed = 0;
      for (int r = 0; r < n; ++r) {
        if (!rows[r])
          continue;
        const hipError_t q = hipEventQuery(m->done[r]);
        if (q != hipSuccess && q != hipErrorNotReady)
          return fail(RT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
        if (q == hipErrorNotReady) {
          all_done = false;
          RT_ON_DEVICE(m->devices[r]);
          unsigned long long head = 0;
          RT_HIP(hipMemcpyAsync(&head, info[r].counters, sizeof head, hipMemcpyDeviceToHost, m->poll_streams[r]));
          RT_HIP(hipStreamSynchronize(m->poll_streams[r]));
          handed += std::min<unsigned long long>(head, info[r].n_samples);
        } else {
          handed += info[r].n_samples;
        }
      }
      (void)hipGetLastError(); // hipErrorNotReady is not an error
      if (all_done)
        break;
      const int due = static_cast<int>(handed * static_cast<unsigned long long>(n_reports) / std::max<uint64_t>(total, 1));
      while (reported < std::min(due, n_reports - 1)) // the last report is the finished frame's
        progress(static_cast<float>(++reported) / static_cast<float>(n_reports), user);
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  // 4. the frame is complete when rank 0's stream is idle; the other streams only have their sends left
  for (int r = n - 1; r >= 0; --r) {
    RT_ON_DEVICE(m->devices[r]);
    RT_HIP(hipStreamSynchronize(m->streams[r]));
  }
  if (progress)
    progress(1.f, user);
  // what the gather adds to the frame: transfer + placement, HIP events on device 0's stream around them (the stream waits
  // for every rank's render first)
  float gather_ms = 0.f;
  {
    RT_ON_DEVICE(m->devices[0]);
    RT_HIP(hipEventElapsedTime(&gather_ms, m->gather_begin, m->gather_end));
  }
  if (stats || timing) {
    if (stats)
      std::memset(stats, 0, sizeof *stats);
    if (timing)
      std::memset(timing, 0, sizeof *timing);
    float slowest = 0.f;
    uint64_t rays_total = 0, capped_total = 0, samples_total = 0;
    for (int r = 0; r < n; ++r) {
      if (!rows[r])
        continue;
      RT_ON_DEVICE(m->devices[r]);
      uint64_t rays = 0, capped = 0;
      if (rt_status st = read_tallies(info[r].counters, &rays, &capped))
        return st;
      float ms = 0.f;
      RT_HIP(hipEventElapsedTime(&ms, m->begun[r], m->done[r]));
      slowest = std::max(slowest, ms);
      rays_total += rays, capped_total += capped, samples_total += info[r].n_samples;
      if (timing && r < RT_MULTI_MAX_TIMED_RANKS)
        timing->render_ms[r] = ms, timing->rays[r] = rays;
    }
    if (stats) {
      stats->rays = rays_total, stats->capped_samples = capped_total, stats->samples = samples_total, stats->kernel_ms = slowest;
      stats->grid = info[0].grid, stats->block = info[0].block;
      std::memcpy(stats->kernel, info[0].kernel, sizeof stats->kernel);
    }
    if (timing)
      timing->n_ranks = n, timing->slowest_render_ms = slowest, timing->gather_ms = gather_ms;
  }
  return RT_OK;
}
} // namespace

extern "C" {

rt_status rt_multi_render(rt_multi *m, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_device0, rt_stats *stats,
                          rt_multi_timing *timing) {
  return multi_render_impl(m, cam, cfg, rgb_out_device0, stats, timing, 0, nullptr, nullptr);
}

rt_status rt_multi_render_frame_host(rt_multi *m, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_host, int n_reports,
                                     rt_progress_fn progress, void *user, rt_stats *stats, rt_multi_timing *timing) {
  if (!m || !cfg || !rgb_out_host)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_multi_render_frame_host: NULL argument");
  if (rt_status st = check_cfg(cfg))
    return st;
  RT_ON_DEVICE(m->devices[0]);
  const size_t bytes = static_cast<size_t>(cfg->height) * cfg->width * 3 * sizeof(float);
  if (rt_status st = ensure(m->frame, bytes))
    return st;
  if (rt_status st = multi_render_impl(m, cam, cfg, static_cast<float *>(m->frame.ptr), stats, timing, std::clamp(n_reports, 1, 256), progress, user))
    return st;
  RT_HIP(hipMemcpy(rgb_out_host, m->frame.ptr, bytes, hipMemcpyDeviceToHost));
  return RT_OK;
}

rt_status rt_multi_render_host(rt_multi *m, const rt_camera *cam, const rt_render_cfg *cfg, float *rgb_out_host, rt_stats *stats,
                               rt_multi_timing *timing) {
  return rt_multi_render_frame_host(m, cam, cfg, rgb_out_host, 1, nullptr, nullptr, stats, timing);
}

} // extern "C"
