// runtime.h -- host runtime under the C-ABI: per-device contexts (constant tables, a pool of "lanes"
// = streams + reusable device buffers + pinned staging), contiguous-range sharding over GPUs, and the
// chunked H2D / kernel / D2H pipeline of the host-pointer entry points.
//
// Why it looks like this (SURVEY.md section 8b "each call uses its own stream"; round-1 review):
//   * hipMalloc / hipFree per call and copies on the NULL stream serialise concurrent callers and cost
//     ~10 % on a 2^20-item batch.  Each call now borrows a Lane from its device's pool: three
//     non-blocking streams (copy-in, kernels, copy-out), device buffers and pinned staging that are
//     reused by later calls, so the steady state does no allocation and never touches the NULL stream.
//   * A batch is cut into chunks that are whole multiples of one full wave of workgroups of the
//     kernel (host::plan_chunks; the quantum comes from the occupancy API), the copy of chunk i + 1
//     runs under the kernel of chunk i, and the device footprint is 3 chunks instead of the whole batch.
//   * ANEMOI_ALL_DEVICES shards a batch into contiguous ranges, one host thread per range.
//     ANEMOI_VIRTUAL_DEVICES=N (a test knob) makes that N ranges mapped round-robin onto the physical
//     devices, so the partition / gather code runs on a one-GPU box exactly as it would on N GPUs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/anemoi_mi355x.h"
#include "anemoi_kernels.h"
#include "host_logic.h"

namespace anemoi {
namespace rt {

inline thread_local std::string g_last_error;

inline int fail_hip(hipError_t e, const char* what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return ANEMOI_ERR_DEVICE;
}

#define HIP_TRY(expr)                                           \
  do {                                                          \
    hipError_t e_ = (expr);                                     \
    if (e_ != hipSuccess) return ::anemoi::rt::fail_hip(e_, #expr); \
  } while (0)

constexpr int kMaxDevices = 64;
constexpr int kSlots = 3;                         // chunks in flight per lane
constexpr int kScratch = 6;                       // ad-hoc device buffers per lane (Merkle drivers, verification)
constexpr size_t kChunkTargetBytes = 24u << 20;   // input bytes per chunk, rounded to the kernel's quantum
constexpr size_t kRetainBytes = 256u << 20;       // buffers above this are returned to HIP when a lane is released
constexpr size_t kMaxIdleLanes = 8;               // lanes kept per device between calls (each holds ~100 MB pinned + device)

struct DeviceGuard {  // restores the caller's current device
  int prev = -1;
  DeviceGuard() { (void)hipGetDevice(&prev); }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// grow-only buffer (device or pinned host)
struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  void release() {
    if (p) (void)(pinned ? hipHostFree(p) : hipFree(p));
    p = nullptr, cap = 0;
  }
  int reserve(size_t bytes) {
    if (bytes <= cap && p) return ANEMOI_OK;
    release();
    if (bytes < 256) bytes = 256;
    hipError_t e = pinned ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      p = nullptr;
      g_last_error = std::string(pinned ? "hipHostMalloc: " : "hipMalloc: ") + hipGetErrorString(e);
      return ANEMOI_ERR_ALLOC;
    }
    cap = bytes;
    return ANEMOI_OK;
  }
  void trim() {
    if (cap > kRetainBytes) release();
  }
};

struct Slot {
  Buf d_in, d_out, d_tmp, p_in, p_out;
  hipEvent_t e_in = nullptr, e_k = nullptr, e_out = nullptr;
};

// One in-flight host-pointer call on one device.
struct Lane {
  int dev = -1;
  // s_k: the lane's kernel stream; a call that fits one chunk runs entirely on it (copy-in, kernel,
  // copy-out), so a small call occupies ONE stream -- HIP multiplexes streams onto a handful of hardware
  // queues (GPU_MAX_HW_QUEUES, 4 by default) and kernels of two streams that share a queue serialise.
  // The other three exist only once a call needed the multi-chunk pipeline: s_in / s_out for the copies,
  // s_k2 so that consecutive chunks' kernels alternate between two streams and the next chunk's
  // workgroups fill the CUs as the previous chunk drains (one stream would put a full barrier -- ~1.3 ms
  // of ragged tail on a 19 ms chunk, measured -- between every two chunks).
  hipStream_t s_k = nullptr, s_k2 = nullptr, s_in = nullptr, s_out = nullptr;
  bool high_priority = false;
  Slot slot[kSlots];
  Buf scratch[kScratch];
  std::vector<hipEvent_t> events;  // extra events (one per tree level), created on demand

  int event(size_t i, hipEvent_t* out) {
    while (events.size() <= i) {
      hipEvent_t e = nullptr;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events.push_back(e);
    }
    *out = events[i];
    return ANEMOI_OK;
  }

  int create(int device) {
    dev = device;
    // HIP keeps a separate set of hardware queues per stream PRIORITY (streams of different priorities never share a queue:
    // profiles/r06/sampler_queue_collision.txt), and this platform has two.  Every second lane takes the other one, so
    // that concurrent callers spread over twice the queues without GPU_MAX_HW_QUEUES in the host's environment
    // (option lane_priorities; tools/exp_lane_priorities.py).
    static std::atomic<unsigned> made{0};
    int least = 0, greatest = 0;
    if (anemoi::opt::get_or(anemoi::opt::kLanePriorities, 1) && (made.fetch_add(1) & 1u) &&
        hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
      HIP_TRY(hipStreamCreateWithPriority(&s_k, hipStreamNonBlocking, greatest));
      high_priority = true;
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    }
    for (auto& s : slot) {
      s.p_in.pinned = s.p_out.pinned = true;
      HIP_TRY(hipEventCreateWithFlags(&s.e_in, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&s.e_k, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&s.e_out, hipEventDisableTiming));
    }
    return ANEMOI_OK;
  }
  int pipeline_streams() {  // the three extra streams of the multi-chunk pipeline, on first need
    if (!s_in) HIP_TRY(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    if (!s_out) HIP_TRY(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    if (!s_k2) {   // the OTHER priority than s_k: the two kernel streams of a lane can then never share a hardware queue
      int least = 0, greatest = 0;
      if (anemoi::opt::get_or(anemoi::opt::kLanePriorities, 1) && !high_priority &&
          hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
        HIP_TRY(hipStreamCreateWithPriority(&s_k2, hipStreamNonBlocking, greatest));
      else
        HIP_TRY(hipStreamCreateWithFlags(&s_k2, hipStreamNonBlocking));
    }
    return ANEMOI_OK;
  }
  void destroy() {
    for (auto& s : slot) {
      s.d_in.release(), s.d_out.release(), s.d_tmp.release(), s.p_in.release(), s.p_out.release();
      if (s.e_in) (void)hipEventDestroy(s.e_in);
      if (s.e_k) (void)hipEventDestroy(s.e_k);
      if (s.e_out) (void)hipEventDestroy(s.e_out);
      s.e_in = s.e_k = s.e_out = nullptr;
    }
    for (auto& b : scratch) b.release();
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    events.clear();
    for (hipStream_t* st : {&s_k, &s_k2, &s_in, &s_out}) {
      if (*st) (void)hipStreamDestroy(*st);
      *st = nullptr;
    }
  }
  void trim() {
    for (auto& s : slot) s.d_in.trim(), s.d_out.trim(), s.d_tmp.trim(), s.p_in.trim(), s.p_out.trim();
    for (auto& b : scratch) b.trim();
  }
};

struct DeviceCtx {
  std::mutex mu;
  bool ready[kNumFields][2] = {};
  PermConsts pc[kNumFields][2] = {};
  void* blob[kNumFields][2] = {};
  std::vector<Lane*> idle;
  int lanes_out = 0;  // lanes currently borrowed
  int num_cus = 0;
};
inline DeviceCtx g_ctx[kMaxDevices];

// Constant tables for (current device, field, width); uploaded once (one allocation, one blocking copy:
// the first use of an instance on a device synchronises -- anemoi_init() does it ahead of time, e.g.
// before a stream capture).  A failed upload frees the blob and leaves the instance "not ready".
inline int get_consts(int field, int width, PermConsts* out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return ANEMOI_ERR_DEVICE;
  DeviceCtx& c = g_ctx[dev];
  const int wi = width == 2 ? 0 : 1;
  std::lock_guard<std::mutex> lock(c.mu);
  if (!c.ready[field][wi]) {
    HostConsts hc;
    field_ops(field)->host_consts(width, &hc);
    // schedules go up as one 32-bit word per step (squarings | op << 8): scalar loads in the kernels
    auto words = [](const std::vector<uint8_t>& pairs) {
      std::vector<uint32_t> w(pairs.size() / 2);
      for (size_t i = 0; i < w.size(); i++) w[i] = uint32_t(pairs[2 * i]) | (uint32_t(pairs[2 * i + 1]) << 8);
      return w;
    };
    const std::vector<uint32_t> s3 = words(hc.sched), s5 = words(hc.sched5), sp = words(hc.sched_plain);
    const std::vector<uint32_t>* parts[9] = {&hc.ark_c, &hc.ark_d, &s3, &s5, &hc.coop_c, &hc.coop_d, &sp, &hc.fold_c, &hc.fold_d};
    size_t off[10] = {0};
    std::vector<uint32_t> host;
    for (int i = 0; i < 9; i++) {
      off[i + 1] = off[i] + parts[i]->size();
      host.insert(host.end(), parts[i]->begin(), parts[i]->end());
    }
    uint32_t* blob = nullptr;
    HIP_TRY(hipMalloc((void**)&blob, host.size() * sizeof(uint32_t) + 16));
    hipError_t e = hipMemcpy(blob, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(blob);
      return fail_hip(e, "hipMemcpy(constant tables)");
    }
    PermConsts pc;
    pc.ark_c = blob + off[0];
    pc.ark_d = blob + off[1];
    pc.sched = blob + off[2];
    pc.steps = hc.steps;
    pc.first = hc.first;
    pc.sched5 = blob + off[3];
    pc.steps5 = hc.steps5;
    pc.first5 = hc.first5;
    pc.coop_c = blob + off[4];
    pc.coop_d = blob + off[5];
    pc.sched_plain = blob + off[6];
    pc.fold_c = hc.fold_c.empty() ? nullptr : blob + off[7];
    pc.fold_d = hc.fold_d.empty() ? nullptr : blob + off[8];
    pc.steps_plain = hc.steps_plain;
    pc.first_plain = hc.first_plain;
    if (!c.num_cus) {  // (device_cus() takes this same mutex)
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      c.num_cus = n;
    }
    pc.simds = 4 * c.num_cus;
    c.pc[field][wi] = pc;
    c.blob[field][wi] = blob;
    c.ready[field][wi] = true;
  }
  *out = c.pc[field][wi];
  return ANEMOI_OK;
}

inline int device_cus(int dev) {
  DeviceCtx& c = g_ctx[dev];
  std::lock_guard<std::mutex> lock(c.mu);
  if (!c.num_cus) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    c.num_cus = n;
  }
  return c.num_cus;
}

// Borrow / return a lane of the CURRENT device (the caller has done hipSetDevice).
inline int acquire_lane(int dev, Lane** out) {
  DeviceCtx& c = g_ctx[dev];
  {
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.idle.empty()) {
      *out = c.idle.back();
      c.idle.pop_back();
      c.lanes_out++;
      return ANEMOI_OK;
    }
  }
  Lane* ln = new Lane();
  int rc = ln->create(dev);
  if (rc) {
    ln->destroy();
    delete ln;
    return rc;
  }
  std::lock_guard<std::mutex> lock(c.mu);
  c.lanes_out++;
  *out = ln;
  return ANEMOI_OK;
}

inline void release_lane(Lane* ln) {
  if (!ln) return;
  ln->trim();
  DeviceCtx& c = g_ctx[ln->dev];
  {
    std::lock_guard<std::mutex> lock(c.mu);
    c.lanes_out--;
    if (c.idle.size() < kMaxIdleLanes) {
      c.idle.push_back(ln);
      return;
    }
  }
  // a burst of concurrent callers is over: do not sit on its staging memory for ever
  DeviceGuard guard;
  if (hipSetDevice(ln->dev) == hipSuccess) ln->destroy();
  delete ln;
}

struct LaneGuard {
  Lane* ln = nullptr;
  ~LaneGuard() { release_lane(ln); }
};

// Frees everything the library holds on `dev` (constant tables, idle lanes).  Fails with ANEMOI_ERR_ARG
// while calls are in flight on that device.
inline int release_device(int dev) {
  DeviceCtx& c = g_ctx[dev];
  std::vector<Lane*> lanes;
  std::vector<void*> blobs;
  {
    // ONE critical section decides and invalidates: a caller that takes a lane after this point finds no
    // ready instance and uploads fresh tables; nobody can still hold a PermConsts into the blobs freed below
    // (they are only handed out under this mutex, to calls that hold a lane -- and none is out)
    std::lock_guard<std::mutex> lock(c.mu);
    if (c.lanes_out != 0) {
      g_last_error = "anemoi_release while calls are in flight on the device";
      return ANEMOI_ERR_ARG;
    }
    lanes.swap(c.idle);
    for (int f = 0; f < kNumFields; f++)
      for (int w = 0; w < 2; w++) {
        if (c.blob[f][w]) blobs.push_back(c.blob[f][w]);
        c.blob[f][w] = nullptr;
        c.ready[f][w] = false;
      }
  }
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(dev));
  for (Lane* ln : lanes) {
    ln->destroy();
    delete ln;
  }
  for (void* b : blobs) (void)hipFree(b);
  return ANEMOI_OK;
}

inline int physical_devices(int* ndev) {
  HIP_TRY(hipGetDeviceCount(ndev));
  if (*ndev <= 0) {
    g_last_error = "no HIP device";
    return ANEMOI_ERR_DEVICE;
  }
  if (*ndev > kMaxDevices) *ndev = kMaxDevices;
  return ANEMOI_OK;
}

// Number of ranges ANEMOI_ALL_DEVICES cuts a batch into: the GPU count, or the option "virtual_devices" (test
// knob: anemoi_set_option, or ANEMOI_VIRTUAL_DEVICES read once; options.h).
inline int shard_parts(int ndev) {
  const long long v = opt::get(opt::kVirtualDevices);
  return v >= 1 && v <= kMaxDevices ? int(v) : ndev;
}

// Runs `body(part, device, first, count)` for every part: one host thread per DEVICE, which works through its
// parts (part i belongs to device i % ndev) one after the other -- two big batches side by side on one GPU
// only slow each other down (LABNOTES.md section 7, item 2).  The first failure (lowest part) is reported with its
// thread's error text.
template <class Body>
int run_parts(int parts, int ndev, size_t n, Body body) {
  std::vector<int> rc(parts, ANEMOI_OK);
  std::vector<std::string> err(parts);
  std::vector<std::thread> th;
  const int workers = parts < ndev ? parts : ndev;
  th.reserve(workers);
  for (int w = 0; w < workers; w++)
    th.emplace_back([&, w] {
      for (int i = w; i < parts; i += ndev) {
        const size_t b = host::shard_begin(n, size_t(i), size_t(parts)), e = host::shard_begin(n, size_t(i) + 1, size_t(parts));
        rc[i] = body(i, w, b, e - b);
        if (rc[i] != ANEMOI_OK) err[i] = g_last_error;
      }
    });
  for (auto& t : th) t.join();
  for (int i = 0; i < parts; i++)
    if (rc[i] != ANEMOI_OK) {
      g_last_error = err[i];
      return rc[i];
    }
  return ANEMOI_OK;
}

// Runs `body(device, first, count)` on one device, or on contiguous ranges over all devices.
template <class Body>
int for_devices(int device, size_t n, Body body) {
  int ndev = 0, rc = physical_devices(&ndev);
  if (rc) return rc;
  if (device != ANEMOI_ALL_DEVICES) {
    if (device < 0 || device >= ndev) {
      g_last_error = "device ordinal out of range";
      return ANEMOI_ERR_DEVICE;
    }
    return body(device, size_t(0), n);
  }
  const int parts = shard_parts(ndev);
  if (parts == 1 || n < size_t(parts)) return body(0, size_t(0), n);
  return run_parts(parts, ndev, n, [&](int, int dev, size_t first, size_t count) { return body(dev, first, count); });
}

// Input bytes per chunk of the host-pointer pipelines.  The option "chunk_target_bytes" is a test knob: small
// values make small batches run through many chunks, so the slot ring wraps in the tests.
inline size_t chunk_target_bytes() { return size_t(opt::get_or(opt::kChunkTargetBytes, (long long)kChunkTargetBytes)); }

// 0 = copy straight from / to the caller's memory (HIP stages pageable memory itself);
// 1 = stage through the lane's pinned buffers (host memcpy + truly asynchronous DMA).
inline int staging_mode() { return int(opt::get_or(opt::kHostStaging, 1)); }

// Is [p, p + bytes) host memory that HIP already knows as pinned (hipHostMalloc / hipHostRegister)?  Then the
// DMA engines can read / write it directly and staging would only add a memcpy.
inline bool is_pinned_host(const void* p, size_t bytes) {
  if (!p || !bytes) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // ordinary pageable memory: "invalid value", not an error of ours
    return false;
  }
  if (a.type != hipMemoryTypeHost) return false;
  hipPointerAttribute_t b;
  if (hipPointerGetAttributes(&b, (const char*)p + bytes - 1) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return b.type == hipMemoryTypeHost;
}

// The chunked pipeline on one device: items [0, n) of `in` -> `out` (host pointers; out == in means in
// place).  launch(d_in, d_out, count, stream) enqueues the kernel for one chunk.
// With d_dst != nullptr the outputs stay on the device (chunk c's at d_dst + first * opi) and `out` is unused:
// the Merkle drivers build their first level this way, under the copy-in of the leaves.
template <class LaunchFn>
int pipeline(Lane& ln, size_t n, const char* in, size_t ipi, char* out, size_t opi, size_t quantum, LaunchFn launch,
             char* d_dst = nullptr) {
  const bool inplace = !d_dst && (const void*)in == (const void*)out;
  const host::ChunkPlan cp = host::plan_chunks(n, quantum, ipi, chunk_target_bytes());
  if (cp.chunks == 0) return ANEMOI_OK;
  bool staged = staging_mode() == 1;
  // the caller's buffers are pinned already (e.g. a pinned tensor, hipHostRegister'ed memory): copy straight
  if (staged && is_pinned_host(in, n * ipi) && (d_dst || is_pinned_host(out, n * opi))) staged = false;
  // Pinned staging is an optimisation: when the host cannot pin that much (locked-memory limits), fall back to
  // copying straight from the caller's memory instead of failing the call.
  auto pin = [&](Slot& sl, size_t in_bytes, size_t out_bytes) {
    if (!staged) return;
    if (sl.p_in.reserve(in_bytes) || (!d_dst && sl.p_out.reserve(out_bytes))) {
      staged = false;
      for (auto& s : ln.slot) s.p_in.release(), s.p_out.release();
    }
  };
  if (cp.chunks == 1) {
    // one chunk: copy-in, kernel, copy-out in order on the lane's kernel stream
    Slot& sl = ln.slot[0];
    int rc = sl.d_in.reserve(n * ipi);
    if (!rc && !inplace && !d_dst) rc = sl.d_out.reserve(n * opi);
    if (rc) return rc;
    pin(sl, n * ipi, n * opi);
    if (staged) memcpy(sl.p_in.p, in, n * ipi);
    HIP_TRY(hipMemcpyAsync(sl.d_in.p, staged ? (const void*)sl.p_in.p : (const void*)in, n * ipi, hipMemcpyHostToDevice,
                           ln.s_k));
    void* dst = d_dst ? (void*)d_dst : (inplace ? sl.d_in.p : sl.d_out.p);
    if ((rc = launch(sl.d_in.p, dst, n, ln.s_k))) return rc;
    if (d_dst) return ANEMOI_OK;  // the caller continues on s_k
    HIP_TRY(hipMemcpyAsync(staged ? sl.p_out.p : (void*)out, dst, n * opi, hipMemcpyDeviceToHost, ln.s_k));
    HIP_TRY(hipStreamSynchronize(ln.s_k));
    if (staged) memcpy(out, sl.p_out.p, n * opi);
    return ANEMOI_OK;
  }
  const int ns = kSlots;
  int rc0 = ln.pipeline_streams();
  if (rc0) return rc0;
  for (int s = 0; s < ns; s++) {
    Slot& sl = ln.slot[s];
    int rc = sl.d_in.reserve(cp.chunk_items * ipi);
    if (!rc && !inplace && !d_dst) rc = sl.d_out.reserve(cp.chunk_items * opi);
    if (rc) return rc;
    pin(sl, cp.chunk_items * ipi, cp.chunk_items * opi);
  }
  auto first_of = [&](size_t c) { return c * cp.chunk_items; };
  auto count_of = [&](size_t c) { return c + 1 == cp.chunks ? n - first_of(c) : cp.chunk_items; };
  auto copy_out = [&](size_t c) -> int {  // enqueue the D2H of chunk c behind its kernel
    if (d_dst) return ANEMOI_OK;
    Slot& sl = ln.slot[c % ns];
    const size_t bytes = count_of(c) * opi;
    HIP_TRY(hipStreamWaitEvent(ln.s_out, sl.e_k, 0));
    void* src = inplace ? sl.d_in.p : sl.d_out.p;
    HIP_TRY(hipMemcpyAsync(staged ? sl.p_out.p : (void*)(out + first_of(c) * opi), src, bytes, hipMemcpyDeviceToHost,
                           ln.s_out));
    HIP_TRY(hipEventRecord(sl.e_out, ln.s_out));
    return ANEMOI_OK;
  };
  auto drain = [&](size_t c) -> int {  // chunk c is complete on the host side; its slot is free again
    Slot& sl = ln.slot[c % ns];
    if (d_dst) {
      HIP_TRY(hipEventSynchronize(sl.e_k));
      return ANEMOI_OK;
    }
    HIP_TRY(hipEventSynchronize(sl.e_out));
    if (staged) memcpy(out + first_of(c) * opi, sl.p_out.p, count_of(c) * opi);
    return ANEMOI_OK;
  };
  for (size_t c = 0; c < cp.chunks; c++) {
    Slot& sl = ln.slot[c % ns];
    int rc;
    if (c >= size_t(ns) && (rc = drain(c - ns))) return rc;
    const size_t cnt = count_of(c), bytes = cnt * ipi;
    if (staged) {
      memcpy(sl.p_in.p, in + first_of(c) * ipi, bytes);
      HIP_TRY(hipMemcpyAsync(sl.d_in.p, sl.p_in.p, bytes, hipMemcpyHostToDevice, ln.s_in));
    } else {
      HIP_TRY(hipMemcpyAsync(sl.d_in.p, in + first_of(c) * ipi, bytes, hipMemcpyHostToDevice, ln.s_in));
    }
    HIP_TRY(hipEventRecord(sl.e_in, ln.s_in));
    hipStream_t ks = (c & 1) ? ln.s_k2 : ln.s_k;  // chunks are independent: their kernels may overlap
    HIP_TRY(hipStreamWaitEvent(ks, sl.e_in, 0));
    void* dst = d_dst ? (void*)(d_dst + first_of(c) * opi) : (inplace ? sl.d_in.p : sl.d_out.p);
    if ((rc = launch(sl.d_in.p, dst, cnt, ks))) return rc;
    HIP_TRY(hipEventRecord(sl.e_k, ks));
    // the copy-out of the previous chunk is enqueued AFTER this chunk's copy-in and kernel: with
    // pageable memory a D2H call blocks the host until its kernel has finished
    if (c >= 1 && (rc = copy_out(c - 1))) return rc;
  }
  int rc = copy_out(cp.chunks - 1);
  if (rc) return rc;
  for (size_t c = cp.chunks > size_t(ns) ? cp.chunks - ns : 0; c < cp.chunks; c++)
    if ((rc = drain(c))) return rc;  // (in d_dst mode this also joins both kernel streams on the host)
  return ANEMOI_OK;
}

// The same three-slot pipeline for calls whose chunks are not one array of equal items: several input arrays
// (path verification: leaves, indices, paths), ragged messages (a byte blob plus offsets rebased per chunk), a
// result that is consumed on the host (verification compares with the root).  The caller describes chunk c by
// its staging sizes, writes its inputs into ONE contiguous staging buffer (`stage`), enqueues its kernels
// (`launch`, on the stream it is given; `d_tmp` is per-slot device scratch) and consumes its outputs
// (`finish`).  Staging is pinned memory of the lane; when that cannot be had (locked-memory limits) or
// ANEMOI_HOST_STAGING=direct, pageable buffers take its place -- slower copies, same results.
// Device footprint: three chunks.
struct StagedChunk {
  size_t in_bytes, out_bytes, tmp_bytes;
};

inline void quiesce(Lane& ln);

template <class Plan, class Stage, class Launch, class Finish>
int pipeline_staged(Lane& ln, size_t chunks, Plan plan, Stage stage, Launch launch, Finish finish) {
  if (chunks == 0) return ANEMOI_OK;
  size_t max_in = 0, max_out = 0, max_tmp = 0;
  for (size_t c = 0; c < chunks; c++) {
    const StagedChunk sc = plan(c);
    max_in = sc.in_bytes > max_in ? sc.in_bytes : max_in;
    max_out = sc.out_bytes > max_out ? sc.out_bytes : max_out;
    max_tmp = sc.tmp_bytes > max_tmp ? sc.tmp_bytes : max_tmp;
  }
  const int ns = chunks < size_t(kSlots) ? int(chunks) : kSlots;
  bool pinned = staging_mode() == 1;
  std::vector<char> pg_in[kSlots], pg_out[kSlots];  // pageable stand-ins for the pinned staging
  // The stand-ins are locals while asynchronous copies may still read / write them when a stage / launch / finish or
  // a HIP call fails half-way: every early return below first waits for the lane's streams (declared after the
  // vectors, so it runs before they are freed).  With pinned staging the buffers belong to the lane and outlive this.
  struct QuiesceBeforeFree {
    Lane& ln;
    bool armed = false;
    ~QuiesceBeforeFree() {
      if (armed) quiesce(ln);
    }
  } guard{ln};
  if (chunks > 1) {
    int rc = ln.pipeline_streams();
    if (rc) return rc;
  }
  for (int i = 0; i < ns; i++) {
    Slot& sl = ln.slot[i];
    int rc = sl.d_in.reserve(max_in);
    if (!rc) rc = sl.d_out.reserve(max_out);
    if (!rc && max_tmp) rc = sl.d_tmp.reserve(max_tmp);
    if (rc) return rc;
    if (pinned && (sl.p_in.reserve(max_in) || sl.p_out.reserve(max_out))) {
      pinned = false;
      for (auto& s2 : ln.slot) s2.p_in.release(), s2.p_out.release();
    }
  }
  if (!pinned) {
    for (int i = 0; i < ns; i++) pg_in[i].resize(max_in ? max_in : 1), pg_out[i].resize(max_out ? max_out : 1);
    guard.armed = true;
  }
  auto host_in = [&](int i) { return pinned ? (char*)ln.slot[i].p_in.p : pg_in[i].data(); };
  auto host_out = [&](int i) { return pinned ? (char*)ln.slot[i].p_out.p : pg_out[i].data(); };
  const bool single = chunks == 1;
  auto copy_out = [&](size_t c) -> int {
    const int i = int(c % ns);
    Slot& sl = ln.slot[i];
    const size_t bytes = plan(c).out_bytes;
    hipStream_t so = single ? ln.s_k : ln.s_out;
    if (!single) HIP_TRY(hipStreamWaitEvent(so, sl.e_k, 0));
    if (bytes) HIP_TRY(hipMemcpyAsync(host_out(i), sl.d_out.p, bytes, hipMemcpyDeviceToHost, so));
    HIP_TRY(hipEventRecord(sl.e_out, so));
    return ANEMOI_OK;
  };
  auto drain = [&](size_t c) -> int {
    const int i = int(c % ns);
    HIP_TRY(hipEventSynchronize(ln.slot[i].e_out));
    return finish(c, (const char*)host_out(i));
  };
  for (size_t c = 0; c < chunks; c++) {
    const int i = int(c % ns);
    Slot& sl = ln.slot[i];
    int rc;
    if (c >= size_t(ns) && (rc = drain(c - ns))) return rc;
    const StagedChunk sc = plan(c);
    if ((rc = stage(c, host_in(i)))) return rc;
    hipStream_t si = single ? ln.s_k : ln.s_in;
    if (sc.in_bytes) HIP_TRY(hipMemcpyAsync(sl.d_in.p, host_in(i), sc.in_bytes, hipMemcpyHostToDevice, si));
    hipStream_t ks = single ? ln.s_k : ((c & 1) ? ln.s_k2 : ln.s_k);
    if (!single) {
      HIP_TRY(hipEventRecord(sl.e_in, si));
      HIP_TRY(hipStreamWaitEvent(ks, sl.e_in, 0));
    }
    if ((rc = launch(c, sl.d_in.p, sl.d_out.p, sl.d_tmp.p, ks))) return rc;
    if (!single) HIP_TRY(hipEventRecord(sl.e_k, ks));
    if (c >= 1 && (rc = copy_out(c - 1))) return rc;
  }
  int rc = copy_out(chunks - 1);
  if (rc) return rc;
  for (size_t c = chunks > size_t(ns) ? chunks - ns : 0; c < chunks; c++)
    if ((rc = drain(c))) return rc;
  guard.armed = false;   // every copy has been waited for (drain synchronises on each slot's copy-out event)
  return ANEMOI_OK;
}

// Waits for everything a failed or finished call left on the lane's streams, so that the lane can be
// reused (and its buffers freed) safely.
inline void quiesce(Lane& ln) {
  for (hipStream_t st : {ln.s_in, ln.s_k, ln.s_k2, ln.s_out})
    if (st) (void)hipStreamSynchronize(st);
}

// Host-pointer batch: shard over devices, borrow a lane per shard, run the pipeline.
template <class QuantumFn, class LaunchFn>
int host_batch(int device, size_t n, const void* in, size_t ipi, void* out, size_t opi, QuantumFn quantum,
               LaunchFn launch) {
  if (n == 0) return ANEMOI_OK;
  return for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (count == 0) return ANEMOI_OK;
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    LaneGuard lg;
    int rc = acquire_lane(dev, &lg.ln);
    if (rc) return rc;
    rc = pipeline(*lg.ln, count, (const char*)in + first * ipi, ipi, (char*)out + first * opi, opi,
                  quantum(dev), launch);
    if (rc) quiesce(*lg.ln);
    return rc;
  });
}

}  // namespace rt
}  // namespace anemoi
