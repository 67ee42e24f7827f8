// capi.hip -- the C-ABI of libanemoi_mi355x.so (declared in include/anemoi_mi355x.h).
//
// Host side only: argument checks (the reference's assert!s -> error codes), the entry points, and the
// Merkle drivers.  The machinery underneath -- per-device constant tables, lanes (streams + reusable
// buffers), the chunked copy/compute pipeline and the contiguous-range sharding over GPUs (no
// collective: items are independent) -- is runtime.h; the index arithmetic is
// host_logic.h.  All field arithmetic runs in the HIP kernels of anemoi_kernels.h; there is no CPU path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/anemoi_mi355x.h"
#include "anemoi_kernels.h"
#include "host_logic.h"
#include "runtime.h"

namespace anemoi {
const FieldOps *field_ops_0(), *field_ops_1(), *field_ops_2(), *field_ops_3(), *field_ops_4(), *field_ops_5(),
    *field_ops_6();

const FieldOps* field_ops(int field) {
  static const FieldOps* const table[kNumFields] = {field_ops_0(), field_ops_1(), field_ops_2(), field_ops_3(),
                                                    field_ops_4(), field_ops_5(), field_ops_6()};
  return field >= 0 && field < kNumFields ? table[field] : nullptr;
}
}  // namespace anemoi

using anemoi::FieldOps;
using anemoi::PermConsts;
namespace host = anemoi::host;
namespace rt = anemoi::rt;
using rt::DeviceGuard;
using rt::g_last_error;
using rt::get_consts;
using rt::Lane;
using rt::LaneGuard;

namespace {

int check_instance(int field, int width) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (width != 2 && width != 4) return ANEMOI_ERR_WIDTH;
  return ANEMOI_OK;
}

int check_k(int width, int k) { return host::valid_k(width, k) ? ANEMOI_OK : ANEMOI_ERR_ARG; }

size_t elem_bytes(int field) { return size_t(anemoi::field_ops(field)->limbs64) * 8; }

// Chunk quantum of a batch kernel on `dev`: items per full wave of workgroups (occupancy API; cached).
size_t quantum_of(int field, int kind, int width, int dev) {
  // per device: the occupancy query answers for the CURRENT device, and the CU count differs between
  // partition modes (SPX / CPX) and parts
  if (const long long v = anemoi::opt::get_or(anemoi::opt::kTestQuantum, 0)) return size_t(v);  // test knob: small quanta make small batches multi-chunk
  static std::mutex mu;
  static size_t cache[rt::kMaxDevices][anemoi::kNumFields][5][2] = {};
  const int wi = width == 2 ? 0 : 1;
  if (dev < 0 || dev >= rt::kMaxDevices) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  size_t& q = cache[dev][field][kind][wi];
  if (!q) {
    DeviceGuard guard;
    (void)hipSetDevice(dev);
    q = anemoi::field_ops(field)->wave_items(kind, width, rt::device_cus(dev));
  }
  return q;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Borrows a lane on `dev` and runs body(lane); waits for the lane's streams if body failed.
template <class Body>
int with_lane(int dev, Body body) {
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(dev));
  LaneGuard lg;
  int rc = rt::acquire_lane(dev, &lg.ln);
  if (rc) return rc;
  rc = body(*lg.ln);
  if (rc) rt::quiesce(*lg.ln);
  return rc;
}

// ---- run-time instances (anemoi_generic.h) ---------------------------------------------------------

int check_generic(const anemoi_generic_instance* inst) {
  if (!inst) return ANEMOI_ERR_ARG;
  if (!anemoi::field_ops(inst->field)) return ANEMOI_ERR_FIELD;
  if (inst->num_columns < 1 || inst->num_columns > ANEMOI_MAX_GENERIC_COLUMNS) return ANEMOI_ERR_WIDTH;
  if (inst->num_rounds < 1 || inst->num_rounds > 255 || !inst->ark_c || !inst->ark_d) return ANEMOI_ERR_ARG;
  if (!inst->mds && inst->num_columns > 6) return ANEMOI_ERR_ARG;  // "NO MDS matrix specified for this instance."
  return ANEMOI_OK;
}

// Device bytes an instance's constants need: the ABI elements as uploaded (ark_c | ark_d | mds) followed by the
// same elements in the kernels' internal form (FieldOps::generic_stride words each).
struct GenericLayout {
  size_t ab, mb, elems, abi_bytes, total;
};
GenericLayout generic_layout(const anemoi_generic_instance* inst) {
  const FieldOps* ops = anemoi::field_ops(inst->field);
  const size_t eb = elem_bytes(inst->field), c = size_t(inst->num_columns);
  GenericLayout g;
  g.ab = size_t(inst->num_rounds) * c * eb;
  g.mb = c * c * eb;
  g.elems = 2 * size_t(inst->num_rounds) * c + c * c;
  g.abi_bytes = align_up(2 * g.ab + g.mb, 256);
  g.total = g.abi_bytes + g.elems * size_t(ops->generic_stride) * 4;
  return g;
}

// Uploads an instance's constants into `d_buf` (generic_layout().total bytes) and converts them, on stream s.
// `canon` (the small-integer matrix of a hard-coded arm) is the source of an asynchronous copy: every exit,
// also the failing ones, waits for the stream before the vector dies.
int build_generic_consts(const anemoi_generic_instance* inst, char* d_buf, hipStream_t st, anemoi::GenericConsts* gc) {
  const FieldOps* ops = anemoi::field_ops(inst->field);
  const size_t eb = elem_bytes(inst->field), c = size_t(inst->num_columns);
  const GenericLayout g = generic_layout(inst);
  char* b = d_buf;
  uint32_t* internal = (uint32_t*)(d_buf + g.abi_bytes);
  std::vector<uint64_t> canon;
  auto body = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(b, inst->ark_c, g.ab, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + g.ab, inst->ark_d, g.ab, hipMemcpyHostToDevice, st));
    if (inst->mds) {
      HIP_TRY(hipMemcpyAsync(b + 2 * g.ab, inst->mds, g.mb, hipMemcpyHostToDevice, st));
    } else {
      std::vector<uint64_t> small;
      if (!host::builtin_mds(inst->num_columns, uint64_t(ops->generator), &small)) return ANEMOI_ERR_ARG;
      canon.assign(c * c * (eb / 8), 0);
      for (size_t i = 0; i < c * c; i++) canon[i * (eb / 8)] = small[i];
      HIP_TRY(hipMemcpyAsync(b + 2 * g.ab, canon.data(), g.mb, hipMemcpyHostToDevice, st));
      HIP_TRY(ops->mont_convert(1, b + 2 * g.ab, b + 2 * g.ab, c * c, st));
    }
    HIP_TRY(ops->generic_prepare(b, internal, g.elems, st));
    return ANEMOI_OK;
  };
  int rc = body();
  const hipError_t e = hipStreamSynchronize(st);
  if (rc) return rc;
  if (e != hipSuccess) return rt::fail_hip(e, "hipStreamSynchronize(constants of a run-time instance)");
  const size_t stride = size_t(ops->generic_stride), per = size_t(inst->num_rounds) * c;
  gc->ark_c = internal;
  gc->ark_d = internal + per * stride;
  gc->mds = internal + 2 * per * stride;
  gc->cols = inst->num_columns;
  gc->rounds = inst->num_rounds;
  return ANEMOI_OK;
}

// per shard of a host-pointer call: the constants live in lane scratch buffer 0
int upload_generic(Lane& ln, const anemoi_generic_instance* inst, anemoi::GenericConsts* gc) {
  int rc = ln.scratch[0].reserve(generic_layout(inst).total);
  if (rc) return rc;
  return build_generic_consts(inst, (char*)ln.scratch[0].p, ln.s_k, gc);
}

// Host-pointer batch over a run-time instance: the constants are uploaded once per shard, then the items go
// through the chunked pipeline (copy of chunk c + 1 under the kernel of chunk c, three chunks on the device).
// One state occupies `cols` lanes, so a full wave of workgroups is the fixed-instance quantum / cols.
template <class LaunchFn>
int generic_batch(const anemoi_generic_instance* inst, int device, size_t n, const void* in, size_t in_per_item,
                  void* out, size_t out_per_item, LaunchFn launch) {
  if (n == 0) return ANEMOI_OK;
  return rt::for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (count == 0) return ANEMOI_OK;
    return with_lane(dev, [&](Lane& ln) -> int {
      anemoi::GenericConsts gc;
      PermConsts pc;
      int rc = upload_generic(ln, inst, &gc);
      if (!rc) rc = get_consts(inst->field, 2, &pc);  // exponent schedule of the field
      if (rc) return rc;
      size_t quantum = quantum_of(inst->field, anemoi::kKindJive, 2, dev) / size_t(inst->num_columns);
      if (quantum < 64) quantum = 64;
      const host::ChunkPlan cp = host::plan_chunks(count, quantum, in_per_item > out_per_item ? in_per_item : out_per_item,
                                                   rt::chunk_target_bytes());
      const char* src = (const char*)in + first * in_per_item;
      char* dst = (char*)out + first * out_per_item;
      auto first_of = [&](size_t c) { return c * cp.chunk_items; };
      auto count_of = [&](size_t c) { return c + 1 == cp.chunks ? count - first_of(c) : cp.chunk_items; };
      return rt::pipeline_staged(
          ln, cp.chunks,
          [&](size_t c) { return rt::StagedChunk{count_of(c) * in_per_item, count_of(c) * out_per_item, 0}; },
          [&](size_t c, char* h) -> int {
            memcpy(h, src + first_of(c) * in_per_item, count_of(c) * in_per_item);
            return ANEMOI_OK;
          },
          [&](size_t c, void* di, void* dout, void*, hipStream_t st) -> int {
            // in place (permutation): the kernel works on the input buffer; its result is copied to d_out
            if (out == in) {
              HIP_TRY(launch(di, di, count_of(c), gc, pc, st));
              HIP_TRY(hipMemcpyAsync(dout, di, count_of(c) * out_per_item, hipMemcpyDeviceToDevice, st));
            } else {
              HIP_TRY(launch(di, dout, count_of(c), gc, pc, st));
            }
            return ANEMOI_OK;
          },
          [&](size_t c, const char* h) -> int {
            memcpy(dst + first_of(c) * out_per_item, h, count_of(c) * out_per_item);
            return ANEMOI_OK;
          });
    });
  });
}

// ---- Merkle drivers ----------------------------------------------------------------------------------

int merkle_levels_dev(int field, const void* d_leaves, unsigned depth, void* d_scratch, void* d_root,
                      hipStream_t s) {
  const FieldOps* ops = anemoi::field_ops(field);
  const size_t eb = elem_bytes(field);
  if (depth == 0) {
    HIP_TRY(hipMemcpyAsync(d_root, d_leaves, eb, hipMemcpyDeviceToDevice, s));
    return ANEMOI_OK;
  }
  PermConsts pc;
  int rc = get_consts(field, 2, &pc);
  if (rc) return rc;
  // ping-pong halves of the scratch: level l (n = 2^(depth-1-l) nodes) reads `src`, writes `dst`
  char* a = (char*)d_scratch;
  char* b = a + (size_t(1) << (depth - 1)) * eb;
  const void* src = d_leaves;
  for (unsigned l = 0; l < depth; l++) {
    const size_t n = size_t(1) << (depth - 1 - l);
    void* dst = n == 1 ? d_root : (void*)((l & 1) ? b : a);
    HIP_TRY(ops->jive(2, 2, src, dst, n, pc, s));
    src = dst;
  }
  return ANEMOI_OK;
}

// A tree (arity 2^alog = 2 with the 2-1 instance's merge, or 4 with the 4-3 instance's Jive-4) over
// arity^depth HOST leaves on the lane's device.  Level 1 is built by the chunked pipeline -- the copy of
// the next chunk of leaves runs under the compression of this one, and the leaves never have to be
// resident as a whole -- and stays on the device; the remaining levels follow on the lane's kernel
// stream.  With `tree_host` every level l >= 1 is copied out to tree_host + (level_off[l] + part *
// nodes_l) elements as soon as it is complete (on the copy-out stream, under the next levels); the root
// is always written to `root_host`.
struct TreeShape {
  int field, alog;  // alog = log2(arity): 1 or 2
  int width() const { return alog == 1 ? 2 : 4; }
  int arity() const { return 1 << alog; }
};

int subtree_host(Lane& ln, TreeShape ts, const char* leaves, unsigned depth, char* root_host, char* tree_host,
                 unsigned full_depth, size_t part) {
  const FieldOps* ops = anemoi::field_ops(ts.field);
  const size_t eb = elem_bytes(ts.field);
  const int W = ts.width(), K = ts.arity();
  if (depth == 0) {
    memcpy(root_host, leaves, eb);
    if (tree_host && tree_host + part * eb != leaves) memcpy(tree_host + part * eb, leaves, eb);
    return ANEMOI_OK;
  }
  PermConsts pc;
  int rc = get_consts(ts.field, W, &pc);
  if (rc) return rc;
  const size_t n1 = size_t(1) << (ts.alog * (depth - 1));  // nodes of level 1
  // device levels 1 .. depth, back to back: n1 + n1/K + ... + 1 <= 2 n1 elements
  size_t total = 0;
  for (unsigned l = 1; l <= depth; l++) total += size_t(1) << (ts.alog * (depth - l));
  if ((rc = ln.scratch[0].reserve(total * eb))) return rc;
  if (tree_host && (rc = ln.pipeline_streams())) return rc;  // the level copies run on s_out
  char* lvl = (char*)ln.scratch[0].p;
  const int dev = ln.dev;
  rc = rt::pipeline(
      ln, n1, leaves, size_t(K) * eb, nullptr, eb, quantum_of(ts.field, anemoi::kKindJive, W, dev),
      [&](void* di, void* dout, size_t cnt, hipStream_t s) -> int {
        HIP_TRY(ops->jive(W, K, di, dout, cnt, pc, s));
        return ANEMOI_OK;
      },
      lvl);
  if (rc) return rc;
  // D2H of a finished level, on the copy-out stream behind the event recorded after that level's kernel
  auto level_out = [&](unsigned l, const char* d_lvl, size_t nodes) -> int {
    if (!tree_host) return ANEMOI_OK;
    hipEvent_t e;
    int r = ln.event(l, &e);
    if (r) return r;
    HIP_TRY(hipStreamWaitEvent(ln.s_out, e, 0));
    const size_t off = ts.alog == 1 ? host::tree2_level_offset(full_depth, l) : host::tree4_level_offset(full_depth, l);
    HIP_TRY(hipMemcpyAsync(tree_host + (off + part * nodes) * eb, d_lvl, nodes * eb, hipMemcpyDeviceToHost, ln.s_out));
    return ANEMOI_OK;
  };
  auto mark = [&](unsigned l) -> int {  // level l's kernel(s) are queued on s_k: remember that point
    if (!tree_host) return ANEMOI_OK;
    hipEvent_t e;
    int r = ln.event(l, &e);
    if (r) return r;
    HIP_TRY(hipEventRecord(e, ln.s_k));
    return ANEMOI_OK;
  };
  if ((rc = mark(1))) return rc;
  const char* src = lvl;
  size_t n = n1;
  // level 0 of the retained tree = this part's leaves: a host-to-host copy, done once level 2 is queued
  const size_t nleaf = size_t(K) * n1;
  char* lvl0 = tree_host ? tree_host + part * nleaf * eb : nullptr;
  bool leaves_copied = !tree_host || lvl0 == leaves;
  for (unsigned l = 2; l <= depth; l++) {
    char* dst = (char*)src + n * eb;
    const size_t below = n;
    n >>= ts.alog;
    HIP_TRY(ops->jive(W, K, src, dst, n, pc, ln.s_k));
    if ((rc = mark(l))) return rc;
    if (!leaves_copied) {
      memcpy(lvl0, leaves, nleaf * eb);
      leaves_copied = true;
    }
    // the copy of level l-1 is issued AFTER level l's kernel is queued: a D2H into pageable memory
    // blocks the host until the data is there, and the GPU should not wait for the host meanwhile
    if ((rc = level_out(l - 1, src, below))) return rc;
    src = dst;
  }
  if (!leaves_copied) memcpy(lvl0, leaves, nleaf * eb);  // (depth 1: no level 2 to hide it under)
  if ((rc = level_out(depth, src, 1))) return rc;
  HIP_TRY(hipMemcpyAsync(root_host, src, eb, hipMemcpyDeviceToHost, ln.s_k));
  HIP_TRY(hipStreamSynchronize(ln.s_k));
  if (tree_host) HIP_TRY(hipStreamSynchronize(ln.s_out));
  return ANEMOI_OK;
}

// ---- sponge over long messages from host memory -------------------------------------------------------
// A batch too small to be cut into message chunks (fewer than two full waves of workgroups: config 3's
// 2^16 messages are 2/3 of one) but with a lot of bytes per message (config 3: 640 MiB) would pay the whole
// copy-in before its single launch.  It is cut ALONG the messages instead: segment c = bytes / elements
// [c S, (c+1) S) of every message, S a multiple of RATE elements so a segment starts on a permutation
// boundary; the kernel of segment c runs while segment c + 1 is gathered (one strided copy per message) and
// copied in; the sponge state travels from launch to launch in a device buffer (SpongeSeg).
constexpr size_t kSegmentMinBytes = size_t(64) << 20;   // below this the single launch is kept

// option "sponge_segment_bytes" (test knob): force the segment path, with (small) segments of this many bytes
bool segments_forced() { return anemoi::opt::get(anemoi::opt::kSpongeSegmentBytes) != anemoi::opt::kAuto; }
size_t segment_target_bytes() {
  return size_t(anemoi::opt::get_or(anemoi::opt::kSpongeSegmentBytes, (long long)rt::kChunkTargetBytes));
}

// unit = bytes of RATE elements of input (BYTES: RATE x chunk bytes; else RATE x element bytes)
bool want_segments(size_t n, size_t per_msg_bytes, size_t unit) {
  const bool forced = segments_forced();
  if (!forced && n * per_msg_bytes < kSegmentMinBytes) return false;
  size_t seg = segment_target_bytes() / (n ? n : 1) / unit * unit;
  if (seg < unit) seg = unit;
  return per_msg_bytes > 2 * seg;   // at least three segments, else nothing overlaps
}

int sponge_segments(Lane& ln, int field, int width, int bytes, const char* src, size_t per_msg, size_t n, char* out) {
  const FieldOps* ops = anemoi::field_ops(field);
  const size_t eb = elem_bytes(field), rate = size_t(width - 1);
  const size_t elem_in = bytes ? size_t(ops->chunk) : eb;        // input bytes per absorbed element
  const size_t unit = rate * elem_in, per_msg_bytes = bytes ? per_msg : per_msg * eb;
  size_t seg_bytes = segment_target_bytes() / n / unit * unit;
  if (seg_bytes < unit) seg_bytes = unit;
  const size_t nseg = (per_msg_bytes + seg_bytes - 1) / seg_bytes;
  PermConsts pc;
  int rc = get_consts(field, width, &pc);
  if (!rc) rc = ln.pipeline_streams();
  if (!rc) rc = ln.scratch[0].reserve(n * width * eb);   // carried sponge state
  if (!rc) rc = ln.scratch[1].reserve(n * eb);           // digests
  for (int s = 0; !rc && s < rt::kSlots; s++) rc = ln.slot[s].d_in.reserve(n * seg_bytes);
  if (rc) return rc;
  // Pinned staging is an optimisation, as in rt::pipeline: without it (ANEMOI_HOST_STAGING=direct, or the host
  // cannot pin that much) the segment goes up as ONE strided copy straight from the caller's memory.
  bool staged = rt::staging_mode() == 1;
  for (int s = 0; staged && s < rt::kSlots; s++)
    if (ln.slot[s].p_in.reserve(n * seg_bytes)) {
      staged = false;
      for (auto& sl : ln.slot) sl.p_in.release();
    }
  for (size_t c = 0; c < nseg; c++) {
    rt::Slot& sl = ln.slot[c % rt::kSlots];
    const size_t off = c * seg_bytes, len = c + 1 == nseg ? per_msg_bytes - off : seg_bytes;
    // the slot's pinned buffer is free once its previous copy-in has completed, its device buffer once the
    // kernel that read it has: the first is waited for here, the second is a stream dependency
    if (c >= size_t(rt::kSlots)) {
      HIP_TRY(hipEventSynchronize(sl.e_in));
      HIP_TRY(hipStreamWaitEvent(ln.s_in, sl.e_k, 0));
    }
    if (staged) {
      char* stage = (char*)sl.p_in.p;
      for (size_t i = 0; i < n; i++) memcpy(stage + i * len, src + i * per_msg_bytes + off, len);   // strided gather
      HIP_TRY(hipMemcpyAsync(sl.d_in.p, stage, n * len, hipMemcpyHostToDevice, ln.s_in));
    } else {
      HIP_TRY(hipMemcpy2DAsync(sl.d_in.p, len, src + off, per_msg_bytes, len, n, hipMemcpyHostToDevice, ln.s_in));
    }
    HIP_TRY(hipEventRecord(sl.e_in, ln.s_in));
    HIP_TRY(hipStreamWaitEvent(ln.s_k, sl.e_in, 0));
    anemoi::SpongeSeg seg{(uint32_t*)ln.scratch[0].p, off / elem_in, per_msg, c == 0 ? 1 : 0, c + 1 == nseg ? 1 : 0};
    // segments of the same messages depend on each other through the state: one kernel stream, in order
    HIP_TRY(ops->sponge_seg(width, bytes, sl.d_in.p, bytes ? len : len / eb, n, ln.scratch[1].p, pc, seg, ln.s_k));
    HIP_TRY(hipEventRecord(sl.e_k, ln.s_k));
  }
  HIP_TRY(hipMemcpyAsync(out, ln.scratch[1].p, n * eb, hipMemcpyDeviceToHost, ln.s_k));
  HIP_TRY(hipStreamSynchronize(ln.s_k));
  return ANEMOI_OK;
}

// Host-pointer sponge batch: by message chunks (rt::host_batch) or, for few long messages, by segments.
int sponge_host(int field, int width, int bytes, const void* src, size_t per_msg, size_t n, uint64_t* out, int device) {
  const FieldOps* ops = anemoi::field_ops(field);
  const size_t eb = elem_bytes(field);
  const size_t elem_in = bytes ? size_t(ops->chunk) : eb, unit = size_t(width - 1) * elem_in;
  const size_t per_msg_bytes = bytes ? per_msg : per_msg * eb;
  static const uint64_t dummy[2] = {0, 0};
  if (!per_msg) src = dummy;
  return rt::for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (count == 0) return ANEMOI_OK;
    return with_lane(dev, [&](Lane& ln) -> int {
      const char* in = (const char*)src + first * per_msg_bytes;
      char* o = (char*)out + first * eb;
      const size_t quantum = quantum_of(field, anemoi::kKindSponge, width, dev);
      // Few messages (cannot be cut by message), or long ones (a message chunk of one quantum would be hundreds of
      // MB of staging): feed blocks of at most one quantum of messages segment by segment.
      const size_t blk = count < 2 * quantum ? count : quantum;
      // (a batch small enough for the cooperative sponge kernels is compute-bound by orders of magnitude -- one
      // permutation per ~1 ms against 93 bytes of input -- so it goes up in one piece; but only while "one piece" is
      // small: a few LONG messages (4 096 x 16 MiB, one multi-GiB message) would otherwise need the whole batch on the
      // device and in pinned staging at once, where the segment path holds three segments.  Either way such a batch
      // takes the latency kernels: since round 5 they absorb segments too, Launch::sponge_seg)
      const bool latency_batch = count <= anemoi::sponge_latency_max_items(width, 4 * rt::device_cus(dev)) &&
                                 count * per_msg_bytes < kSegmentMinBytes && !segments_forced();
      if (!latency_batch && want_segments(blk, per_msg_bytes, unit) &&
          (count < 2 * quantum || quantum * per_msg_bytes > (size_t(256) << 20))) {
        for (size_t b = 0; b < count; b += blk) {
          const size_t m = count - b < blk ? count - b : blk;
          int r = sponge_segments(ln, field, width, bytes, in + b * per_msg_bytes, per_msg, m, o + b * eb);
          if (r) return r;
        }
        return ANEMOI_OK;
      }
      return rt::pipeline(ln, count, in, per_msg_bytes, o, eb, quantum, [&](void* i, void* d, size_t cnt, hipStream_t s) {
        return bytes ? anemoi_hash_bytes_dev(field, width, i, per_msg, cnt, d, s)
                     : anemoi_hash_field_dev(field, width, i, per_msg, cnt, d, s);
      });
    });
  });
}

// One level of arity-4 path verification: states[i] = the 4 children of item i's next node = its current
// node at slot (index >> 2 level) & 3, the path's 3 siblings of that level in the other slots (child order).
// Pure data movement, `quads` uint4 per element; one thread per (item, child, quad).
__global__ void k_assemble4(const uint4* __restrict__ cur, const uint4* __restrict__ paths,
                            const uint64_t* __restrict__ index, unsigned level, unsigned depth4, size_t n, int quads,
                            uint4* __restrict__ states) {
  const size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t per = size_t(4) * quads;
  if (t >= n * per) return;
  const size_t item = t / per;
  const int child = int((t % per) / quads), q = int(t % quads);
  const int pos = int((index[item] >> (2 * level)) & 3);
  uint4 v;
  if (child == pos) v = cur[item * quads + q];
  else v = paths[((item * depth4 + level) * 3 + (child < pos ? child : child - 1)) * quads + q];
  states[t] = v;
}

// ---- device-side bucketing of a ragged batch (anemoi_hash_bytes_ragged_bucketed_dev) -----------------------------------
// A wavefront of the ragged sponge kernels runs as many rate-blocks as its LONGEST message; host::ragged_order (the host
// path's bucketing, host_logic.h) is the specification: messages by descending block count.  Here as a counting sort on
// the device: histogram of the block counts (clamped to kRaggedBins - 1: longer messages share the first bucket),
// exclusive scan from the longest bucket down, placement by atomic cursor (four launches, nothing else).  Not stable -- equal block counts land in
// any order, which costs nothing -- and every digest goes back to its message's own index.
constexpr int kRaggedBins = 1 << 16;
// The scratch of the bucketed forms: [0] the STATUS word (+ 3 words of padding), kRaggedBins counters, n message indices.
// Status: 0, or ANEMOI_RAGGED_DECREASING | ANEMOI_RAGGED_BEYOND_EXTENT -- the offsets are device memory, so only the device
// can see that they are malformed: k_ragged_hist, which reads every pair anyway, flags a decreasing pair and a last offset
// beyond the caller-given extent of the blob; the sponge kernels read the word (one scalar load) and, if it is set, write
// zero digests without reading a message byte.  A malformed pair counts as an empty message here, so that histogram and
// placement agree and `order` is a permutation of 0 .. n-1 whatever the offsets hold.
constexpr int kRaggedHead = 4;
__device__ __forceinline__ uint32_t ragged_bin(const uint64_t* __restrict__ off, size_t i, uint32_t block_units) {
  const uint64_t o0 = off[i], o1 = off[i + 1];
  const uint64_t len = o1 >= o0 ? o1 - o0 : 0;
  const uint64_t blocks = len / block_units + (len % block_units ? 1 : 0);   // (no len + block_units - 1: len may be near 2^64)
  return uint32_t(kRaggedBins - 1) - uint32_t(blocks < uint64_t(kRaggedBins - 1) ? blocks : uint64_t(kRaggedBins - 1));   // bin 0 = the longest
}
// (a kernel, not hipMemsetAsync: captured into a hipGraph, the memset node in front of these kernels did not zero the
// counters on replay -- the second replay then placed its indices behind the first one's and wrote past the scratch)
__global__ void k_ragged_zero(uint32_t* __restrict__ head) {   // status word, padding and counters
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < size_t(kRaggedHead + kRaggedBins)) head[i] = 0;
}
__global__ void k_ragged_hist(const uint64_t* __restrict__ off, size_t n, uint64_t extent, uint32_t block_units,
                              uint32_t* __restrict__ status, uint32_t* __restrict__ bins) {
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t bad = off[i + 1] < off[i] ? uint32_t(ANEMOI_RAGGED_DECREASING) : 0u;
  if (i == n - 1 && off[n] > extent) bad |= uint32_t(ANEMOI_RAGGED_BEYOND_EXTENT);
  if (bad) atomicOr(status, bad);
  atomicAdd(&bins[ragged_bin(off, i, block_units)], 1u);
}
// counts -> first slot of each bin (one workgroup of 1 024 threads, 64 bins each)
__global__ __launch_bounds__(1024) void k_ragged_scan(uint32_t* __restrict__ bins) {
  __shared__ uint32_t part[1024];
  constexpr int PER = kRaggedBins / 1024;
  uint32_t sum = 0;
  for (int j = 0; j < PER; j++) sum += bins[threadIdx.x * PER + j];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {   // inclusive scan of the 1 024 partial sums
    const uint32_t v = threadIdx.x >= unsigned(d) ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
  for (int j = 0; j < PER; j++) {
    const uint32_t c = bins[threadIdx.x * PER + j];
    bins[threadIdx.x * PER + j] = run;
    run += c;
  }
}
__global__ void k_ragged_place(const uint64_t* __restrict__ off, size_t n, uint32_t block_units, uint32_t* __restrict__ bins,
                               uint32_t* __restrict__ order) {
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) order[atomicAdd(&bins[ragged_bin(off, i, block_units)], 1u)] = uint32_t(i);
}

// The do-nothing launch in front of a launch that does not fill the chip (anemoi_kernels.h: balance_before): single-wavefront
// workgroups asleep for ~20 us, which all retire together.
__global__ void k_balance(unsigned sleeps) {
  for (unsigned i = 0; i < sleeps; i++) __builtin_amdgcn_s_sleep(127);   // 127 x 64 clocks ~ 3.4 us
}
}  // namespace
namespace anemoi {
void balance_launch(unsigned wgs, hipStream_t s) { k_balance<<<wgs, 64, 0, s>>>(6); }
}  // namespace anemoi
namespace {

// anemoi_probe_issue_rate: every lane runs ONE dependent chain of v_mad_u64_u32 -- the instruction that carries the
// throughput kernels (75 % of their instructions) -- with the register footprint of those kernels (161 VGPRs claimed:
// three wavefronts per SIMD, launched as exactly three per SIMD), and stamps the shader clock (s_memtime) and the
// 100 MHz wall clock (s_memrealtime) around the loop.
__global__ __launch_bounds__(64) void k_issue_probe(uint64_t* __restrict__ rec, int iters) {
  asm volatile("v_mov_b32 v160, 0" ::: "v160");
  uint64_t acc = threadIdx.x + 1;
  const uint32_t a = 0x9e3779b9u ^ threadIdx.x, b = blockIdx.x | 1u;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int i = 0; i < iters; i++) {
#define ANEMOI_PROBE_MAD4 "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
#define ANEMOI_PROBE_MAD16 ANEMOI_PROBE_MAD4 ANEMOI_PROBE_MAD4 ANEMOI_PROBE_MAD4 ANEMOI_PROBE_MAD4
    asm volatile(ANEMOI_PROBE_MAD16 ANEMOI_PROBE_MAD16 ANEMOI_PROBE_MAD16 ANEMOI_PROBE_MAD16 : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    rec[3 * blockIdx.x] = t1 - t0;
    rec[3 * blockIdx.x + 1] = c1 - c0;
  }
  if (acc == 0x5a5a5a5a5a5a5a5aull) rec[3 * blockIdx.x + 2] = acc;   // keeps the chain alive; never true in practice
}
constexpr int kProbeMadsPerIter = 64;

// ... and the same grid running what the throughput kernels are MADE of: a chain of the generated BLS12-381 squaring
// (mont29_asm_gen.h: 260 multiply-adds among 345 instructions, the mix and the power draw of the headline kernel, which
// is 80 % squarings).  The chip holds a different clock under this mix than under bare multiply-adds, and boxes differ in
// how much: this is the probe a kernel's rate should be read against.
__global__ __launch_bounds__(64) void k_issue_probe_sqr(uint64_t* __restrict__ rec, int iters) {
  asm volatile("v_mov_b32 v160, 0" ::: "v160");
  using L = anemoi::FieldC<0>::R30;
  uint32_t a[L::NL];
#pragma unroll
  for (int i = 0; i < L::NL; i++) a[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) & ((1u << 30) - 1);
  a[L::NL - 1] &= 0xffff;   // below p: every later value is a squaring's result, below 2p
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int i = 0; i < iters; i++) anemoi::AsmMont<0, 30>::sqr(a);
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    rec[3 * blockIdx.x] = t1 - t0;
    rec[3 * blockIdx.x + 1] = c1 - c0;
  }
  if (a[0] == 0x7fffffffu) rec[3 * blockIdx.x + 2] = a[1];   // keeps the chain alive; a limb is below 2^30
}
constexpr int kProbeMadsPerSquaring = 260;   // 13 x 14 / 2 products + 13 x 13 reduction (tools/gen_asm_mul.py; buildinfo.py)

// ---- the shader clock WHILE other work runs (anemoi_clock_sampler_*) ---------------------------------------------------
// kSamplerGroups single-wavefront workgroups (consecutive workgroups land on consecutive XCDs) sit beside whatever the
// caller launches afterwards -- one wave slot and a handful of registers each, asleep between samples -- and log
// (s_memrealtime, s_memtime) every `period`: the 100 MHz wall clock and the shader-cycle counter.  Between two wall-clock
// stamps of the caller's own stream (anemoi_clock_stamp_dev) that is the clock the chip HELD under the caller's kernels,
// per XCD: the one quantity that differs from box to box under this load (profiles/r05/bench_probe_across_boxes.txt)
// and that a process cannot otherwise read.  Every workgroup ends when the stop flag is set, when its log is full, or
// after max_ms -- whichever comes first.
constexpr int kSamplerGroups = 16, kSamplerMaxRecords = 4096;
struct SamplerGroup {
  uint32_t count, xcc;
  uint64_t rec[kSamplerMaxRecords][2];
};
struct SamplerBuf {
  uint32_t stop, groups;
  uint64_t reserved;
  SamplerGroup g[kSamplerGroups];
};
// (kernels, not hipMemsetAsync, set the log up and stop it: fewer operations in front of the sampler, and see k_ragged_zero)
__global__ __launch_bounds__(64) void k_sampler_init(SamplerBuf* buf) {
  if (threadIdx.x == 0) buf->stop = 0, buf->groups = kSamplerGroups, buf->reserved = 0;   // reserved: 1 once the sampler runs
  if (threadIdx.x < kSamplerGroups) buf->g[threadIdx.x].count = 0, buf->g[threadIdx.x].xcc = 0;
}
__global__ void k_sampler_stop(SamplerBuf* buf) { __hip_atomic_store(&buf->stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// holds the CALLER's stream until the sampler has taken its first sample (or timeout): without it a sampler whose
// stream is slow to start -- a queue created on first use -- can begin after a short piece of work has already ended
__global__ void k_sampler_wait(SamplerBuf* buf, uint64_t max_ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(&buf->reserved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 &&
         __builtin_amdgcn_s_memrealtime() - t0 < max_ticks)
    __builtin_amdgcn_s_sleep(32);
}
__global__ __launch_bounds__(64) void k_clock_sampler(SamplerBuf* buf, uint32_t period_ticks, uint64_t max_ticks) {
  if (threadIdx.x) return;
  SamplerGroup& g = buf->g[blockIdx.x];
  g.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7u;   // XCC_ID
  const uint64_t t_begin = __builtin_amdgcn_s_memrealtime();
  uint64_t next = t_begin;
  uint32_t n = 0;
  for (uint32_t it = 0;; it++) {
    const uint64_t t = __builtin_amdgcn_s_memrealtime();
    // the stop flag is read past the caches, so only every 16th turn (~45 us): polled on every turn, sixteen workgroups
    // fetch ~35 MB per 100 ms -- a quarter of the headline kernel's own traffic, and it lands in that kernel's FETCH_SIZE
    bool last = t - t_begin > max_ticks;
    if ((it & 15u) == 0) last = last || __hip_atomic_load(&buf->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
    if (t >= next || last) {
      g.rec[n][0] = __builtin_amdgcn_s_memrealtime();
      g.rec[n][1] = __builtin_amdgcn_s_memtime();
      n++;
      next += period_ticks;
      if (n == 1 && blockIdx.x == 0) __hip_atomic_store(&buf->reserved, uint64_t(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (last || n >= uint32_t(kSamplerMaxRecords)) break;
    __builtin_amdgcn_s_sleep(100);   // ~6 400 cycles: the sampler takes an issue slot every few microseconds
  }
  g.count = n;
}
__global__ void k_clock_stamp(uint64_t* out) { *out = __builtin_amdgcn_s_memrealtime(); }

// Root / retained tree over arity^depth host leaves, on one device or sharded into subtrees over all
// devices (the only cross-GPU data: one subtree root per part, finished on the first device).
int merkle_host(TreeShape ts, const uint64_t* leaves, unsigned depth, uint64_t* root, uint64_t* tree, int device) {
  const size_t eb = elem_bytes(ts.field);
  int ndev = 0, rc = rt::physical_devices(&ndev);
  if (rc) return rc;
  if (device != ANEMOI_ALL_DEVICES && (device < 0 || device >= ndev)) {
    g_last_error = "device ordinal out of range";
    return ANEMOI_ERR_DEVICE;
  }
  // (level 0 of a retained tree -- the leaves themselves -- is copied by subtree_host, under the GPU's work)
  const unsigned sub_lv = device == ANEMOI_ALL_DEVICES ? host::subtree_levels(depth, ts.alog, size_t(rt::shard_parts(ndev))) : 0;
  std::vector<uint64_t> root_buf(eb / 8);
  if (sub_lv == 0) {
    const int dev = device == ANEMOI_ALL_DEVICES ? 0 : device;
    rc = with_lane(dev, [&](Lane& ln) {
      return subtree_host(ln, ts, (const char*)leaves, depth, (char*)root_buf.data(), (char*)tree, depth, 0);
    });
    if (rc) return rc;
    if (root) memcpy(root, root_buf.data(), eb);
    return ANEMOI_OK;
  }
  const size_t nsub = size_t(1) << (ts.alog * sub_lv);
  const unsigned sub_depth = depth - sub_lv;
  const size_t sub_leaves = size_t(1) << (ts.alog * sub_depth);
  std::vector<uint64_t> tops(nsub * (eb / 8));
  rc = rt::run_parts(int(nsub), ndev, nsub, [&](int part, int dev, size_t, size_t) -> int {
    return with_lane(dev, [&](Lane& ln) {
      return subtree_host(ln, ts, (const char*)leaves + size_t(part) * sub_leaves * eb, sub_depth,
                          (char*)tops.data() + size_t(part) * eb, (char*)tree, depth, size_t(part));
    });
  });
  if (rc) return rc;
  // the top sub_lv levels over the nsub subtree roots, on the first device.  In a retained tree the
  // subtree roots are level sub_depth, and levels sub_depth .. depth form exactly the layout of a tree
  // over nsub leaves, so the top is built in place.
  if (tree) {
    const size_t off = ts.alog == 1 ? host::tree2_level_offset(depth, sub_depth) : host::tree4_level_offset(depth, sub_depth);
    uint64_t* top = tree + off * (eb / 8);
    memcpy(top, tops.data(), nsub * eb);  // (already there for sub_depth >= 1: level sub_depth was copied out)
    return merkle_host(ts, top, sub_lv, root, top, 0);
  }
  return merkle_host(ts, tops.data(), sub_lv, root, nullptr, 0);
}

}  // namespace

extern "C" {

int anemoi_abi_version(void) { return 100 * ANEMOI_ABI_MAJOR + ANEMOI_ABI_MINOR; }   // the rule: include/anemoi_mi355x.h

int anemoi_device_count(void) {
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  return n;
}

const char* anemoi_strerror(int code) {
  switch (code) {
    case ANEMOI_OK: return "ok";
    case ANEMOI_ERR_FIELD: return "unknown field id";
    case ANEMOI_ERR_WIDTH: return "state width must be 2 or 4";
    case ANEMOI_ERR_ARG: return "invalid argument (null pointer, unsupported k, overlapping buffers, or size out of range)";
    case ANEMOI_ERR_DEVICE: return "HIP device error";
    case ANEMOI_ERR_ALLOC: return "allocation failed";
    default: return "unknown error code";
  }
}

const char* anemoi_last_error(void) { return g_last_error.c_str(); }

int anemoi_field_id(const char* name) {
  if (!name) return ANEMOI_ERR_ARG;
  for (int f = 0; f < anemoi::kNumFields; f++)
    if (!strcmp(name, anemoi::field_ops(f)->name)) return f;
  return ANEMOI_ERR_FIELD;
}

const char* anemoi_field_name(int field) {
  const FieldOps* o = anemoi::field_ops(field);
  return o ? o->name : nullptr;
}

int anemoi_field_limbs(int field) {
  const FieldOps* o = anemoi::field_ops(field);
  return o ? o->limbs64 : ANEMOI_ERR_FIELD;
}

int anemoi_field_chunk_bytes(int field) {
  const FieldOps* o = anemoi::field_ops(field);
  return o ? o->chunk : ANEMOI_ERR_FIELD;
}

int anemoi_num_rounds(int field, int width) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  const FieldOps* o = anemoi::field_ops(field);
  return width == 2 ? o->rounds21 : o->rounds43;
}

/* ---- lifecycle ---- */

int anemoi_init(int device, int field, int width) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  int ndev = 0;
  if ((rc = rt::physical_devices(&ndev))) return rc;
  const int lo = device == ANEMOI_ALL_DEVICES ? 0 : device, hi = device == ANEMOI_ALL_DEVICES ? ndev : device + 1;
  if (lo < 0 || hi > ndev) {
    g_last_error = "device ordinal out of range";
    return ANEMOI_ERR_DEVICE;
  }
  DeviceGuard guard;
  for (int d = lo; d < hi; d++) {
    HIP_TRY(hipSetDevice(d));
    PermConsts pc;
    if ((rc = get_consts(field, width, &pc))) return rc;
    Lane* ln = nullptr;  // one warm lane: streams and events exist before the first real call
    if ((rc = rt::acquire_lane(d, &ln))) return rc;
    rt::release_lane(ln);
  }
  return ANEMOI_OK;
}

int anemoi_warmup(int device, int field, int width) {
  int rc = anemoi_init(device, field, width);
  if (rc) return rc;
  int ndev = 0;
  if ((rc = rt::physical_devices(&ndev))) return rc;
  const int lo = device == ANEMOI_ALL_DEVICES ? 0 : device, hi = device == ANEMOI_ALL_DEVICES ? ndev : device + 1;
  const anemoi::FieldOps* ops = anemoi::field_ops(field);
  for (int d = lo; d < hi; d++) {
    rc = with_lane(d, [&](Lane& ln) -> int {
      PermConsts pc;
      int r = get_consts(field, width, &pc);
      if (r) return r;
      const size_t n = size_t(16) * size_t(pc.simds), bytes = n * size_t(width) * elem_bytes(field);
      if ((r = ln.slot[0].d_in.reserve(bytes)) || (r = ln.slot[0].d_out.reserve(bytes))) return r;
      HIP_TRY(hipMemsetAsync(ln.slot[0].d_in.p, 0, bytes, ln.s_k));   // zeros are canonical elements and valid messages
      HIP_TRY(ops->warmup(width, ln.slot[0].d_in.p, ln.slot[0].d_out.p, n, pc, ln.s_k));
      HIP_TRY(hipStreamSynchronize(ln.s_k));
      return ANEMOI_OK;
    });
    if (rc) return rc;
  }
  return ANEMOI_OK;
}

int anemoi_probe_issue_rate(int device, double* lane_mad_per_s, double* shader_clock_ghz, double* sqr_lane_mad_per_s,
                            double* sqr_shader_clock_ghz) {
  if (!lane_mad_per_s || !shader_clock_ghz || !sqr_lane_mad_per_s || !sqr_shader_clock_ghz) return ANEMOI_ERR_ARG;
  int ndev = 0, rc = rt::physical_devices(&ndev);
  if (rc) return rc;
  if (device < 0 || device >= ndev) {
    g_last_error = "device ordinal out of range";
    return ANEMOI_ERR_DEVICE;
  }
  return with_lane(device, [&](Lane& ln) -> int {
    const int simds = 4 * rt::device_cus(device), grid = 3 * simds;
    int r = ln.slot[0].d_out.reserve(size_t(grid) * 3 * sizeof(uint64_t));
    if (r) return r;
    uint64_t* rec = (uint64_t*)ln.slot[0].d_out.p;
    // one probe: an untimed launch first (the clock the timed launch sees is a loaded one), then the timed one
    auto run = [&](bool squarings, int iters, int mads_per_iter, double* rate, double* clock) -> int {
      hipEvent_t a = nullptr, b = nullptr;
      HIP_TRY(hipEventCreate(&a));
      if (hipError_t eb = hipEventCreate(&b); eb != hipSuccess) {
        (void)hipEventDestroy(a);
        HIP_TRY(eb);
      }
      if (squarings) k_issue_probe_sqr<<<grid, 64, 0, ln.s_k>>>(rec, iters / 4);
      else k_issue_probe<<<grid, 64, 0, ln.s_k>>>(rec, iters / 8);
      (void)hipEventRecord(a, ln.s_k);
      if (squarings) k_issue_probe_sqr<<<grid, 64, 0, ln.s_k>>>(rec, iters);
      else k_issue_probe<<<grid, 64, 0, ln.s_k>>>(rec, iters);
      (void)hipEventRecord(b, ln.s_k);
      hipError_t e = hipEventSynchronize(b);
      float ms = 0;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
      std::vector<uint64_t> h(size_t(grid) * 3);
      if (e == hipSuccess) e = hipMemcpy(h.data(), rec, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
      HIP_TRY(e);
      std::vector<double> ghz;
      for (int i = 0; i < grid; i++)
        if (h[3 * i]) ghz.push_back(double(h[3 * i + 1]) / double(h[3 * i]) * 0.1);   // cycles per 10 ns tick
      std::sort(ghz.begin(), ghz.end());
      *clock = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
      *rate = ms > 0 ? double(grid) * 64.0 * double(iters) * double(mads_per_iter) / (double(ms) * 1e-3) : 0.0;
      return ANEMOI_OK;
    };
    if ((r = run(false, 60000, kProbeMadsPerIter, lane_mad_per_s, shader_clock_ghz))) return r;   // ~20 ms
    // ~200 ms (after ~50 ms untimed): under this mix the chip's clock settles over tens of milliseconds, and a 20 ms
    // sample of it wanders by +-1.5 % from run to run on one box (profiles/r05/bench_probe_across_boxes.txt)
    return run(true, 110000, kProbeMadsPerSquaring, sqr_lane_mad_per_s, sqr_shader_clock_ghz);
  });
}

size_t anemoi_clock_sampler_bytes(void) { return sizeof(SamplerBuf); }

int anemoi_clock_sampler_start_dev(void* d_buf, size_t bytes, unsigned period_us, unsigned max_ms, void* stream) {
  if (!d_buf || bytes < sizeof(SamplerBuf) || (uintptr_t)d_buf % 8 || period_us < 10 || period_us > 1000000 ||
      max_ms < 1 || max_ms > 600000)
    return ANEMOI_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  k_sampler_init<<<1, 64, 0, s>>>((SamplerBuf*)d_buf);                                // stop = 0, every group's count = 0
  k_clock_sampler<<<kSamplerGroups, 64, 0, s>>>((SamplerBuf*)d_buf, period_us * 100u, uint64_t(max_ms) * 100000ull);
  HIP_TRY(hipGetLastError());
  return ANEMOI_OK;
}

int anemoi_clock_sampler_wait_dev(void* d_buf, unsigned timeout_ms, void* stream) {
  if (!d_buf || timeout_ms < 1 || timeout_ms > 10000) return ANEMOI_ERR_ARG;
  k_sampler_wait<<<1, 1, 0, (hipStream_t)stream>>>((SamplerBuf*)d_buf, uint64_t(timeout_ms) * 100000ull);
  HIP_TRY(hipGetLastError());
  return ANEMOI_OK;
}

int anemoi_clock_sampler_stop_dev(void* d_buf, void* stream) {
  if (!d_buf) return ANEMOI_ERR_ARG;
  k_sampler_stop<<<1, 1, 0, (hipStream_t)stream>>>((SamplerBuf*)d_buf);
  HIP_TRY(hipGetLastError());
  return ANEMOI_OK;
}

int anemoi_clock_stamp_dev(void* d_u64, void* stream) {
  if (!d_u64 || (uintptr_t)d_u64 % 8) return ANEMOI_ERR_ARG;
  k_clock_stamp<<<1, 1, 0, (hipStream_t)stream>>>((uint64_t*)d_u64);
  HIP_TRY(hipGetLastError());
  return ANEMOI_OK;
}

int anemoi_clock_sampler_read(const void* h_buf, size_t bytes, unsigned long long t0, unsigned long long t1, double* ghz_mean,
                              double* ghz_min, double* ghz_max, int* groups_used) {
  if (!h_buf || bytes < sizeof(SamplerBuf) || !ghz_mean || !ghz_min || !ghz_max || !groups_used || t1 <= t0) return ANEMOI_ERR_ARG;
  const SamplerBuf* b = (const SamplerBuf*)h_buf;
  double sum = 0, lo = 1e30, hi = 0;
  int used = 0;
  for (int i = 0; i < kSamplerGroups; i++) {
    const SamplerGroup& g = b->g[i];
    if (g.count < 2 || g.count > uint32_t(kSamplerMaxRecords)) continue;
    // the samples inside [t0, t1]; the clock of every interval between two of them; the mean of the middle 80 % of the
    // intervals (the cycle counter of an XCD occasionally jumps: one sampler workgroup then reads 8 % high and its
    // neighbour 8 % low over half a second -- profiles/r05/clock_sampler_across_boxes.txt -- and a jump spoils ONE interval)
    std::vector<double> iv;
    for (uint32_t k = 1; k < g.count; k++) {
      if (g.rec[k - 1][0] < t0 || g.rec[k][0] > t1 || g.rec[k][0] <= g.rec[k - 1][0]) continue;
      iv.push_back(double(g.rec[k][1] - g.rec[k - 1][1]) / double(g.rec[k][0] - g.rec[k - 1][0]) * 0.1);   // cycles per 10 ns tick
    }
    if (iv.size() < 5) continue;
    std::sort(iv.begin(), iv.end());
    const size_t cut = iv.size() / 10;
    double acc = 0;
    for (size_t k = cut; k < iv.size() - cut; k++) acc += iv[k];
    const double ghz = acc / double(iv.size() - 2 * cut);
    sum += ghz, lo = ghz < lo ? ghz : lo, hi = ghz > hi ? ghz : hi;
    used++;
  }
  *groups_used = used;
  if (!used) {
    *ghz_mean = *ghz_min = *ghz_max = 0.0;
    return ANEMOI_OK;
  }
  *ghz_mean = sum / used, *ghz_min = lo, *ghz_max = hi;
  return ANEMOI_OK;
}

int anemoi_release(int device) {
  int ndev = 0, rc = rt::physical_devices(&ndev);
  if (rc) return rc;
  const int lo = device == ANEMOI_ALL_DEVICES ? 0 : device, hi = device == ANEMOI_ALL_DEVICES ? ndev : device + 1;
  if (lo < 0 || hi > ndev) {
    g_last_error = "device ordinal out of range";
    return ANEMOI_ERR_DEVICE;
  }
  for (int d = lo; d < hi; d++)
    if ((rc = rt::release_device(d))) return rc;
  return ANEMOI_OK;
}

/* ---- options (options.h) ---- */

int anemoi_set_option(const char* name, long long value) {
  const int id = anemoi::opt::find(name);
  if (id < 0 || !anemoi::opt::set(id, value)) return ANEMOI_ERR_ARG;
  return ANEMOI_OK;
}

int anemoi_get_option(const char* name, long long* value) {
  const int id = anemoi::opt::find(name);
  if (id < 0 || !value) return ANEMOI_ERR_ARG;
  *value = anemoi::opt::get(anemoi::opt::Id(id));
  // an environment value rejected at start-up is reported here; otherwise the thread's last error text is left alone
  if (!anemoi::opt::env_error().empty()) g_last_error = anemoi::opt::env_error();
  return ANEMOI_OK;
}

/* ---- run-time instances ---- */

int anemoi_generic_mds_matrix(int field, int num_columns, uint64_t* mds, int device) {
  const FieldOps* ops = anemoi::field_ops(field);
  if (!ops) return ANEMOI_ERR_FIELD;
  std::vector<uint64_t> small;
  if (!mds || !host::builtin_mds(num_columns, uint64_t(ops->generator), &small)) return ANEMOI_ERR_ARG;
  const size_t L = size_t(ops->limbs64), cnt = small.size();
  std::vector<uint64_t> canon(cnt * L, 0);
  for (size_t i = 0; i < cnt; i++) canon[i * L] = small[i];
  return anemoi_to_montgomery(field, canon.data(), mds, cnt, device == ANEMOI_ALL_DEVICES ? 0 : device);
}

int anemoi_generic_permutation_batch(const anemoi_generic_instance* inst, uint64_t* states, size_t n, int device) {
  int rc = check_generic(inst);
  if (rc) return rc;
  if (n && !states) return ANEMOI_ERR_ARG;
  const size_t per = elem_bytes(inst->field) * 2 * size_t(inst->num_columns);
  return generic_batch(inst, device, n, states, per, states, per,
                       [&](void* i, void*, size_t cnt, anemoi::GenericConsts gc, PermConsts pc, hipStream_t s) {
                         return anemoi::field_ops(inst->field)->generic_permutation(i, cnt, gc, pc, s);
                       });
}

int anemoi_generic_jive_compress_k_batch(const anemoi_generic_instance* inst, int k, const uint64_t* in,
                                         uint64_t* out, size_t n, int device) {
  int rc = check_generic(inst);
  if (rc) return rc;
  const int w = 2 * inst->num_columns;
  // the reference's asserts (anemoi_4_3/hasher.rs:163-165): k <= width, k | width, k even
  if (!host::valid_generic_k(w, k)) return ANEMOI_ERR_ARG;
  if (n && (!in || !out)) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(inst->field);
  if (host::ranges_overlap(in, n * eb * w, out, n * eb * (w / k))) return ANEMOI_ERR_ARG;
  return generic_batch(inst, device, n, in, eb * w, out, eb * (w / k),
                       [&](void* i, void* o, size_t cnt, anemoi::GenericConsts gc, PermConsts pc, hipStream_t s) {
                         return anemoi::field_ops(inst->field)->generic_jive(i, o, cnt, k, gc, pc, s);
                       });
}

static int generic_hash(const anemoi_generic_instance* inst, int rate, int bytes, const void* src, size_t per_msg,
                        size_t n, uint64_t* out, int device) {
  int rc = check_generic(inst);
  if (rc) return rc;
  if (rate < 1 || rate >= 2 * inst->num_columns) return ANEMOI_ERR_ARG;
  if (n && (!out || (per_msg && !src))) return ANEMOI_ERR_ARG;
  static const uint64_t dummy[2] = {0, 0};
  const size_t eb = elem_bytes(inst->field);
  return generic_batch(inst, device, n, src ? src : (const void*)dummy, bytes ? per_msg : eb * per_msg, out, eb,
                       [&](void* i, void* o, size_t cnt, anemoi::GenericConsts gc, PermConsts pc, hipStream_t s) {
                         return anemoi::field_ops(inst->field)->generic_sponge(bytes, i, per_msg, cnt, o, rate, gc, pc, s);
                       });
}

int anemoi_generic_hash_field_batch(const anemoi_generic_instance* inst, int rate, const uint64_t* elems,
                                    size_t elems_per_msg, size_t n, uint64_t* out, int device) {
  return generic_hash(inst, rate, 0, elems, elems_per_msg, n, out, device);
}

int anemoi_generic_hash_bytes_batch(const anemoi_generic_instance* inst, int rate, const uint8_t* msgs, size_t msg_len,
                                    size_t n, uint64_t* out, int device) {
  return generic_hash(inst, rate, 1, msgs, msg_len, n, out, device);
}

/* ---- run-time instances, device-pointer form ---- */

struct anemoi_generic_handle {
  uint32_t magic;
  int field, cols, rounds, device;
  void* blob;
  anemoi::GenericConsts gc;
};
static constexpr uint32_t kHandleMagic = 0xA9E30C01u;

int anemoi_generic_prepare(const anemoi_generic_instance* inst, int device, anemoi_generic_handle** handle) {
  if (!handle) return ANEMOI_ERR_ARG;
  *handle = nullptr;
  int rc = check_generic(inst);
  if (rc) return rc;
  int ndev = 0;
  if ((rc = rt::physical_devices(&ndev))) return rc;
  if (device < 0 || device >= ndev) {
    g_last_error = "device ordinal out of range";
    return ANEMOI_ERR_DEVICE;
  }
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(device));
  PermConsts pc;
  if ((rc = get_consts(inst->field, 2, &pc))) return rc;  // the field's exponent schedule, uploaded ahead of time
  void* blob = nullptr;
  if (hipMalloc(&blob, generic_layout(inst).total) != hipSuccess) {
    (void)hipGetLastError();
    g_last_error = "hipMalloc(constants of a run-time instance)";
    return ANEMOI_ERR_ALLOC;
  }
  anemoi::GenericConsts gc;
  rc = with_lane(device, [&](Lane& ln) { return build_generic_consts(inst, (char*)blob, ln.s_k, &gc); });
  if (rc) {
    (void)hipFree(blob);
    return rc;
  }
  *handle = new anemoi_generic_handle{kHandleMagic, inst->field, inst->num_columns, inst->num_rounds, device, blob, gc};
  return ANEMOI_OK;
}

int anemoi_generic_destroy(anemoi_generic_handle* h) {
  if (!h) return ANEMOI_OK;
  if (h->magic != kHandleMagic) return ANEMOI_ERR_ARG;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(h->device));
  (void)hipFree(h->blob);
  h->magic = 0;
  delete h;
  return ANEMOI_OK;
}

// the handle is valid, the current device is its device; fetches the field's exponent schedule
static int check_handle(const anemoi_generic_handle* h, PermConsts* pc) {
  if (!h || h->magic != kHandleMagic) return ANEMOI_ERR_ARG;
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != h->device) {
    g_last_error = "the current device is not the device the handle was prepared on";
    return ANEMOI_ERR_DEVICE;
  }
  return get_consts(h->field, 2, pc);
}

int anemoi_generic_permutation_dev(const anemoi_generic_handle* h, void* d_states, size_t n, void* stream) {
  PermConsts pc;
  int rc = check_handle(h, &pc);
  if (rc) return rc;
  if (n && !d_states) return ANEMOI_ERR_ARG;
  HIP_TRY(anemoi::field_ops(h->field)->generic_permutation(d_states, n, h->gc, pc, (hipStream_t)stream));
  return ANEMOI_OK;
}

int anemoi_generic_jive_compress_k_dev(const anemoi_generic_handle* h, int k, const void* d_in, void* d_out, size_t n,
                                       void* stream) {
  PermConsts pc;
  int rc = check_handle(h, &pc);
  if (rc) return rc;
  const int w = 2 * h->cols;
  if (!host::valid_generic_k(w, k)) return ANEMOI_ERR_ARG;
  if (n && (!d_in || !d_out)) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(h->field);
  if (host::ranges_overlap(d_in, n * eb * w, d_out, n * eb * (w / k))) return ANEMOI_ERR_ARG;
  HIP_TRY(anemoi::field_ops(h->field)->generic_jive(d_in, d_out, n, k, h->gc, pc, (hipStream_t)stream));
  return ANEMOI_OK;
}

static int generic_hash_dev(const anemoi_generic_handle* h, int rate, int bytes, const void* d_src, size_t per_msg,
                            size_t n, void* d_out, void* stream) {
  PermConsts pc;
  int rc = check_handle(h, &pc);
  if (rc) return rc;
  if (rate < 1 || rate >= 2 * h->cols) return ANEMOI_ERR_ARG;
  if (n && (!d_out || (per_msg && !d_src))) return ANEMOI_ERR_ARG;
  HIP_TRY(anemoi::field_ops(h->field)->generic_sponge(bytes, d_src, per_msg, n, d_out, rate, h->gc, pc,
                                                      (hipStream_t)stream));
  return ANEMOI_OK;
}

int anemoi_generic_hash_field_dev(const anemoi_generic_handle* h, int rate, const void* d_elems, size_t elems_per_msg,
                                  size_t n, void* d_out, void* stream) {
  return generic_hash_dev(h, rate, 0, d_elems, elems_per_msg, n, d_out, stream);
}

int anemoi_generic_hash_bytes_dev(const anemoi_generic_handle* h, int rate, const void* d_msgs, size_t msg_len, size_t n,
                                  void* d_out, void* stream) {
  return generic_hash_dev(h, rate, 1, d_msgs, msg_len, n, d_out, stream);
}

int anemoi_exp_alpha_batch(int field, int inverse, uint64_t* elems, size_t n, int device) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (n && !elems) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  return rt::host_batch(
      device, n, elems, eb, elems, eb, [&](int dev) { return quantum_of(field, anemoi::kKindExpAlpha, 2, dev); },
      [&](void* i, void*, size_t cnt, hipStream_t s) -> int {
        PermConsts pc;
        int rc = get_consts(field, 2, &pc);
        if (rc) return rc;
        HIP_TRY(anemoi::field_ops(field)->exp_alpha(inverse ? 1 : 0, i, cnt, pc, s));
        return ANEMOI_OK;
      });
}

/* ---- device-pointer API ---- */

int anemoi_permutation_dev(int field, int width, void* d_states, size_t n, void* stream) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && !d_states) return ANEMOI_ERR_ARG;
  PermConsts pc;
  if ((rc = get_consts(field, width, &pc))) return rc;
  HIP_TRY(anemoi::field_ops(field)->permutation(width, 0, d_states, n, pc, (hipStream_t)stream));
  return ANEMOI_OK;
}

int anemoi_sbox_layer_dev(int field, int width, void* d_states, size_t n, void* stream) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && !d_states) return ANEMOI_ERR_ARG;
  PermConsts pc;
  if ((rc = get_consts(field, width, &pc))) return rc;
  HIP_TRY(anemoi::field_ops(field)->permutation(width, 1, d_states, n, pc, (hipStream_t)stream));
  return ANEMOI_OK;
}

int anemoi_jive_compress_k_dev(int field, int width, int k, const void* d_in, void* d_out, size_t n, void* stream) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if ((rc = check_k(width, k))) return rc;
  if (n && (!d_in || !d_out)) return ANEMOI_ERR_ARG;
  // the kernels read and write through __restrict__ pointers and a workgroup's outputs land where another
  // workgroup's inputs are: overlapping buffers are a cross-workgroup race, so they are rejected
  const size_t eb = elem_bytes(field);
  if (host::ranges_overlap(d_in, n * width * eb, d_out, n * (width / k) * eb)) return ANEMOI_ERR_ARG;
  PermConsts pc;
  if ((rc = get_consts(field, width, &pc))) return rc;
  HIP_TRY(anemoi::field_ops(field)->jive(width, k, d_in, d_out, n, pc, (hipStream_t)stream));
  return ANEMOI_OK;
}

#if ANEMOI_AB_BUILD
// laboratory builds only (not in the header): Jive 2-1 with a work queue; d_queue = one uint32 owned by the caller
static int x_jive_queue(int field, const void* d_in, void* d_out, size_t n, void* d_queue, unsigned wgs, void* d_acct, void* stream) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  PermConsts pc;
  if ((rc = get_consts(field, 2, &pc))) return rc;
  if (wgs) HIP_TRY(hipMemsetAsync(d_queue, 0, 4, (hipStream_t)stream));
  HIP_TRY(anemoi::field_ops(field)->jive_queue(d_in, d_out, n, pc, (uint32_t*)d_queue, wgs, d_acct, (hipStream_t)stream));
  return ANEMOI_OK;
}
int anemoi_x_jive_queue_dev(int field, const void* d_in, void* d_out, size_t n, void* d_queue, unsigned wgs, void* stream) {
  if (!wgs) return ANEMOI_ERR_ARG;
  return x_jive_queue(field, d_in, d_out, n, d_queue, wgs, nullptr, stream);
}
// EXPERIMENT (tools/exp_balance_launch.py): a launch that does nothing but occupy wave slots for `sleeps` x ~3.4 us, in front of
// an underfilled launch that follows a different kernel -- does it change where the dispatcher puts the next launch's waves?
__global__ void k_x_balance(unsigned sleeps) {
  for (unsigned i = 0; i < sleeps; i++) __builtin_amdgcn_s_sleep(127);
}
int anemoi_x_balance_dev(unsigned wgs, unsigned threads, unsigned sleeps, void* stream) {
  if (!wgs || !threads || threads > 1024) return ANEMOI_ERR_ARG;
  k_x_balance<<<wgs, threads, 0, (hipStream_t)stream>>>(sleeps);
  HIP_TRY(hipGetLastError());
  return ANEMOI_OK;
}
// ... and with the per-XCD accounting (anemoi_kernels.h: XcdAcct; d_acct = 8 records of 5 uint64, prepared by the caller):
// wgs = 0 is the SHIPPED static dealing, wgs < blocks the queue, wgs >= blocks the tickets
int anemoi_x_jive_acct_dev(int field, const void* d_in, void* d_out, size_t n, void* d_queue, unsigned wgs, void* d_acct, void* stream) {
  if (!d_acct) return ANEMOI_ERR_ARG;
  return x_jive_queue(field, d_in, d_out, n, d_queue, wgs, d_acct, stream);
}
#endif

int anemoi_hash_field_dev(int field, int width, const void* d_elems, size_t elems_per_msg, size_t n, void* d_out,
                          void* stream) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!d_out || (elems_per_msg && !d_elems))) return ANEMOI_ERR_ARG;
  PermConsts pc;
  if ((rc = get_consts(field, width, &pc))) return rc;
  HIP_TRY(anemoi::field_ops(field)->sponge(width, 0, d_elems, elems_per_msg, n, d_out, pc, (hipStream_t)stream));
  return ANEMOI_OK;
}

int anemoi_hash_bytes_dev(int field, int width, const void* d_msgs, size_t msg_len, size_t n, void* d_out,
                          void* stream) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!d_out || (msg_len && !d_msgs))) return ANEMOI_ERR_ARG;
  PermConsts pc;
  if ((rc = get_consts(field, width, &pc))) return rc;
  HIP_TRY(anemoi::field_ops(field)->sponge(width, 1, d_msgs, msg_len, n, d_out, pc, (hipStream_t)stream));
  return ANEMOI_OK;
}

// the ragged sponge on device-resident messages: bytes (offsets in bytes) or ABI elements (offsets in elements)
static int ragged_dev(int field, int width, int bytes, const void* d_msgs, const void* d_offsets, size_t n, void* d_out,
                      void* stream) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!d_out || !d_offsets || !d_msgs)) return ANEMOI_ERR_ARG;
  PermConsts pc;
  if ((rc = get_consts(field, width, &pc))) return rc;
  HIP_TRY(anemoi::field_ops(field)->sponge_ragged(width, bytes, d_msgs, d_offsets, n, d_out, pc, nullptr, nullptr, (hipStream_t)stream));
  return ANEMOI_OK;
}
int anemoi_hash_bytes_ragged_dev(int field, int width, const void* d_msgs, const void* d_offsets, size_t n, void* d_out,
                                 void* stream) {
  return ragged_dev(field, width, 1, d_msgs, d_offsets, n, d_out, stream);
}
int anemoi_hash_field_ragged_dev(int field, int width, const void* d_elems, const void* d_offsets, size_t n, void* d_out,
                                 void* stream) {
  return ragged_dev(field, width, 0, d_elems, d_offsets, n, d_out, stream);
}

size_t anemoi_ragged_scratch_bytes(size_t n) { return (size_t(kRaggedHead) + size_t(kRaggedBins) + n) * sizeof(uint32_t); }

static int ragged_bucketed_dev(int field, int width, int bytes, const void* d_msgs, size_t msgs_len, const void* d_offsets, size_t n,
                               void* d_out, void* d_scratch, size_t scratch_bytes, void* stream) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!d_out || !d_offsets || !d_msgs || !d_scratch)) return ANEMOI_ERR_ARG;
  if (n >= (size_t(1) << 32) || scratch_bytes < anemoi_ragged_scratch_bytes(n) || (uintptr_t)d_scratch % sizeof(uint32_t))
    return ANEMOI_ERR_ARG;   // (the scratch is an array of 32-bit counters and indices)
  if (!n) return ANEMOI_OK;
  PermConsts pc;
  if ((rc = get_consts(field, width, &pc))) return rc;
  hipStream_t s = (hipStream_t)stream;
  uint32_t* status = (uint32_t*)d_scratch;    // the status word (+ padding), kRaggedBins counters, then the order: n message indices
  uint32_t* bins = status + kRaggedHead;
  uint32_t* order = bins + kRaggedBins;
  // what one permutation absorbs, in the unit of the offsets: bytes, or elements
  const uint32_t block_units = uint32_t(width - 1) * (bytes ? uint32_t(anemoi::field_ops(field)->chunk) : 1u);
  const unsigned grid = unsigned((n + 255) / 256);
  k_ragged_zero<<<(kRaggedHead + kRaggedBins + 255) / 256, 256, 0, s>>>(status);
  k_ragged_hist<<<grid, 256, 0, s>>>((const uint64_t*)d_offsets, n, uint64_t(msgs_len), block_units, status, bins);
  k_ragged_scan<<<1, 1024, 0, s>>>(bins);
  k_ragged_place<<<grid, 256, 0, s>>>((const uint64_t*)d_offsets, n, block_units, bins, order);
  HIP_TRY(hipGetLastError());
  HIP_TRY(anemoi::field_ops(field)->sponge_ragged(width, bytes, d_msgs, d_offsets, n, d_out, pc, order, status, s));
  return ANEMOI_OK;
}
int anemoi_hash_bytes_ragged_bucketed_dev(int field, int width, const void* d_msgs, size_t msgs_len, const void* d_offsets, size_t n,
                                          void* d_out, void* d_scratch, size_t scratch_bytes, void* stream) {
  return ragged_bucketed_dev(field, width, 1, d_msgs, msgs_len, d_offsets, n, d_out, d_scratch, scratch_bytes, stream);
}
int anemoi_hash_field_ragged_bucketed_dev(int field, int width, const void* d_elems, size_t elems_len, const void* d_offsets, size_t n,
                                          void* d_out, void* d_scratch, size_t scratch_bytes, void* stream) {
  return ragged_bucketed_dev(field, width, 0, d_elems, elems_len, d_offsets, n, d_out, d_scratch, scratch_bytes, stream);
}

int anemoi_merkle_root_dev(int field, const void* d_leaves, unsigned depth, void* d_scratch, void* d_root,
                           void* stream) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (!d_leaves || !d_root || depth > 30 || (depth && !d_scratch)) return ANEMOI_ERR_ARG;
  return merkle_levels_dev(field, d_leaves, depth, d_scratch, d_root, (hipStream_t)stream);
}

/* levels retained: d_tree = level 0 (2^depth leaves, copied) | level 1 | ... | level depth (root) */
int anemoi_merkle_tree_dev(int field, const void* d_leaves, unsigned depth, void* d_tree, void* stream) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (!d_leaves || !d_tree || depth > 30) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  hipStream_t s = (hipStream_t)stream;
  if (d_tree != d_leaves)
    HIP_TRY(hipMemcpyAsync(d_tree, d_leaves, (size_t(1) << depth) * eb, hipMemcpyDeviceToDevice, s));
  PermConsts pc;
  if ((rc = get_consts(field, 2, &pc))) return rc;
  char* lvl = (char*)d_tree;
  for (unsigned l = 0; l < depth; l++) {
    const size_t n = size_t(1) << (depth - 1 - l);
    char* next = lvl + 2 * n * eb;
    HIP_TRY(anemoi::field_ops(field)->jive(2, 2, lvl, next, n, pc, s));
    lvl = next;
  }
  return ANEMOI_OK;
}

int anemoi_merkle_climb_dev(int field, const void* d_leaves, const void* d_index, const void* d_paths, unsigned depth,
                            size_t n, void* d_roots, void* stream) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (depth > 63 || (n && (!d_leaves || !d_index || !d_roots || (depth && !d_paths)))) return ANEMOI_ERR_ARG;
  PermConsts pc;
  if ((rc = get_consts(field, 2, &pc))) return rc;
  HIP_TRY(anemoi::field_ops(field)->merkle_climb(d_leaves, d_index, d_paths, depth, n, d_roots, pc, (hipStream_t)stream));
  return ANEMOI_OK;
}

int anemoi_to_montgomery_dev(int field, const void* d_in, void* d_out, size_t count, void* stream) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (count && (!d_in || !d_out)) return ANEMOI_ERR_ARG;
  HIP_TRY(anemoi::field_ops(field)->mont_convert(1, d_in, d_out, count, (hipStream_t)stream));
  return ANEMOI_OK;
}

int anemoi_from_montgomery_dev(int field, const void* d_in, void* d_out, size_t count, void* stream) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (count && (!d_in || !d_out)) return ANEMOI_ERR_ARG;
  HIP_TRY(anemoi::field_ops(field)->mont_convert(0, d_in, d_out, count, (hipStream_t)stream));
  return ANEMOI_OK;
}

/* ---- host-pointer API ---- */

int anemoi_permutation_batch(int field, int width, uint64_t* states, size_t n, int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && !states) return ANEMOI_ERR_ARG;
  const size_t per = elem_bytes(field) * width;
  return rt::host_batch(
      device, n, states, per, states, per, [&](int dev) { return quantum_of(field, anemoi::kKindPermutation, width, dev); },
      [&](void* in, void*, size_t cnt, hipStream_t s) { return anemoi_permutation_dev(field, width, in, cnt, s); });
}

int anemoi_sbox_layer_batch(int field, int width, uint64_t* states, size_t n, int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && !states) return ANEMOI_ERR_ARG;
  const size_t per = elem_bytes(field) * width;
  return rt::host_batch(
      device, n, states, per, states, per, [&](int dev) { return quantum_of(field, anemoi::kKindPermutation, width, dev); },
      [&](void* in, void*, size_t cnt, hipStream_t s) { return anemoi_sbox_layer_dev(field, width, in, cnt, s); });
}

int anemoi_jive_compress_k_batch(int field, int width, int k, const uint64_t* in, uint64_t* out, size_t n,
                                 int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if ((rc = check_k(width, k))) return rc;
  if (n && (!in || !out)) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  if (host::ranges_overlap(in, n * width * eb, out, n * (width / k) * eb)) return ANEMOI_ERR_ARG;
  return rt::host_batch(
      device, n, in, eb * width, out, eb * (width / k), [&](int dev) { return quantum_of(field, anemoi::kKindJive, width, dev); },
      [&](void* i, void* o, size_t cnt, hipStream_t s) { return anemoi_jive_compress_k_dev(field, width, k, i, o, cnt, s); });
}

int anemoi_jive_compress_batch(int field, int width, const uint64_t* in, uint64_t* out, size_t n, int device) {
  return anemoi_jive_compress_k_batch(field, width, 2, in, out, n, device);
}

int anemoi_merge_batch(int field, const uint64_t* pairs, uint64_t* out, size_t n, int device) {
  return anemoi_jive_compress_k_batch(field, 2, 2, pairs, out, n, device);
}

int anemoi_hash_field_batch(int field, int width, const uint64_t* elems, size_t elems_per_msg, size_t n,
                            uint64_t* out, int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!out || (elems_per_msg && !elems))) return ANEMOI_ERR_ARG;
  if (n == 0) return ANEMOI_OK;
  return sponge_host(field, width, 0, elems, elems_per_msg, n, out, device);
}

int anemoi_hash_bytes_batch(int field, int width, const uint8_t* msgs, size_t msg_len, size_t n, uint64_t* out,
                            int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!out || (msg_len && !msgs))) return ANEMOI_ERR_ARG;
  if (n == 0) return ANEMOI_OK;
  return sponge_host(field, width, 1, msgs, msg_len, n, out, device);
}

// Sponge::hash / hash_field on n host messages of different lengths: bytes with byte offsets, or (bytes = 0) ABI
// elements with offsets in ELEMENTS.  Everything below works on BYTE offsets (element offsets x the element size); the
// offsets a chunk carries to the device are converted back to the kernel's unit.
static int ragged_host(int field, int width, int bytes, const uint8_t* msgs, const uint64_t* offsets, size_t n, uint64_t* out,
                       int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!out || !offsets)) return ANEMOI_ERR_ARG;
  if (n == 0) return ANEMOI_OK;
  for (size_t i = 0; i < n; i++)
    if (offsets[i + 1] < offsets[i]) return ANEMOI_ERR_ARG;  // offsets must be non-decreasing
  if (offsets[n] > offsets[0] && !msgs) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field), unit = bytes ? 1 : eb;
  if (offsets[n] > UINT64_MAX / unit) return ANEMOI_ERR_ARG;
  std::vector<uint64_t> scaled;   // element offsets as byte offsets
  if (unit != 1) {
    scaled.resize(n + 1);
    for (size_t i = 0; i <= n; i++) scaled[i] = offsets[i] * unit;
  }
  const uint64_t* boff = unit == 1 ? offsets : scaled.data();
  // bytes one permutation absorbs
  const size_t block_bytes = size_t(width - 1) * (bytes ? size_t(anemoi::field_ops(field)->chunk) : eb);
  // Chunked like the fixed-length batches (rt::pipeline_staged): cut at message boundaries near the chunk
  // target, whole wavefronts of messages per chunk, offsets rebased to the chunk's first byte; the copy of
  // chunk c + 1 runs under the kernel of chunk c and the device holds three chunks, not the batch.
  // Staging of a chunk: [count + 1 offsets | pad to 256 B | the chunk's bytes].
  const size_t per_wave = width == 2 ? size_t(anemoi::kBlock) : size_t(anemoi::kPairStates);
  return rt::for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (!count) return ANEMOI_OK;
    return with_lane(dev, [&](Lane& ln) -> int {
      const uint64_t* off = boff + first;
      const size_t quantum = quantum_of(field, anemoi::kKindSponge, width, dev);
      // Length bucketing: a wavefront costs what its longest message costs, and a realistic ragged batch arrives
      // unsorted.  The host walks every offset anyway, so messages are STAGED by descending block count (stable;
      // host::ragged_order returns nothing when the order as given is already within ~3 % of that) and the digests
      // are scattered back to the caller's order: per-message semantics unchanged (src/<f>/anemoi_x/hasher.rs hash()).
      const std::vector<size_t> order = host::ragged_order(off, count, block_bytes, per_wave);
      std::vector<uint64_t> poff;     // offsets of the permuted sequence (virtual: the bytes are gathered while staging)
      if (!order.empty()) {
        poff.resize(count + 1);
        poff[0] = 0;
        for (size_t i = 0; i < count; i++) poff[i + 1] = poff[i] + (off[order[i] + 1] - off[order[i]]);
      }
      const uint64_t* voff = order.empty() ? off : poff.data();
      // at least half a wave of workgroups of messages per chunk (two chunks' kernels run side by side on the
      // lane's two kernel streams), at most 8 chunk targets of bytes
      const size_t target = rt::chunk_target_bytes();
      const std::vector<size_t> cuts =
          host::plan_ragged_chunks(voff, count, target, per_wave, quantum / 2, 4 * quantum, 8 * target);
      auto cnt_of = [&](size_t c) { return cuts[c + 1] - cuts[c]; };
      auto off_bytes = [&](size_t c) { return align_up((cnt_of(c) + 1) * 8, 256); };
      auto msg_bytes = [&](size_t c) { return size_t(voff[cuts[c + 1]] - voff[cuts[c]]); };
      return rt::pipeline_staged(
          ln, cuts.size() - 1,
          [&](size_t c) { return rt::StagedChunk{off_bytes(c) + msg_bytes(c) + 16, cnt_of(c) * eb, 0}; },
          [&](size_t c, char* h) -> int {
            uint64_t* rel = (uint64_t*)h;
            const uint64_t base = voff[cuts[c]];
            for (size_t i = 0; i <= cnt_of(c); i++) rel[i] = voff[cuts[c] + i] - base;
            if (order.empty()) {
              if (msg_bytes(c)) memcpy(h + off_bytes(c), msgs + base, msg_bytes(c));
            } else {   // gather: message order[k] goes where the permuted sequence puts it
              char* dst = h + off_bytes(c);
              for (size_t i = 0; i < cnt_of(c); i++) {
                const size_t m = order[cuts[c] + i];
                const size_t len = size_t(off[m + 1] - off[m]);
                if (len) memcpy(dst + rel[i], msgs + off[m], len);
              }
            }
            if (unit != 1)   // the kernel reads offsets in elements
              for (size_t i = 0; i <= cnt_of(c); i++) rel[i] /= unit;
            return ANEMOI_OK;
          },
          [&](size_t c, void* di, void* dout, void*, hipStream_t st) -> int {
            return ragged_dev(field, width, bytes, (char*)di + off_bytes(c), di, cnt_of(c), dout, st);
          },
          [&](size_t c, const char* h) -> int {
            if (order.empty()) {
              memcpy((char*)out + (first + cuts[c]) * eb, h, cnt_of(c) * eb);
            } else {
              for (size_t i = 0; i < cnt_of(c); i++)
                memcpy((char*)out + (first + order[cuts[c] + i]) * eb, h + i * eb, eb);
            }
            return ANEMOI_OK;
          });
    });
  });
}
int anemoi_hash_bytes_ragged_batch(int field, int width, const uint8_t* msgs, const uint64_t* offsets, size_t n,
                                   uint64_t* out, int device) {
  return ragged_host(field, width, 1, msgs, offsets, n, out, device);
}
int anemoi_hash_field_ragged_batch(int field, int width, const uint64_t* elems, const uint64_t* offsets, size_t n,
                                   uint64_t* out, int device) {
  return ragged_host(field, width, 0, (const uint8_t*)elems, offsets, n, out, device);
}

int anemoi_to_montgomery(int field, const uint64_t* in, uint64_t* out, size_t count, int device) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (count && (!in || !out)) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  return rt::host_batch(
      device, count, in, eb, out, eb, [&](int dev) { return quantum_of(field, anemoi::kKindConvert, 2, dev); },
      [&](void* i, void* o, size_t cnt, hipStream_t s) { return anemoi_to_montgomery_dev(field, i, o, cnt, s); });
}

int anemoi_from_montgomery(int field, const uint64_t* in, uint64_t* out, size_t count, int device) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (count && (!in || !out)) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  return rt::host_batch(
      device, count, in, eb, out, eb, [&](int dev) { return quantum_of(field, anemoi::kKindConvert, 2, dev); },
      [&](void* i, void* o, size_t cnt, hipStream_t s) { return anemoi_from_montgomery_dev(field, i, o, cnt, s); });
}

/* ---- Merkle trees ---- */

int anemoi_merkle_root(int field, const uint64_t* leaves, unsigned depth, uint64_t* root, int device) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (!leaves || !root || depth > 30) return ANEMOI_ERR_ARG;
  return merkle_host(TreeShape{field, 1}, leaves, depth, root, nullptr, device);
}

int anemoi_merkle_tree(int field, const uint64_t* leaves, unsigned depth, uint64_t* tree, int device) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (!leaves || !tree || depth > 30) return ANEMOI_ERR_ARG;
  return merkle_host(TreeShape{field, 1}, leaves, depth, nullptr, tree, device);
}

/* host-side indexing only: the sibling of node (level l, position index >> l) for l = 0 .. depth-1 */
int anemoi_merkle_path(int field, const uint64_t* tree, unsigned depth, size_t index, uint64_t* path) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (!tree || (depth && !path) || depth > 30 || index >= (size_t(1) << depth)) return ANEMOI_ERR_ARG;
  host::merkle_path2(tree, depth, index, size_t(anemoi::field_ops(field)->limbs64), path);
  return ANEMOI_OK;
}

int anemoi_merkle_verify_batch(int field, const uint64_t* leaves, const uint64_t* indices, const uint64_t* paths,
                               unsigned depth, size_t n, const uint64_t* root, uint8_t* ok, int device) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (depth > 63 || (n && (!leaves || !indices || !root || !ok || (depth && !paths)))) return ANEMOI_ERR_ARG;
  if (n == 0) return ANEMOI_OK;
  const size_t eb = elem_bytes(field);
  // chunked (rt::pipeline_staged): a chunk's staging is [leaves | indices | paths], each part 256-byte aligned
  return rt::for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (!count) return ANEMOI_OK;
    return with_lane(dev, [&](Lane& ln) -> int {
      const size_t per_item = eb + 8 + size_t(depth) * eb;
      const host::ChunkPlan cp = host::plan_chunks(count, quantum_of(field, anemoi::kKindJive, 2, dev), per_item,
                                                   rt::chunk_target_bytes());
      auto first_of = [&](size_t c) { return c * cp.chunk_items; };
      auto count_of = [&](size_t c) { return c + 1 == cp.chunks ? count - first_of(c) : cp.chunk_items; };
      auto o_idx = [&](size_t c) { return align_up(count_of(c) * eb, 256); };
      auto o_path = [&](size_t c) { return o_idx(c) + align_up(count_of(c) * 8, 256); };
      return rt::pipeline_staged(
          ln, cp.chunks,
          [&](size_t c) { return rt::StagedChunk{o_path(c) + count_of(c) * depth * eb + 16, count_of(c) * eb, 0}; },
          [&](size_t c, char* h) -> int {
            const size_t f = first + first_of(c), m = count_of(c);
            memcpy(h, (const char*)leaves + f * eb, m * eb);
            memcpy(h + o_idx(c), indices + f, m * 8);
            if (depth) memcpy(h + o_path(c), (const char*)paths + f * depth * eb, m * depth * eb);
            return ANEMOI_OK;
          },
          [&](size_t c, void* di, void* dout, void*, hipStream_t st) -> int {
            char* b = (char*)di;
            return anemoi_merkle_climb_dev(field, b, b + o_idx(c), b + o_path(c), depth, count_of(c), dout, st);
          },
          [&](size_t c, const char* h) -> int {
            const size_t f = first + first_of(c);
            for (size_t i = 0; i < count_of(c); i++) ok[f + i] = memcmp(h + i * eb, root, eb) == 0 ? 1 : 0;
            return ANEMOI_OK;
          });
    });
  });
}

/* Arity-4 tree with the 4-3 instance's Jive-4 (compress_k(.,4), anemoi_4_3/hasher.rs:162-179):
 * 4^depth4 leaf digests -> root, level by level. */
int anemoi_merkle_root_arity4(int field, const uint64_t* leaves, unsigned depth4, uint64_t* root, int device) {
  int rc = check_instance(field, 4);
  if (rc) return rc;
  if (!leaves || !root || depth4 > 15) return ANEMOI_ERR_ARG;
  return merkle_host(TreeShape{field, 2}, leaves, depth4, root, nullptr, device);
}

int anemoi_merkle_tree_arity4(int field, const uint64_t* leaves, unsigned depth4, uint64_t* tree, int device) {
  int rc = check_instance(field, 4);
  if (rc) return rc;
  if (!leaves || !tree || depth4 > 15) return ANEMOI_ERR_ARG;
  return merkle_host(TreeShape{field, 2}, leaves, depth4, nullptr, tree, device);
}

int anemoi_merkle_path_arity4(int field, const uint64_t* tree, unsigned depth4, size_t index, uint64_t* path) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (!tree || (depth4 && !path) || depth4 > 15 || index >= (size_t(1) << (2 * depth4))) return ANEMOI_ERR_ARG;
  host::merkle_path4(tree, depth4, index, size_t(anemoi::field_ops(field)->limbs64), path);
  return ANEMOI_OK;
}

int anemoi_merkle_verify_arity4_batch(int field, const uint64_t* leaves, const uint64_t* indices, const uint64_t* paths,
                                      unsigned depth4, size_t n, const uint64_t* root, uint8_t* ok, int device) {
  int rc = check_instance(field, 4);
  if (rc) return rc;
  if (depth4 > 31 || (n && (!leaves || !indices || !root || !ok || (depth4 && !paths)))) return ANEMOI_ERR_ARG;
  if (n == 0) return ANEMOI_OK;
  const size_t eb = elem_bytes(field);
  const int quads = int(eb / 16);
  // chunked like the binary form; per chunk the current nodes live in the slot's output buffer and the
  // assembled 4-element states in the slot's scratch
  return rt::for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (!count) return ANEMOI_OK;
    return with_lane(dev, [&](Lane& ln) -> int {
      PermConsts pc;
      int r = get_consts(field, 4, &pc);
      if (r) return r;
      const size_t per_item = eb + 8 + size_t(depth4) * 3 * eb;
      const host::ChunkPlan cp = host::plan_chunks(count, quantum_of(field, anemoi::kKindJive, 4, dev), per_item,
                                                   rt::chunk_target_bytes());
      auto first_of = [&](size_t c) { return c * cp.chunk_items; };
      auto count_of = [&](size_t c) { return c + 1 == cp.chunks ? count - first_of(c) : cp.chunk_items; };
      auto o_idx = [&](size_t c) { return align_up(count_of(c) * eb, 256); };
      auto o_path = [&](size_t c) { return o_idx(c) + align_up(count_of(c) * 8, 256); };
      return rt::pipeline_staged(
          ln, cp.chunks,
          [&](size_t c) {
            return rt::StagedChunk{o_path(c) + count_of(c) * depth4 * 3 * eb + 16, count_of(c) * eb, count_of(c) * 4 * eb};
          },
          [&](size_t c, char* h) -> int {
            const size_t f = first + first_of(c), m = count_of(c);
            memcpy(h, (const char*)leaves + f * eb, m * eb);
            memcpy(h + o_idx(c), indices + f, m * 8);
            if (depth4) memcpy(h + o_path(c), (const char*)paths + f * depth4 * 3 * eb, m * depth4 * 3 * eb);
            return ANEMOI_OK;
          },
          [&](size_t c, void* di, void* dout, void* dtmp, hipStream_t st) -> int {
            char* b = (char*)di;
            const size_t m = count_of(c), threads = m * 4 * size_t(quads);
            if (depth4 == 0) HIP_TRY(hipMemcpyAsync(dout, b, m * eb, hipMemcpyDeviceToDevice, st));
            for (unsigned l = 0; l < depth4; l++) {
              // level 0 reads the leaves where they were staged; the following levels the previous level's nodes
              k_assemble4<<<unsigned((threads + 255) / 256), 256, 0, st>>>(
                  (const uint4*)(l == 0 ? (void*)b : dout), (const uint4*)(b + o_path(c)), (const uint64_t*)(b + o_idx(c)), l,
                  depth4, m, quads, (uint4*)dtmp);
              HIP_TRY(hipGetLastError());
              HIP_TRY(anemoi::field_ops(field)->jive(4, 4, dtmp, dout, m, pc, st));
            }
            return ANEMOI_OK;
          },
          [&](size_t c, const char* h) -> int {
            const size_t f = first + first_of(c);
            for (size_t i = 0; i < count_of(c); i++) ok[f + i] = memcmp(h + i * eb, root, eb) == 0 ? 1 : 0;
            return ANEMOI_OK;
          });
    });
  });
}

}  // extern "C"
