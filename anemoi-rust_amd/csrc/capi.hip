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
  static std::mutex mu;
  static size_t cache[anemoi::kNumFields][5][2] = {};
  const int wi = width == 2 ? 0 : 1;
  std::lock_guard<std::mutex> lock(mu);
  size_t& q = cache[field][kind][wi];
  if (!q) q = anemoi::field_ops(field)->wave_items(kind, width, rt::device_cus(dev));
  return q;
}

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

// Uploads an instance's constants into lane scratch buffer 0 (stream-ordered on the lane's kernel stream).
int upload_generic(Lane& ln, const anemoi_generic_instance* inst, anemoi::GenericConsts* gc) {
  const FieldOps* ops = anemoi::field_ops(inst->field);
  const size_t eb = elem_bytes(inst->field), c = size_t(inst->num_columns);
  const size_t ab = size_t(inst->num_rounds) * c * eb, mb = c * c * eb;
  int rc = ln.scratch[0].reserve(2 * ab + mb);
  if (rc) return rc;
  char* b = (char*)ln.scratch[0].p;
  HIP_TRY(hipMemcpyAsync(b, inst->ark_c, ab, hipMemcpyHostToDevice, ln.s_k));
  HIP_TRY(hipMemcpyAsync(b + ab, inst->ark_d, ab, hipMemcpyHostToDevice, ln.s_k));
  std::vector<uint64_t> canon;  // must outlive the asynchronous copy below: synchronised before returning
  if (inst->mds) {
    HIP_TRY(hipMemcpyAsync(b + 2 * ab, inst->mds, mb, hipMemcpyHostToDevice, ln.s_k));
  } else {
    std::vector<uint64_t> small;
    if (!host::builtin_mds(inst->num_columns, uint64_t(ops->generator), &small)) return ANEMOI_ERR_ARG;
    canon.assign(c * c * (eb / 8), 0);
    for (size_t i = 0; i < c * c; i++) canon[i * (eb / 8)] = small[i];
    HIP_TRY(hipMemcpyAsync(b + 2 * ab, canon.data(), mb, hipMemcpyHostToDevice, ln.s_k));
    HIP_TRY(ops->mont_convert(1, b + 2 * ab, b + 2 * ab, c * c, ln.s_k));
  }
  HIP_TRY(hipStreamSynchronize(ln.s_k));
  gc->ark_c = (const uint32_t*)b;
  gc->ark_d = (const uint32_t*)(b + ab);
  gc->mds = (const uint32_t*)(b + 2 * ab);
  gc->cols = inst->num_columns;
  gc->rounds = inst->num_rounds;
  return ANEMOI_OK;
}

// host-pointer batch over a run-time instance: constants are uploaded per shard, then `launch`
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
      if (!rc) rc = ln.scratch[1].reserve(count * in_per_item);
      if (!rc && out != in) rc = ln.scratch[2].reserve(count * out_per_item);
      if (rc) return rc;
      void* di = ln.scratch[1].p;
      void* o = out != in ? ln.scratch[2].p : di;
      HIP_TRY(hipMemcpyAsync(di, (const char*)in + first * in_per_item, count * in_per_item, hipMemcpyHostToDevice,
                             ln.s_k));
      HIP_TRY(launch(di, o, count, gc, pc, ln.s_k));
      HIP_TRY(hipMemcpyAsync((char*)out + first * out_per_item, o, count * out_per_item, hipMemcpyDeviceToHost,
                             ln.s_k));
      HIP_TRY(hipStreamSynchronize(ln.s_k));
      return ANEMOI_OK;
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

size_t segment_target_bytes() {
  const char* e = getenv("ANEMOI_SPONGE_SEGMENT_BYTES");   // test knob: force (small) segments
  if (e && *e) return size_t(strtoull(e, nullptr, 10));
  return rt::kChunkTargetBytes;
}

// unit = bytes of RATE elements of input (BYTES: RATE x chunk bytes; else RATE x element bytes)
bool want_segments(size_t n, size_t per_msg_bytes, size_t unit) {
  const bool forced = getenv("ANEMOI_SPONGE_SEGMENT_BYTES") != nullptr;
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
  for (int s = 0; !rc && s < rt::kSlots; s++) {
    rc = ln.slot[s].d_in.reserve(n * seg_bytes);
    if (!rc) rc = ln.slot[s].p_in.reserve(n * seg_bytes);
  }
  if (rc) return rc;
  for (size_t c = 0; c < nseg; c++) {
    rt::Slot& sl = ln.slot[c % rt::kSlots];
    const size_t off = c * seg_bytes, len = c + 1 == nseg ? per_msg_bytes - off : seg_bytes;
    // the slot's pinned buffer is free once its previous copy-in has completed, its device buffer once the
    // kernel that read it has: the first is waited for here, the second is a stream dependency
    if (c >= size_t(rt::kSlots)) {
      HIP_TRY(hipEventSynchronize(sl.e_in));
      HIP_TRY(hipStreamWaitEvent(ln.s_in, sl.e_k, 0));
    }
    char* stage = (char*)sl.p_in.p;
    for (size_t i = 0; i < n; i++) memcpy(stage + i * len, src + i * per_msg_bytes + off, len);   // strided gather
    HIP_TRY(hipMemcpyAsync(sl.d_in.p, stage, n * len, hipMemcpyHostToDevice, ln.s_in));
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
      if (want_segments(blk, per_msg_bytes, unit) && (count < 2 * quantum || quantum * per_msg_bytes > (size_t(256) << 20))) {
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

int anemoi_abi_version(void) { return 2; }

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
  if ((const void*)in == (void*)out) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(inst->field);
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

int anemoi_hash_bytes_ragged_dev(int field, int width, const void* d_msgs, const void* d_offsets, size_t n, void* d_out,
                                 void* stream) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!d_out || !d_offsets || !d_msgs)) return ANEMOI_ERR_ARG;
  PermConsts pc;
  if ((rc = get_consts(field, width, &pc))) return rc;
  HIP_TRY(anemoi::field_ops(field)->sponge_ragged(width, d_msgs, d_offsets, n, d_out, pc, (hipStream_t)stream));
  return ANEMOI_OK;
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

int anemoi_hash_bytes_ragged_batch(int field, int width, const uint8_t* msgs, const uint64_t* offsets, size_t n,
                                   uint64_t* out, int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!out || !offsets)) return ANEMOI_ERR_ARG;
  if (n == 0) return ANEMOI_OK;
  for (size_t i = 0; i < n; i++)
    if (offsets[i + 1] < offsets[i]) return ANEMOI_ERR_ARG;  // offsets must be non-decreasing
  if (offsets[n] > offsets[0] && !msgs) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  static const uint8_t dummy[16] = {0};
  return rt::for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (!count) return ANEMOI_OK;
    return with_lane(dev, [&](Lane& ln) -> int {
      const uint64_t base = offsets[first], bytes = offsets[first + count] - base;
      std::vector<uint64_t> rel(count + 1);
      for (size_t i = 0; i <= count; i++) rel[i] = offsets[first + i] - base;
      rt::Buf &dm = ln.scratch[0], &dof = ln.scratch[1], &dd = ln.scratch[2];
      int r = dm.reserve(bytes + 16);
      if (!r) r = dof.reserve((count + 1) * 8);
      if (!r) r = dd.reserve(count * eb);
      if (r) return r;
      hipStream_t s = ln.s_k;
      HIP_TRY(hipMemcpyAsync(dm.p, bytes ? (const void*)(msgs + base) : (const void*)dummy, bytes ? bytes : 16,
                             hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync(dof.p, rel.data(), (count + 1) * 8, hipMemcpyHostToDevice, s));
      r = anemoi_hash_bytes_ragged_dev(field, width, dm.p, dof.p, count, dd.p, s);
      if (r) return r;
      HIP_TRY(hipMemcpyAsync((char*)out + first * eb, dd.p, count * eb, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));  // `rel` must outlive the copy-in
      return ANEMOI_OK;
    });
  });
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
  return rt::for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (!count) return ANEMOI_OK;
    return with_lane(dev, [&](Lane& ln) -> int {
      rt::Buf &dl = ln.scratch[0], &di = ln.scratch[1], &dp = ln.scratch[2], &dr = ln.scratch[3];
      int r = dl.reserve(count * eb);
      if (!r) r = di.reserve(count * 8);
      if (!r) r = dp.reserve(count * depth * eb);
      if (!r) r = dr.reserve(count * eb);
      if (r) return r;
      hipStream_t s = ln.s_k;
      HIP_TRY(hipMemcpyAsync(dl.p, (const char*)leaves + first * eb, count * eb, hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync(di.p, indices + first, count * 8, hipMemcpyHostToDevice, s));
      if (depth)
        HIP_TRY(hipMemcpyAsync(dp.p, (const char*)paths + first * depth * eb, count * depth * eb, hipMemcpyHostToDevice, s));
      r = anemoi_merkle_climb_dev(field, dl.p, di.p, dp.p, depth, count, dr.p, s);
      if (r) return r;
      std::vector<uint64_t> got(count * (eb / 8));
      HIP_TRY(hipMemcpyAsync(got.data(), dr.p, count * eb, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      for (size_t i = 0; i < count; i++) ok[first + i] = memcmp(&got[i * (eb / 8)], root, eb) == 0 ? 1 : 0;
      return ANEMOI_OK;
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
  return rt::for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (!count) return ANEMOI_OK;
    return with_lane(dev, [&](Lane& ln) -> int {
      rt::Buf &dcur = ln.scratch[0], &di = ln.scratch[1], &dp = ln.scratch[2], &dst = ln.scratch[3];
      int r = dcur.reserve(count * eb);
      if (!r) r = di.reserve(count * 8);
      if (!r) r = dp.reserve(count * depth4 * 3 * eb);
      if (!r) r = dst.reserve(count * 4 * eb);
      if (r) return r;
      hipStream_t s = ln.s_k;
      HIP_TRY(hipMemcpyAsync(dcur.p, (const char*)leaves + first * eb, count * eb, hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync(di.p, indices + first, count * 8, hipMemcpyHostToDevice, s));
      if (depth4)
        HIP_TRY(hipMemcpyAsync(dp.p, (const char*)paths + first * depth4 * 3 * eb, count * depth4 * 3 * eb,
                               hipMemcpyHostToDevice, s));
      PermConsts pc;
      if ((r = get_consts(field, 4, &pc))) return r;
      const size_t threads = count * 4 * size_t(quads);
      for (unsigned l = 0; l < depth4; l++) {
        k_assemble4<<<unsigned((threads + 255) / 256), 256, 0, s>>>((const uint4*)dcur.p, (const uint4*)dp.p,
                                                                     (const uint64_t*)di.p, l, depth4, count, quads,
                                                                     (uint4*)dst.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(anemoi::field_ops(field)->jive(4, 4, dst.p, dcur.p, count, pc, s));
      }
      std::vector<uint64_t> got(count * (eb / 8));
      HIP_TRY(hipMemcpyAsync(got.data(), dcur.p, count * eb, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      for (size_t i = 0; i < count; i++) ok[first + i] = memcmp(&got[i * (eb / 8)], root, eb) == 0 ? 1 : 0;
      return ANEMOI_OK;
    });
  });
}

}  // extern "C"
