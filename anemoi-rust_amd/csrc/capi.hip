// capi.hip -- the C-ABI of libanemoi_mi355x.so (declared in include/anemoi_mi355x.h).
//
// Host side only: argument checks (the reference's assert!s -> error codes), the lazily created
// per-device constant tables, H2D/D2H staging for the host-pointer entry points, contiguous-range
// sharding over GPUs (no collective: items are independent) and the level-by-level
// Merkle driver.  All arithmetic runs in the HIP kernels of anemoi_kernels.h; there is no CPU path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/anemoi_mi355x.h"
#include "anemoi_kernels.h"

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

namespace {

thread_local std::string g_last_error;

int fail_hip(hipError_t e, const char* what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return ANEMOI_ERR_DEVICE;
}

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return fail_hip(e_, #expr);  \
  } while (0)

constexpr int kMaxDevices = 64;

struct DeviceCtx {
  std::mutex mu;
  bool ready[anemoi::kNumFields][2] = {};
  PermConsts pc[anemoi::kNumFields][2] = {};
};
DeviceCtx g_ctx[kMaxDevices];

// Constant tables for (current device, field, width); uploaded once.
int get_consts(int field, int width, PermConsts* out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return ANEMOI_ERR_DEVICE;
  DeviceCtx& c = g_ctx[dev];
  const int wi = width == 2 ? 0 : 1;
  std::lock_guard<std::mutex> lock(c.mu);
  if (!c.ready[field][wi]) {
    anemoi::HostConsts hc;
    anemoi::field_ops(field)->host_consts(width, &hc);
    // schedules go up as one 32-bit word per step (squarings | op << 8): scalar loads in the kernels
    auto words = [](const std::vector<uint8_t>& pairs) {
      std::vector<uint32_t> w(pairs.size() / 2);
      for (size_t i = 0; i < w.size(); i++) w[i] = uint32_t(pairs[2 * i]) | (uint32_t(pairs[2 * i + 1]) << 8);
      return w;
    };
    const std::vector<uint32_t> s3 = words(hc.sched), s5 = words(hc.sched5);
    const std::vector<uint32_t>* parts[6] = {&hc.ark_c, &hc.ark_d, &s3, &s5, &hc.coop_c, &hc.coop_d};
    size_t off[7] = {0};
    for (int i = 0; i < 6; i++) off[i + 1] = off[i] + parts[i]->size() * sizeof(uint32_t);
    char* blob = nullptr;
    HIP_TRY(hipMalloc((void**)&blob, off[6]));
    for (int i = 0; i < 6; i++)
      HIP_TRY(hipMemcpy(blob + off[i], parts[i]->data(), parts[i]->size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    PermConsts pc;
    pc.ark_c = (const uint32_t*)(blob + off[0]);
    pc.ark_d = (const uint32_t*)(blob + off[1]);
    pc.sched = (const uint32_t*)(blob + off[2]);
    pc.steps = hc.steps;
    pc.first = hc.first;
    pc.sched5 = (const uint32_t*)(blob + off[3]);
    pc.steps5 = hc.steps5;
    pc.first5 = hc.first5;
    pc.coop_c = (const uint32_t*)(blob + off[4]);
    pc.coop_d = (const uint32_t*)(blob + off[5]);
    c.pc[field][wi] = pc;
    c.ready[field][wi] = true;
  }
  *out = c.pc[field][wi];
  return ANEMOI_OK;
}

int check_instance(int field, int width) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (width != 2 && width != 4) return ANEMOI_ERR_WIDTH;
  return ANEMOI_OK;
}

// the reference's compress_k asserts (anemoi_2_1/hasher.rs:107; anemoi_4_3/hasher.rs:163-165)
int check_k(int width, int k) {
  if (width == 2) return k == 2 ? ANEMOI_OK : ANEMOI_ERR_ARG;
  return (k == 2 || k == 4) ? ANEMOI_OK : ANEMOI_ERR_ARG;
}

size_t elem_bytes(int field) { return size_t(anemoi::field_ops(field)->limbs64) * 8; }

struct DeviceGuard {  // restores the caller's current device
  int prev = -1;
  DeviceGuard() { (void)hipGetDevice(&prev); }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) {
      g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
      return ANEMOI_ERR_ALLOC;
    }
    return ANEMOI_OK;
  }
};

// Runs `body(first, count)` on one device, or on contiguous ranges over all devices (one host
// thread per GPU).  body must be thread-safe and set its own device.
template <class Body>
int for_devices(int device, size_t n, Body body) {
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0) {
    g_last_error = "no HIP device";
    return ANEMOI_ERR_DEVICE;
  }
  if (device != ANEMOI_ALL_DEVICES) {
    if (device < 0 || device >= ndev || device >= kMaxDevices) {
      g_last_error = "device ordinal out of range";
      return ANEMOI_ERR_DEVICE;
    }
    return body(device, size_t(0), n);
  }
  if (ndev > kMaxDevices) ndev = kMaxDevices;
  if (ndev == 1 || n < size_t(ndev)) return body(0, size_t(0), n);
  std::vector<int> rc(ndev, ANEMOI_OK);
  std::vector<std::string> err(ndev);
  std::vector<std::thread> th;
  for (int d = 0; d < ndev; d++) {
    const size_t b = n * size_t(d) / size_t(ndev), e = n * size_t(d + 1) / size_t(ndev);
    th.emplace_back([&, d, b, e] {
      rc[d] = body(d, b, e - b);
      if (rc[d] != ANEMOI_OK) err[d] = g_last_error;
    });
  }
  for (auto& t : th) t.join();
  for (int d = 0; d < ndev; d++)
    if (rc[d] != ANEMOI_OK) {
      g_last_error = err[d];
      return rc[d];
    }
  return ANEMOI_OK;
}

// Generic host-pointer batch: copy `in_per_item` bytes per item in, run `launch`, copy out, all on the
// device's default stream.  Cutting a batch into chunks on two streams (copy of chunk i+1 under the
// kernel of chunk i) was measured and dropped: 2^20 BLS12-381 compressions from pageable memory took
// 138.8 ms in 8 chunks against 130.2 ms unchunked (kernel 119.6 ms) -- every chunk ends in a partially
// filled wave of workgroups on this ALU-bound kernel, which costs more than the ~10 ms of copies that
// could be hidden.
template <class LaunchFn>
int host_batch(int device, size_t n, const void* in, size_t in_per_item, void* out, size_t out_per_item,
               LaunchFn launch) {
  if (n == 0) return ANEMOI_OK;
  return for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (count == 0) return ANEMOI_OK;
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    DevBuf din, dout;
    int rc = din.alloc(count * in_per_item);
    if (rc) return rc;
    if (out != in) {
      rc = dout.alloc(count * out_per_item);
      if (rc) return rc;
    }
    void* o = out != in ? dout.p : din.p;
    HIP_TRY(hipMemcpyAsync(din.p, (const char*)in + first * in_per_item, count * in_per_item, hipMemcpyHostToDevice,
                           nullptr));
    rc = launch(din.p, o, count, (hipStream_t) nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync((char*)out + first * out_per_item, o, count * out_per_item, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return ANEMOI_OK;
  });
}

// ---- run-time instances (anemoi_generic.h) ---------------------------------------------------------

// The matrix a hard-coded mds_layer arm applies to each half of the state, as small integers: the arm's
// statements (src/traits.rs:136-279; mds_internal :307-323) applied to the unit vectors.
bool builtin_mds(int c, uint64_t g, std::vector<uint64_t>* m) {
  if (c < 1 || c > 6) return false;
  m->assign(size_t(c) * c, 0);
  for (int j = 0; j < c; j++) {
    uint64_t s[6] = {0, 0, 0, 0, 0, 0}, o[6];
    s[j] = 1;
    switch (c) {
      case 1: o[0] = s[0]; break;
      case 2:
        s[0] += g * s[1];
        s[1] += g * s[0];
        o[0] = s[0], o[1] = s[1];
        break;
      case 3: {
        const uint64_t tmp = s[0] + g * s[2];
        s[2] += s[1];
        s[2] += g * s[0];
        s[0] = tmp + s[2];
        s[1] += tmp;
        o[0] = s[0], o[1] = s[1], o[2] = s[2];
        break;
      }
      case 4:
        s[0] += s[1];
        s[2] += s[3];
        s[3] += g * s[0];
        s[1] = g * (s[1] + s[2]);
        s[0] += s[1];
        s[2] += g * s[3];
        s[1] += s[2];
        s[3] += s[0];
        for (int i = 0; i < 4; i++) o[i] = s[i];
        break;
      case 5: {
        const uint64_t tot = s[0] + s[1] + s[2] + s[3] + s[4];
        for (int i = 0; i < 5; i++)
          o[i] = tot + s[(i + 3) % 5] + 2 * (s[(i + 2) % 5] + s[(i + 3) % 5] + 2 * s[(i + 4) % 5]);
        break;
      }
      default: {
        const uint64_t tot = s[0] + s[1] + s[2] + s[3] + s[4] + s[5];
        for (int i = 0; i < 6; i++)
          o[i] = tot + s[(i + 3) % 6] + s[(i + 5) % 6] +
                 2 * (s[(i + 2) % 6] + s[(i + 3) % 6] + 2 * (s[(i + 4) % 6] + s[(i + 5) % 6]));
        break;
      }
    }
    for (int i = 0; i < c; i++) (*m)[size_t(i) * c + j] = o[i];
  }
  return true;
}

int check_generic(const anemoi_generic_instance* inst) {
  if (!inst) return ANEMOI_ERR_ARG;
  if (!anemoi::field_ops(inst->field)) return ANEMOI_ERR_FIELD;
  if (inst->num_columns < 1 || inst->num_columns > ANEMOI_MAX_GENERIC_COLUMNS) return ANEMOI_ERR_WIDTH;
  if (inst->num_rounds < 1 || inst->num_rounds > 255 || !inst->ark_c || !inst->ark_d) return ANEMOI_ERR_ARG;
  if (!inst->mds && inst->num_columns > 6) return ANEMOI_ERR_ARG;  // "NO MDS matrix specified for this instance."
  return ANEMOI_OK;
}

// Uploads an instance's constants to the current device (stream-ordered on the default stream).
int upload_generic(const anemoi_generic_instance* inst, DevBuf* blob, anemoi::GenericConsts* gc) {
  const FieldOps* ops = anemoi::field_ops(inst->field);
  const size_t eb = elem_bytes(inst->field), c = size_t(inst->num_columns);
  const size_t ab = size_t(inst->num_rounds) * c * eb, mb = c * c * eb;
  int rc = blob->alloc(2 * ab + mb);
  if (rc) return rc;
  char* b = (char*)blob->p;
  HIP_TRY(hipMemcpy(b, inst->ark_c, ab, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b + ab, inst->ark_d, ab, hipMemcpyHostToDevice));
  if (inst->mds) {
    HIP_TRY(hipMemcpy(b + 2 * ab, inst->mds, mb, hipMemcpyHostToDevice));
  } else {
    std::vector<uint64_t> small, canon(c * c * (eb / 8), 0);
    if (!builtin_mds(inst->num_columns, uint64_t(ops->generator), &small)) return ANEMOI_ERR_ARG;
    for (size_t i = 0; i < c * c; i++) canon[i * (eb / 8)] = small[i];
    HIP_TRY(hipMemcpy(b + 2 * ab, canon.data(), mb, hipMemcpyHostToDevice));
    HIP_TRY(ops->mont_convert(1, b + 2 * ab, b + 2 * ab, c * c, nullptr));
  }
  gc->ark_c = (const uint32_t*)b;
  gc->ark_d = (const uint32_t*)(b + ab);
  gc->mds = (const uint32_t*)(b + 2 * ab);
  gc->cols = inst->num_columns;
  gc->rounds = inst->num_rounds;
  return ANEMOI_OK;
}

// host-pointer batch over a run-time instance: constants are uploaded per device, then `launch`
template <class LaunchFn>
int generic_batch(const anemoi_generic_instance* inst, int device, size_t n, const void* in, size_t in_per_item,
                  void* out, size_t out_per_item, LaunchFn launch) {
  if (n == 0) return ANEMOI_OK;
  return for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (count == 0) return ANEMOI_OK;
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    DevBuf blob, din, dout;
    anemoi::GenericConsts gc;
    PermConsts pc;
    int rc = upload_generic(inst, &blob, &gc);
    if (!rc) rc = get_consts(inst->field, 2, &pc);  // exponent schedule of the field
    if (!rc) rc = din.alloc(count * in_per_item);
    if (!rc && out != in) rc = dout.alloc(count * out_per_item);
    if (rc) return rc;
    void* o = out != in ? dout.p : din.p;
    HIP_TRY(hipMemcpy(din.p, (const char*)in + first * in_per_item, count * in_per_item, hipMemcpyHostToDevice));
    HIP_TRY(launch(din.p, o, count, gc, pc));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy((char*)out + first * out_per_item, o, count * out_per_item, hipMemcpyDeviceToHost));
    return ANEMOI_OK;
  });
}

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

}  // namespace

extern "C" {

int anemoi_abi_version(void) { return 1; }

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
    case ANEMOI_ERR_ARG: return "invalid argument (null pointer, unsupported k, or size out of range)";
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

/* ---- run-time instances ---- */

int anemoi_generic_mds_matrix(int field, int num_columns, uint64_t* mds, int device) {
  const FieldOps* ops = anemoi::field_ops(field);
  if (!ops) return ANEMOI_ERR_FIELD;
  std::vector<uint64_t> small;
  if (!mds || !builtin_mds(num_columns, uint64_t(ops->generator), &small)) return ANEMOI_ERR_ARG;
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
                       [&](void* i, void*, size_t cnt, anemoi::GenericConsts gc, PermConsts pc) {
                         return anemoi::field_ops(inst->field)->generic_permutation(i, cnt, gc, pc, nullptr);
                       });
}

int anemoi_generic_jive_compress_k_batch(const anemoi_generic_instance* inst, int k, const uint64_t* in,
                                         uint64_t* out, size_t n, int device) {
  int rc = check_generic(inst);
  if (rc) return rc;
  const int w = 2 * inst->num_columns;
  // the reference's asserts (anemoi_4_3/hasher.rs:163-165): k <= width, k | width, k even
  if (k < 2 || k > w || w % k != 0 || k % 2 != 0) return ANEMOI_ERR_ARG;
  if (n && (!in || !out)) return ANEMOI_ERR_ARG;
  if ((const void*)in == (void*)out) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(inst->field);
  return generic_batch(inst, device, n, in, eb * w, out, eb * (w / k),
                       [&](void* i, void* o, size_t cnt, anemoi::GenericConsts gc, PermConsts pc) {
                         return anemoi::field_ops(inst->field)->generic_jive(i, o, cnt, k, gc, pc, nullptr);
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
                       [&](void* i, void* o, size_t cnt, anemoi::GenericConsts gc, PermConsts pc) {
                         return anemoi::field_ops(inst->field)->generic_sponge(bytes, i, per_msg, cnt, o, rate, gc, pc,
                                                                               nullptr);
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
  return host_batch(device, n, elems, eb, elems, eb, [&](void* i, void*, size_t cnt, hipStream_t s) -> int {
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
  return host_batch(device, n, states, per, states, per, [&](void* in, void*, size_t cnt, hipStream_t s) {
    return anemoi_permutation_dev(field, width, in, cnt, s);
  });
}

int anemoi_sbox_layer_batch(int field, int width, uint64_t* states, size_t n, int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && !states) return ANEMOI_ERR_ARG;
  const size_t per = elem_bytes(field) * width;
  return host_batch(device, n, states, per, states, per, [&](void* in, void*, size_t cnt, hipStream_t s) {
    return anemoi_sbox_layer_dev(field, width, in, cnt, s);
  });
}

int anemoi_jive_compress_k_batch(int field, int width, int k, const uint64_t* in, uint64_t* out, size_t n,
                                 int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if ((rc = check_k(width, k))) return rc;
  if (n && (!in || !out)) return ANEMOI_ERR_ARG;
  if ((const void*)in == (void*)out) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  return host_batch(device, n, in, eb * width, out, eb * (width / k), [&](void* i, void* o, size_t cnt, hipStream_t s) {
    return anemoi_jive_compress_k_dev(field, width, k, i, o, cnt, s);
  });
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
  const size_t eb = elem_bytes(field);
  static const uint64_t dummy[2] = {0, 0};
  return host_batch(device, n, elems ? (const void*)elems : (const void*)dummy, eb * elems_per_msg, out, eb,
                    [&](void* i, void* o, size_t cnt, hipStream_t s) {
                      return anemoi_hash_field_dev(field, width, i, elems_per_msg, cnt, o, s);
                    });
}

int anemoi_hash_bytes_batch(int field, int width, const uint8_t* msgs, size_t msg_len, size_t n, uint64_t* out,
                            int device) {
  int rc = check_instance(field, width);
  if (rc) return rc;
  if (n && (!out || (msg_len && !msgs))) return ANEMOI_ERR_ARG;
  static const uint64_t dummy[2] = {0, 0};
  return host_batch(device, n, msgs ? (const void*)msgs : (const void*)dummy, msg_len, out, elem_bytes(field),
                    [&](void* i, void* o, size_t cnt, hipStream_t s) {
                      return anemoi_hash_bytes_dev(field, width, i, msg_len, cnt, o, s);
                    });
}

int anemoi_to_montgomery(int field, const uint64_t* in, uint64_t* out, size_t count, int device) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (count && (!in || !out)) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  return host_batch(device, count, in, eb, out, eb, [&](void* i, void* o, size_t cnt, hipStream_t s) {
    return anemoi_to_montgomery_dev(field, i, o, cnt, s);
  });
}

int anemoi_from_montgomery(int field, const uint64_t* in, uint64_t* out, size_t count, int device) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (count && (!in || !out)) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  return host_batch(device, count, in, eb, out, eb, [&](void* i, void* o, size_t cnt, hipStream_t s) {
    return anemoi_from_montgomery_dev(field, i, o, cnt, s);
  });
}

int anemoi_merkle_tree(int field, const uint64_t* leaves, unsigned depth, uint64_t* tree, int device) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (!leaves || !tree || depth > 30) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field), total = (size_t(2) << depth) - 1;
  return for_devices(device == ANEMOI_ALL_DEVICES ? 0 : device, 1, [&](int dev, size_t, size_t) -> int {
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    DevBuf dt;
    int r = dt.alloc(total * eb);
    if (r) return r;
    HIP_TRY(hipMemcpy(dt.p, leaves, (size_t(1) << depth) * eb, hipMemcpyHostToDevice));
    r = anemoi_merkle_tree_dev(field, dt.p, depth, dt.p, nullptr);
    if (r) return r;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(tree, dt.p, total * eb, hipMemcpyDeviceToHost));
    return ANEMOI_OK;
  });
}

/* host-side indexing only: the sibling of node (level l, position index >> l) for l = 0 .. depth-1 */
int anemoi_merkle_path(int field, const uint64_t* tree, unsigned depth, size_t index, uint64_t* path) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (!tree || (depth && !path) || depth > 30 || index >= (size_t(1) << depth)) return ANEMOI_ERR_ARG;
  const size_t L = anemoi::field_ops(field)->limbs64;
  size_t off = 0;
  for (unsigned l = 0; l < depth; l++) {
    const size_t pos = (index >> l) ^ 1;
    memcpy(path + l * L, tree + (off + pos) * L, L * 8);
    off += size_t(1) << (depth - l);
  }
  return ANEMOI_OK;
}

int anemoi_merkle_verify_batch(int field, const uint64_t* leaves, const uint64_t* indices, const uint64_t* paths,
                               unsigned depth, size_t n, const uint64_t* root, uint8_t* ok, int device) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (depth > 63 || (n && (!leaves || !indices || !root || !ok || (depth && !paths)))) return ANEMOI_ERR_ARG;
  if (n == 0) return ANEMOI_OK;
  const size_t eb = elem_bytes(field);
  return for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (!count) return ANEMOI_OK;
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    DevBuf dl, di, dp, dr;
    int r = dl.alloc(count * eb);
    if (!r) r = di.alloc(count * 8);
    if (!r) r = dp.alloc(count * depth * eb);
    if (!r) r = dr.alloc(count * eb);
    if (r) return r;
    HIP_TRY(hipMemcpy(dl.p, (const char*)leaves + first * eb, count * eb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(di.p, indices + first, count * 8, hipMemcpyHostToDevice));
    if (depth)
      HIP_TRY(hipMemcpy(dp.p, (const char*)paths + first * depth * eb, count * depth * eb, hipMemcpyHostToDevice));
    r = anemoi_merkle_climb_dev(field, dl.p, di.p, dp.p, depth, count, dr.p, nullptr);
    if (r) return r;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint64_t> got(count * (eb / 8));
    HIP_TRY(hipMemcpy(got.data(), dr.p, count * eb, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < count; i++) ok[first + i] = memcmp(&got[i * (eb / 8)], root, eb) == 0 ? 1 : 0;
    return ANEMOI_OK;
  });
}

/* Arity-4 tree with the 4-3 instance's Jive-4 (compress_k(.,4), anemoi_4_3/hasher.rs:162-179):
 * 4^depth4 leaf digests -> root, level by level. */
int anemoi_merkle_root_arity4(int field, const uint64_t* leaves, unsigned depth4, uint64_t* root, int device) {
  int rc = check_instance(field, 4);
  if (rc) return rc;
  if (!leaves || !root || depth4 > 15) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field), nleaf = size_t(1) << (2 * depth4);
  if (depth4 == 0) {
    memcpy(root, leaves, eb);
    return ANEMOI_OK;
  }
  return for_devices(device == ANEMOI_ALL_DEVICES ? 0 : device, 1, [&](int dev, size_t, size_t) -> int {
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    DevBuf da, db;
    int r = da.alloc(nleaf * eb);
    if (!r) r = db.alloc(nleaf / 4 * eb);
    if (r) return r;
    HIP_TRY(hipMemcpy(da.p, leaves, nleaf * eb, hipMemcpyHostToDevice));
    PermConsts pc;
    if ((r = get_consts(field, 4, &pc))) return r;
    void *src = da.p, *dst = db.p;
    for (size_t n = nleaf / 4; n >= 1; n /= 4) {
      HIP_TRY(anemoi::field_ops(field)->jive(4, 4, src, dst, n, pc, nullptr));
      std::swap(src, dst);
      if (n == 1) break;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(root, src, eb, hipMemcpyDeviceToHost));
    return ANEMOI_OK;
  });
}

int anemoi_merkle_tree_arity4(int field, const uint64_t* leaves, unsigned depth4, uint64_t* tree, int device) {
  int rc = check_instance(field, 4);
  if (rc) return rc;
  if (!leaves || !tree || depth4 > 15) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field), nleaf = size_t(1) << (2 * depth4), total = ((nleaf << 2) - 1) / 3;
  return for_devices(device == ANEMOI_ALL_DEVICES ? 0 : device, 1, [&](int dev, size_t, size_t) -> int {
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    DevBuf dt;
    int r = dt.alloc(total * eb);
    if (r) return r;
    HIP_TRY(hipMemcpy(dt.p, leaves, nleaf * eb, hipMemcpyHostToDevice));
    PermConsts pc;
    if ((r = get_consts(field, 4, &pc))) return r;
    char* lvl = (char*)dt.p;
    for (size_t n = nleaf / 4; n >= 1; n /= 4) {  // n nodes of the next level from 4 n of this one
      char* next = lvl + 4 * n * eb;
      HIP_TRY(anemoi::field_ops(field)->jive(4, 4, lvl, next, n, pc, nullptr));
      lvl = next;
      if (n == 1) break;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(tree, dt.p, total * eb, hipMemcpyDeviceToHost));
    return ANEMOI_OK;
  });
}

int anemoi_merkle_path_arity4(int field, const uint64_t* tree, unsigned depth4, size_t index, uint64_t* path) {
  if (!anemoi::field_ops(field)) return ANEMOI_ERR_FIELD;
  if (!tree || (depth4 && !path) || depth4 > 15 || index >= (size_t(1) << (2 * depth4))) return ANEMOI_ERR_ARG;
  const size_t L = anemoi::field_ops(field)->limbs64;
  size_t off = 0;
  for (unsigned l = 0; l < depth4; l++) {
    const size_t node = index >> (2 * l), first = node & ~size_t(3);
    int k = 0;
    for (size_t c = 0; c < 4; c++)
      if (first + c != node) memcpy(path + (size_t(l) * 3 + k++) * L, tree + (off + first + c) * L, L * 8);
    off += size_t(1) << (2 * (depth4 - l));
  }
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
  return for_devices(device, n, [&](int dev, size_t first, size_t count) -> int {
    if (!count) return ANEMOI_OK;
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    DevBuf dcur, di, dp, dst;
    int r = dcur.alloc(count * eb);
    if (!r) r = di.alloc(count * 8);
    if (!r) r = dp.alloc(count * depth4 * 3 * eb);
    if (!r) r = dst.alloc(count * 4 * eb);
    if (r) return r;
    HIP_TRY(hipMemcpy(dcur.p, (const char*)leaves + first * eb, count * eb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(di.p, indices + first, count * 8, hipMemcpyHostToDevice));
    if (depth4)
      HIP_TRY(hipMemcpy(dp.p, (const char*)paths + first * depth4 * 3 * eb, count * depth4 * 3 * eb,
                        hipMemcpyHostToDevice));
    PermConsts pc;
    if ((r = get_consts(field, 4, &pc))) return r;
    const size_t threads = count * 4 * size_t(quads);
    for (unsigned l = 0; l < depth4; l++) {
      k_assemble4<<<unsigned((threads + 255) / 256), 256, 0, nullptr>>>((const uint4*)dcur.p, (const uint4*)dp.p,
                                                                        (const uint64_t*)di.p, l, depth4, count, quads,
                                                                        (uint4*)dst.p);
      HIP_TRY(hipGetLastError());
      HIP_TRY(anemoi::field_ops(field)->jive(4, 4, dst.p, dcur.p, count, pc, nullptr));
    }
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint64_t> got(count * (eb / 8));
    HIP_TRY(hipMemcpy(got.data(), dcur.p, count * eb, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < count; i++) ok[first + i] = memcmp(&got[i * (eb / 8)], root, eb) == 0 ? 1 : 0;
    return ANEMOI_OK;
  });
}

int anemoi_merkle_root(int field, const uint64_t* leaves, unsigned depth, uint64_t* root, int device) {
  int rc = check_instance(field, 2);
  if (rc) return rc;
  if (!leaves || !root || depth > 30) return ANEMOI_ERR_ARG;
  const size_t eb = elem_bytes(field);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0) {
    g_last_error = "no HIP device";
    return ANEMOI_ERR_DEVICE;
  }
  // number of subtrees: a power of two, at most the GPU count and at most 2^depth
  unsigned sub_log = 0;
  if (device == ANEMOI_ALL_DEVICES)
    while ((2u << sub_log) <= unsigned(ndev) && sub_log + 1 <= depth) sub_log++;
  const unsigned nsub = 1u << sub_log, sub_depth = depth - sub_log;
  std::vector<uint64_t> tops(size_t(nsub) * (eb / 8));
  // each subtree is one "item" of the sharded loop: subtree i runs on device i (or `device`)
  auto subtree = [&](int dev, size_t idx) -> int {
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(dev));
    const size_t nleaf = size_t(1) << sub_depth;
    DevBuf dl, ds, dr;
    int r = dl.alloc(nleaf * eb);
    if (!r) r = ds.alloc(nleaf * eb);
    if (!r) r = dr.alloc(eb);
    if (r) return r;
    HIP_TRY(hipMemcpy(dl.p, (const char*)leaves + idx * nleaf * eb, nleaf * eb, hipMemcpyHostToDevice));
    r = merkle_levels_dev(field, dl.p, sub_depth, ds.p, dr.p, nullptr);
    if (r) return r;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy((char*)tops.data() + idx * eb, dr.p, eb, hipMemcpyDeviceToHost));
    return ANEMOI_OK;
  };
  if (nsub == 1) {
    const int dev = device == ANEMOI_ALL_DEVICES ? 0 : device;
    if (dev < 0 || dev >= ndev) {
      g_last_error = "device ordinal out of range";
      return ANEMOI_ERR_DEVICE;
    }
    if ((rc = subtree(dev, 0))) return rc;
    memcpy(root, tops.data(), eb);
    return ANEMOI_OK;
  }
  std::vector<int> rcs(nsub, ANEMOI_OK);
  std::vector<std::string> errs(nsub);
  std::vector<std::thread> th;
  for (unsigned i = 0; i < nsub; i++)
    th.emplace_back([&, i] {
      rcs[i] = subtree(int(i), i);
      if (rcs[i]) errs[i] = g_last_error;
    });
  for (auto& t : th) t.join();
  for (unsigned i = 0; i < nsub; i++)
    if (rcs[i]) {
      g_last_error = errs[i];
      return rcs[i];
    }
  // the only cross-GPU data: nsub subtree roots (<= 8 x 48 B), finished on device 0
  return anemoi_merkle_root(field, tops.data(), sub_log, root, 0);
}

}  // extern "C"
