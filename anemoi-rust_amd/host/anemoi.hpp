// anemoi.hpp -- C++ host-side mirror of the reference crate's operator surface for this path,
// implemented on the C-ABI of libanemoi_mi355x.so (include/anemoi_mi355x.h).
//
// The reference is Rust; no Rust toolchain exists in the build image, so the host layer above the
// C-ABI is written in C++ with the reference's names, argument meaning and error behaviour
// (INTEGRATION.md shows the equivalent Rust `-sys` shim):
//
//   reference (src/traits.rs:8-33, src/<field>/anemoi_X_Y/{mod,hasher,digest}.rs)   here
//   ------------------------------------------------------------------------------  ---------------------------
//   pub struct AnemoiBls12_381_2_1; impl Sponge<Felt> + impl Jive<Felt>               anemoi::AnemoiBls12_381_2_1
//   Sponge::hash(&[u8]) / hash_field(&[Felt]) / merge(&[Digest; 2])                   ::hash / ::hash_field / ::merge
//   Jive::compress(&[Felt]) -> Vec<Felt> / compress_k(&[Felt], k)                     ::compress / ::compress_k
//   STATE_WIDTH, RATE_WIDTH, NUM_COLUMNS, DIGEST_SIZE, NUM_HASH_ROUNDS                same-named constants
//   AnemoiDigest::{new, as_elements, to_elements, digests_to_elements, to_bytes}      anemoi::Digest<F>
//   assert!(..) panics (hasher.rs:97,107; 4-3: :149,163-165)                          throw std::invalid_argument
//
// plus the batched forms a GPU needs (compress_batch, hash_batch, ...), which equal the single-item
// functions applied to every item.  `Felt` is the arkworks in-memory form (Montgomery limbs).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/anemoi_mi355x.h"

namespace anemoi {

struct DeviceError : std::runtime_error {
  int code;
  DeviceError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc) {
  if (rc == ANEMOI_OK) return;
  std::string msg = std::string("anemoi_mi355x: ") + anemoi_strerror(rc);
  if (rc == ANEMOI_ERR_ARG) throw std::invalid_argument(msg);  // the reference's assert! panics
  if (rc == ANEMOI_ERR_DEVICE) msg += std::string(" (") + anemoi_last_error() + ")";
  throw DeviceError(rc, msg);
}

template <int LIMBS>
struct Felt {  // = arkworks Fp<MontBackend<_, LIMBS>, LIMBS>: little-endian u64 limbs, Montgomery form
  std::array<uint64_t, LIMBS> limbs{};
  bool operator==(const Felt& o) const { return limbs == o.limbs; }
  bool operator!=(const Felt& o) const { return !(*this == o); }
};

// src/<field>/anemoi_X_Y/digest.rs: AnemoiDigest([Felt; DIGEST_SIZE]), DIGEST_SIZE = 1
template <int FIELD, int LIMBS>
struct Digest {
  static constexpr size_t DIGEST_SIZE = 1;
  std::array<Felt<LIMBS>, 1> elements{};
  Digest() = default;
  explicit Digest(const std::array<Felt<LIMBS>, 1>& e) : elements(e) {}
  const std::array<Felt<LIMBS>, 1>& as_elements() const { return elements; }
  std::array<Felt<LIMBS>, 1> to_elements() const { return elements; }
  static std::array<Felt<LIMBS>, 2> digests_to_elements(const std::array<Digest, 2>& d) {
    return {d[0].elements[0], d[1].elements[0]};
  }
  // digest.rs:42-46: serialize_compressed = canonical little-endian bytes (32 or 48)
  std::array<uint8_t, 8 * LIMBS> to_bytes(int device = 0) const {
    std::array<uint64_t, LIMBS> canon{};
    check(anemoi_from_montgomery(FIELD, elements[0].limbs.data(), canon.data(), 1, device));
    std::array<uint8_t, 8 * LIMBS> out{};
    for (int i = 0; i < LIMBS; i++)
      for (int b = 0; b < 8; b++) out[8 * i + b] = uint8_t(canon[i] >> (8 * b));
    return out;
  }
  bool operator==(const Digest& o) const { return elements == o.elements; }
};

template <int FIELD, int WIDTH, int LIMBS, int ROUNDS>
struct Instance {
  using F = Felt<LIMBS>;
  using D = Digest<FIELD, LIMBS>;
  static constexpr size_t STATE_WIDTH = WIDTH;
  static constexpr size_t RATE_WIDTH = WIDTH - 1;
  static constexpr size_t NUM_COLUMNS = WIDTH / 2;
  static constexpr size_t DIGEST_SIZE = 1;
  static constexpr size_t NUM_HASH_ROUNDS = ROUNDS;

  // ---- Jive (src/traits.rs:23-33) --------------------------------------------------------------
  static std::vector<F> compress(const std::vector<F>& elems, int device = 0) {
    if (elems.size() != STATE_WIDTH) throw std::invalid_argument("compress: elems.len() != STATE_WIDTH");
    return compress_batch(elems, device);
  }
  static std::vector<F> compress_k(const std::vector<F>& elems, size_t k, int device = 0) {
    if (elems.size() != STATE_WIDTH) throw std::invalid_argument("compress_k: elems.len() != STATE_WIDTH");
    return compress_k_batch(elems, k, device);
  }
  // ---- Sponge (src/traits.rs:8-20) -------------------------------------------------------------
  static D hash(const uint8_t* bytes, size_t len, int device = 0) {
    D d;
    check(anemoi_hash_bytes_batch(FIELD, WIDTH, bytes, len, 1, d.elements[0].limbs.data(), device));
    return d;
  }
  static D hash(const std::vector<uint8_t>& bytes, int device = 0) { return hash(bytes.data(), bytes.size(), device); }
  static D hash_field(const std::vector<F>& elems, int device = 0) {
    D d;
    check(anemoi_hash_field_batch(FIELD, WIDTH, elems.empty() ? nullptr : elems[0].limbs.data(), elems.size(), 1,
                                  d.elements[0].limbs.data(), device));
    return d;
  }
  static D merge(const std::array<D, 2>& digests, int device = 0) {
    D d;
    if (WIDTH == 2) {  // anemoi_2_1/hasher.rs:87-92: Jive compression of the two digests
      auto e = D::digests_to_elements(digests);
      check(anemoi_merge_batch(FIELD, e[0].limbs.data(), d.elements[0].limbs.data(), 1, device));
    } else {
      // anemoi_4_3/hasher.rs:131-145 writes digests[0] into BOTH rate cells (line :138 reads
      // digests[0] again); kept bit-for-bit, on top of the batched permutation
      std::array<F, 4> st{};
      st[0] = digests[0].elements[0];
      st[1] = digests[0].elements[0];
      check(anemoi_permutation_batch(FIELD, WIDTH, st[0].limbs.data(), 1, device));
      d.elements[0] = st[0];
    }
    return d;
  }

  // ---- batched forms (new surface; item i of the result == the single-item function on item i) ---
  static std::vector<F> compress_k_batch(const std::vector<F>& states, size_t k, int device = 0) {
    if (states.size() % STATE_WIDTH) throw std::invalid_argument("compress_batch: not a whole number of states");
    if (k == 0 || STATE_WIDTH % k) throw std::invalid_argument("compress_k: STATE_WIDTH % k != 0");
    const size_t n = states.size() / STATE_WIDTH;
    std::vector<F> out(n * (STATE_WIDTH / k));
    check(anemoi_jive_compress_k_batch(FIELD, WIDTH, int(k), n ? states[0].limbs.data() : nullptr,
                                       n ? out[0].limbs.data() : nullptr, n, device));
    return out;
  }
  static std::vector<F> compress_batch(const std::vector<F>& states, int device = 0) {
    return compress_k_batch(states, 2, device);
  }
  static void permutation_batch(std::vector<F>& states, int device = 0) {
    if (states.size() % STATE_WIDTH) throw std::invalid_argument("permutation_batch: not a whole number of states");
    check(anemoi_permutation_batch(FIELD, WIDTH, states.empty() ? nullptr : states[0].limbs.data(),
                                   states.size() / STATE_WIDTH, device));
  }
  static std::vector<D> hash_batch(const uint8_t* msgs, size_t msg_len, size_t n, int device = 0) {
    std::vector<D> out(n);
    static_assert(sizeof(D) == sizeof(uint64_t) * LIMBS, "Digest must be layout-compatible with one element");
    check(anemoi_hash_bytes_batch(FIELD, WIDTH, msgs, msg_len, n, n ? out[0].elements[0].limbs.data() : nullptr, device));
    return out;
  }
  // Sponge::hash of each of `msgs` (any lengths) in one launch: item i == hash(msgs[i])
  static std::vector<D> hash_ragged(const std::vector<std::vector<uint8_t>>& msgs, int device = 0) {
    std::vector<uint64_t> off(msgs.size() + 1, 0);
    for (size_t i = 0; i < msgs.size(); i++) off[i + 1] = off[i] + msgs[i].size();
    std::vector<uint8_t> blob;
    blob.reserve(off.back());
    for (const auto& m : msgs) blob.insert(blob.end(), m.begin(), m.end());
    std::vector<D> out(msgs.size());
    check(anemoi_hash_bytes_ragged_batch(FIELD, WIDTH, blob.empty() ? nullptr : blob.data(), off.data(), msgs.size(),
                                         msgs.empty() ? nullptr : out[0].elements[0].limbs.data(), device));
    return out;
  }
  // Sponge::hash_field of each of `msgs` (any numbers of elements) in one launch: item i == hash_field(msgs[i])
  static std::vector<D> hash_field_ragged(const std::vector<std::vector<F>>& msgs, int device = 0) {
    std::vector<uint64_t> off(msgs.size() + 1, 0);
    for (size_t i = 0; i < msgs.size(); i++) off[i + 1] = off[i] + msgs[i].size();
    std::vector<F> blob;
    blob.reserve(off.back());
    for (const auto& m : msgs) blob.insert(blob.end(), m.begin(), m.end());
    std::vector<D> out(msgs.size());
    static_assert(sizeof(F) == sizeof(uint64_t) * LIMBS, "an element is its limbs");
    check(anemoi_hash_field_ragged_batch(FIELD, WIDTH, blob.empty() ? nullptr : blob[0].limbs.data(), off.data(), msgs.size(),
                                         msgs.empty() ? nullptr : out[0].elements[0].limbs.data(), device));
    return out;
  }
  // n pairs of digests -> n digests, item i == merge(pairs[i]) (the 4-3 form keeps the reference's behaviour, see merge)
  static std::vector<D> merge_batch(const std::vector<std::array<D, 2>>& pairs, int device = 0) {
    std::vector<D> out(pairs.size());
    if (pairs.empty()) return out;
    if (WIDTH == 2) {
      static_assert(sizeof(std::array<D, 2>) == 2 * sizeof(uint64_t) * LIMBS, "pairs must be contiguous elements");
      check(anemoi_merge_batch(FIELD, pairs[0][0].elements[0].limbs.data(), out[0].elements[0].limbs.data(), pairs.size(),
                               device));
    } else {
      std::vector<F> st(pairs.size() * 4);
      for (size_t i = 0; i < pairs.size(); i++) st[4 * i] = st[4 * i + 1] = pairs[i][0].elements[0];
      check(anemoi_permutation_batch(FIELD, WIDTH, st[0].limbs.data(), pairs.size(), device));
      for (size_t i = 0; i < pairs.size(); i++) out[i].elements[0] = st[4 * i];
    }
    return out;
  }
  // optional warm-up / release of the library's per-device state (constant tables, lanes)
  static void init(int device = ANEMOI_ALL_DEVICES) { check(anemoi_init(device, FIELD, WIDTH)); }
  static void release(int device = ANEMOI_ALL_DEVICES) { check(anemoi_release(device)); }

  static D merkle_root(const std::vector<D>& leaves, unsigned depth, int device = 0) {
    static_assert(WIDTH == 2, "the Merkle driver uses the 2-1 instance's merge");
    if (leaves.size() != (size_t(1) << depth)) throw std::invalid_argument("merkle_root: need 2^depth leaves");
    D root;
    check(anemoi_merkle_root(FIELD, leaves[0].elements[0].limbs.data(), depth, root.elements[0].limbs.data(), device));
    return root;
  }

  // canonical integers (little-endian u64 limbs) <-> Felt, e.g. for MontFp!-style constants
  static F from_canonical(const std::array<uint64_t, LIMBS>& c, int device = 0) {
    F f;
    check(anemoi_to_montgomery(FIELD, c.data(), f.limbs.data(), 1, device));
    return f;
  }
  static std::array<uint64_t, LIMBS> to_canonical(const F& f, int device = 0) {
    std::array<uint64_t, LIMBS> c{};
    check(anemoi_from_montgomery(FIELD, f.limbs.data(), c.data(), 1, device));
    return c;
  }
};

// An instance described by the `Anemoi` trait's constants (src/traits.rs:36-76) instead of by a type:
// what a maintainer adding e.g. an Anemoi-8-7 instance would fill in.  `mds` empty = the reference's
// hard-coded mds_layer arm (NUM_COLUMNS <= 6, src/traits.rs:136-279).
template <int FIELD, int LIMBS>
struct GenericInstance {
  using F = Felt<LIMBS>;
  size_t num_columns, num_rounds;
  std::vector<F> ark_c, ark_d, mds;

  size_t state_width() const { return 2 * num_columns; }
  anemoi_generic_instance raw() const {
    if (ark_c.size() != num_columns * num_rounds || ark_d.size() != ark_c.size() ||
        (!mds.empty() && mds.size() != num_columns * num_columns))
      throw std::invalid_argument("GenericInstance: constant tables do not match NUM_COLUMNS / NUM_ROUNDS");
    return anemoi_generic_instance{FIELD, int(num_columns), int(num_rounds),
                                   ark_c.empty() ? nullptr : ark_c[0].limbs.data(),
                                   ark_d.empty() ? nullptr : ark_d[0].limbs.data(),
                                   mds.empty() ? nullptr : mds[0].limbs.data()};
  }
  void permutation_batch(std::vector<F>& states, int device = 0) const {
    if (states.size() % state_width()) throw std::invalid_argument("permutation_batch: not a whole number of states");
    const auto inst = raw();
    check(anemoi_generic_permutation_batch(&inst, states.empty() ? nullptr : states[0].limbs.data(),
                                           states.size() / state_width(), device));
  }
  std::vector<F> compress_k_batch(const std::vector<F>& states, size_t k, int device = 0) const {
    if (states.size() % state_width()) throw std::invalid_argument("compress_k_batch: not a whole number of states");
    if (k == 0 || state_width() % k) throw std::invalid_argument("compress_k: STATE_WIDTH % k != 0");
    const size_t n = states.size() / state_width();
    std::vector<F> out(n * (state_width() / k));
    const auto inst = raw();
    check(anemoi_generic_jive_compress_k_batch(&inst, int(k), n ? states[0].limbs.data() : nullptr,
                                               n ? out[0].limbs.data() : nullptr, n, device));
    return out;
  }
  Digest<FIELD, LIMBS> hash_field(const std::vector<F>& elems, size_t rate, int device = 0) const {
    Digest<FIELD, LIMBS> d;
    const auto inst = raw();
    check(anemoi_generic_hash_field_batch(&inst, int(rate), elems.empty() ? nullptr : elems[0].limbs.data(), elems.size(),
                                          1, d.elements[0].limbs.data(), device));
    return d;
  }
};

// A run-time instance whose constants live on one device in the kernels' own form (anemoi_generic_prepare): the
// device-pointer face of GenericInstance.  Buffers are raw device pointers (the caller's hipMalloc / tensor), calls are
// asynchronous on `stream` (a hipStream_t as void*); the caller's current device must be `device`.
template <int FIELD, int LIMBS>
class PreparedGeneric {
 public:
  PreparedGeneric(const GenericInstance<FIELD, LIMBS>& g, int device = 0) : width_(g.state_width()) {
    const auto inst = g.raw();
    check(anemoi_generic_prepare(&inst, device, &h_));
  }
  ~PreparedGeneric() { (void)anemoi_generic_destroy(h_); }
  PreparedGeneric(const PreparedGeneric&) = delete;
  PreparedGeneric& operator=(const PreparedGeneric&) = delete;

  size_t state_width() const { return width_; }
  void permutation_dev(void* d_states, size_t n, void* stream = nullptr) const {
    check(anemoi_generic_permutation_dev(h_, d_states, n, stream));
  }
  void compress_k_dev(size_t k, const void* d_in, void* d_out, size_t n, void* stream = nullptr) const {
    if (k == 0 || width_ % k) throw std::invalid_argument("compress_k: STATE_WIDTH % k != 0");
    check(anemoi_generic_jive_compress_k_dev(h_, int(k), d_in, d_out, n, stream));
  }
  void hash_field_dev(size_t rate, const void* d_elems, size_t elems_per_msg, size_t n, void* d_out, void* stream = nullptr) const {
    check(anemoi_generic_hash_field_dev(h_, int(rate), d_elems, elems_per_msg, n, d_out, stream));
  }
  void hash_bytes_dev(size_t rate, const void* d_msgs, size_t msg_len, size_t n, void* d_out, void* stream = nullptr) const {
    check(anemoi_generic_hash_bytes_dev(h_, int(rate), d_msgs, msg_len, n, d_out, stream));
  }

 private:
  anemoi_generic_handle* h_ = nullptr;
  size_t width_;
};

// the reference's 14 unit structs (src/<field>/anemoi_X_Y/mod.rs:37-38); rounds from :31
using AnemoiBls12_381_2_1 = Instance<ANEMOI_BLS12_381, 2, 6, 21>;
using AnemoiBls12_381_4_3 = Instance<ANEMOI_BLS12_381, 4, 6, 14>;
using AnemoiBls12_377_2_1 = Instance<ANEMOI_BLS12_377, 2, 6, 21>;
using AnemoiBls12_377_4_3 = Instance<ANEMOI_BLS12_377, 4, 6, 14>;
using AnemoiBn254_2_1 = Instance<ANEMOI_BN_254, 2, 4, 21>;
using AnemoiBn254_4_3 = Instance<ANEMOI_BN_254, 4, 4, 14>;
using AnemoiEdOnBls12_377_2_1 = Instance<ANEMOI_ED_ON_BLS12_377, 2, 4, 19>;
using AnemoiEdOnBls12_377_4_3 = Instance<ANEMOI_ED_ON_BLS12_377, 4, 4, 13>;
using AnemoiJubjub_2_1 = Instance<ANEMOI_JUBJUB, 2, 4, 21>;
using AnemoiJubjub_4_3 = Instance<ANEMOI_JUBJUB, 4, 4, 14>;
using AnemoiPallas_2_1 = Instance<ANEMOI_PALLAS, 2, 4, 21>;
using AnemoiPallas_4_3 = Instance<ANEMOI_PALLAS, 4, 4, 14>;
using AnemoiVesta_2_1 = Instance<ANEMOI_VESTA, 2, 4, 21>;
using AnemoiVesta_4_3 = Instance<ANEMOI_VESTA, 4, 4, 14>;

}  // namespace anemoi
