// test_reference_style.cpp -- the reference's hasher tests, restated against the C++ host mirror
// (anemoi-rust_amd/host/anemoi.hpp) running on the GPU.  Mirrors, per instance:
//   test_anemoi_hash        src/<f>/anemoi_X_Y/hasher.rs  (hash_field on the 10 KAT inputs)
//   test_anemoi_hash_bytes  (hash on the 4 structured inputs packed as full chunks)
//   test_anemoi_jive        (compress, compress_k(.,2), merge on 2-1, compress_k(.,4) on 4-3,
//                            and the assert! on a wrong k / wrong length)
// plus the run-time instances of tests/golden/generic.json through GenericInstance (kind "generic").
// Vectors come from tests/golden/kats.json, flattened by tests/test_cpp_host.py into a text file:
//   <field_id> <width> <kind> <n_in> <n_out> <decimal>...      (kind hash_bytes: one hex string as input)
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../anemoi-rust_amd/host/anemoi.hpp"

using namespace anemoi;

template <size_t L>
static std::array<uint64_t, L> parse_dec(const std::string& s) {  // MontFp!("...") analogue: decimal -> limbs
  std::array<uint64_t, L> v{};
  for (char ch : s) {
    unsigned __int128 carry = unsigned(ch - '0');
    for (size_t i = 0; i < L; i++) {
      unsigned __int128 t = (unsigned __int128)v[i] * 10 + carry;
      v[i] = (uint64_t)t;
      carry = t >> 64;
    }
  }
  return v;
}

static int failures = 0;
#define EXPECT(cond, what)                                                     \
  do {                                                                         \
    if (!(cond)) {                                                             \
      failures++;                                                              \
      std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, what);                \
    }                                                                          \
  } while (0)

template <class I>
static void run_line(const std::string& kind, const std::vector<std::string>& in, const std::vector<std::string>& out) {
  using F = typename I::F;
  constexpr size_t L = sizeof(F) / 8;
  auto felt = [](const std::string& d) { return I::from_canonical(parse_dec<L>(d)); };
  std::vector<F> expected;
  for (auto& o : out) expected.push_back(felt(o));
  if (kind == "hash_field") {
    std::vector<F> input;
    for (auto& s : in) input.push_back(felt(s));
    EXPECT(I::hash_field(input).to_elements()[0] == expected[0], "hash_field");
  } else if (kind == "hash_bytes") {
    std::vector<uint8_t> bytes;
    for (size_t i = 0; i + 1 < in[0].size(); i += 2) bytes.push_back(uint8_t(std::stoi(in[0].substr(i, 2), nullptr, 16)));
    EXPECT(I::hash(bytes).to_elements()[0] == expected[0], "hash(bytes)");
  } else if (kind == "jive") {
    std::vector<F> input;
    for (auto& s : in) input.push_back(felt(s));
    EXPECT(I::compress(input) == expected, "compress");
    EXPECT(I::compress_k(input, 2) == expected, "compress_k(.,2)");
    if (I::STATE_WIDTH == 2) {
      typename I::D a, b;
      a.elements[0] = input[0];
      b.elements[0] = input[1];
      EXPECT(I::merge({a, b}).to_elements()[0] == expected[0], "merge");
    }
    // the reference's assert!s become exceptions
    bool threw = false;
    try { I::compress_k(input, 3); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw, "compress_k(.,3) must be rejected");
    threw = false;
    try { input.pop_back(); I::compress(input); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw, "compress on a short slice must be rejected");
  } else if (kind == "jive_k4") {
    std::vector<F> input;
    for (auto& s : in) input.push_back(felt(s));
    EXPECT(I::compress_k(input, 4) == expected, "compress_k(.,4)");
  }
}

// the three HIP runtime entry points the prepared-handle case needs (this file is built by g++, without the HIP headers)
extern "C" {
int hipMalloc(void** ptr, size_t size);
int hipFree(void* ptr);
int hipMemcpy(void* dst, const void* src, size_t size, int kind);
}

// a run-time instance (tests/golden/generic.json): in = rounds, has_mds, ARK_C, ARK_D, [MDS], state
template <int FIELD, int L>
static void run_generic(size_t cols, const std::vector<std::string>& in, const std::vector<std::string>& out) {
  using I = Instance<FIELD, 2, L, 1>;  // only for from_canonical
  auto felt = [](const std::string& d) { return I::from_canonical(parse_dec<L>(d)); };
  GenericInstance<FIELD, L> g;
  g.num_columns = cols;
  g.num_rounds = size_t(std::stoul(in[0]));
  const bool has_mds = in[1] == "1";
  size_t pos = 2;
  for (size_t i = 0; i < cols * g.num_rounds; i++) g.ark_c.push_back(felt(in[pos++]));
  for (size_t i = 0; i < cols * g.num_rounds; i++) g.ark_d.push_back(felt(in[pos++]));
  if (has_mds)
    for (size_t i = 0; i < cols * cols; i++) g.mds.push_back(felt(in[pos++]));
  std::vector<Felt<L>> st, expected;
  for (size_t i = 0; i < 2 * cols; i++) st.push_back(felt(in[pos++]));
  for (auto& o : out) expected.push_back(felt(o));
  const std::vector<Felt<L>> input = st;
  g.permutation_batch(st);
  EXPECT(st == expected, "generic permutation");
  {  // the same instance prepared once, on device pointers (the HIP runtime's C entry points, declared below)
    PreparedGeneric<FIELD, L> prep(g, 0);
    const size_t bytes = input.size() * sizeof(Felt<L>);
    void* d = nullptr;
    EXPECT(hipMalloc(&d, bytes) == 0, "hipMalloc");
    EXPECT(hipMemcpy(d, input.data(), bytes, 1 /* hipMemcpyHostToDevice */) == 0, "hipMemcpy H2D");
    prep.permutation_dev(d, 1);
    std::vector<Felt<L>> got(input.size());
    EXPECT(hipMemcpy(got.data(), d, bytes, 2 /* hipMemcpyDeviceToHost */) == 0, "hipMemcpy D2H");   // synchronises
    EXPECT(got == expected, "generic permutation through a prepared handle");
    (void)hipFree(d);
  }
  bool threw = false;
  try { g.compress_k_batch(expected, 2 * cols + 2); } catch (const std::invalid_argument&) { threw = true; }
  EXPECT(threw, "compress_k with k > STATE_WIDTH must be rejected");
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream f(argv[1]);
  std::string line;
  int lines = 0;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    int field, width, nin, nout;
    std::string kind;
    ss >> field >> width >> kind >> nin >> nout;
    std::vector<std::string> in(nin), out(nout);
    for (auto& s : in) ss >> s;
    for (auto& s : out) ss >> s;
    lines++;
    if (kind == "generic") {
#define GCASE(FID, L) if (field == FID) run_generic<FID, L>(size_t(width), in, out);
      GCASE(0, 6) GCASE(1, 6) GCASE(2, 4) GCASE(3, 4) GCASE(4, 4) GCASE(5, 4) GCASE(6, 4)
      continue;
    }
#define CASE(FID, W, T) if (field == FID && width == W) run_line<T>(kind, in, out);
    CASE(0, 2, AnemoiBls12_381_2_1) CASE(0, 4, AnemoiBls12_381_4_3) CASE(1, 2, AnemoiBls12_377_2_1)
    CASE(1, 4, AnemoiBls12_377_4_3) CASE(2, 2, AnemoiBn254_2_1) CASE(2, 4, AnemoiBn254_4_3)
    CASE(3, 2, AnemoiEdOnBls12_377_2_1) CASE(3, 4, AnemoiEdOnBls12_377_4_3) CASE(4, 2, AnemoiJubjub_2_1)
    CASE(4, 4, AnemoiJubjub_4_3) CASE(5, 2, AnemoiPallas_2_1) CASE(5, 4, AnemoiPallas_4_3)
    CASE(6, 2, AnemoiVesta_2_1) CASE(6, 4, AnemoiVesta_4_3)
  }
  // the batched forms reduce to the single-item functions: ragged hash, merge_batch (2-1 and the 4-3 form)
  {
    using I = AnemoiBn254_4_3;
    I::init(0);
    std::vector<std::vector<uint8_t>> msgs = {{}, {1}, std::vector<uint8_t>(93, 7), std::vector<uint8_t>(200, 9)};
    for (size_t i = 0; i < msgs[3].size(); i++) msgs[3][i] = uint8_t(i * 31 + 5);
    auto got = I::hash_ragged(msgs);
    for (size_t i = 0; i < msgs.size(); i++) EXPECT(got[i] == I::hash(msgs[i]), "hash_ragged item");
    // ... and hash_field over messages of 0, 1, 2, 3, 4, 7 elements (the digests above serve as elements)
    std::vector<std::vector<I::F>> emsgs;
    for (size_t k : {0, 1, 2, 3, 4, 7}) {
      std::vector<I::F> m;
      for (size_t i = 0; i < k; i++) m.push_back(got[(i + k) % got.size()].elements[0]);
      emsgs.push_back(m);
    }
    auto egot = I::hash_field_ragged(emsgs);
    for (size_t i = 0; i < emsgs.size(); i++) EXPECT(egot[i] == I::hash_field(emsgs[i]), "hash_field_ragged item");
    using J = AnemoiJubjub_2_1;
    std::vector<std::array<J::D, 2>> pairs(5);
    for (size_t i = 0; i < pairs.size(); i++) {
      pairs[i][0] = J::hash(std::vector<uint8_t>(10 + i, uint8_t(i)));
      pairs[i][1] = J::hash(std::vector<uint8_t>(40 + i, uint8_t(3 * i)));
    }
    auto mb = J::merge_batch(pairs);
    for (size_t i = 0; i < pairs.size(); i++) EXPECT(mb[i] == J::merge(pairs[i]), "merge_batch 2-1 item");
    std::vector<std::array<I::D, 2>> pairs4(3);
    for (size_t i = 0; i < pairs4.size(); i++) pairs4[i] = {got[i], got[i + 1]};
    auto mb4 = I::merge_batch(pairs4);
    for (size_t i = 0; i < pairs4.size(); i++) EXPECT(mb4[i] == I::merge(pairs4[i]), "merge_batch 4-3 item");
  }
  // The symbols the Rust patch binds (integration/rust/reference-patch/mi355x.rs), called the way the patch calls
  // them: `&[Felt]` as one contiguous buffer of L u64 limbs per element cast to `*const u64` / `*mut u64`, lengths as
  // size_t, at the patch's small-batch boundary MI355X_MIN_BATCH = 32 and one item either side of it.  Every item must
  // equal the single-item trait function (what the CPU branch of the patch computes below the boundary).
  {
    using I = AnemoiBls12_381_2_1;
    using J = AnemoiBn254_4_3;
    static_assert(sizeof(I::F) == 6 * 8 && alignof(I::F) == 8 && sizeof(J::F) == 4 * 8, "Felt = L u64 limbs, as the patch asserts");
    // (and the sixth: `H::mi355x_warmup()` = anemoi_warmup(ANEMOI_ALL_DEVICES, FIELD, WIDTH), once at start-up; no result changes)
    EXPECT(anemoi_warmup(ANEMOI_ALL_DEVICES, ANEMOI_BLS12_381, 2) == 0 && anemoi_warmup(ANEMOI_ALL_DEVICES, ANEMOI_BN_254, 4) == 0, "anemoi_warmup");
    EXPECT(anemoi_warmup(0, 99, 2) == ANEMOI_ERR_FIELD && anemoi_warmup(0, ANEMOI_BN_254, 3) == ANEMOI_ERR_WIDTH, "anemoi_warmup rejects");
    for (size_t n : {size_t(31), size_t(32), size_t(33)}) {
      std::vector<I::F> st(2 * n), out(n);
      for (size_t i = 0; i < st.size(); i++) st[i] = I::hash(std::vector<uint8_t>(1 + i % 7, uint8_t(i))).to_elements()[0];
      EXPECT(anemoi_jive_compress_k_batch(ANEMOI_BLS12_381, 2, 2, (const uint64_t*)st.data(), (uint64_t*)out.data(), n, 0) == 0,
             "anemoi_jive_compress_k_batch");
      EXPECT(out[n - 1] == I::compress({st[2 * n - 2], st[2 * n - 1]})[0] && out[0] == I::compress({st[0], st[1]})[0], "compress_k_batch item");
      std::vector<I::F> mo(n);
      EXPECT(anemoi_merge_batch(ANEMOI_BLS12_381, (const uint64_t*)st.data(), (uint64_t*)mo.data(), n, 0) == 0, "anemoi_merge_batch");
      EXPECT(mo == out, "merge = compress on Anemoi-2-1 (hasher.rs:86-92)");
      std::vector<J::F> st4(4 * n), before;
      for (size_t i = 0; i < st4.size(); i++) st4[i] = J::hash(std::vector<uint8_t>(2 + i % 5, uint8_t(3 * i))).to_elements()[0];
      before = st4;
      EXPECT(anemoi_permutation_batch(ANEMOI_BN_254, 4, (uint64_t*)st4.data(), n, 0) == 0, "anemoi_permutation_batch");
      std::vector<J::F> one(before.begin() + 4 * (n - 1), before.end());
      J::permutation_batch(one);
      EXPECT(std::vector<J::F>(st4.begin() + 4 * (n - 1), st4.end()) == one, "permutation_batch item");
      const size_t msg_len = 77;
      std::vector<uint8_t> bytes(n * msg_len);
      for (size_t i = 0; i < bytes.size(); i++) bytes[i] = uint8_t(i * 131 + 7);
      std::vector<J::F> dig(n);
      EXPECT(anemoi_hash_bytes_batch(ANEMOI_BN_254, 4, bytes.data(), msg_len, n, (uint64_t*)dig.data(), 0) == 0, "anemoi_hash_bytes_batch");
      EXPECT(dig[n - 1] == J::hash(std::vector<uint8_t>(bytes.end() - msg_len, bytes.end())).to_elements()[0], "hash_batch item");
      const size_t m = 5;
      std::vector<J::F> hf(n);
      EXPECT(anemoi_hash_field_batch(ANEMOI_BN_254, 4, (const uint64_t*)before.data(), m, (4 * n) / m, (uint64_t*)hf.data(), 0) == 0,
             "anemoi_hash_field_batch");
      EXPECT(hf[0] == J::hash_field(std::vector<J::F>(before.begin(), before.begin() + m)).to_elements()[0], "hash_field_batch item");
    }
  }
  // digest_elements / to_bytes (digest.rs:66-88): the zero digest serialises to zero bytes
  AnemoiBls12_381_2_1::D zero;
  for (auto b : zero.to_bytes()) EXPECT(b == 0, "to_bytes(zero)");
  static_assert(AnemoiBn254_4_3::RATE_WIDTH == 3 && AnemoiBn254_4_3::NUM_COLUMNS == 2, "sizes");
  static_assert(AnemoiEdOnBls12_377_2_1::NUM_HASH_ROUNDS == 19, "rounds");
  std::printf("%d vector lines, %d failures\n", lines, failures);
  return failures ? 1 : 0;
}
