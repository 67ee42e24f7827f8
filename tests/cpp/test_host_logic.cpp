// test_host_logic.cpp -- the pure host side of the C-ABI (anemoi-rust_amd/csrc/host_logic.h) on the CPU,
// built with -fsanitize=address,undefined by tools/sanitize_host.sh / tests/test_host_logic.py.
// Covers: shard ranges (coverage, balance, no overflow), subtree planning, retained-tree layout and the
// authentication-path indexing of both arities (against a brute-force tree of integers), overlap checks,
// the chunk plan, the small-integer MDS matrices (against the reference's matrices, src/traits.rs:161-279)
// and the compress_k argument rules (anemoi_2_1/hasher.rs:107, anemoi_4_3/hasher.rs:163-165).
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "../../anemoi-rust_amd/csrc/host_logic.h"

using namespace anemoi::host;

#define CHECK(c)                                                       \
  do {                                                                 \
    if (!(c)) {                                                        \
      std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                    \
    }                                                                  \
  } while (0)

static void test_shards() {
  for (size_t n : {size_t(0), size_t(1), size_t(7), size_t(8), size_t(1000), size_t(1) << 24, (size_t(1) << 62) + 12345})
    for (size_t parts : {size_t(1), size_t(2), size_t(3), size_t(8), size_t(64), size_t(1000)}) {
      CHECK(shard_begin(n, 0, parts) == 0 && shard_begin(n, parts, parts) == n);
      size_t lo = ~size_t(0), hi = 0;
      for (size_t i = 0; i < parts; i++) {
        const size_t b = shard_begin(n, i, parts), e = shard_begin(n, i + 1, parts);
        CHECK(b <= e);
        lo = e - b < lo ? e - b : lo;
        hi = e - b > hi ? e - b : hi;
      }
      CHECK(hi - lo <= 1);  // balanced to +-1
    }
  // agrees with the plain formula where that does not overflow
  for (size_t n = 0; n < 300; n++)
    for (size_t parts = 1; parts < 20; parts++)
      for (size_t i = 0; i <= parts; i++) CHECK(shard_begin(n, i, parts) == n * i / parts);
}

static void test_subtrees() {
  CHECK(subtree_levels(24, 1, 8) == 3 && subtree_levels(24, 1, 7) == 2 && subtree_levels(24, 1, 1) == 0);
  CHECK(subtree_levels(2, 1, 8) == 2 && subtree_levels(0, 1, 8) == 0 && subtree_levels(1, 1, 1000) == 1);
  // arity 4: 8 workers take 16 subtrees (2 each) rather than 4 (half of them idle); 2 workers take 4; 3 stay at 1
  CHECK(subtree_levels(10, 2, 8) == 2 && subtree_levels(10, 2, 16) == 2 && subtree_levels(10, 2, 3) == 0);
  CHECK(subtree_levels(10, 2, 2) == 1 && subtree_levels(10, 2, 4) == 1 && subtree_levels(10, 2, 5) == 1);
  CHECK(subtree_levels(1, 2, 64) == 1 && subtree_levels(1, 2, 8) == 1 && subtree_levels(0, 2, 8) == 0);
  CHECK(subtree_levels(24, 1, 6) == 2 && subtree_levels(24, 1, 3) == 1);   // binary: 8 % 6 != 0, 4 % 3 != 0
}

// brute-force trees over integers: node value = a hash-free stand-in (position-coded), only indices matter
static void test_paths() {
  for (unsigned depth = 0; depth <= 6; depth++) {
    const size_t L = 3, total = tree2_total(depth);
    CHECK(tree2_level_offset(depth, 0) == 0 && tree2_level_offset(depth, depth) == total - 1);
    std::vector<uint64_t> tree(total * L);
    for (unsigned l = 0; l <= depth; l++)
      for (size_t j = 0; j < (size_t(1) << (depth - l)); j++)
        for (size_t w = 0; w < L; w++) tree[(tree2_level_offset(depth, l) + j) * L + w] = (uint64_t(l) << 32) | (j << 4) | w;
    for (size_t index = 0; index < (size_t(1) << depth); index++) {
      std::vector<uint64_t> path(depth * L + 1, 0xdeadbeef);
      merkle_path2(tree.data(), depth, index, L, path.data());
      for (unsigned l = 0; l < depth; l++)
        for (size_t w = 0; w < L; w++) CHECK(path[l * L + w] == ((uint64_t(l) << 32) | (((index >> l) ^ 1) << 4) | w));
      CHECK(path[depth * L] == 0xdeadbeef);  // nothing written past the path
    }
  }
  for (unsigned d4 = 0; d4 <= 3; d4++) {
    const size_t L = 2, total = tree4_total(d4);
    CHECK(tree4_level_offset(d4, d4) == total - 1);
    std::vector<uint64_t> tree(total * L);
    for (unsigned l = 0; l <= d4; l++)
      for (size_t j = 0; j < (size_t(1) << (2 * (d4 - l))); j++)
        for (size_t w = 0; w < L; w++) tree[(tree4_level_offset(d4, l) + j) * L + w] = (uint64_t(l) << 32) | (j << 4) | w;
    for (size_t index = 0; index < (size_t(1) << (2 * d4)); index++) {
      std::vector<uint64_t> path(d4 * 3 * L + 1, 0xdeadbeef);
      merkle_path4(tree.data(), d4, index, L, path.data());
      for (unsigned l = 0; l < d4; l++) {
        const size_t node = index >> (2 * l);
        int k = 0;
        for (size_t c = 0; c < 4; c++) {
          const size_t sib = (node & ~size_t(3)) + c;
          if (sib == node) continue;
          for (size_t w = 0; w < L; w++) CHECK(path[(l * 3 + k) * L + w] == ((uint64_t(l) << 32) | (sib << 4) | w));
          k++;
        }
        CHECK(k == 3);
      }
      CHECK(path[d4 * 3 * L] == 0xdeadbeef);
    }
  }
}

static void test_overlap_and_chunks() {
  char buf[256];
  CHECK(ranges_overlap(buf, 64, buf, 64) && ranges_overlap(buf, 64, buf + 63, 1) && !ranges_overlap(buf, 64, buf + 64, 64));
  CHECK(!ranges_overlap(buf, 0, buf, 64) && !ranges_overlap(buf + 10, 5, buf, 10) && ranges_overlap(buf + 10, 5, buf, 11));
  // chunk plan: sizes are multiples of the quantum, cover n, and small batches are not cut
  for (size_t q : {size_t(1), size_t(196608), size_t(262144)})
    for (size_t n : {size_t(0), size_t(1), q, 2 * q - 1, 2 * q, 5 * q + 17, size_t(1) << 24})
      for (size_t ipi : {size_t(32), size_t(96), size_t(10240)}) {
        const ChunkPlan cp = plan_chunks(n, q, ipi, size_t(24) << 20);
        if (n == 0) { CHECK(cp.chunks == 0); continue; }
        CHECK(cp.chunks >= 1 && cp.chunk_items >= 1);
        CHECK((cp.chunks - 1) * cp.chunk_items < n && cp.chunks * cp.chunk_items >= n);
        if (n < 2 * q) CHECK(cp.chunks == 1 && cp.chunk_items == n);
        else CHECK(cp.chunk_items % q == 0);
      }
  CHECK(plan_chunks(1000, 0, 8, 1 << 20).chunks >= 1);  // quantum 0 is treated as 1
}

// ragged chunk plan against its contract on random offset tables (incl. empty messages, one huge message)
static void test_ragged_chunks() {
  uint64_t seed = 12345;
  auto rnd = [&]() { return seed = seed * 6364136223846793005ull + 1442695040888963407ull, seed >> 33; };
  for (int trial = 0; trial < 600; trial++) {
    const size_t n = rnd() % 700;
    const size_t target = 1 + rnd() % 5000, align = trial % 3 == 0 ? 1 : 64, max_items = 64 + rnd() % 400;
    const size_t min_items = rnd() % 200, max_bytes = target * (1 + rnd() % 8);
    std::vector<uint64_t> off(n + 1);
    off[0] = rnd() % 100;   // offsets need not start at 0
    for (size_t i = 0; i < n; i++) {
      const unsigned kind = rnd() % 10;
      const uint64_t len = kind == 0 ? 0 : kind == 1 ? rnd() % 20000 : rnd() % 200;
      off[i + 1] = off[i] + len;
    }
    const std::vector<size_t> cuts = plan_ragged_chunks(off.data(), n, target, align, min_items, max_items, max_bytes);
    const size_t cap_items = max_items < align ? align : max_items;
    const size_t want_items = min_items > cap_items ? cap_items : min_items;
    CHECK(!cuts.empty() && cuts.back() == n && (n == 0 ? cuts.size() == 1 : cuts.front() == 0));
    for (size_t c = 0; c + 1 < cuts.size(); c++) {
      const size_t a = cuts[c], b = cuts[c + 1];
      CHECK(a < b);                                              // non-empty, increasing
      CHECK(b - a <= cap_items);                                 // bounded offsets table
      const uint64_t bytes = off[b] - off[a];
      CHECK(bytes <= max_bytes || b - a == 1);                   // bounded staging, or one long message alone
      if (b - a > align) CHECK((b - a) % align == 0);            // whole wavefronts
      // beyond the byte target only to reach the minimum message count
      if (bytes > target && b - a > 1) CHECK(b - a <= want_items + align);
      // greedy: a chunk that stopped early (not at the end, not at a cap) could not have taken `align` more messages
      if (b + align <= n && b - a + align <= cap_items && b - a >= align) {
        const uint64_t more = off[b + align] - off[a];
        CHECK(more > max_bytes || (more > target && b - a + align > want_items));
      }
    }
  }
}

static void test_ragged_order() {
  // the order is a permutation by descending block count, stable, and is only produced when it pays
  uint64_t seed = 99;
  auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return seed >> 33; };
  for (int trial = 0; trial < 300; trial++) {
    const size_t n = rnd() % 700, per_wave = (rnd() % 2) ? 64 : 32, block = 31 * (1 + rnd() % 3);
    std::vector<uint64_t> off(n + 1, 0);
    // 0: equal lengths, 1: already sorted, 2: long-tailed, 3: uniform, 4: one HUGE message among small ones -- its block
    // count stretches the key range far beyond 4 n, so the comparison-sort branch orders the batch (shapes 2 and 3 take
    // the counting sort whenever n is large enough for their range)
    const unsigned shape = rnd() % 5;
    const size_t huge_at = n ? rnd() % n : 0;
    for (size_t i = 0; i < n; i++) {
      uint64_t len = shape == 0 ? 500 : shape == 2 ? (rnd() % 16 == 0 ? rnd() % 20000 : rnd() % 300) : rnd() % 3000;
      if (shape == 4) len = i == huge_at ? 390000000ull + rnd() % 1000 : rnd() % 400;
      off[i + 1] = off[i] + len;
    }
    if (shape == 1) {
      std::vector<uint64_t> lens(n);
      for (size_t i = 0; i < n; i++) lens[i] = off[i + 1] - off[i];
      std::sort(lens.begin(), lens.end(), [](uint64_t a, uint64_t b) { return a > b; });
      for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + lens[i];
    }
    auto blocks = [&](size_t i) { return (off[i + 1] - off[i] + block - 1) / block + 1; };
    const std::vector<size_t> order = ragged_order(off.data(), n, block, per_wave);
    auto cost = [&](const std::vector<size_t>& ord) {
      unsigned long long c = 0;
      for (size_t w = 0; w < n; w += per_wave) {
        uint64_t mx = 0;      // a wavefront runs as many block-steps as its longest message, however many lanes are live
        for (size_t i = w; i < n && i < w + per_wave; i++) mx = std::max<uint64_t>(mx, blocks(ord[i]));
        c += mx;
      }
      return c;
    };
    std::vector<size_t> ident(n);
    for (size_t i = 0; i < n; i++) ident[i] = i;
    if (order.empty()) {
      // left alone: the live lanes of the given order idle for at most 1/32 of their block-steps, or there is at most
      // one wavefront
      unsigned long long total = 0, given = 0;
      for (size_t w = 0; w < n; w += per_wave) {
        uint64_t mx = 0, cnt = 0;
        for (size_t i = w; i < n && i < w + per_wave; i++, cnt++) mx = std::max<uint64_t>(mx, blocks(i)), total += blocks(i);
        given += mx * cnt;
      }
      CHECK(n <= per_wave || given - total <= given / 32);
      if (shape == 0 || shape == 1) continue;
    } else {
      CHECK(order.size() == n);
      std::vector<char> seen(n, 0);
      for (size_t i = 0; i < n; i++) {
        CHECK(order[i] < n && !seen[order[i]]);
        seen[order[i]] = 1;
        if (i) CHECK(blocks(order[i - 1]) > blocks(order[i]) || (blocks(order[i - 1]) == blocks(order[i]) && order[i - 1] < order[i]));
      }
      CHECK(cost(order) <= cost(ident));
      CHECK(shape != 0);   // equal lengths are never reordered
      if (shape == 4) CHECK(order[0] == huge_at);
      // what the host path does with it (capi.hip: messages staged in `order`, digests scattered back): slot j of the
      // staged batch holds message order[j], its digest goes to out[order[j]] -- every message's digest lands at its own
      // index exactly once, whatever the order
      std::vector<uint64_t> staged_len(n), out(n, ~0ull);
      for (size_t j = 0; j < n; j++) staged_len[j] = off[order[j] + 1] - off[order[j]];
      for (size_t j = 0; j < n; j++) {
        CHECK(out[order[j]] == ~0ull);
        out[order[j]] = staged_len[j] * 2654435761ull + 7;      // stand-in for "the digest of a message of this length"
      }
      for (size_t i = 0; i < n; i++) CHECK(out[i] == (off[i + 1] - off[i]) * 2654435761ull + 7);
    }
  }
  // both sort branches on the SAME keys give the same stable order: a batch whose key range is just inside / just outside 4 n
  for (int wide = 0; wide < 2; wide++) {
    const size_t n = 200, block = 31;
    std::vector<uint64_t> off(n + 1, 0);
    for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + block * ((i * 37) % 50) + (wide && i == 17 ? block * 100000ull : 0);
    const std::vector<size_t> order = ragged_order(off.data(), n, block, 64);
    CHECK(order.size() == n);
    std::vector<size_t> ref(n);
    for (size_t i = 0; i < n; i++) ref[i] = i;
    std::stable_sort(ref.begin(), ref.end(), [&](size_t a, size_t b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
    CHECK(order == ref);
  }
}

static void test_mds_and_k() {
  std::vector<uint64_t> m;
  CHECK(!builtin_mds(0, 2, &m) && !builtin_mds(7, 2, &m));
  CHECK(builtin_mds(1, 5, &m) && m == std::vector<uint64_t>({1}));
  // NUM_COLUMNS = 2 (src/traits.rs:143-147 applied to unit vectors): [[1, g], [g, g^2 + 1]]
  for (uint64_t g : {2, 3, 5, 7, 15, 22}) {
    CHECK(builtin_mds(2, g, &m) && m == std::vector<uint64_t>({1, g, g, g * g + 1}));
    // 3 columns, the reference's matrix (src/traits.rs:161-163): [[g+1, 1, g+1], [1, 1, g], [g, 1, 1]]
    CHECK(builtin_mds(3, g, &m) && m == std::vector<uint64_t>({g + 1, 1, g + 1, 1, 1, g, g, 1, 1}));
    // 4 columns (src/traits.rs:166-189)
    CHECK(builtin_mds(4, g, &m));
    // the arm's statements on a test vector == the matrix applied to it
    uint64_t s[4] = {3, 5, 7, 11}, t[4] = {3, 5, 7, 11};
    t[0] += t[1]; t[2] += t[3]; t[3] += g * t[0]; t[1] = g * (t[1] + t[2]); t[0] += t[1]; t[2] += g * t[3]; t[1] += t[2]; t[3] += t[0];
    for (int i = 0; i < 4; i++) {
      uint64_t acc = 0;
      for (int j = 0; j < 4; j++) acc += m[i * 4 + j] * s[j];
      CHECK(acc == t[i]);
    }
  }
  // circulant arms: 5 columns circ(1, 2, 1+... ) -- every row is a rotation of the first
  for (int c : {5, 6}) {
    CHECK(builtin_mds(c, 2, &m));
    for (int i = 1; i < c; i++)
      for (int j = 0; j < c; j++) CHECK(m[i * c + j] == m[((j - i) % c + c) % c]);
  }
  CHECK(valid_k(2, 2) && !valid_k(2, 4) && !valid_k(2, 1) && valid_k(4, 2) && valid_k(4, 4) && !valid_k(4, 3) && !valid_k(4, 8));
  CHECK(valid_generic_k(6, 2) && valid_generic_k(6, 6) && !valid_generic_k(6, 3) && !valid_generic_k(6, 4) && !valid_generic_k(6, 0));
}

int main() {
  test_shards();
  test_subtrees();
  test_paths();
  test_overlap_and_chunks();
  test_ragged_chunks();
  test_ragged_order();
  test_mds_and_k();
  std::printf("host logic ok\n");
  return 0;
}
