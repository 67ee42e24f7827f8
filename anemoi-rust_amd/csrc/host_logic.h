// host_logic.h -- the pure host-side logic of the C-ABI: no HIP, no device, plain C++17.
//
// Everything here is arithmetic on sizes, indices and small integers -- shard ranges, the Merkle
// level/path indexing, the chunk plan of the host-pointer pipeline, the small-integer form of the
// reference's hard-coded mds_layer arms -- kept apart from capi.hip so that tests/cpp/test_host_logic.cpp
// can compile it with gcc under AddressSanitizer/UBSan (tools/sanitize_host.sh); GPU sanitizers are not
// available on the pool.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>

namespace anemoi {
namespace host {

// ---- contiguous range sharding (no collective, items are independent) ---------
// shard i of `parts` owns items [shard_begin(n, i, parts), shard_begin(n, i + 1, parts))
inline size_t shard_begin(size_t n, size_t i, size_t parts) {
  // n * i can overflow 64 bits for huge n: split n = q * parts + r
  const size_t q = n / parts, r = n % parts;
  return q * i + (r * i) / parts;
}

// Merkle tree over arity^depth leaves (arity = 2^arity_log) shared by `parts` workers.  Returns
// log_arity(#subtrees): the largest power of the arity that is <= parts and <= the leaf count -- or, when
// the NEXT power is a whole multiple of `parts` (arity 4 on 2 or 8 GPUs: 4 or 16 subtrees), that one, so that
// every worker gets the same number of subtrees instead of half of them idling (subtrees of one worker run
// one after the other, rt::run_parts).
inline unsigned subtree_levels(unsigned depth, unsigned arity_log, size_t parts) {
  unsigned lv = 0;
  while (lv + 1 <= depth && (size_t(1) << (arity_log * (lv + 1))) <= parts) lv++;
  const size_t have = size_t(1) << (arity_log * lv);
  if (have < parts && lv + 1 <= depth) {
    const size_t next = size_t(1) << (arity_log * (lv + 1));
    if (next % parts == 0) lv++;
  }
  return lv;
}

// ---- retained-tree layout ------------------------------------------------------------------------
// Binary: level l (0 = leaves) has 2^(depth - l) nodes and starts at element offset
//   off(l) = sum_{i < l} 2^(depth - i) = 2^(depth + 1) - 2^(depth - l + 1).
inline size_t tree2_level_offset(unsigned depth, unsigned l) {
  return (size_t(2) << depth) - (size_t(2) << (depth - l));
}
inline size_t tree2_total(unsigned depth) { return (size_t(2) << depth) - 1; }
// Arity 4: level l has 4^(depth4 - l) nodes.
inline size_t tree4_level_offset(unsigned depth4, unsigned l) {
  size_t off = 0;
  for (unsigned i = 0; i < l; i++) off += size_t(1) << (2 * (depth4 - i));
  return off;
}
inline size_t tree4_total(unsigned depth4) { return ((size_t(4) << (2 * depth4)) - 1) / 3; }

// Authentication path of leaf `index`: the sibling of node (level l, position index >> l), l = 0 .. depth-1.
// `L` u64 limbs per element.  Caller has checked index < 2^depth.
inline void merkle_path2(const uint64_t* tree, unsigned depth, size_t index, size_t L, uint64_t* path) {
  for (unsigned l = 0; l < depth; l++) {
    const size_t pos = (index >> l) ^ 1;
    std::memcpy(path + size_t(l) * L, tree + (tree2_level_offset(depth, l) + pos) * L, L * 8);
  }
}

// Arity 4: per level the 3 siblings of the node in child order, its own slot left out.
inline void merkle_path4(const uint64_t* tree, unsigned depth4, size_t index, size_t L, uint64_t* path) {
  for (unsigned l = 0; l < depth4; l++) {
    const size_t node = index >> (2 * l), first = node & ~size_t(3), off = tree4_level_offset(depth4, l);
    int k = 0;
    for (size_t c = 0; c < 4; c++)
      if (first + c != node) std::memcpy(path + (size_t(l) * 3 + k++) * L, tree + (off + first + c) * L, L * 8);
  }
}

// ---- byte ranges ----------------------------------------------------------------------------------
inline bool ranges_overlap(const void* a, size_t a_bytes, const void* b, size_t b_bytes) {
  const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
  return a_bytes && b_bytes && a0 < b0 + b_bytes && b0 < a0 + a_bytes;
}

// ---- chunk plan of the host-pointer pipeline ------------------------------------------------------
// A batch of n items is cut into chunks whose sizes are multiples of `quantum` (= the items one full
// wave of workgroups processes: every CU at its resident-workgroup limit), so that no chunk but the
// last ends in a partially filled wave of workgroups on these ALU-bound kernels.  A chunk holds about
// `target_bytes` of input.  Batches of fewer than 2 quanta are not cut.
struct ChunkPlan {
  size_t chunk_items;  // items per chunk (the last chunk takes the remainder)
  size_t chunks;
};
inline ChunkPlan plan_chunks(size_t n, size_t quantum, size_t in_bytes_per_item, size_t target_bytes) {
  if (quantum == 0) quantum = 1;
  if (n < 2 * quantum) return {n, n ? size_t(1) : size_t(0)};
  size_t per = in_bytes_per_item ? target_bytes / in_bytes_per_item : n;
  size_t q = per / quantum;
  if (q == 0) q = 1;
  const size_t items = q * quantum;
  return {items, (n + items - 1) / items};
}

// Chunk plan of a RAGGED batch (message i = bytes [off[i], off[i+1]) of one blob), cut at message boundaries.
// A chunk takes messages until it holds `target_bytes` AND `min_items` messages (a chunk with fewer messages than
// half a wave of workgroups leaves the machine underfilled however many bytes it has: these kernels cost what
// their wavefronts' longest messages cost), but never more than `max_bytes` (staging stays bounded; a single
// longer message is a chunk of its own) nor more than `max_items` messages (many empty messages must not make
// an unbounded offsets table).  Its message count is a multiple of `align` (one wavefront's worth of messages)
// whenever it has at least `align`.  Returns the first message of every chunk plus n at the end; every chunk
// is non-empty.
inline std::vector<size_t> plan_ragged_chunks(const uint64_t* off, size_t n, size_t target_bytes, size_t align,
                                              size_t min_items, size_t max_items, size_t max_bytes) {
  std::vector<size_t> cuts;
  if (align == 0) align = 1;
  if (max_items < align) max_items = align;
  if (min_items > max_items) min_items = max_items;
  if (max_bytes < target_bytes) max_bytes = target_bytes;
  size_t first = 0;
  while (first < n) {
    cuts.push_back(first);
    size_t end = first;
    while (end < n && end - first < max_items) {
      const uint64_t with = off[end + 1] - off[first];
      if (with > max_bytes) break;
      if (with > target_bytes && end - first >= min_items) break;
      end++;
    }
    if (end == first) end = first + 1;  // one message longer than the cap
    size_t cnt = end - first;
    if (cnt >= align) cnt = cnt / align * align;
    first += cnt;
  }
  cuts.push_back(n);
  return cuts;
}

// Order in which a ragged batch should be fed to the sponge kernels: every lane of a wavefront walks its own message
// and the wavefront runs as many rate-blocks as its LONGEST message, so `per_wave` consecutive messages should have
// the same block count.  Returns the message indices by descending block count (stable) -- or an empty vector when the
// given order already wastes less than 1/32 of the block-steps (sorted input, equal lengths, tiny batches).
//   off[0..n]      message i = bytes [off[i], off[i+1])
//   block_bytes    input bytes one permutation absorbs (RATE x chunk bytes)
inline std::vector<size_t> ragged_order(const uint64_t* off, size_t n, size_t block_bytes, size_t per_wave) {
  std::vector<size_t> order;
  if (n <= per_wave || block_bytes == 0 || per_wave == 0) return order;
  auto blocks = [&](size_t i) { return size_t((off[i + 1] - off[i] + block_bytes - 1) / block_bytes) + 1; };  // + the final permutation
  // block-steps as given (sum over wavefronts of the longest message) against the minimum (the sorted order's)
  unsigned long long given = 0, total = 0;
  for (size_t w = 0; w < n; w += per_wave) {
    size_t mx = 0, cnt = 0;
    for (size_t i = w; i < n && i < w + per_wave; i++, cnt++) {
      const size_t b = blocks(i);
      mx = b > mx ? b : mx;
      total += b;
    }
    given += (unsigned long long)mx * cnt;   // (the idle lanes of a partly filled last wavefront are nobody's fault)
  }
  if (given - total <= given / 32) return order;   // at most ~3 % of the lanes' block-steps idle: leave it
  // stable, by descending block count: a counting sort when the counts span a small range (the usual case: O(n), a
  // 2^19-message batch in ~2 ms where a comparison sort takes ~100 ms before the first chunk can be staged)
  std::vector<size_t> key(n);
  size_t lo = ~size_t(0), hi = 0;
  for (size_t i = 0; i < n; i++) {
    key[i] = blocks(i);
    lo = key[i] < lo ? key[i] : lo;
    hi = key[i] > hi ? key[i] : hi;
  }
  order.resize(n);
  // (only while the key range is of the batch's order -- span <= 4 n: a 65-message batch with one 390 MB message must not
  // allocate and prefix-sum four million buckets to order 65 items -- else the comparison sort)
  if (hi - lo <= 4 * n && hi - lo < (size_t(1) << 22)) {
    std::vector<size_t> start(hi - lo + 2, 0);
    for (size_t i = 0; i < n; i++) start[hi - key[i] + 1]++;          // bucket 0 = the longest messages
    for (size_t b = 1; b < start.size(); b++) start[b] += start[b - 1];
    for (size_t i = 0; i < n; i++) order[start[hi - key[i]]++] = i;
  } else {
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] > key[b]; });
  }
  return order;
}

// ---- the reference's hard-coded mds_layer arms as small-integer matrices --------------------------
// The matrix a hard-coded arm applies to each half of the state: the arm's statements
// (src/traits.rs:136-279; mds_internal :307-323) applied to the unit vectors.  Row-major c x c.
inline bool builtin_mds(int c, uint64_t g, std::vector<uint64_t>* m) {
  if (c < 1 || c > 6) return false;
  m->assign(size_t(c) * c, 0);
  for (int j = 0; j < c; j++) {
    uint64_t s[6] = {0, 0, 0, 0, 0, 0}, o[6] = {0, 0, 0, 0, 0, 0};
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

// the reference's compress_k asserts (anemoi_2_1/hasher.rs:107; anemoi_4_3/hasher.rs:163-165)
inline bool valid_k(int width, int k) { return width == 2 ? k == 2 : (k == 2 || k == 4); }
// generic instances: k <= width, k | width, k even
inline bool valid_generic_k(int width, int k) { return k >= 2 && k <= width && width % k == 0 && k % 2 == 0; }

}  // namespace host
}  // namespace anemoi
