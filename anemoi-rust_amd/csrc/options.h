// options.h -- the library's run-time knobs (include/anemoi_mi355x.h lists them for callers).
//
// Every knob is ONE atomic value.  It is initialised once, on first use, from its environment variable
// (validated: a value that is not a whole number in range is ignored and reported by anemoi_last_error() of the
// first anemoi_get_option / anemoi_set_option call that looks at it) and changed afterwards only through
// anemoi_set_option(name, value).  No entry point calls getenv() on its launch path, so a host thread that calls
// setenv() cannot race with a launch, and a test switches kernels through the API instead of os.environ.
// kAuto (-1) means "derive it": the cooperative cut-offs from the device's SIMD count, the chunk quantum from the
// occupancy API, the shard count from the GPU count.
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "build_config.h"

namespace anemoi {
namespace opt {

constexpr long long kAuto = -1;

enum Id {
  kCoopMax = 0,         // A/B BUILDS ONLY (the product does not know the name): largest Jive 2-1 batch on the one-item-per-wavefront kernel
  kCoop2dMax,           // largest batch (Jive 2-1, permutation 2-1, sponge 2-1, path climb) on the two-row 2-D kernels
  kCoop4Max,            // largest Jive 2-1 / permutation batch on the row-cooperative kernel (four items per wavefront)
  kCoop43Max,           // largest Jive 4-3 / permutation batch on the row-cooperative 4-3 kernel (two states per wavefront)
  kCoop2d43Max,         // largest 4-3 batch (Jive, permutation, sponge) on the two-row fold kernels (one state per wavefront)
  kCoopSpongeMax,       // largest equal-length sponge batch on k_sponge_coop
  kCoopClimbMax,        // largest batch of authentication paths on k_merkle_climb_coop
  kTestQuantum,         // items per "full wave of workgroups" of the chunked host pipelines (auto: occupancy API)
  kChunkTargetBytes,    // input bytes per chunk of the host pipelines (auto: 24 MiB)
  kSpongeSegmentBytes,  // forces the segment-fed sponge with this many bytes per segment (auto: by batch shape)
  kVirtualDevices,      // ANEMOI_ALL_DEVICES cuts into this many parts (auto: the GPU count)
  kHostStaging,         // 1 = stage host buffers through pinned memory (auto), 0 = copy straight from / to the caller's memory
  kBalanceUnderfilled,  // 1 (auto) = a launch of 2 ... 16 workgroups per CU is preceded by a do-nothing launch (k_balance: even placement)
  kLanePriorities,      // 1 (auto) = every second lane's kernel stream is a high-priority stream (its own set of hardware queues)
  kCount
};

struct Spec {
  const char* name;
  const char* env;
  long long lo, hi;  // accepted range (kAuto is always accepted)
  bool ab_only = false;  // exists in `make AB=1` libraries only (routes to a kernel the product does not contain)
};
inline bool known(const Spec& s) { return ANEMOI_AB_BUILD || !s.ab_only; }

inline const Spec& spec(int id) {
  static const Spec table[kCount] = {
      {"coop_max", "ANEMOI_COOP_MAX", 0, 1ll << 62, true},
      {"coop2d_max", "ANEMOI_COOP2D_MAX", 0, 1ll << 62},
      {"coop4_max", "ANEMOI_COOP4_MAX", 0, 1ll << 62},
      {"coop43_max", "ANEMOI_COOP43_MAX", 0, 1ll << 62},
      {"coop2d43_max", "ANEMOI_COOP2D43_MAX", 0, 1ll << 62},
      {"coop_sponge_max", "ANEMOI_COOP_SPONGE_MAX", 0, 1ll << 62},
      {"coop_climb_max", "ANEMOI_COOP_CLIMB_MAX", 0, 1ll << 62},
      {"test_quantum", "ANEMOI_TEST_QUANTUM", 1, 1ll << 40},
      {"chunk_target_bytes", "ANEMOI_CHUNK_TARGET_BYTES", 1, 1ll << 40},
      {"sponge_segment_bytes", "ANEMOI_SPONGE_SEGMENT_BYTES", 1, 1ll << 40},
      {"virtual_devices", "ANEMOI_VIRTUAL_DEVICES", 1, 64},
      {"host_staging", "ANEMOI_HOST_STAGING", 0, 1},
      {"balance_underfilled", "ANEMOI_BALANCE_UNDERFILLED", 0, 1},
      {"lane_priorities", "ANEMOI_LANE_PRIORITIES", 0, 1},
  };
  return table[id];
}

struct State {
  std::atomic<long long> v[kCount];
  std::once_flag once;
  std::string env_error;  // written once under `once`, read-only afterwards
};
inline State& state() {
  static State s;
  return s;
}

// whole-string decimal (or "pinned" / "direct" for host_staging); false = not a valid value
inline bool parse(int id, const char* text, long long* out) {
  if (!text || !*text) return false;
  if (id == kHostStaging) {
    if (!strcmp(text, "pinned")) return *out = 1, true;
    if (!strcmp(text, "direct")) return *out = 0, true;
  }
  if (!strcmp(text, "auto") || !strcmp(text, "default")) return *out = kAuto, true;
  char* end = nullptr;
  const long long v = strtoll(text, &end, 10);
  if (end == text || *end) return false;
  if (v != kAuto && (v < spec(id).lo || v > spec(id).hi)) return false;
  return *out = v, true;
}

inline void init_once() {
  State& s = state();
  std::call_once(s.once, [&] {
    for (int i = 0; i < kCount; i++) {
      long long v = kAuto;
      const char* e = known(spec(i)) ? getenv(spec(i).env) : nullptr;
      if (e) {
        if (!parse(i, e, &v)) {
          v = kAuto;
          s.env_error += std::string(s.env_error.empty() ? "" : "; ") + spec(i).env + "=\"" + e + "\" ignored (not a value in range)";
        }
      }
      s.v[i].store(v, std::memory_order_relaxed);
    }
  });
}

inline long long get(Id id) {
  init_once();
  return state().v[id].load(std::memory_order_relaxed);
}
inline long long get_or(Id id, long long automatic) {
  const long long v = get(id);
  return v == kAuto ? automatic : v;
}
inline int find(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < kCount; i++)
    if (known(spec(i)) && (!strcmp(name, spec(i).name) || !strcmp(name, spec(i).env))) return i;
  return -1;
}
inline bool set(int id, long long value) {
  init_once();
  if (id < 0 || id >= kCount) return false;
  if (value != kAuto && (value < spec(id).lo || value > spec(id).hi)) return false;
  state().v[id].store(value, std::memory_order_relaxed);
  return true;
}
inline const std::string& env_error() {
  init_once();
  return state().env_error;
}

}  // namespace opt
}  // namespace anemoi
