#!/bin/bash
# Builds the pure host side of the C-ABI (csrc/host_logic.h: shard ranges, Merkle layout and paths, chunk
# plan, MDS arms, argument rules) with AddressSanitizer + UndefinedBehaviorSanitizer and runs its test
# program (CPU only; GPU sanitizers are not available on the pool).   tools/sanitize_host.sh
set -e
cd "$(dirname "$0")/.."
g++ -std=c++17 -O1 -g -Wall -Wextra -fsanitize=address,undefined -fno-sanitize-recover=all \
    tests/cpp/test_host_logic.cpp -o /tmp/anemoi_test_host_logic
ASAN_OPTIONS=detect_leaks=1 /tmp/anemoi_test_host_logic
