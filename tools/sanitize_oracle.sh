#!/bin/bash
# Builds the C oracle with AddressSanitizer + UndefinedBehaviorSanitizer and replays the reference KATs,
# a threaded batch, Merkle roots and the empty / long byte messages through it (CPU only; GPU ASan is not
# available on the pool).   tools/sanitize_oracle.sh
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fsanitize=undefined,address -fno-sanitize-recover=all -fPIC -shared -pthread \
    oracle/anemoi_oracle.c -o /tmp/liboracle_san.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python - <<'PY'
import json, random, sys
sys.path.insert(0, "oracle")
import orc
o = orc.Oracle("/tmp/liboracle_san.so")
P = json.load(open("tests/golden/params.json"))
K = json.load(open("tests/golden/kats.json"))
FIELDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
n = 0
for fid, f in enumerate(FIELDS):
    p = int(P[f]["modulus"])
    for w in (2, 4):
        k = K["%s/anemoi_%s" % (f, "2_1" if w == 2 else "4_3")]
        for a, b in zip(k["hash_field"]["in"], k["hash_field"]["out"]):
            assert o.mont_to_ints(fid, o.hash_field(fid, w, o.ints_to_mont(fid, [int(x) for x in a]))) == [int(b)]
            n += 1
        for a, b in zip(k["hash_bytes"]["in_hex"], k["hash_bytes"]["out"]):
            assert o.mont_to_ints(fid, o.hash_bytes(fid, w, bytes.fromhex(a))) == [int(b)]
            n += 1
        for a, b in zip(k["jive"]["in"], k["jive"]["out"]):
            e = o.ints_to_mont(fid, [int(x) for x in a])
            assert o.mont_to_ints(fid, o.compress_k(fid, w, e, 2)) == [int(x) for x in b]
            n += 1
        rng = random.Random(fid)
        st = o.ints_to_mont(fid, [rng.randrange(p) for _ in range(37 * w)]).reshape(37, w, -1)
        o.compress_batch(fid, w, st, k=2, threads=4)
        o.hash_bytes(fid, w, b"")
        o.hash_bytes(fid, w, bytes(range(200)))
    o.merkle_root(fid, o.ints_to_mont(fid, [random.randrange(p) for _ in range(8)]), 3)
print("sanitized oracle: %d KAT checks + batch / Merkle / edge calls ok" % n)
PY
