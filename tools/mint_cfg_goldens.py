#!/usr/bin/env python3
"""Mint tests/golden/cfg_full.json: oracle-derived goldens for the BASELINE.json configs AT FULL SIZE.

Runs the pinned C oracle (oracle/anemoi_oracle.c, threaded) over the seeded inputs of
anemoi_amd/synth.py -- the same generator bench.py and the -m gpu tests draw from -- and records

  cfg2  2^20 BLS12-381 Anemoi-2-1 Jive compressions   (reference: src/bls12_381/anemoi_2_1/hasher.rs:96-103)
  cfg4  2^24 of the same, 8 contiguous shards of 2^21
        -> SHA-256 of all outputs (ABI bytes, item order), SHA-256 per shard, every `stride`-th output
  cfg3  2^16 BN-254 Anemoi-4-3 sponge hashes of 10 240-byte messages (src/bn_254/anemoi_4_3/hasher.rs:19-91)
        -> SHA-256 of all digests, every 64-th digest
  cfg5  depth-24 Jubjub Merkle tree, node = Sponge::merge 2-1 (src/jubjub/anemoi_2_1/hasher.rs:86-92)
        -> root, the 8 depth-21 subtree roots, SHA-256 of every level, the whole level 12 (4 096 nodes)

so the GPU tests compare full-size runs with oracle values instead of GPU-vs-GPU properties.  Takes
~40 min of 8 host threads in the build container (BLS12-381: ~11 k compress/s on 8 threads); each
config is cached under --cache so an interrupted run resumes.

    python tools/mint_cfg_goldens.py [--threads 8] [--only cfg2,cfg3,cfg4,cfg5] [--cache /tmp/anemoi_mint]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "anemoi-rust_amd", "anemoi_amd"))
import orc  # noqa: E402
import synth  # noqa: E402  (imported as a plain module: the package itself needs the HIP library)

FID = {n: i for i, n in enumerate(orc.FIELD_IDS)}
OUT = os.path.join(ROOT, "tests", "golden", "cfg_full.json")


def hexrows(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return [row.tobytes().hex() for row in a.reshape(len(a), -1)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def log(msg):
    print("[%s] %s" % (time.strftime("%H:%M:%S"), msg), flush=True)


def compress_all(oracle, cfg, threads, piece=1 << 18):
    """All outputs of a compress config, piece by piece (progress + bounded memory for the inputs)."""
    f, n = cfg["field"], cfg["n"]
    L = synth.limbs_of(f)
    out = np.empty((n, L), dtype=np.uint64)
    t0 = time.time()
    for b in range(0, n, piece):
        c = min(piece, n - b)
        st = synth.states(f, 2, cfg["seed"], b, c)
        out[b:b + c] = oracle.compress_batch(FID[f], 2, st, threads=threads).reshape(c, L)
        log("  %s: %d / %d  (%.0f /s)" % (f, b + c, n, (b + c) / (time.time() - t0)))
    return out


def mint_compress(oracle, cfg, threads, stride, shards):
    out = compress_all(oracle, cfg, threads)
    n = cfg["n"]
    g = {"field": cfg["field"], "width": 2, "seed": cfg["seed"], "n": n, "sha256": sha(out),
         "sample_stride": stride, "sample": hexrows(out[::stride])}
    if shards > 1:
        per = n // shards
        g["shards"] = shards
        g["shard_sha256"] = [sha(out[i * per:(i + 1) * per]) for i in range(shards)]
    return g


def mint_cfg3(oracle, threads, piece=1 << 12):
    cfg = synth.CFG3
    n, L = cfg["n"], synth.limbs_of(cfg["field"])
    out = np.empty((n, L), dtype=np.uint64)
    t0 = time.time()
    for b in range(0, n, piece):
        msgs = synth.messages(cfg["seed"], b, piece, cfg["msg_len"])
        out[b:b + piece] = oracle.hash_bytes_batch(FID[cfg["field"]], 4, msgs, threads=threads)
        log("  cfg3: %d / %d  (%.0f msg/s)" % (b + piece, n, (b + piece) / (time.time() - t0)))
    return {"field": cfg["field"], "width": 4, "seed": cfg["seed"], "n": n, "msg_len": cfg["msg_len"],
            "sha256": sha(out), "sample_stride": 64, "sample": hexrows(out[::64])}


def mint_cfg5(oracle, threads):
    cfg = synth.CFG5
    f, depth = cfg["field"], cfg["depth"]
    L = synth.limbs_of(f)
    lvl = synth.elements(f, cfg["seed"], 0, 1 << depth)
    g = {"field": f, "seed": cfg["seed"], "depth": depth, "level_sha256": [sha(lvl)]}
    for l in range(depth):
        t0 = time.time()
        n = len(lvl) // 2
        # level l+1 node i = merge(node 2i, node 2i+1) = compress([left, right])
        lvl = oracle.compress_batch(FID[f], 2, lvl.reshape(n, 2, L), threads=threads if n >= 64 else 1).reshape(n, L)
        g["level_sha256"].append(sha(lvl))
        if n == 4096:
            g["level12"] = hexrows(lvl)
        if n == 8:
            g["subtree_roots_depth21"] = hexrows(lvl)
        log("  cfg5: level %d (%d nodes) %.1f s" % (l + 1, n, time.time() - t0))
    g["root"] = hexrows(lvl)[0]
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--only", default="cfg2,cfg3,cfg5,cfg4")
    ap.add_argument("--cache", default="/tmp/anemoi_mint")
    args = ap.parse_args()
    os.makedirs(args.cache, exist_ok=True)
    orc.build()
    oracle = orc.Oracle()
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {}
    doc["generator"] = ("anemoi_amd/synth.py: counter-based SplitMix64, seed 0xA9E30100 + config; element = L u64 "
                        "stream words, top limb mod the modulus's top limb; values are ABI (Montgomery, R = 2^(64 L)) "
                        "limbs; hex strings = little-endian bytes of the u64 limbs in order")
    doc["minted_by"] = "tools/mint_cfg_goldens.py with oracle/anemoi_oracle.c (pinned on tests/golden/kats.json)"
    for name in args.only.split(","):
        cache = os.path.join(args.cache, name + ".json")
        if os.path.exists(cache):
            doc[name] = json.load(open(cache))
            log("%s: from cache" % name)
            continue
        log("%s: minting" % name)
        t0 = time.time()
        if name == "cfg2":
            g = mint_compress(oracle, synth.CFG2, args.threads, 256, 1)
        elif name == "cfg4":
            g = mint_compress(oracle, synth.CFG4, args.threads, 4096, synth.CFG4["shards"])
        elif name == "cfg3":
            g = mint_cfg3(oracle, args.threads)
        elif name == "cfg5":
            g = mint_cfg5(oracle, args.threads)
        else:
            raise SystemExit("unknown config " + name)
        g["oracle_seconds"] = round(time.time() - t0, 1)
        g["oracle_threads"] = args.threads
        json.dump(g, open(cache, "w"))
        doc[name] = g
        with open(OUT, "w") as fh:
            json.dump(doc, fh, indent=0, sort_keys=True)
            fh.write("\n")
        log("%s: done in %.0f s" % (name, time.time() - t0))
    with open(OUT, "w") as fh:
        json.dump(doc, fh, indent=0, sort_keys=True)
        fh.write("\n")
    log("wrote " + OUT)


if __name__ == "__main__":
    main()
