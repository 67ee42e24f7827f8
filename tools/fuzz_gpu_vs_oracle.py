#!/usr/bin/env python3
"""One-off differential fuzz: many random and structured states per field through the GPU kernels
(lane-private, row-cooperative and wave-cooperative Jive 2-1, Jive 4-3, permutation) against the C oracle.
Structured states stress carry patterns: limbs of all ones, values next to p and to 2^k, sparse values.
    python tools/fuzz_gpu_vs_oracle.py [items_per_field] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "anemoi-rust_amd")):
    sys.path.insert(0, p)
import json
import numpy as np
import orc
import anemoi_amd as A

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000   # > 8192: the lane-private kernel
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
params = json.load(open(os.path.join(ROOT, "tests", "golden", "params.json")))
oracle = orc.Oracle()
threads = 16
bad = 0
for fid, field in enumerate(A.FIELD_IDS):
    p, L = int(params[field]["modulus"]), params[field]["u64_limbs"]
    rng = random.Random(1000 * seed + fid)
    vals = []
    bits = p.bit_length()
    for k in range(0, bits, 7):
        for d in (-2, -1, 0, 1, 2):
            vals.append(((1 << k) + d) % p)
            vals.append((p - (1 << k) + d) % p)
    vals += [((1 << bits) - 1) % p, (1 << (bits - 1)) % p, p - 1, p - 2, 0, 1, 2]
    for w in (29, 30, 32, 58, 60, 64):
        ones = 0
        for i in range(0, bits, 2 * w):
            ones |= ((1 << w) - 1) << i
        vals += [ones % p, (ones << w) % p]
    structured = vals
    for width in (2, 4):
        cnt = n if width == 2 else n // 4
        items = []
        for i in range(cnt):
            if i < len(structured):
                st = [structured[(i + j * 17) % len(structured)] for j in range(width)]
            else:
                st = [rng.randrange(p) for _ in range(width)]
            items.extend(st)
        st = oracle.ints_to_mont(fid, items).reshape(cnt, width, L)
        inst = A.Anemoi(field, width)
        exp = oracle.compress_batch(fid, width, st, threads=threads)
        got = inst.compress_batch(st)
        ok = (got == exp).all()
        msg = "%-16s W=%d compress %6d items: %s" % (field, width, cnt, "ok" if ok else "MISMATCH")
        if width == 2:  # the latency kernels on slices (the selection knobs are read at every call)
            ok2 = (inst.compress_batch(st[:3001]) == exp[:3001]).all()  # <= 8192 -> row-cooperative kernel, 4 items per wavefront
            os.environ["ANEMOI_COOP_MAX"] = "1000000"
            ok4 = (inst.compress_batch(st[:700]) == exp[:700]).all()    # forced: one item per wavefront
            del os.environ["ANEMOI_COOP_MAX"]
            msg += "  row-coop(3001): %s  wave-coop(700): %s" % ("ok" if ok2 else "MISMATCH", "ok" if ok4 else "MISMATCH")
            ok = ok and ok2 and ok4
        else:  # Anemoi-4-3: a slice small enough for the row-cooperative kernel (two states per wavefront), k = 2 and 4
            ok2 = (inst.compress_batch(st[:1501]) == exp[:1501]).all()
            ok4 = (inst.compress_k_batch(st[:1501], 4) == oracle.compress_batch(fid, 4, st[:1501], k=4, threads=threads)).all()
            msg += "  row-coop(1501): k=2 %s k=4 %s" % ("ok" if ok2 else "MISMATCH", "ok" if ok4 else "MISMATCH")
            ok = ok and ok2 and ok4
        pg = inst.permutation_batch(st[:256])
        ok3 = all((pg[i] == oracle.permutation(fid, width, st[i])).all() for i in range(0, 256, 5))
        msg += "  permutation: %s" % ("ok" if ok3 else "MISMATCH")
        print(msg, flush=True)
        bad += 0 if (ok and ok3) else 1
# sponge: random byte messages of structured lengths (around the chunk and rate-block boundaries) through the
# row-cooperative sponge (small equal-length batches), the lane-private sponge (forced) and the ragged kernel
nprng = np.random.default_rng(seed)
for fid, field in enumerate(A.FIELD_IDS):
    for width in (2, 4):
        inst = A.Anemoi(field, width)
        ch, r = inst.chunk, width - 1
        lens = sorted({0, 1, ch - 1, ch, ch + 1, r * ch, r * ch + 1, 2 * r * ch - 1, 5 * ch + 3, 333} |
                      {int(v) for v in nprng.integers(0, 400, size=6)})
        ok = True
        ragged, want = [], []
        for ln in lens:
            msgs = nprng.integers(0, 256, size=(9, ln), dtype=np.uint8)
            if ln:
                msgs[0], msgs[1] = 0, 255
            exp = oracle.hash_bytes_batch(fid, width, msgs, threads=threads)
            ok = ok and (inst.hash_batch(msgs) == exp).all()                      # cooperative sponge
            os.environ["ANEMOI_COOP_SPONGE_MAX"] = "0"
            ok = ok and (inst.hash_batch(msgs) == exp).all()                      # lane-private sponge
            del os.environ["ANEMOI_COOP_SPONGE_MAX"]
            ragged += [m.tobytes() for m in msgs]
            want.append(exp)
        ok = ok and (inst.hash_ragged(ragged) == np.concatenate(want)).all()      # ragged kernel, all lengths in one batch
        print("%-16s W=%d sponge, %d lengths x 9 messages (cooperative, lane-private, ragged): %s"
              % (field, width, len(lens), "ok" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
print("FUZZ", "FAILED" if bad else "PASSED")
sys.exit(1 if bad else 0)
