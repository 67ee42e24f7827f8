#!/usr/bin/env python3
"""Differential fuzz of EVERY kernel family against the C oracle: random and structured states per field through

  Jive 2-1      lane-private, two-row fold (coop2d), row-cooperative scan, one item per wavefront (four-row fold on the
                11-limb fields, the scan on the 15-limb ones)
  Jive 4-3      lane-pair, row-cooperative and two-row fold (one state per wavefront), k = 2 and 4
  permutation   the default routing of the batch size
  sponge        two-row / row-cooperative / lane-private kernels on equal-length batches, the three ragged kernels on all
                lengths in one batch (byte messages, and hash_field's element messages), the segment-fed host path (tiny forced segments)
  generic       the run-time-instance kernels fed with the shipped constants (3 fields)

Structured states stress carry patterns: limbs of all ones, values next to p and to 2^k, sparse values; every third element the
GPU receives is UNREDUCED (X + k p up to the top of the 64 L-bit range: round 6's input contract) while the oracle keeps X.
Kernels are forced through anemoi_set_option.  `run()` is what tests/test_gpu_fuzz.py calls (fixed seed, small sizes);

    python tools/fuzz_gpu_vs_oracle.py [items_per_field] [seed]

runs it big."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "anemoi-rust_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np

BIG = 1 << 40
LANE = dict(coop2d_max=0, coop4_max=0, coop43_max=0, coop2d43_max=0, coop_sponge_max=0, coop_climb_max=0)


def structured_values(p):
    vals, bits = [], p.bit_length()
    for k in range(0, bits, 7):
        for d in (-2, -1, 0, 1, 2):
            vals.append(((1 << k) + d) % p)
            vals.append((p - (1 << k) + d) % p)
    vals += [((1 << bits) - 1) % p, (1 << (bits - 1)) % p, p - 1, p - 2, 0, 1, 2]
    for w in (28, 29, 30, 32, 56, 58, 60, 64):
        ones = 0
        for i in range(0, bits, 2 * w):
            ones |= ((1 << w) - 1) << i
        vals += [ones % p, (ones << w) % p]
    return vals


def run(n2=20000, n4=5000, seed=1, threads=16, log=print, generic=True):
    """Returns the list of failed checks (empty = all bit-exact)."""
    import orc
    import anemoi_amd as A
    params = json.load(open(os.path.join(ROOT, "tests", "golden", "params.json")))
    oracle = orc.Oracle()
    failed = []

    def lifted(arr, p, L, rng):
        """round 6's input contract (any 64 L-bit pattern is taken mod p): every third element of what the GPU gets is
        X + k p, k up to the top of the 64 L-bit range; the oracle keeps the reduced X"""
        flat = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, L).copy()
        room = ((1 << (64 * L)) - 1) // p
        idx = list(range(seed % 3, len(flat), 3))
        vals = A.limbs_to_ints(flat[idx])
        flat[idx] = A.ints_to_limbs([v + rng.randrange(1, room) * p for v in vals], L)
        return flat.reshape(np.shape(arr))

    def check(ok, what):
        if not ok:
            failed.append(what)
        return "ok" if ok else "MISMATCH"

    for fid, field in enumerate(A.FIELD_IDS):
        p, L = int(params[field]["modulus"]), params[field]["u64_limbs"]
        rng = random.Random(1000 * seed + fid)
        structured = structured_values(p)
        for width in (2, 4):
            cnt = n2 if width == 2 else n4
            items = []
            for i in range(cnt):
                if i < len(structured):
                    items.extend(structured[(i + j * 17) % len(structured)] for j in range(width))
                else:
                    items.extend(rng.randrange(p) for _ in range(width))
            st = oracle.ints_to_mont(fid, items).reshape(cnt, width, L)
            inst = A.Anemoi(field, width)
            exp = oracle.compress_batch(fid, width, st, threads=threads)
            st_red, st = st, lifted(st, p, L, rng)          # the GPU gets unreduced patterns, the oracle got the reduced ones
            msg = "%-16s W=%d %6d items:" % (field, width, cnt)
            if width == 2:
                with A.options(**LANE):
                    msg += " lane-private %s" % check((inst.compress_batch(st) == exp).all(), (field, width, "lane-private"))
                m = min(cnt, 1501)
                with A.options(coop2d_max=BIG, coop4_max=0):
                    msg += "  two-row(%d) %s" % (m, check((inst.compress_batch(st[:m]) == exp[:m]).all(), (field, width, "two-row")))
                m = min(cnt, 3001)
                with A.options(coop2d_max=0, coop4_max=BIG):
                    msg += "  row-coop(%d) %s" % (m, check((inst.compress_batch(st[:m]) == exp[:m]).all(), (field, width, "row-coop")))
                if A.is_ab_build():   # the recorded negative exists in `make AB=1` libraries only
                    m = min(cnt, 300)
                    with A.options(coop_max=BIG, coop2d_max=0, coop4_max=0):
                        msg += "  one-per-wave(%d) %s" % (m, check((inst.compress_batch(st[:m]) == exp[:m]).all(), (field, width, "one-per-wave")))
            else:
                exp4 = oracle.compress_batch(fid, 4, st_red, k=4, threads=threads)
                with A.options(**LANE):
                    ok = (inst.compress_batch(st) == exp).all() and (inst.compress_k_batch(st, 4) == exp4).all()
                msg += " lane-pair %s" % check(ok, (field, width, "lane-pair"))
                m = min(cnt, 1501)
                with A.options(coop43_max=BIG, coop2d43_max=0):
                    ok = (inst.compress_batch(st[:m]) == exp[:m]).all() and (inst.compress_k_batch(st[:m], 4) == exp4[:m]).all()
                msg += "  row-coop(%d) k=2,4 %s" % (m, check(ok, (field, width, "row-coop 4-3")))
                m = min(cnt, 700)
                with A.options(coop2d43_max=BIG):
                    ok = (inst.compress_batch(st[:m]) == exp[:m]).all() and (inst.compress_k_batch(st[:m], 4) == exp4[:m]).all()
                msg += "  two-row(%d) k=2,4 %s" % (m, check(ok, (field, width, "two-row 4-3")))
            m = min(cnt, 256)
            pg = inst.permutation_batch(st[:m])
            ok = all((pg[i] == oracle.permutation(fid, width, st_red[i])).all() for i in range(0, m, 5))
            msg += "  permutation %s" % check(ok, (field, width, "permutation"))
            log(msg)
    # sponge: random byte messages of structured lengths (around the chunk and rate-block boundaries)
    nprng = np.random.default_rng(seed)
    for fid, field in enumerate(A.FIELD_IDS):
        for width in (2, 4):
            inst = A.Anemoi(field, width)
            ch, r = inst.chunk, width - 1
            lens = sorted({0, 1, ch - 1, ch, ch + 1, r * ch, r * ch + 1, 2 * r * ch - 1, 5 * ch + 3, 333} |
                          {int(v) for v in nprng.integers(0, 400, size=3)})
            ok = True
            ragged, want = [], []
            for ln in lens:
                msgs = nprng.integers(0, 256, size=(5, ln), dtype=np.uint8)
                if ln:
                    msgs[0], msgs[1] = 0, 255
                exp = oracle.hash_bytes_batch(fid, width, msgs, threads=threads)
                ok = ok and (inst.hash_batch(msgs) == exp).all()                      # default: the two-row fold kernels
                with A.options(coop2d_max=0, coop2d43_max=0):
                    ok = ok and (inst.hash_batch(msgs) == exp).all()                  # row-cooperative sponge
                with A.options(**LANE):
                    ok = ok and (inst.hash_batch(msgs) == exp).all()                  # lane-private sponge
                if ln > 3 * r * ch:
                    with A.options(sponge_segment_bytes=5 * r * ch):                   # segment-fed host path
                        ok = ok and (inst.hash_batch(msgs) == exp).all()
                ragged += [m.tobytes() for m in msgs]
                want.append(exp)
            # all lengths in one ragged batch: the two-row fold, the row-cooperative and the lane-private ragged kernels
            ok = ok and (inst.hash_ragged(ragged) == np.concatenate(want)).all()
            with A.options(coop2d_max=0, coop2d43_max=0):
                ok = ok and (inst.hash_ragged(ragged) == np.concatenate(want)).all()
            with A.options(**LANE):
                ok = ok and (inst.hash_ragged(ragged) == np.concatenate(want)).all()
            # hash_field over element messages of structured values, 0 .. 3 rate blocks + 1 elements, all in one ragged batch
            pval = int(params[field]["modulus"])
            sv = structured_values(pval)
            emsgs = [oracle.ints_to_mont(fid, [sv[(7 * k + i) % len(sv)] for i in range(k)]).reshape(k, inst.limbs)
                     for k in list(range(0, 3 * r + 2)) * 3]
            ewant = np.stack([oracle.hash_field(fid, width, m) for m in emsgs])
            erng = random.Random(31 * seed + fid)
            emsgs = [lifted(m, pval, inst.limbs, erng) if len(m) else m for m in emsgs]      # (unreduced elements: see lifted)
            ok = ok and (inst.hash_field_ragged(emsgs) == ewant).all()
            with A.options(coop2d_max=0, coop2d43_max=0):
                ok = ok and (inst.hash_field_ragged(emsgs) == ewant).all()
            with A.options(**LANE):
                ok = ok and (inst.hash_field_ragged(emsgs) == ewant).all()
            log("%-16s W=%d sponge, %d lengths x 5 messages (default, row-coop, lane-private, segments; ragged bytes and ragged elements on the three): %s"
                % (field, width, len(lens), check(ok, (field, width, "sponge"))))
    if generic:   # the run-time-instance kernels fed with the shipped constants reproduce the fixed instances
        for field in ("bn_254", "bls12_381", "ed_on_bls12_377"):
            fid = A.FIELD_IDS.index(field)
            p, L = int(params[field]["modulus"]), params[field]["u64_limbs"]
            rng = random.Random(77 * seed + fid)
            for width, key in ((2, "anemoi_2_1"), (4, "anemoi_4_3")):
                ins = params[field]["instances"][key]
                c = width // 2
                enc = lambda vals: oracle.ints_to_mont(fid, [int(v) for v in vals])
                g = A.GenericAnemoi(field, c, ins["num_rounds"], enc(ins["ark_c"]), enc(ins["ark_d"]))
                st = oracle.ints_to_mont(fid, [rng.randrange(p) for _ in range(width * 70)]).reshape(70, width, L)
                ok = (g.compress_k_batch(st, 2) == oracle.compress_batch(fid, width, st, threads=threads)).all()
                log("%-16s W=%d generic (run-time instance) Jive: %s" % (field, width, check(ok, (field, width, "generic"))))
    return failed


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000   # > 8192: the lane-private kernel at its real grid
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    bad = run(n2=n, n4=n // 4, seed=seed, log=lambda m: print(m, flush=True))
    print("FUZZ %s in %.1f s" % ("FAILED: %r" % bad if bad else "PASSED", time.time() - t0))
    sys.exit(1 if bad else 0)
