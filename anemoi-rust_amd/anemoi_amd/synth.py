"""Seeded synthetic workloads for the BASELINE.json configs.

Counter-based SplitMix64: word i of the stream with seed s is mix(s + (i + 1) * GAMMA) -- exactly the
i-th output of the sequential SplitMix64 generator -- so any contiguous shard of a config's input can be
produced on its own (one rank's slice of config 4, one subtree's leaves of config 5) and is identical
to the same slice of the whole.  bench.py, the parity tests and tools/mint_cfg_goldens.py (which runs
the CPU oracle over these inputs and commits tests/golden/cfg_full.json) all draw from here, so the
committed goldens describe exactly the batches the GPU runs.

Element = `L` little-endian u64 limbs read straight from the stream, the top limb reduced modulo the
modulus's top limb: every value is < p, i.e. a valid arkworks Montgomery-form `Felt` (any residue is).
"""
import numpy as np

GAMMA = 0x9E3779B97F4A7C15
SEED_BASE = 0xA9E30100  # + config number

MODULI = {
    "bls12_381": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    "bls12_377": 0x1ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001,
    "bn_254": 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    "ed_on_bls12_377": 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001,
    "jubjub": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    "pallas": 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
    "vesta": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
}


def limbs_of(field):
    return 6 if MODULI[field].bit_length() > 256 else 4


def words(seed, first, count):
    """SplitMix64 outputs [first, first + count) of the stream seeded `seed`, as uint64."""
    with np.errstate(over="ignore"):
        z = (np.arange(first + 1, first + 1 + count, dtype=np.uint64) * np.uint64(GAMMA)) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def elements(field, seed, first, count, chunk=1 << 20):
    """Elements [first, first + count) of the config's element stream: (count, L) uint64, each < p."""
    L = limbs_of(field)
    top = np.uint64(MODULI[field] >> (64 * (L - 1)))
    out = np.empty((count, L), dtype=np.uint64)
    for b in range(0, count, chunk):
        c = min(chunk, count - b)
        w = words(seed, (first + b) * L, c * L).reshape(c, L)
        w[:, L - 1] %= top
        out[b:b + c] = w
    return out


def states(field, width, seed, first, count):
    """States [first, first + count): (count, width, L) uint64."""
    L = limbs_of(field)
    return elements(field, seed, first * width, count * width).reshape(count, width, L)


def messages(seed, first, count, msg_len):
    """Byte messages [first, first + count) of msg_len bytes (msg_len a multiple of 8): (count, msg_len) uint8."""
    assert msg_len % 8 == 0
    per = msg_len // 8
    out = np.empty((count, msg_len), dtype=np.uint8)
    step = max(1, (1 << 20) // per)
    for b in range(0, count, step):
        c = min(step, count - b)
        out[b:b + c] = words(seed, (first + b) * per, c * per).view(np.uint8).reshape(c, msg_len)
    return out


# The BASELINE.json configs as (field, width, seed, size) -- sizes are the full ones.
CFG2 = dict(field="bls12_381", width=2, seed=SEED_BASE + 2, n=1 << 20)
CFG3 = dict(field="bn_254", width=4, seed=SEED_BASE + 3, n=1 << 16, msg_len=10240)
CFG4 = dict(field="bls12_381", width=2, seed=SEED_BASE + 4, n=1 << 24, shards=8)
CFG5 = dict(field="jubjub", width=2, seed=SEED_BASE + 5, depth=24, shards=8)
