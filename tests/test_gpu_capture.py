"""The `_dev` API under stream capture (include/anemoi_mi355x.h: "`_dev` functions ... enqueue asynchronously and do not
synchronise -- except that the FIRST use of a (device, field, width) uploads that instance's constant tables with a
blocking copy (not legal inside a stream capture): call anemoi_init() beforehand").

After anemoi_init, four calls are captured into ONE hipGraph on a side stream (torch.cuda.graph is plumbing: the
library only ever sees the hipStream_t it is handed):
  * anemoi_merkle_root_dev, depth 15 Jubjub: a chain of 15 launches that picks three different kernels by level size --
    lane-private (2^14 nodes), the row-cooperative scan (8 192, 4 096) and the two-row fold (2 048 ... 1);
  * anemoi_hash_bytes_dev (BN-254 Anemoi-4-3, 700 messages of 200 bytes: the two-row fold sponge);
  * anemoi_jive_compress_k_dev (BLS12-381, 20 000 states: the lane-private throughput kernel);
  * anemoi_hash_bytes_ragged_bucketed_dev (Jubjub, 300 messages of 0 ... 400 bytes: the four bucketing launches and the
    ragged latency kernel reading the order they wrote.  Round 5 found a bug here: the counters were zeroed with
    hipMemsetAsync, and the captured memset node did not zero them on replay -- every replay placed its indices behind
    the previous one's and the third or fourth wrote past the scratch; they are zeroed by a kernel now, and the test
    checks after every replay that the order array is a permutation of 0 ... n - 1).
The graph is replayed three times on FRESH inputs written into the captured buffers, every output compared with the
oracle.  Then the cut-offs are changed through anemoi_set_option -- which would send the same sizes to other kernels --
and the captured graph must still give the right bits: capture froze the kernel CHOICE made at capture time (the
routing runs on the host), not the options, and every kernel of a family computes the same function.  A call made
AFTER the change, outside the graph, takes the new route and agrees as well.
"""
import numpy as np
import pytest

from conftest import FIELD_IDS

pytestmark = pytest.mark.gpu


def test_dev_calls_captured_into_a_graph_and_replayed(oracle, params):
    import torch
    import anemoi_amd as A
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    jub, bn, bls = FIELD_IDS.index("jubjub"), FIELD_IDS.index("bn_254"), FIELD_IDS.index("bls12_381")
    depth, nmsg, mlen, nst, nrag = 15, 700, 200, 20000, 300
    for f, w in ((jub, 2), (bn, 4), (bls, 2)):
        assert A.lib.anemoi_init(0, f, w) == 0          # constant tables up front: nothing but launches below

    def fresh():
        leaves = rng.integers(0, 1 << 60, size=(1 << depth, 4), dtype=np.uint64)      # limbs < 2^60: canonical elements
        msgs = rng.integers(0, 256, size=(nmsg, mlen), dtype=np.uint8)
        states = rng.integers(0, 1 << 60, size=(nst, 2, 6), dtype=np.uint64)
        return leaves, msgs, states

    # the ragged batch: the lengths stay (offsets are captured data like everything else), the bytes change per replay
    rag_lens = rng.integers(0, 400, size=nrag)
    rag_offs = np.zeros(nrag + 1, dtype=np.uint64)
    rag_offs[1:] = np.cumsum(rag_lens, dtype=np.uint64)
    d_rag_offs = torch.from_numpy(rag_offs.view(np.int64)).to(dev)
    d_rag = torch.zeros(int(rag_offs[-1]) + 1, dtype=torch.uint8, device=dev)
    d_rag_out = torch.zeros(nrag * 4, dtype=torch.int64, device=dev)
    rag_need = A.lib.anemoi_ragged_scratch_bytes(nrag)
    d_rag_scr = torch.empty(rag_need, dtype=torch.uint8, device=dev)
    rag = {}

    def fresh_ragged():
        rag["blob"] = rng.integers(0, 256, size=int(rag_offs[-1]) + 1, dtype=np.uint8)
        d_rag.copy_(torch.from_numpy(rag["blob"]).to(dev))

    def i64(a):
        return torch.from_numpy(a.view(np.int64).reshape(-1)).to(dev)

    leaves, msgs, states = fresh()
    d_leaves, d_msgs, d_states = i64(leaves), torch.from_numpy(msgs.reshape(-1)).to(dev), i64(states)
    d_scratch = torch.empty((1 << depth) * 4, dtype=torch.int64, device=dev)
    d_root = torch.zeros(4, dtype=torch.int64, device=dev)
    d_dig = torch.zeros(nmsg * 4, dtype=torch.int64, device=dev)
    d_out = torch.zeros(nst * 6, dtype=torch.int64, device=dev)

    def enqueue(stream):
        s = stream.cuda_stream
        assert A.lib.anemoi_merkle_root_dev(jub, d_leaves.data_ptr(), depth, d_scratch.data_ptr(), d_root.data_ptr(), s) == 0
        assert A.lib.anemoi_hash_bytes_dev(bn, 4, d_msgs.data_ptr(), mlen, nmsg, d_dig.data_ptr(), s) == 0
        assert A.lib.anemoi_jive_compress_k_dev(bls, 2, 2, d_states.data_ptr(), d_out.data_ptr(), nst, s) == 0
        assert A.lib.anemoi_hash_bytes_ragged_bucketed_dev(jub, 2, d_rag.data_ptr(), d_rag.numel(), d_rag_offs.data_ptr(), nrag, d_rag_out.data_ptr(),
                                                           d_rag_scr.data_ptr(), rag_need, s) == 0

    def check(what):
        torch.cuda.synchronize()
        got_root = d_root.cpu().numpy().view(np.uint64)
        assert (got_root == oracle.merkle_root(jub, leaves, depth)).all(), what + ": Merkle root"
        got_dig = d_dig.cpu().numpy().view(np.uint64).reshape(nmsg, 4)
        assert (got_dig == oracle.hash_bytes_batch(bn, 4, msgs, threads=8)).all(), what + ": sponge digests"
        got_out = d_out.cpu().numpy().view(np.uint64).reshape(nst, 6)
        assert (got_out == oracle.compress_batch(bls, 2, states, threads=8).reshape(nst, 6)).all(), what + ": Jive outputs"
        got_rag = d_rag_out.cpu().numpy().view(np.uint64).reshape(nrag, 4)
        assert d_rag_scr[:4].cpu().numpy().view(np.uint32)[0] == 0, "%s: the status word is set" % what
        order = d_rag_scr.cpu().numpy()[(4 + 65536) * 4:].view(np.uint32)     # status word (+ padding), counters, order
        assert sorted(order.tolist()) == list(range(nrag)), what + ": the bucketing's order is not a permutation"
        for i in range(0, nrag, 7):
            m = rag["blob"][int(rag_offs[i]):int(rag_offs[i + 1])].tobytes()
            assert (got_rag[i] == oracle.hash_bytes(jub, 2, m)).all(), what + ": ragged digest %d" % i

    fresh_ragged()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        enqueue(torch.cuda.current_stream())
    for replay in range(3):
        if replay:
            leaves, msgs, states = fresh()
            d_leaves.copy_(i64(leaves)), d_msgs.copy_(torch.from_numpy(msgs.reshape(-1)).to(dev)), d_states.copy_(i64(states))
        d_root.zero_(), d_dig.zero_(), d_out.zero_(), d_rag_out.zero_(), fresh_ragged()
        graph.replay()
        check("replay %d" % replay)

    # other cut-offs: the same sizes would now take other kernels (no fold kernels, no scan kernels: everything lane-private)
    with A.options(coop2d_max=0, coop2d43_max=0, coop4_max=0, coop43_max=0, coop_sponge_max=0):
        leaves, msgs, states = fresh()
        d_leaves.copy_(i64(leaves)), d_msgs.copy_(torch.from_numpy(msgs.reshape(-1)).to(dev)), d_states.copy_(i64(states))
        d_root.zero_(), d_dig.zero_(), d_out.zero_(), d_rag_out.zero_(), fresh_ragged()
        graph.replay()                                   # still the kernels chosen at capture time
        check("replay after the options changed")
        d_root.zero_(), d_dig.zero_(), d_out.zero_(), d_rag_out.zero_(), fresh_ragged()
        enqueue(torch.cuda.current_stream())             # a direct call takes the new route
        check("direct call on the new route")
    d_root.zero_(), d_dig.zero_(), d_out.zero_(), d_rag_out.zero_(), fresh_ragged()
    graph.replay()
    check("replay after the options were restored")
    # round 6: the offsets are captured DATA -- corrupt them in device memory (a pair that decreases), replay: the status word
    # says so, the ragged digests are zero, nothing faulted and the graph's other work is right; restore them, replay: the
    # word is 0 again (a kernel of the graph zeroes it on every replay) and the digests are the messages' own
    bad = rag_offs.copy()
    bad[nrag // 3 + 1] = np.uint64(1) << np.uint64(63)               # (the NEXT pair decreases; followed, it would read 2^63 bytes)
    d_rag_offs.copy_(torch.from_numpy(bad.view(np.int64)).to(dev))
    d_rag_out.fill_(-1)
    graph.replay()
    torch.cuda.synchronize()
    assert d_rag_scr[:4].cpu().numpy().view(np.uint32)[0] == 1, "a decreasing pair under replay: status word"
    assert not d_rag_out.cpu().numpy().any(), "malformed offsets under replay: zero digests"
    assert (d_root.cpu().numpy().view(np.uint64) == oracle.merkle_root(jub, leaves, depth)).all()
    d_rag_offs.copy_(torch.from_numpy(rag_offs.view(np.int64)).to(dev))
    d_root.zero_(), d_dig.zero_(), d_out.zero_(), d_rag_out.zero_(), fresh_ragged()
    graph.replay()
    check("replay after the offsets were repaired")


def test_every_other_dev_entry_point_replays_like_a_direct_call(oracle, params):
    """The rest of the `_dev` API in ONE captured graph -- permutation (in place), sbox_layer, hash_field, the retained
    Merkle tree (a device-to-device copy node in front of its launches), path climbing, both Montgomery conversions, the
    three remaining ragged forms (bytes in order, elements in order, elements bucketed) and the four entry points of a
    prepared run-time instance -- replayed three times on fresh inputs: every output must equal what the same calls give
    when made directly on the same inputs (the direct calls are what the rest of the suite checks against the oracle; one
    output is checked against it here as well), and must not be the zeros the buffers held before."""
    import torch
    import anemoi_amd as A
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(88)
    jub = FIELD_IDS.index("jubjub")
    L, n, depth, epm = 4, 300, 7, 5
    for w in (2, 4):
        assert A.lib.anemoi_init(0, jub, w) == 0
    ins = params["jubjub"]["instances"]["anemoi_2_1"]
    enc = lambda vals: oracle.ints_to_mont(jub, [int(v) for v in vals])
    prep = A.GenericAnemoi("jubjub", 1, ins["num_rounds"], enc(ins["ark_c"]), enc(ins["ark_d"])).prepare()

    def elems(*shape):
        return rng.integers(0, 1 << 60, size=shape + (L,), dtype=np.uint64)      # limbs < 2^60: canonical

    counts = rng.integers(0, 9, size=n)
    eoffs = np.zeros(n + 1, dtype=np.uint64)
    eoffs[1:] = np.cumsum(counts, dtype=np.uint64)
    blens = rng.integers(0, 200, size=n)
    boffs = np.zeros(n + 1, dtype=np.uint64)
    boffs[1:] = np.cumsum(blens, dtype=np.uint64)
    host = {}

    def fresh():
        host.update(perm=elems(n, 2), sbox=elems(n, 4), hf=elems(n, epm), leaves=elems(1 << depth), climb_leaves=elems(n),
                    paths=elems(n, depth), mont=elems(n), relems=elems(int(eoffs[-1]) + 1),
                    rbytes=rng.integers(0, 256, size=int(boffs[-1]) + 1, dtype=np.uint8), gmsgs=rng.integers(0, 256, size=(n, 77), dtype=np.uint8))

        host["gperm"] = host["perm"].copy()      # the run-time instance holds the shipped constants: same permutation

    fresh()
    t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1)).to(dev)
    zeros = lambda k: torch.zeros(k * L, dtype=torch.int64, device=dev)
    d = {k: (torch.from_numpy(v.reshape(-1)).to(dev) if v.dtype == np.uint8 else t64(v)) for k, v in host.items()}
    d_index = t64(rng.integers(0, 1 << depth, size=n, dtype=np.uint64))
    d_eoffs, d_boffs = t64(eoffs), t64(boffs)
    need = A.lib.anemoi_ragged_scratch_bytes(n)
    d_scr = torch.empty(need, dtype=torch.uint8, device=dev)
    out = dict(hf=zeros(n), tree=zeros(2 << depth), roots=zeros(n), to_m=zeros(n), from_m=zeros(n), rb=zeros(n), re=zeros(n),
               reb=zeros(n), gjive=zeros(n), ghf=zeros(n), ghb=zeros(n))
    p = lambda t: t.data_ptr()
    lib = A.lib

    def enqueue(stream):
        s = stream.cuda_stream
        assert lib.anemoi_permutation_dev(jub, 2, p(d["perm"]), n, s) == 0
        assert lib.anemoi_sbox_layer_dev(jub, 4, p(d["sbox"]), n, s) == 0
        assert lib.anemoi_hash_field_dev(jub, 4, p(d["hf"]), epm, n, p(out["hf"]), s) == 0
        assert lib.anemoi_merkle_tree_dev(jub, p(d["leaves"]), depth, p(out["tree"]), s) == 0
        assert lib.anemoi_merkle_climb_dev(jub, p(d["climb_leaves"]), p(d_index), p(d["paths"]), depth, n, p(out["roots"]), s) == 0
        assert lib.anemoi_from_montgomery_dev(jub, p(d["mont"]), p(out["from_m"]), n, s) == 0
        assert lib.anemoi_to_montgomery_dev(jub, p(out["from_m"]), p(out["to_m"]), n, s) == 0
        assert lib.anemoi_hash_bytes_ragged_dev(jub, 2, p(d["rbytes"]), p(d_boffs), n, p(out["rb"]), s) == 0
        assert lib.anemoi_hash_field_ragged_dev(jub, 4, p(d["relems"]), p(d_eoffs), n, p(out["re"]), s) == 0
        assert lib.anemoi_hash_field_ragged_bucketed_dev(jub, 2, p(d["relems"]), d["relems"].numel() // L, p(d_eoffs), n, p(out["reb"]), p(d_scr), need, s) == 0
        prep.permutation_dev(p(d["gperm"]), n, s)
        prep.compress_k_dev(2, p(d["perm"]), p(out["gjive"]), n, s)       # (reads what anemoi_permutation_dev left in place)
        prep.hash_field_dev(1, p(d["hf"]), epm, n, p(out["ghf"]), s)
        prep.hash_bytes_dev(1, p(d["gmsgs"]), 77, n, p(out["ghb"]), s)

    def load():
        for k, v in host.items():
            d[k].copy_(torch.from_numpy(v.reshape(-1)).to(dev) if v.dtype == np.uint8 else t64(v))
        for t in out.values():
            t.zero_()

    def collect():
        torch.cuda.synchronize()
        got = {k: t.cpu().numpy().copy() for k, t in out.items()}
        got.update(perm=d["perm"].cpu().numpy().copy(), sbox=d["sbox"].cpu().numpy().copy(), gperm=d["gperm"].cpu().numpy().copy())
        return got

    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        enqueue(torch.cuda.current_stream())
    for replay in range(3):
        fresh()
        load()
        graph.replay()
        got = collect()
        load()
        enqueue(torch.cuda.current_stream())
        want = collect()
        for k in want:
            assert (got[k] == want[k]).all(), "replay %d: %s differs from the direct call" % (replay, k)
            assert got[k].any(), k
        assert (got["to_m"].view(np.uint64).reshape(n, L) == host["mont"]).all()          # from_montgomery then to_montgomery
        i = int(np.argmax(counts))
        m = host["relems"][int(eoffs[i]):int(eoffs[i + 1])]
        assert (got["re"].view(np.uint64).reshape(n, L)[i] == oracle.hash_field(jub, 4, m)).all()
        assert (got["reb"].view(np.uint64).reshape(n, L)[i] == oracle.hash_field(jub, 2, m)).all()
        assert (got["perm"] == got["gperm"]).all()          # fixed instance = run-time instance fed with its constants
