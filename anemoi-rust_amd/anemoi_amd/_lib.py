import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)

FIELD_IDS = ["bls12_381", "bls12_377", "bn_254", "ed_on_bls12_377", "jubjub", "pallas", "vesta"]
ALL_DEVICES = -1

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_sz = ctypes.c_size_t
_int = ctypes.c_int
_vp = ctypes.c_void_p


class AnemoiError(RuntimeError):
    """An error code from the C-ABI (the reference's assert! panics map to ANEMOI_ERR_ARG = -3)."""

    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("anemoi_mi355x error %d: %s%s" % (code, _strerror(code), (" (%s)" % detail) if detail else ""))


def lib_path():
    return os.environ.get("ANEMOI_MI355X_LIB", os.path.join(_ROOT, "lib", "libanemoi_mi355x.so"))


def _load():
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "libanemoi_mi355x.so not found at %s: build it with `make -C anemoi-rust_amd -j8` "
            "(or __graft_entry__.build()). There is no CPU fallback." % path)
    # PyTorch (device memory / streams plumbing in bench.py and tests) bundles its own HIP runtime;
    # load it first so this library binds to the same libamdhip64 instance.
    if os.environ.get("ANEMOI_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    return ctypes.CDLL(path)


lib = _load()

class _GenericInstance(ctypes.Structure):
    """struct anemoi_generic_instance (include/anemoi_mi355x.h)"""
    _fields_ = [("field", _int), ("num_columns", _int), ("num_rounds", _int),
                ("ark_c", _u64p), ("ark_d", _u64p), ("mds", _u64p)]


_gip = ctypes.POINTER(_GenericInstance)

_SIGS = {
    "anemoi_abi_version": ([], _int),
    "anemoi_device_count": ([], _int),
    "anemoi_strerror": ([_int], ctypes.c_char_p),
    "anemoi_last_error": ([], ctypes.c_char_p),
    "anemoi_field_id": ([ctypes.c_char_p], _int),
    "anemoi_field_name": ([_int], ctypes.c_char_p),
    "anemoi_field_limbs": ([_int], _int),
    "anemoi_field_chunk_bytes": ([_int], _int),
    "anemoi_num_rounds": ([_int, _int], _int),
    "anemoi_init": ([_int, _int, _int], _int),
    "anemoi_release": ([_int], _int),
    "anemoi_warmup": ([_int, _int, _int], _int),
    "anemoi_probe_issue_rate": ([_int] + [ctypes.POINTER(ctypes.c_double)] * 4, _int),
    "anemoi_clock_sampler_bytes": ([], _sz),
    "anemoi_clock_sampler_start_dev": ([_vp, _sz, ctypes.c_uint, ctypes.c_uint, _vp], _int),
    "anemoi_clock_sampler_wait_dev": ([_vp, ctypes.c_uint, _vp], _int),
    "anemoi_clock_sampler_stop_dev": ([_vp, _vp], _int),
    "anemoi_clock_stamp_dev": ([_vp, _vp], _int),
    "anemoi_clock_sampler_read": ([_vp, _sz, ctypes.c_ulonglong, ctypes.c_ulonglong] + [ctypes.POINTER(ctypes.c_double)] * 3
                                  + [ctypes.POINTER(_int)], _int),
    "anemoi_set_option": ([ctypes.c_char_p, ctypes.c_longlong], _int),
    "anemoi_get_option": ([ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)], _int),
    "anemoi_permutation_batch": ([_int, _int, _u64p, _sz, _int], _int),
    "anemoi_sbox_layer_batch": ([_int, _int, _u64p, _sz, _int], _int),
    "anemoi_sbox_layer_dev": ([_int, _int, _vp, _sz, _vp], _int),
    "anemoi_jive_compress_batch": ([_int, _int, _u64p, _u64p, _sz, _int], _int),
    "anemoi_jive_compress_k_batch": ([_int, _int, _int, _u64p, _u64p, _sz, _int], _int),
    "anemoi_merge_batch": ([_int, _u64p, _u64p, _sz, _int], _int),
    "anemoi_hash_field_batch": ([_int, _int, _u64p, _sz, _sz, _u64p, _int], _int),
    "anemoi_hash_bytes_batch": ([_int, _int, _u8p, _sz, _sz, _u64p, _int], _int),
    "anemoi_hash_bytes_ragged_batch": ([_int, _int, _u8p, _u64p, _sz, _u64p, _int], _int),
    "anemoi_hash_field_ragged_batch": ([_int, _int, _u64p, _u64p, _sz, _u64p, _int], _int),
    "anemoi_hash_bytes_ragged_dev": ([_int, _int, _vp, _vp, _sz, _vp, _vp], _int),
    "anemoi_ragged_scratch_bytes": ([_sz], _sz),
    "anemoi_hash_bytes_ragged_bucketed_dev": ([_int, _int, _vp, _sz, _vp, _sz, _vp, _vp, _sz, _vp], _int),
    "anemoi_hash_field_ragged_dev": ([_int, _int, _vp, _vp, _sz, _vp, _vp], _int),
    "anemoi_hash_field_ragged_bucketed_dev": ([_int, _int, _vp, _sz, _vp, _sz, _vp, _vp, _sz, _vp], _int),
    "anemoi_merkle_root": ([_int, _u64p, ctypes.c_uint, _u64p, _int], _int),
    "anemoi_merkle_tree": ([_int, _u64p, ctypes.c_uint, _u64p, _int], _int),
    "anemoi_merkle_path": ([_int, _u64p, ctypes.c_uint, _sz, _u64p], _int),
    "anemoi_merkle_verify_batch": ([_int, _u64p, _u64p, _u64p, ctypes.c_uint, _sz, _u64p, _u8p, _int], _int),
    "anemoi_merkle_root_arity4": ([_int, _u64p, ctypes.c_uint, _u64p, _int], _int),
    "anemoi_merkle_tree_arity4": ([_int, _u64p, ctypes.c_uint, _u64p, _int], _int),
    "anemoi_merkle_path_arity4": ([_int, _u64p, ctypes.c_uint, _sz, _u64p], _int),
    "anemoi_merkle_verify_arity4_batch": ([_int, _u64p, _u64p, _u64p, ctypes.c_uint, _sz, _u64p, _u8p, _int], _int),
    "anemoi_merkle_tree_dev": ([_int, _vp, ctypes.c_uint, _vp, _vp], _int),
    "anemoi_merkle_climb_dev": ([_int, _vp, _vp, _vp, ctypes.c_uint, _sz, _vp, _vp], _int),
    "anemoi_to_montgomery": ([_int, _u64p, _u64p, _sz, _int], _int),
    "anemoi_from_montgomery": ([_int, _u64p, _u64p, _sz, _int], _int),
    "anemoi_permutation_dev": ([_int, _int, _vp, _sz, _vp], _int),
    "anemoi_jive_compress_k_dev": ([_int, _int, _int, _vp, _vp, _sz, _vp], _int),
    "anemoi_hash_field_dev": ([_int, _int, _vp, _sz, _sz, _vp, _vp], _int),
    "anemoi_hash_bytes_dev": ([_int, _int, _vp, _sz, _sz, _vp, _vp], _int),
    "anemoi_merkle_root_dev": ([_int, _vp, ctypes.c_uint, _vp, _vp, _vp], _int),
    "anemoi_to_montgomery_dev": ([_int, _vp, _vp, _sz, _vp], _int),
    "anemoi_generic_mds_matrix": ([_int, _int, _u64p, _int], _int),
    "anemoi_generic_permutation_batch": ([_gip, _u64p, _sz, _int], _int),
    "anemoi_generic_jive_compress_k_batch": ([_gip, _int, _u64p, _u64p, _sz, _int], _int),
    "anemoi_generic_hash_field_batch": ([_gip, _int, _u64p, _sz, _sz, _u64p, _int], _int),
    "anemoi_generic_hash_bytes_batch": ([_gip, _int, _u8p, _sz, _sz, _u64p, _int], _int),
    "anemoi_exp_alpha_batch": ([_int, _int, _u64p, _sz, _int], _int),
    "anemoi_generic_prepare": ([_gip, _int, ctypes.POINTER(_vp)], _int),
    "anemoi_generic_destroy": ([_vp], _int),
    "anemoi_generic_permutation_dev": ([_vp, _vp, _sz, _vp], _int),
    "anemoi_generic_jive_compress_k_dev": ([_vp, _int, _vp, _vp, _sz, _vp], _int),
    "anemoi_generic_hash_field_dev": ([_vp, _int, _vp, _sz, _sz, _vp, _vp], _int),
    "anemoi_generic_hash_bytes_dev": ([_vp, _int, _vp, _sz, _sz, _vp, _vp], _int),
    "anemoi_from_montgomery_dev": ([_int, _vp, _vp, _sz, _vp], _int),
}
for _name, (_args, _res) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export the header's symbol
    _fn.argtypes, _fn.restype = _args, _res


def _strerror(code):
    return lib.anemoi_strerror(code).decode()


def _check(rc):
    if rc != 0:
        raise AnemoiError(rc, lib.anemoi_last_error().decode() if rc == -4 else "")


def device_count():
    n = lib.anemoi_device_count()
    if n < 0:
        raise AnemoiError(n, lib.anemoi_last_error().decode())
    return n


def init(field, width, device=ALL_DEVICES):
    """Upload the constant tables of (field, width) and create one lane ahead of time (anemoi_init)."""
    rc = lib.anemoi_init(device, field_id(field), width)
    if rc != 0:
        raise AnemoiError(rc, lib.anemoi_last_error().decode())


def warmup(field, width, device=ALL_DEVICES):
    """anemoi_init, then one small launch of every throughput kernel of (field, width) (anemoi_warmup): a service that cares
    about the duration of its FIRST large call does this at start-up."""
    rc = lib.anemoi_warmup(device, field_id(field), width)
    if rc != 0:
        raise AnemoiError(rc, lib.anemoi_last_error().decode())


def probe_issue_rate(device=0):
    """anemoi_probe_issue_rate on `device`: (lane multiply-adds per second, shader clock in GHz) of a full grid of bare
    dependent v_mad_u64_u32 chains, then the same two figures for a chain of the generated BLS12-381 squaring -- what
    this box delivers of the instruction, and of the instruction mix, the throughput kernels are made of."""
    v = [ctypes.c_double(0) for _ in range(4)]
    rc = lib.anemoi_probe_issue_rate(device, *[ctypes.byref(x) for x in v])
    if rc != 0:
        raise AnemoiError(rc, lib.anemoi_last_error().decode())
    return tuple(x.value for x in v)


class ClockSampler:
    """The shader clock the chip holds WHILE work runs on `work_stream` (anemoi_clock_sampler_*; torch is plumbing: device
    memory and streams).  start() before the work is enqueued, finish() after it is enqueued -- everything is ordered on
    the device, the host never waits in between -- then read() once the device is idle:
        cs = ClockSampler(device); cs.start(work_stream); ...enqueue work...; cs.finish(work_stream); torch.cuda.synchronize()
        mean_ghz, min_ghz, max_ghz, groups = cs.read()"""

    def __init__(self, device, period_us=2000, max_ms=120000, wait_ms=1000, stream_priorities="auto", reuse_streams=True):
        import torch
        self.torch, self.period_us, self.max_ms, self.wait_ms = torch, period_us, max_ms, wait_ms
        self.bytes = lib.anemoi_clock_sampler_bytes()
        self.buf = torch.zeros(self.bytes, dtype=torch.uint8, device=device)
        self.stamps = torch.zeros(2, dtype=torch.int64, device=device)
        # The sampler never ends by itself, so its stream must not share a HARDWARE QUEUE with the work's stream: HIP
        # multiplexes the streams of one priority onto GPU_MAX_HW_QUEUES (4) queues, round-robin in the order they are
        # created, and runs the kernels of streams that share a queue one after the other -- the work then waits until the
        # sampler's log is full (4 096 periods: 0.4 ... 8 s) and runs without a sampler (round 6: "no sample beside the
        # work" in some test orders; tools/exp_sampler_queue_collision.py: every fourth sampler when each takes ONE
        # default-priority stream).  Streams of different priorities never share a queue, and this platform has two
        # (torch.cuda.Stream.priority_range() = (0, -1)): the sampler runs on a HIGH-priority stream -- it sleeps, so its
        # priority costs the work nothing -- and its stop on a default-priority one (which may share a queue with the
        # work: it is ordered behind the work anyway).  The work (the caller's stream) is assumed to have the default
        # priority.  stream_priorities=None: two default-priority streams, as round 5 did (the A/B).
        # AND the streams are used once, and waited for, BEFORE anything is measured, and kept for the next sampler of the
        # process: the hardware queue behind a stream is created at its first submission, and a queue created while a
        # launch is in flight disturbs that launch's placement (tools/exp_sampler_priority_placement.py: config 3 488 ms
        # instead of 335 behind a sampler whose queue was new, back to back; never behind one whose queue existed).
        if os.environ.get("ANEMOI_SAMPLER_STREAMS") == "default":      # (the A/B from outside: tools/measure_cycles.py)
            stream_priorities = None
        if stream_priorities == "auto":
            lo, hi = self.priority_range()
            stream_priorities = (hi, lo)
        key = (str(device), stream_priorities)
        ent = ClockSampler._streams.get(key) if reuse_streams else None      # (reuse_streams=False: the experiments)
        if ent is None or ent["busy"]:
            if stream_priorities is None:
                pair = (torch.cuda.Stream(device), torch.cuda.Stream(device))
            else:
                pair = (torch.cuda.Stream(device, priority=stream_priorities[0]), torch.cuda.Stream(device, priority=stream_priorities[1]))
            for st in pair:
                with torch.cuda.stream(st):
                    torch.zeros(1, device=device)
                st.synchronize()
            fresh = {"pair": pair, "busy": False}
            if ent is None and reuse_streams:
                ClockSampler._streams[key] = fresh
            ent = fresh
        self._ent = ent
        self.side, self.third = ent["pair"]

    _streams = {}

    @staticmethod
    def priority_range():
        """(lowest, highest) stream priority of the device as torch numbers them (a smaller number is a higher priority)"""
        import torch
        try:
            least, greatest = torch.cuda.Stream.priority_range()
        except Exception:   # noqa: BLE001  (older torch: -1 = high, 0 = default)
            least, greatest = 0, -1
        return least, greatest

    def start(self, work_stream):
        self._ent["busy"] = True
        self.side.wait_stream(self.torch.cuda.current_stream())     # (the buffers were zero-filled on the current stream)
        _check(lib.anemoi_clock_sampler_start_dev(self.buf.data_ptr(), self.bytes, self.period_us, self.max_ms, self.side.cuda_stream))
        # the work starts once the sampler runs: the wait ends with the sampler's first sample, so a generous bound costs
        # nothing -- with 50 ms a sampler whose (new) stream took longer to start was seen to miss 100 ms of work entirely
        _check(lib.anemoi_clock_sampler_wait_dev(self.buf.data_ptr(), self.wait_ms, work_stream.cuda_stream))
        _check(lib.anemoi_clock_stamp_dev(self.stamps.data_ptr(), work_stream.cuda_stream))

    def finish(self, work_stream):
        """second stamp behind the work, then the stop ordered behind THAT on a stream of its own (the sampler's stream never
        ends by itself, so nothing may wait for it before the stop is queued)"""
        _check(lib.anemoi_clock_stamp_dev(self.stamps.data_ptr() + 8, work_stream.cuda_stream))
        ev = self.torch.cuda.Event()
        ev.record(work_stream)
        self.third.wait_event(ev)
        _check(lib.anemoi_clock_sampler_stop_dev(self.buf.data_ptr(), self.third.cuda_stream))

    def read(self):
        host = self.buf.cpu().numpy()       # (waits for the device: the sampler has been stopped, its streams are free again)
        self._ent["busy"] = False
        st = self.stamps.cpu().numpy().view(np.uint64)
        v = [ctypes.c_double(0) for _ in range(3)]
        g = ctypes.c_int(0)
        _check(lib.anemoi_clock_sampler_read(host.ctypes.data, self.bytes, int(st[0]), int(st[1]), ctypes.byref(v[0]),
                                             ctypes.byref(v[1]), ctypes.byref(v[2]), ctypes.byref(g)))
        return v[0].value, v[1].value, v[2].value, g.value

    def diagnose(self):
        """why read() found no usable group: per sampler workgroup, how many samples it took and how many of them lie between
        the caller's two stamps (layout of the log: csrc/capi.hip, SamplerBuf)"""
        host = self.buf.cpu().numpy()
        st = self.stamps.cpu().numpy().view(np.uint64)
        out, off, per = [], 16, 8 + 4096 * 16
        for i in range(16):
            g = host[off + i * per: off + (i + 1) * per]
            count = int(g[:4].view(np.uint32)[0])
            wall = g[8:8 + 16 * min(count, 4096)].view(np.uint64)[0::2]
            inside = int(((wall >= st[0]) & (wall <= st[1])).sum())
            out.append((count, inside))
        return {"stamps_ticks": (int(st[0]), int(st[1])), "work_ms": (int(st[1]) - int(st[0])) * 1e-5, "samples_taken_and_inside": out}


def kernel_Mcycles(device, enqueue, reps=3, warmup=1, best=False):
    """The box-independent cost of a piece of device work: `enqueue(stream)` (one or more `_dev` calls) run `reps` times with
    the clock sampler beside it.  Returns (ms per repetition from HIP events on the work's stream, the SLOWEST sampled XCD
    clock in GHz, their product in millions of shader cycles) -- bench.py's `alu.kernel_Mcycles_slowest_xcd` for any
    kernel: a slower box moves the clock, a slower build moves the cycles.  best: the fastest
    repetition instead of the mean (a budget test: one repetition that met a launch-order effect must not fail a build)."""
    import torch
    stream = torch.cuda.current_stream(device)
    for _ in range(warmup):
        enqueue(stream)
    torch.cuda.synchronize(device)
    for attempt in (0, 1):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        cs = ClockSampler(device)
        cs.start(stream)
        for a, b in evs:
            a.record(stream)
            enqueue(stream)
            b.record(stream)
        cs.finish(stream)
        torch.cuda.synchronize(device)
        _, ghz_min, _, groups = cs.read()
        if groups and ghz_min > 0:
            break
        kernel_Mcycles.last_miss = cs.diagnose()        # (kept for the caller: a measurement without its clock is repeated once)
        if attempt:
            raise RuntimeError("the clock sampler took no sample beside the work, twice: %r" % (kernel_Mcycles.last_miss,))
    each = [a.elapsed_time(b) for a, b in evs]
    ms = min(each) if best else sum(each) / reps
    kernel_Mcycles.last_each_ms = each
    return ms, ghz_min, ms * ghz_min


def release(device=ALL_DEVICES):
    """Free everything the library holds on `device` (anemoi_release); it re-initialises lazily afterwards."""
    rc = lib.anemoi_release(device)
    if rc != 0:
        raise AnemoiError(rc, lib.anemoi_last_error().decode())


OPTIONS = ("coop2d_max", "coop4_max", "coop43_max", "coop2d43_max", "coop_sponge_max", "coop_climb_max",
           "virtual_devices", "host_staging", "chunk_target_bytes", "test_quantum",
           "sponge_segment_bytes", "balance_underfilled", "lane_priorities")
AUTO = -1


def is_ab_build():
    """True when the loaded library is a laboratory build (`make AB=1`): it knows the option `coop_max`, which routes to the
    one-item-per-wavefront kernels the product does not contain."""
    v = ctypes.c_longlong(0)
    return lib.anemoi_get_option(b"coop_max", ctypes.byref(v)) == 0


def set_option(name, value):
    """anemoi_set_option: value AUTO (-1) / None restores the automatic default; host_staging also takes
    "pinned" / "direct".  Options are process-wide and affect calls that start afterwards."""
    if value is None:
        value = AUTO
    if name == "host_staging" and isinstance(value, str):
        value = {"pinned": 1, "direct": 0}[value]
    _check(lib.anemoi_set_option(name.encode(), int(value)))


def get_option(name):
    v = ctypes.c_longlong(0)
    _check(lib.anemoi_get_option(name.encode(), ctypes.byref(v)))
    return v.value


class options:
    """Context manager: `with options(coop4_max=0, virtual_devices=8): ...` sets the options and restores the
    previous values on exit (what the tests used to do through os.environ)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def field_id(field):
    if isinstance(field, str):
        fid = lib.anemoi_field_id(field.encode())
        if fid < 0:
            raise AnemoiError(fid)
        return fid
    return int(field)


def _p64(a):
    return a.ctypes.data_as(_u64p)


def _p8(a):
    return a.ctypes.data_as(_u8p)


def ints_to_limbs(ints, limbs):
    return np.array([[(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(limbs)] for v in ints],
                    dtype=np.uint64).reshape(-1, limbs)


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    arr = arr.reshape(-1, arr.shape[-1])
    return [sum(int(arr[r, i]) << (64 * i) for i in range(arr.shape[1])) for r in range(arr.shape[0])]


def to_montgomery(field, canon, device=0):
    """canonical little-endian limbs (rows of `limbs` uint64) -> Montgomery elements, on the GPU."""
    fid = field_id(field)
    a = np.ascontiguousarray(canon, dtype=np.uint64)
    out = np.empty_like(a)
    _check(lib.anemoi_to_montgomery(fid, _p64(a), _p64(out), a.size // lib.anemoi_field_limbs(fid), device))
    return out


def from_montgomery(field, mont, device=0):
    fid = field_id(field)
    a = np.ascontiguousarray(mont, dtype=np.uint64)
    out = np.empty_like(a)
    _check(lib.anemoi_from_montgomery(fid, _p64(a), _p64(out), a.size // lib.anemoi_field_limbs(fid), device))
    return out


class Anemoi:
    """One (field, width) instantiation, e.g. Anemoi("bls12_381", 2) == the reference's AnemoiBls12_381_2_1.

    Single-item methods keep the reference's names and argument meaning and run as a batch of one;
    `*_batch` methods are the GPU-shaped forms.  `device` is a HIP ordinal or ALL_DEVICES.
    """

    def __init__(self, field, width, device=0):
        self.field = field_id(field)
        if lib.anemoi_field_limbs(self.field) < 0:
            raise AnemoiError(-1)
        if width not in (2, 4):
            raise AnemoiError(-2)
        self.width, self.device = width, device
        self.limbs = lib.anemoi_field_limbs(self.field)
        self.chunk = lib.anemoi_field_chunk_bytes(self.field)
        self.rate = width - 1
        self.num_columns = width // 2
        self.num_rounds = lib.anemoi_num_rounds(self.field, width)

    # ---- encoding helpers
    def encode(self, ints):
        """canonical Python ints -> Montgomery limb rows (via the GPU conversion kernel)."""
        if len(ints) == 0:
            return np.zeros((0, self.limbs), dtype=np.uint64)
        return to_montgomery(self.field, ints_to_limbs(ints, self.limbs), self.device if self.device >= 0 else 0)

    def decode(self, mont):
        a = np.asarray(mont, dtype=np.uint64)
        if a.size == 0:
            return []
        return limbs_to_ints(from_montgomery(self.field, a.reshape(-1, self.limbs),
                                             self.device if self.device >= 0 else 0))

    # ---- batched operators
    def permutation_batch(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, self.width, self.limbs).copy()
        _check(lib.anemoi_permutation_batch(self.field, self.width, _p64(s), len(s), self.device))
        return s

    def sbox_layer_batch(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, self.width, self.limbs).copy()
        _check(lib.anemoi_sbox_layer_batch(self.field, self.width, _p64(s), len(s), self.device))
        return s

    def compress_k_batch(self, states, k):
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, self.width, self.limbs)
        if k <= 0 or self.width % k:
            raise AnemoiError(-3)
        out = np.empty((len(s), self.width // k, self.limbs), dtype=np.uint64)
        _check(lib.anemoi_jive_compress_k_batch(self.field, self.width, k, _p64(s), _p64(out), len(s), self.device))
        return out

    def compress_batch(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, self.width, self.limbs)
        out = np.empty((len(s), self.width // 2, self.limbs), dtype=np.uint64)
        _check(lib.anemoi_jive_compress_batch(self.field, self.width, _p64(s), _p64(out), len(s), self.device))
        return out

    def merge_batch(self, pairs):
        """2-1 instances: n x [left, right] digests -> n digests (anemoi_2_1/hasher.rs:87-92).
        4-3 instances: the reference's merge (hasher.rs:131-145) puts digests[0] in BOTH rate cells
        and returns permutation(state)[0]; reproduced here on top of permutation_batch."""
        p = np.ascontiguousarray(pairs, dtype=np.uint64).reshape(-1, 2, self.limbs)
        if self.width == 2:
            out = np.empty((len(p), self.limbs), dtype=np.uint64)
            _check(lib.anemoi_merge_batch(self.field, _p64(p), _p64(out), len(p), self.device))
            return out
        st = np.zeros((len(p), self.width, self.limbs), dtype=np.uint64)
        st[:, 0] = p[:, 0]
        st[:, 1] = p[:, 0]
        return self.permutation_batch(st)[:, 0].copy()

    def hash_field_batch(self, elems):
        e = np.ascontiguousarray(elems, dtype=np.uint64)
        assert e.ndim == 3 and e.shape[2] == self.limbs, "expected [n][elems_per_msg][limbs]"
        out = np.empty((e.shape[0], self.limbs), dtype=np.uint64)
        ptr = _p64(e) if e.size else None
        _check(lib.anemoi_hash_field_batch(self.field, self.width, ptr, e.shape[1], e.shape[0], _p64(out),
                                           self.device))
        return out

    def hash_batch(self, msgs):
        m = np.ascontiguousarray(msgs, dtype=np.uint8)
        assert m.ndim == 2, "expected [n][msg_len] bytes"
        out = np.empty((m.shape[0], self.limbs), dtype=np.uint64)
        ptr = _p8(m) if m.size else None
        _check(lib.anemoi_hash_bytes_batch(self.field, self.width, ptr, m.shape[1], m.shape[0], _p64(out),
                                           self.device))
        return out

    def hash_ragged(self, messages):
        """Sponge::hash of each of `messages` (a sequence of bytes-like objects of any lengths) in one launch."""
        lens = [len(m) for m in messages]
        offs = np.zeros(len(lens) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens, dtype=np.uint64)
        blob = np.frombuffer(b"".join(bytes(m) for m in messages), dtype=np.uint8)
        out = np.empty((len(lens), self.limbs), dtype=np.uint64)
        _check(lib.anemoi_hash_bytes_ragged_batch(self.field, self.width, _p8(blob) if blob.size else None, _p64(offs),
                                                  len(lens), _p64(out) if len(lens) else None, self.device))
        return out

    def hash_field_ragged(self, messages):
        """Sponge::hash_field of each of `messages` (a sequence of arrays of elements, (k_i, limbs) each, any k_i >= 0) in
        one launch."""
        arrs = [np.ascontiguousarray(m, dtype=np.uint64).reshape(-1, self.limbs) for m in messages]
        offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(a) for a in arrs], dtype=np.uint64)
        blob = np.concatenate(arrs) if arrs and int(offs[-1]) else np.zeros((0, self.limbs), dtype=np.uint64)
        out = np.empty((len(arrs), self.limbs), dtype=np.uint64)
        _check(lib.anemoi_hash_field_ragged_batch(self.field, self.width, _p64(blob) if blob.size else None, _p64(offs),
                                                  len(arrs), _p64(out) if len(arrs) else None, self.device))
        return out

    def merkle_root(self, leaves, depth):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, self.limbs)
        if len(lv) != 1 << depth:
            raise AnemoiError(-3)
        out = np.empty(self.limbs, dtype=np.uint64)
        _check(lib.anemoi_merkle_root(self.field, _p64(lv), depth, _p64(out), self.device))
        return out

    def merkle_tree(self, leaves, depth):
        """All levels: list [level0 (leaves), level1, ..., [root]] of numpy arrays."""
        lv = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, self.limbs)
        if len(lv) != 1 << depth:
            raise AnemoiError(-3)
        flat = np.empty(((2 << depth) - 1, self.limbs), dtype=np.uint64)
        _check(lib.anemoi_merkle_tree(self.field, _p64(lv), depth, _p64(flat), self.device))
        self._last_tree = flat
        out, off = [], 0
        for l in range(depth + 1):
            out.append(flat[off: off + (1 << (depth - l))])
            off += 1 << (depth - l)
        return out

    def merkle_path(self, tree_levels, depth, index):
        flat = np.ascontiguousarray(np.concatenate(tree_levels), dtype=np.uint64)
        path = np.empty((depth, self.limbs), dtype=np.uint64)
        rc = lib.anemoi_merkle_path(self.field, _p64(flat), depth, index, _p64(path) if depth else None)
        _check(rc)
        return path

    def merkle_verify_batch(self, leaves, indices, paths, depth, root):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, self.limbs)
        ix = np.ascontiguousarray(indices, dtype=np.uint64)
        pa = np.ascontiguousarray(paths, dtype=np.uint64).reshape(len(lv), depth, self.limbs)
        rt = np.ascontiguousarray(root, dtype=np.uint64).reshape(self.limbs)
        ok = np.zeros(len(lv), dtype=np.uint8)
        _check(lib.anemoi_merkle_verify_batch(self.field, _p64(lv), _p64(ix), _p64(pa) if pa.size else None, depth,
                                              len(lv), _p64(rt), _p8(ok), self.device))
        return ok.astype(bool)

    def merkle_root_arity4(self, leaves, depth4):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, self.limbs)
        if len(lv) != 1 << (2 * depth4):
            raise AnemoiError(-3)
        out = np.empty(self.limbs, dtype=np.uint64)
        _check(lib.anemoi_merkle_root_arity4(self.field, _p64(lv), depth4, _p64(out), self.device))
        return out

    def merkle_tree_arity4(self, leaves, depth4):
        """All levels of the arity-4 tree: list [leaves, level1, ..., [root]] of numpy arrays."""
        lv = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, self.limbs)
        if len(lv) != 1 << (2 * depth4):
            raise AnemoiError(-3)
        flat = np.empty((((len(lv) << 2) - 1) // 3, self.limbs), dtype=np.uint64)
        _check(lib.anemoi_merkle_tree_arity4(self.field, _p64(lv), depth4, _p64(flat), self.device))
        out, off = [], 0
        for l in range(depth4 + 1):
            out.append(flat[off: off + (1 << (2 * (depth4 - l)))])
            off += 1 << (2 * (depth4 - l))
        return out

    def merkle_path_arity4(self, tree_levels, depth4, index):
        flat = np.ascontiguousarray(np.concatenate(tree_levels), dtype=np.uint64)
        path = np.empty((3 * depth4, self.limbs), dtype=np.uint64)
        _check(lib.anemoi_merkle_path_arity4(self.field, _p64(flat), depth4, index, _p64(path) if depth4 else None))
        return path

    def merkle_verify_arity4_batch(self, leaves, indices, paths, depth4, root):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, self.limbs)
        ix = np.ascontiguousarray(indices, dtype=np.uint64)
        pa = np.ascontiguousarray(paths, dtype=np.uint64).reshape(len(lv), 3 * depth4, self.limbs)
        rt = np.ascontiguousarray(root, dtype=np.uint64).reshape(self.limbs)
        ok = np.zeros(len(lv), dtype=np.uint8)
        _check(lib.anemoi_merkle_verify_arity4_batch(self.field, _p64(lv), _p64(ix), _p64(pa) if pa.size else None,
                                                     depth4, len(lv), _p64(rt), _p8(ok), self.device))
        return ok.astype(bool)

    # ---- the reference's single-item surface (src/traits.rs:8-33)
    def permutation(self, state):
        return self.permutation_batch(np.asarray(state, dtype=np.uint64).reshape(1, self.width, self.limbs))[0]

    def compress(self, elems):
        e = np.asarray(elems, dtype=np.uint64).reshape(-1, self.limbs)
        if len(e) != self.width:  # assert!(elems.len() == STATE_WIDTH)
            raise AnemoiError(-3)
        return self.compress_batch(e[None])[0]

    def compress_k(self, elems, k):
        e = np.asarray(elems, dtype=np.uint64).reshape(-1, self.limbs)
        if len(e) != self.width:
            raise AnemoiError(-3)
        return self.compress_k_batch(e[None], k)[0]

    def hash(self, data):
        b = np.frombuffer(bytes(data), dtype=np.uint8).reshape(1, -1)
        return self.hash_batch(b)[0]

    def hash_field(self, elems):
        e = np.asarray(elems, dtype=np.uint64).reshape(1, -1, self.limbs)
        return self.hash_field_batch(e)[0]

    def merge(self, digests):
        d = np.asarray(digests, dtype=np.uint64).reshape(1, 2, self.limbs)
        return self.merge_batch(d)[0]

    def digest_to_bytes(self, digest):
        """AnemoiDigest::to_bytes (digest.rs:42-46): canonical little-endian bytes."""
        d = np.asarray(digest, dtype=np.uint64).reshape(1, self.limbs)
        return from_montgomery(self.field, d, self.device if self.device >= 0 else 0).tobytes()


def exp_alpha_batch(field, elems, inverse=False, device=0):
    """element-wise x^ALPHA (exp_by_alpha, src/traits.rs:94-104) or x^(1/ALPHA) (exp_by_inv_alpha)"""
    fid = field_id(field)
    limbs = lib.anemoi_field_limbs(fid)
    if limbs < 0:
        raise AnemoiError(-1)
    e = np.ascontiguousarray(elems, dtype=np.uint64).reshape(-1, limbs).copy()
    _check(lib.anemoi_exp_alpha_batch(fid, 1 if inverse else 0, _p64(e) if e.size else None, len(e), device))
    return e


def builtin_mds_matrix(field, num_columns, device=0):
    """the matrix of the reference's hard-coded mds_layer arm for 1..6 columns (Montgomery rows)"""
    fid = field_id(field)
    limbs = lib.anemoi_field_limbs(fid)
    if limbs < 0:
        raise AnemoiError(-1)
    out = np.zeros((max(num_columns, 0) ** 2, limbs), dtype=np.uint64)
    _check(lib.anemoi_generic_mds_matrix(fid, num_columns, _p64(out) if out.size else None, device))
    return out


class GenericAnemoi:
    """An Anemoi instance described by its trait constants (src/traits.rs:36-76): NUM_COLUMNS, NUM_ROUNDS,
    ARK_C, ARK_D and optionally an MDS matrix (None = the reference's hard-coded arm, <= 6 columns).
    Constants are Montgomery limb rows."""

    def __init__(self, field, num_columns, num_rounds, ark_c, ark_d, mds=None, device=0):
        self.field = field_id(field)
        self.limbs = lib.anemoi_field_limbs(self.field)
        if self.limbs < 0:
            raise AnemoiError(-1)
        self.num_columns, self.num_rounds, self.device = num_columns, num_rounds, device
        self.width = 2 * num_columns
        self._c = np.ascontiguousarray(ark_c, dtype=np.uint64).reshape(-1, self.limbs)
        self._d = np.ascontiguousarray(ark_d, dtype=np.uint64).reshape(-1, self.limbs)
        self._m = None if mds is None else np.ascontiguousarray(mds, dtype=np.uint64).reshape(-1, self.limbs)
        if len(self._c) != num_columns * num_rounds or len(self._d) != len(self._c):
            raise AnemoiError(-3)
        if self._m is not None and len(self._m) != num_columns * num_columns:
            raise AnemoiError(-3)
        self._inst = _GenericInstance(self.field, num_columns, num_rounds, _p64(self._c), _p64(self._d),
                                      _p64(self._m) if self._m is not None else None)

    def _inst_ref(self):
        return ctypes.byref(self._inst)

    def prepare(self, device=None):
        """anemoi_generic_prepare: constants uploaded and converted once -> PreparedGenericAnemoi (device pointers)"""
        return PreparedGenericAnemoi(self, self.device if device is None else device)

    def permutation_batch(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, self.width, self.limbs).copy()
        _check(lib.anemoi_generic_permutation_batch(ctypes.byref(self._inst), _p64(s), len(s), self.device))
        return s

    def compress_k_batch(self, states, k):
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, self.width, self.limbs)
        out = np.empty((len(s), self.width // k if k > 0 and self.width % k == 0 else 1, self.limbs), dtype=np.uint64)
        _check(lib.anemoi_generic_jive_compress_k_batch(ctypes.byref(self._inst), k, _p64(s), _p64(out), len(s),
                                                        self.device))
        return out

    def hash_field_batch(self, elems, rate):
        e = np.ascontiguousarray(elems, dtype=np.uint64)
        assert e.ndim == 3 and e.shape[2] == self.limbs, "expected [n][elems_per_msg][limbs]"
        out = np.empty((e.shape[0], self.limbs), dtype=np.uint64)
        _check(lib.anemoi_generic_hash_field_batch(ctypes.byref(self._inst), rate, _p64(e) if e.size else None,
                                                   e.shape[1], e.shape[0], _p64(out), self.device))
        return out

    def hash_batch(self, msgs, rate):
        m = np.ascontiguousarray(msgs, dtype=np.uint8)
        assert m.ndim == 2, "expected [n][msg_len] bytes"
        out = np.empty((m.shape[0], self.limbs), dtype=np.uint64)
        _check(lib.anemoi_generic_hash_bytes_batch(ctypes.byref(self._inst), rate, _p8(m) if m.size else None,
                                                   m.shape[1], m.shape[0], _p64(out), self.device))
        return out


class PreparedGenericAnemoi:
    """A run-time instance whose constants live on the device in the kernels' own form (anemoi_generic_prepare).
    The methods take raw device pointers (ints) and a HIP stream handle (int, 0 = the NULL stream), like the
    library's other `_dev` entry points; the caller's current device must be `device`."""

    def __init__(self, generic, device):
        self.generic, self.device = generic, device
        self.limbs, self.width = generic.limbs, generic.width
        h = ctypes.c_void_p()
        _check(lib.anemoi_generic_prepare(generic._inst_ref(), device, ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h is not None and self._h.value:
            _check(lib.anemoi_generic_destroy(self._h))
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def permutation_dev(self, d_states, n, stream=0):
        _check(lib.anemoi_generic_permutation_dev(self._h, d_states, n, stream))

    def compress_k_dev(self, k, d_in, d_out, n, stream=0):
        _check(lib.anemoi_generic_jive_compress_k_dev(self._h, k, d_in, d_out, n, stream))

    def hash_field_dev(self, rate, d_elems, elems_per_msg, n, d_out, stream=0):
        _check(lib.anemoi_generic_hash_field_dev(self._h, rate, d_elems, elems_per_msg, n, d_out, stream))

    def hash_bytes_dev(self, rate, d_msgs, msg_len, n, d_out, stream=0):
        _check(lib.anemoi_generic_hash_bytes_dev(self._h, rate, d_msgs, msg_len, n, d_out, stream))
