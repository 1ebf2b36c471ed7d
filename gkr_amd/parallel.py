"""One sumcheck split across GPUs.

GKR layer (the default form, `prove_sumcheck_opt_gate_sharded`): the layer's GATES are partitioned over the
ranks -- any partition works, contiguous ranges are used -- because every table the linear-time layer sumcheck
works on is a sum over gates: U(b), V(b) before the rounds that bind b, the row a_u(c), m_u(c) before the rounds
that bind c.  Two sum-over-ranks exchanges of 2 * 2^k field elements per layer, then every rank runs the 2k rounds
on the completed tables and holds the whole transcript (gkr_sumcheck_layer_sharded in include/gkr_amd.h; the
reference's counterpart is the rayon map-reduce over the gate list, sumcheck.rs:50-63, 97-124).

Plain multilinear sumcheck, and the older dense form of the layer: trailing-variable shards + one tiny all-reduce
per round, described next.

The reference sums each round's per-assignment polynomials with a rayon map-reduce
(rust/src/gkr/sumcheck.rs:50-63, 65-78, 97-124).  Across GPUs that reduce becomes one
all-reduce of <= 3 field elements per round:

  * rank p of P owns the hypercube entries whose index LOW bits are p (for the GKR layer: the
    gates whose right operand % P == p; W is replicated);
  * rounds bind the LEADING variable, so every pair (i, i + h) is rank-local for the first
    v - log2(P) rounds: each rank computes partial sums on its shard (the library's session
    API), the partials are all-reduced, every rank derives the same round vector and the same
    MiMC7 challenge, and folds its shard;
  * then every rank holds one entry per table; they are all-gathered and the last log2(P) rounds
    run redundantly on every rank.

RCCL has no modular sum: field elements travel as eight 32-bit limbs widened to int64, are summed
with ReduceOp.SUM (exact for < 2^31 ranks) and normalised mod r by every rank.

The algorithm is written once against two small interfaces -- a *shard* (sums / bind / tail) and a
*collective* -- so that the same code runs (a) one process per GPU over torch.distributed (RCCL on
the GPU box, gloo in the CPU tests) and (b) P logical ranks inside one process on one GPU, which is
how bit-exactness of the sharded algorithm is checked where only one device is visible.
"""

import ctypes
from typing import List, Sequence

import numpy as np

from . import _native as N
from .field import MODULUS, as_limbs, from_limbs, to_limbs
from .prover import Context, GkrError, Layer, multi_hash

_M32 = 0xFFFFFFFF


# ----------------------------------------------------------------------------- collectives

def _to_limb_list(values: Sequence[int]) -> List[int]:
    out = []
    for v in values:
        v %= MODULUS
        out.extend((v >> (32 * i)) & _M32 for i in range(8))
    return out


def _from_limb_sums(limbs: Sequence[int]) -> List[int]:
    """Eight (possibly > 32-bit) limb sums per element -> value mod r."""
    return [sum(int(limbs[8 * e + i]) << (32 * i) for i in range(8)) % MODULUS for e in range(len(limbs) // 8)]


class SingleProcess:
    """World of one rank."""
    rank, world = 0, 1

    def all_reduce_fr(self, values):
        return [v % MODULUS for v in values]

    def all_gather_fr(self, values):
        return [[v % MODULUS for v in values]]

    def all_reduce_or(self, flag):
        return bool(flag)


class TorchCollective:
    """torch.distributed backend: "nccl" (= RCCL over xGMI on the GPU box) or "gloo" (CPU tests)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if device is None:
            device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        self._device = device

    def all_reduce_fr(self, values):
        t = self._torch.tensor(_to_limb_list(values), dtype=self._torch.int64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        return _from_limb_sums(t.cpu().tolist())

    def all_gather_fr(self, values):
        t = self._torch.tensor(_to_limb_list(values), dtype=self._torch.int64, device=self._device)
        outs = [self._torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(outs, t, group=self._group)
        return [_from_limb_sums(o.cpu().tolist()) for o in outs]

    def all_reduce_or(self, flag):
        t = self._torch.tensor([1 if flag else 0], dtype=self._torch.int64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self._group)
        return bool(int(t.item()))

    def sum_limbs(self, limbs):
        """SUM all-reduce of an int64 limb array (the transport of make_allreduce_hook)."""
        t = self._torch.from_numpy(np.ascontiguousarray(limbs, dtype=np.int64)).to(self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        return t.cpu().numpy()

    def exchange(self, k_max=13, limbs=0):
        """What ResidentGates.sumcheck_raw takes as its sum over ranks: on RCCL the device exchange (the limbs never
        leave the GPU, the all-reduce is queued on the library's own stream), on gloo the host transport."""
        if self._dist.get_backend(self._group) == "nccl":
            return DeviceExchange(self, k_max, limbs)
        return self.sum_limbs

    def device_exchange(self, limbs):
        """A gkr_exchange_dev of `limbs` int64 whatever the backend (what sumcheck_mle_sharded_raw takes): RCCL queues the
        all-reduce on the library's stream; gloo -- ranks sharing a GPU in the tests, hosts without RCCL -- waits for the
        stream, sums on the host and copies back (StagedDeviceExchange)."""
        if self._dist.get_backend(self._group) == "nccl":
            return DeviceExchange(self, 1, limbs)
        return StagedDeviceExchange(self, limbs)


class DeviceExchange:
    """(Order of initialisation in a process: torch's GPU runtime first -- init_process_group / torch.cuda.init() -- then
    the first gkr Context; the other way round torch finds no device on this image.)
    gkr_exchange_dev over torch.distributed with backend nccl (= RCCL over xGMI): the limb buffer is a CUDA int64
    tensor the library widens into and narrows from; the hook queues ONE in-place SUM all-reduce on the library's HIP
    stream (wrapped as torch.cuda.ExternalStream), so nothing crosses PCIe and no stream is synchronised
    (include/gkr_amd.h, gkr_resident_layer_sumcheck_dev; the reference's counterpart is the rayon reduce of
    sumcheck.rs:50-63, 97-124)."""

    def __init__(self, coll: TorchCollective, k_max=13, limbs=0):
        torch, dist, group = coll._torch, coll._dist, coll._group
        self._buf = torch.zeros(max(int(limbs), int(N.lib().gkr_exchange_limbs(ctypes.c_int(k_max)))), dtype=torch.int64, device=coll._device)
        self.errors = []
        self.calls = 0
        self.backend = "nccl"
        self.world = coll.world

        def fn(_user, count, stream):
            try:
                self.calls += 1
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream or 0), device=self._buf.device)):
                    dist.all_reduce(self._buf[:count], op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception as e:   # never unwind through the C frame
                self.errors.append(e)
                return 1
        self._fn = N.ALLREDUCE_DEV_FN(fn)
        self.struct = N.ExchangeDev(self._fn, None, self._buf.data_ptr(), self._buf.numel())


class StagedDeviceExchange(DeviceExchange):
    """The same contract over a backend that cannot reduce device memory (gloo): the hook waits for the library's stream,
    sums the limbs on the host and copies them back before it returns -- correct, not fast; RCCL is the product path."""

    def __init__(self, coll: TorchCollective, limbs):
        import torch
        dist, group = coll._dist, coll._group
        self._buf = torch.zeros(int(limbs), dtype=torch.int64, device="cuda:%d" % torch.cuda.current_device())
        self.errors = []
        self.calls = 0
        self.backend = dist.get_backend(group)
        self.world = coll.world

        def fn(_user, count, stream):
            try:
                self.calls += 1
                ext = torch.cuda.ExternalStream(int(stream or 0), device=self._buf.device)
                ext.synchronize()
                host = self._buf[:count].cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                with torch.cuda.stream(ext):
                    self._buf[:count].copy_(host.to(self._buf.device))
                ext.synchronize()
                return 0
            except Exception as e:
                self.errors.append(e)
                return 1
        self._fn = N.ALLREDUCE_DEV_FN(fn)
        self.struct = N.ExchangeDev(self._fn, None, self._buf.data_ptr(), self._buf.numel())


class RcclExchange:
    """gkr_exchange_dev owned by the LIBRARY (csrc/exchange_rccl.cpp): RCCL loaded by the library itself, the all-reduce
    queued by its own hook -- no torch in the data path, the form a non-Python host uses.  unique_id: the 128 bytes of
    RcclExchange.unique_id() made on one rank and handed to the others (here: by torch.distributed's broadcast when there
    is a process group, see from_torch_group); create() blocks until every rank has called it."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(RcclExchange.ID_BYTES)
        rc = N.lib().gkr_exchange_rccl_unique_id(buf)
        if rc:
            raise GkrError(rc, RcclExchange._error())
        return buf.raw

    @staticmethod
    def _error():
        N.lib().gkr_exchange_rccl_error.restype = ctypes.c_char_p
        return (N.lib().gkr_exchange_rccl_error() or b"").decode()

    def __init__(self, device, unique_id, rank, world, limbs):
        self._h = ctypes.c_void_p()
        rc = N.lib().gkr_exchange_rccl_create(ctypes.c_int(device), ctypes.c_char_p(unique_id), ctypes.c_int(rank), ctypes.c_int(world),
                                              ctypes.c_size_t(int(limbs)), ctypes.byref(self._h))
        if rc:
            self._h = None
            raise GkrError(rc, RcclExchange._error())
        N.lib().gkr_exchange_rccl_dev.restype = ctypes.POINTER(N.ExchangeDev)
        N.lib().gkr_exchange_rccl_dev.argtypes = [ctypes.c_void_p]
        N.lib().gkr_exchange_rccl_calls.restype = ctypes.c_uint64
        N.lib().gkr_exchange_rccl_calls.argtypes = [ctypes.c_void_p]
        N.lib().gkr_exchange_rccl_destroy.restype = None
        N.lib().gkr_exchange_rccl_destroy.argtypes = [ctypes.c_void_p]
        self.struct = N.lib().gkr_exchange_rccl_dev(self._h).contents
        self.errors = []
        self.backend = "rccl (library-owned communicator)"
        self.world = world

    @property
    def calls(self):
        return int(N.lib().gkr_exchange_rccl_calls(self._h)) if self._h else 0

    @classmethod
    def from_torch_group(cls, device, limbs):
        """The id made on rank 0 of torch.distributed's default group and broadcast through it (torch only carries those
        128 bytes; every all-reduce of the proving path then goes through the library's own communicator)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.zeros(cls.ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            t = torch.tensor(list(cls.unique_id()), dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0)
        return cls(device, bytes(t.cpu().tolist()), rank, world, limbs)

    def close(self):
        if self._h:
            N.lib().gkr_exchange_rccl_destroy(self._h)
            self._h = None


class NoExchange:
    """gkr_exchange_dev for ONE rank (log2_shards = 0): a device buffer and a hook that has nothing to add."""

    def __init__(self, ctx: Context, limbs):
        self._ctx = ctx
        self._d = ctx.alloc(8 * int(limbs))
        self.errors = []
        self.calls = 0

        def fn(_user, count, stream):
            self.calls += 1
            return 0
        self._fn = N.ALLREDUCE_DEV_FN(fn)
        self.struct = N.ExchangeDev(self._fn, None, self._d.value, int(limbs))

    def close(self):
        if self._d is not None:
            self._ctx.free(self._d)
            self._d = None


# ----------------------------------------------------------------------------- gate-sharded layer sumcheck

def gate_range(k_i, rank, world):
    """Contiguous share of the layer's 2^k_i gates held by `rank` of `world` (any partition would do)."""
    r = shard_units(1 << k_i, rank, world)
    return r.start, len(r)


def widen(values_limbs):
    """(n, 4) uint64 canonical field elements -> (n, 8) int64 32-bit limbs (gkr_fr_widen)."""
    v = np.ascontiguousarray(values_limbs, dtype=np.uint64).reshape(-1, 4)
    out = np.empty((v.shape[0], 8), dtype=np.int64)
    rc = N.lib().gkr_fr_widen(_ptr(v), ctypes.c_size_t(v.shape[0]), _ptr(out))
    if rc:
        raise GkrError(rc, "gkr_fr_widen")
    return out


def narrow(limb_sums):
    """(n, 8) int64 limb sums (< 2^31 addends) -> (n, 4) uint64 canonical values mod r (gkr_fr_narrow)."""
    w = np.ascontiguousarray(limb_sums, dtype=np.int64).reshape(-1, 8)
    out = np.empty((w.shape[0], 4), dtype=np.uint64)
    rc = N.lib().gkr_fr_narrow(_ptr(w), ctypes.c_size_t(w.shape[0]), _ptr(out))
    if rc:
        raise GkrError(rc, "gkr_fr_narrow")
    return out


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


def make_allreduce_hook(sum_limbs):
    """The gkr_allreduce_fn the library calls back: `sum_limbs(int64 array (n, 8)) -> summed array` is the
    transport (torch.distributed SUM all-reduce, or an in-process sum for logical ranks).  Returns the ctypes
    function object (keep it alive for the duration of the call) and a list that collects exceptions."""
    errors = []

    def hook(_user, values, count):
        try:
            buf = np.ctypeslib.as_array(ctypes.cast(values, ctypes.POINTER(ctypes.c_uint64)), shape=(count, 4))
            buf[:] = narrow(sum_limbs(widen(buf)))
            return 0
        except Exception as e:   # never unwind through the C frame
            errors.append(e)
            return 1
    return ALLREDUCE_FN(hook), errors


def prove_sumcheck_opt_gate_sharded(ctx: Context, k_i, k_next, gate_first, gate_type, left, right, z, W, sum_limbs):
    """One rank's side of a layer sumcheck split by gates: this rank holds gates gate_first .. of the layer.
    Every rank returns the same (proof, r) as the unsharded prove_sumcheck_opt."""
    gt = np.ascontiguousarray(gate_type, dtype=np.uint8)
    l = np.ascontiguousarray(left, dtype=np.uint32)
    r = np.ascontiguousarray(right, dtype=np.uint32)
    if not (len(gt) == len(l) == len(r)):
        raise GkrError(N.GKR_ERR_INVALID, "gate arrays of different lengths")
    zl = as_limbs(z) if k_i else np.zeros((0, 4), dtype=np.uint64)
    wl = as_limbs(W)
    v = 2 * max(k_next, 0)
    C = np.zeros((max(v, 1), 3, 4), dtype=np.uint64)
    L = np.zeros(max(v, 1), dtype=np.uint32)
    R = np.zeros((max(v, 1), 4), dtype=np.uint64)
    hook, errors = make_allreduce_hook(sum_limbs)
    rc = N.lib().gkr_sumcheck_layer_sharded(ctx._h, ctypes.c_int(k_i), ctypes.c_int(k_next), ctypes.c_uint64(gate_first),
                                            ctypes.c_uint64(len(gt)), _ptr(gt), _ptr(l), _ptr(r), _ptr(zl), _ptr(wl), hook, None,
                                            _ptr(C), _ptr(L), _ptr(R))
    if errors:
        raise errors[0]
    ctx._check(rc)
    return [from_limbs(C[j])[3 - int(L[j]):] for j in range(v)], from_limbs(R[:v])


class ResidentGates:
    """A contiguous range of a layer's gates kept in device memory across sumchecks, with the gate lists sorted from
    them on first use (gkr_resident_layer_*): the whole layer, or one rank's share of it."""

    def __init__(self, ctx: Context, k_i, gate_first, gate_type, left, right):
        self._ctx, self.k_i, self.first = ctx, k_i, gate_first
        self._arrays = (np.ascontiguousarray(gate_type, dtype=np.uint8), np.ascontiguousarray(left, dtype=np.uint32),
                        np.ascontiguousarray(right, dtype=np.uint32))
        if not (len(self._arrays[0]) == len(self._arrays[1]) == len(self._arrays[2])):
            raise GkrError(N.GKR_ERR_INVALID, "gate arrays of different lengths")
        self.count = len(self._arrays[0])
        self._handles = {}      # k_next -> gkr_resident_layer (the lists depend on the next layer's width)

    def _layer(self, k_next):
        if k_next not in self._handles:
            h = ctypes.c_void_p()
            gt, l, r = self._arrays
            self._ctx._check(N.lib().gkr_resident_layer_create(self._ctx._h, ctypes.c_int(self.k_i), ctypes.c_int(k_next),
                                                                ctypes.c_uint64(self.first), ctypes.c_uint64(self.count), _ptr(gt), _ptr(l),
                                                                _ptr(r), ctypes.byref(h)))
            self._handles[k_next] = h
        return self._handles[k_next]

    def sumcheck_raw(self, k_next, z_limbs, w_limbs, sum_limbs=None):
        """-> (C, L, R) like Context.sumcheck_layer_raw; sum_limbs: the sum over ranks -- a DeviceExchange (RCCL, on the
        device), or the host transport `int64 limb array -> summed array` (None: the arrays hold the whole layer)."""
        v = 2 * k_next
        C = np.zeros((v, 3, 4), dtype=np.uint64)
        L = np.zeros(v, dtype=np.uint32)
        R = np.zeros((v, 4), dtype=np.uint64)
        zl = np.ascontiguousarray(z_limbs, dtype=np.uint64).reshape(-1, 4)
        wl = np.ascontiguousarray(w_limbs, dtype=np.uint64)
        if hasattr(sum_limbs, "struct"):   # a gkr_exchange_dev: DeviceExchange (torch), RcclExchange (the library's own), ...
            rc = N.lib().gkr_resident_layer_sumcheck_dev(self._ctx._h, self._layer(k_next), _ptr(zl), _ptr(wl), ctypes.byref(sum_limbs.struct),
                                                         _ptr(C), _ptr(L), _ptr(R))
            if sum_limbs.errors:   # (taken off the list: a later call with the same exchange object starts clean)
                errs = list(sum_limbs.errors)
                del sum_limbs.errors[:]
                raise errs[0]
            self._ctx._check(rc)
            return C, L, R
        hook, errors = make_allreduce_hook(sum_limbs) if sum_limbs is not None else (None, [])
        rc = N.lib().gkr_resident_layer_sumcheck(self._ctx._h, self._layer(k_next), _ptr(zl), _ptr(wl), hook, None, _ptr(C), _ptr(L), _ptr(R))
        if errors:
            raise errors[0]
        self._ctx._check(rc)
        return C, L, R

    def sumcheck_raw_device_w(self, k_next, z_limbs, d_w):
        """The same with W (2^k_next values) already in device memory (d_w: a Context.alloc pointer): nothing but z and the
        transcript crosses PCIe.  Whole layers only."""
        v = 2 * k_next
        C = np.zeros((v, 3, 4), dtype=np.uint64)
        L = np.zeros(v, dtype=np.uint32)
        R = np.zeros((v, 4), dtype=np.uint64)
        zl = np.ascontiguousarray(z_limbs, dtype=np.uint64).reshape(-1, 4)
        self._ctx._check(N.lib().gkr_resident_layer_sumcheck_wdev(self._ctx._h, self._layer(k_next), _ptr(zl), d_w, _ptr(C), _ptr(L), _ptr(R)))
        return C, L, R

    def close(self):
        for h in self._handles.values():
            N.lib().gkr_resident_layer_free(self._ctx._h, h)
        self._handles = {}


class ThreadedSum:
    """Sum-over-ranks for P logical ranks that are P threads of one process (one GPU): every rank deposits its
    limbs, the last one in adds them up, all leave with the total."""

    def __init__(self, world):
        import threading
        self._barrier = threading.Barrier(world)
        self._parts = [None] * world
        self._total = None

    def for_rank(self, rank):
        def sum_limbs(limbs):
            self._parts[rank] = limbs
            if self._barrier.wait() == 0:
                self._total = np.sum(np.stack(self._parts), axis=0)
            self._barrier.wait()
            return self._total
        return sum_limbs

    def abort(self):
        self._barrier.abort()


class ThreadedDeviceSum:
    """The device exchange (gkr_exchange_dev) for P logical ranks that are P threads of one process on one GPU: every
    rank's limb buffer is a CUDA tensor; the hook waits for its rank's stream, the last rank in adds the P buffers up on
    the device, every rank copies the total into its own buffer on its own stream.  Same widen / narrow kernels and the
    same flag handling as with RCCL; only the transport differs."""

    def __init__(self, world, device=0, k_max=13, limbs=0):
        import threading

        import torch
        self._torch = torch
        self._barrier = threading.Barrier(world)
        n = max(int(limbs), int(N.lib().gkr_exchange_limbs(ctypes.c_int(k_max))))
        self._bufs = [torch.zeros(n, dtype=torch.int64, device="cuda:%d" % device) for _ in range(world)]
        self._total = None
        self.errors = []

    def for_rank(self, rank):
        torch = self._torch
        buf = self._bufs[rank]
        holder = type("Exchange", (DeviceExchange,), {})
        ex = holder.__new__(holder)
        ex.errors = self.errors
        ex.calls = 0

        def fn(_user, count, stream):
            try:
                ext = torch.cuda.ExternalStream(int(stream or 0), device=buf.device)
                ext.synchronize()
                if self._barrier.wait() == 0:
                    self._total = torch.stack([b[:count] for b in self._bufs]).sum(dim=0)
                    torch.cuda.synchronize(buf.device)
                self._barrier.wait()
                with torch.cuda.stream(ext):
                    buf[:count].copy_(self._total)
                ext.synchronize()
                self._barrier.wait()    # nobody starts the next sum while a rank still reads this total
                return 0
            except Exception as e:
                self.errors.append(e)
                self._barrier.abort()
                return 1
        ex._fn = N.ALLREDUCE_DEV_FN(fn)
        ex._buf = buf
        ex.struct = N.ExchangeDev(ex._fn, None, buf.data_ptr(), buf.numel())
        return ex

    def abort(self):
        self._barrier.abort()


def prove_sumcheck_opt_logical_gates_dev(device, layer: Layer, k_next, z, W, nshards):
    """prove_sumcheck_opt_logical_gates with the DEVICE exchange: resident gate shards, gkr_resident_layer_sumcheck_dev.
    Returns every rank's (C, L, R)."""
    import threading
    gt, l, r = layer.arrays()
    coll = ThreadedDeviceSum(nshards, device, max(k_next, 1))
    out, errs = [None] * nshards, []
    zl = as_limbs(z) if layer.k else np.zeros((0, 4), dtype=np.uint64)
    wl = as_limbs(W)

    def run(rank):
        try:
            first, count = gate_range(layer.k, rank, nshards)
            with Context(device) as ctx:
                gates = ResidentGates(ctx, layer.k, first, gt[first:first + count], l[first:first + count], r[first:first + count])
                try:
                    out[rank] = gates.sumcheck_raw(k_next, zl, wl, coll.for_rank(rank))
                finally:
                    gates.close()
        except Exception as e:
            errs.append(e)
            coll.abort()
    threads = [threading.Thread(target=run, args=(p,)) for p in range(nshards)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        real = [e for e in errs if not isinstance(e, threading.BrokenBarrierError)]
        raise (real or errs)[0]
    return out


def prove_sumcheck_opt_logical_gates(device, layer: Layer, k_next, z, W, nshards):
    """prove_sumcheck_opt with the gates cut into `nshards` logical ranks on ONE GPU (one thread and one context
    per rank, as one process per GPU would have): the gate-sharded algorithm end to end, the collective being an
    in-process sum.  Returns the list of every rank's (proof, r) -- they must all be equal."""
    import threading
    gt, l, r = layer.arrays()
    coll = ThreadedSum(nshards)
    out, errs = [None] * nshards, []

    def run(rank):
        try:
            first, count = gate_range(layer.k, rank, nshards)
            with Context(device) as ctx:
                out[rank] = prove_sumcheck_opt_gate_sharded(ctx, layer.k, k_next, first, gt[first:first + count],
                                                            l[first:first + count], r[first:first + count], z, W,
                                                            coll.for_rank(rank))
        except Exception as e:
            errs.append(e)
            coll.abort()
    threads = [threading.Thread(target=run, args=(p,)) for p in range(nshards)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        real = [e for e in errs if not isinstance(e, threading.BrokenBarrierError)]
        raise (real or errs)[0]
    return out


# ----------------------------------------------------------------------------- round vectors

def layer_round_vector(tot, dep_flag):
    """[c2, c1, c0] from the summed (c0, g(1), c2), with the reference's length 2 + dep
    (get_univariate_coeff, rust/src/gkr/poly.rs:388-420); c1 = g(1) - c0 - c2."""
    c0, g1, c2 = (x % MODULUS for x in tot)
    lin = (g1 - c0 - c2) % MODULUS
    return [c2, lin, c0] if dep_flag else [lin, c0]


def mle_round_vector(tot, last_round, dep_last):
    """[c1, c0] from the summed (low, high) half sums with prove_sumcheck's length rule
    (rust/src/gkr/sumcheck.rs:158-214: add_poly drops a zero linear term except in the last round)."""
    lo, hi = (x % MODULUS for x in tot)
    c1 = (hi - lo) % MODULUS
    if last_round:
        return [c1, lo] if dep_last else [lo]
    return [c1, lo] if c1 else [lo]


# ----------------------------------------------------------------------------- GPU shards (library sessions)

def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class LayerSession:
    """One rank's shard of a GKR layer sumcheck (gkr_layer_session_* in include/gkr_amd.h)."""

    def __init__(self, ctx: Context, handle):
        self._ctx, self._h = ctx, handle

    @classmethod
    def open(cls, ctx: Context, layer: Layer, k_next, z, W, nshards=1, shard=0):
        gt, l, r = layer.arrays()
        zl = as_limbs(z) if layer.k else np.zeros((0, 4), dtype=np.uint64)
        wl = as_limbs(W)
        h = ctypes.c_void_p()
        ctx._check(N.lib().gkr_layer_session_open(ctx._h, ctypes.c_int(layer.k), ctypes.c_int(k_next), _ptr(gt), _ptr(l),
                                                  _ptr(r), _ptr(zl), _ptr(wl), ctypes.c_uint32(nshards),
                                                  ctypes.c_uint32(shard), ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def open_tables(cls, ctx: Context, kc, A, M, wb, Wc):
        h = ctypes.c_void_p()
        ctx._check(N.lib().gkr_layer_session_open_tables(ctx._h, ctypes.c_int(kc), _ptr(to_limbs(A)), _ptr(to_limbs(M)),
                                                         _ptr(to_limbs([wb])), _ptr(to_limbs(Wc)), ctypes.byref(h)))
        return cls(ctx, h)

    def dep(self, k):
        out = np.zeros(k, dtype=np.uint32)
        self._ctx._check(N.lib().gkr_layer_session_dep(self._ctx._h, self._h, _ptr(out), ctypes.c_uint32(k)))
        return [bool(x) for x in out]

    def sums(self):
        out = np.zeros((3, 4), dtype=np.uint64)
        self._ctx._check(N.lib().gkr_layer_session_sums(self._ctx._h, self._h, _ptr(out)))
        return from_limbs(out)

    def bind(self, r):
        self._ctx._check(N.lib().gkr_layer_session_bind(self._ctx._h, self._h, _ptr(to_limbs([r]))))

    def tail(self):
        out = np.zeros((4, 4), dtype=np.uint64)
        self._ctx._check(N.lib().gkr_layer_session_tail(self._ctx._h, self._h, _ptr(out)))
        return from_limbs(out)

    def close(self):
        if self._h:
            N.lib().gkr_layer_session_close(self._ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MleSession:
    """One rank's shard of a plain multilinear sumcheck; the table lives in device memory."""

    def __init__(self, ctx: Context, d_table, n):
        self._ctx = ctx
        self._h = ctypes.c_void_p()
        ctx._check(N.lib().gkr_mle_session_open(ctx._h, d_table, ctypes.c_int(n), ctypes.byref(self._h)))

    def sums(self):
        out = np.zeros((2, 4), dtype=np.uint64)
        dep = ctypes.c_uint32(0)
        self._ctx._check(N.lib().gkr_mle_session_sums(self._ctx._h, self._h, _ptr(out), ctypes.byref(dep)))
        return from_limbs(out), bool(dep.value)

    def bind(self, r):
        self._ctx._check(N.lib().gkr_mle_session_bind(self._ctx._h, self._h, _ptr(to_limbs([r]))))

    def value(self):
        out = np.zeros((1, 4), dtype=np.uint64)
        self._ctx._check(N.lib().gkr_mle_session_value(self._ctx._h, self._h, _ptr(out)))
        return from_limbs(out)[0]

    def close(self):
        if self._h:
            N.lib().gkr_mle_session_close(self._ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------- the sharded algorithms

def _log2(p):
    lp = p.bit_length() - 1
    if p < 1 or (1 << lp) != p:
        raise GkrError(N.GKR_ERR_INVALID, "the number of shards must be a power of two")
    return lp


def layer_sumcheck_rounds(shards, reduce_fr, k, log_p, dep, hasher=multi_hash):
    """Local rounds of the layer sumcheck over `shards` (one per logical rank held by THIS process).

    reduce_fr(list_of_partial_vectors) -> summed vector over ALL ranks of the world.
    Returns (proof, r) for the 2k - log_p rounds that are shard-local.
    """
    proof, rs = [], []
    for j in range(2 * k - log_p):
        tot = reduce_fr([s.sums() for s in shards])
        g = layer_round_vector(tot, dep[j % k])
        r = hasher(g, 0)
        proof.append(g)
        rs.append(r)
        for s in shards:
            s.bind(r)
    return proof, rs


def _finish_layer(open_tail, tails, k, log_p, dep, proof, rs, hasher):
    """Last log_p rounds on the gathered one-entry-per-rank tables (identical on every rank)."""
    if log_p == 0:
        return proof, rs
    A = [t[0] for t in tails]
    M = [t[1] for t in tails]
    Wc = [t[2] for t in tails]
    wb = tails[0][3]
    tail = open_tail(log_p, A, M, wb, Wc)
    try:
        for j in range(2 * k - log_p, 2 * k):
            g = layer_round_vector(tail.sums(), dep[j % k])
            r = hasher(g, 0)
            proof.append(g)
            rs.append(r)
            tail.bind(r)
    finally:
        if hasattr(tail, "close"):
            tail.close()
    return proof, rs


def prove_sumcheck_opt_logical(ctx: Context, layer: Layer, k_next, z, W, nshards):
    """prove_sumcheck_opt with the hypercube cut into `nshards` logical ranks on ONE GPU: the
    partitioned algorithm end to end (shard build, per-round reduce, gather, redundant tail), the
    collective degenerating to an in-process sum.  Used to prove bit-exactness of the sharded
    path on a single device."""
    log_p = _log2(nshards)
    shards = [LayerSession.open(ctx, layer, k_next, z, W, nshards, p) for p in range(nshards)]
    try:
        dep = shards[0].dep(k_next)

        def reduce_fr(parts):
            return [sum(col) % MODULUS for col in zip(*parts)]
        proof, rs = layer_sumcheck_rounds(shards, reduce_fr, k_next, log_p, dep)
        tails = [s.tail() for s in shards]
    finally:
        for s in shards:
            s.close()
    return _finish_layer(lambda kc, A, M, wb, Wc: LayerSession.open_tables(ctx, kc, A, M, wb, Wc), tails, k_next, log_p,
                         dep, proof, rs, multi_hash)


def prove_sumcheck_opt_distributed(shard, coll, k_next, dep, open_tail, hasher=multi_hash):
    """One rank's side of the distributed layer sumcheck.  `shard` is this rank's shard object
    (LayerSession on a GPU), `coll` a collective (TorchCollective), `open_tail(kc, A, M, wb, Wc)`
    builds the tail object.  Every rank returns the same (proof, r)."""
    log_p = _log2(coll.world)
    proof, rs = layer_sumcheck_rounds([shard], lambda parts: coll.all_reduce_fr(parts[0]), k_next, log_p, dep, hasher)
    tails = coll.all_gather_fr(shard.tail()) if log_p else []
    return _finish_layer(open_tail, tails, k_next, log_p, dep, proof, rs, hasher)


def mle_sumcheck_rounds(shards, reduce_fr, reduce_or, n, log_p, dep_rank_bit, hasher=multi_hash):
    """prove_sumcheck on a table of 2^n entries cut into 2^log_p trailing-variable shards held by
    `shards` (those of this process).  dep_rank_bit: for log_p >= 1, whether the full table depends
    on its LAST variable (a rank bit: shards p and p ^ 1 differ) -- the last round's length rule."""
    proof, rs = [], []
    local = n - log_p
    dep_local = False
    for j in range(local):
        parts = [s.sums() for s in shards]
        if j == 0:
            dep_local = reduce_or(any(d for _, d in parts))
        tot = reduce_fr([p for p, _ in parts])
        last = (j == n - 1)
        g = mle_round_vector(tot, last, dep_local if log_p == 0 else dep_rank_bit)
        r = hasher(g, 0)
        proof.append(g)
        rs.append(r)
        for s in shards:
            s.bind(r)
    return proof, rs


class _DeviceTail:
    """The gathered per-rank values as a small device table (index = rank = the trailing bits)."""

    def __init__(self, ctx: Context, values):
        self._ctx = ctx
        self._d = ctx.alloc(32 * len(values))
        ctx.upload(self._d, to_limbs(values))
        self._s = MleSession(ctx, self._d, _log2(len(values)))

    def sums(self):
        return self._s.sums()

    def bind(self, r):
        self._s.bind(r)

    def close(self):
        self._s.close()
        if self._d is not None:
            self._ctx.free(self._d)
            self._d = None


def _finish_mle(open_tail, values, n, log_p, dep_rank_bit, proof, rs, hasher):
    """Last log_p rounds on the gathered one-entry-per-rank table (identical on every rank)."""
    if log_p == 0:
        return proof, rs
    tail = open_tail(values)
    try:
        for j in range(n - log_p, n):
            tot, _ = tail.sums()
            g = mle_round_vector(tot, j == n - 1, dep_rank_bit)
            r = hasher(g, 0)
            proof.append(g)
            rs.append(r)
            tail.bind(r)
    finally:
        if hasattr(tail, "close"):
            tail.close()
    return proof, rs


def prove_sumcheck_logical(ctx: Context, d_shards, n, dep_rank_bit=None):
    """prove_sumcheck on 2^n entries held as len(d_shards) device-resident trailing-variable shards
    (shard p = entries p, p + P, p + 2P, ...), all on one GPU."""
    nshards = len(d_shards)
    log_p = _log2(nshards)
    if dep_rank_bit is None and log_p:
        dep_rank_bit = any(tables_differ(ctx, d_shards[p], d_shards[p + 1], 1 << (n - log_p)) for p in range(0, nshards, 2))
    sess = [MleSession(ctx, d, n - log_p) for d in d_shards]
    try:
        proof, rs = mle_sumcheck_rounds(sess, lambda parts: [sum(c) % MODULUS for c in zip(*parts)], bool, n, log_p,
                                        dep_rank_bit)
        values = [s.value() for s in sess] if log_p else []
    finally:
        for s in sess:
            s.close()
    return _finish_mle(lambda v: _DeviceTail(ctx, v), values, n, log_p, dep_rank_bit, proof, rs, multi_hash)


def prove_sumcheck_distributed(shard, coll, n, dep_rank_bit, open_tail, hasher=multi_hash):
    """One rank's side of the distributed plain sumcheck (shard: MleSession-like; open_tail(values)
    builds the object that runs the last log2(P) rounds on the gathered values)."""
    log_p = _log2(coll.world)
    proof, rs = mle_sumcheck_rounds([shard], lambda parts: coll.all_reduce_fr(parts[0]), coll.all_reduce_or, n, log_p,
                                    dep_rank_bit, hasher)
    values = [v[0] for v in coll.all_gather_fr([shard.value()])] if log_p else []
    return _finish_mle(open_tail, values, n, log_p, dep_rank_bit, proof, rs, hasher)


# ----------------------------------------------------------------------------- one plain sumcheck split over ranks, multi-round passes

def mle_shard(table_limbs, n, log_p, p):
    """Shard p of 2^log_p of a table of 2^n entries as gkr_sumcheck_mle_sharded_dev wants it: index bits log_p .. 1 of an
    entry are its rank, the last variable stays inside the shard -- T_p[h * 2 + x_n] = T[h * 2P + 2p + x_n]."""
    t = np.ascontiguousarray(table_limbs, dtype=np.uint64).reshape(1 << (n - log_p - 1), 1 << log_p, 2, 4)
    return np.ascontiguousarray(t[:, p]).reshape(-1, 4)


def exchange_limbs_mle(n, log_p, batch=1):
    N.lib().gkr_exchange_limbs_mle.restype = ctypes.c_size_t
    return int(N.lib().gkr_exchange_limbs_mle(ctypes.c_int(n), ctypes.c_int(log_p), ctypes.c_int(batch)))


def sumcheck_mle_sharded_raw(ctx: Context, d_shards, n, log_p, shard, exchange, batch=1):
    """gkr_sumcheck_mle_sharded_dev: this rank's `batch` shards (device memory, 2^(n - log_p) entries each) of `batch`
    tables of 2^n entries -> the whole transcript (C (batch, n, 2, 4), L (batch, n), R (batch, n, 4)) and the number of
    exchanges.  exchange: a DeviceExchange / StagedDeviceExchange / NoExchange / ThreadedDeviceSum.for_rank()."""
    C = np.zeros((batch, n, 2, 4), dtype=np.uint64)
    L = np.zeros((batch, n), dtype=np.uint32)
    R = np.zeros((batch, n, 4), dtype=np.uint64)
    nx = ctypes.c_uint32(0)
    rc = N.lib().gkr_sumcheck_mle_sharded_dev(ctx._h, d_shards, ctypes.c_int(n), ctypes.c_int(log_p), ctypes.c_int(shard),
                                              ctypes.c_int(batch), ctypes.byref(exchange.struct), _ptr(C), _ptr(L), _ptr(R), ctypes.byref(nx))
    if exchange.errors:
        errs = list(exchange.errors)
        del exchange.errors[:]
        raise errs[0]
    ctx._check(rc)
    return C, L, R, int(nx.value)


def prove_sumcheck_logical_dev(device, tables_limbs, n, nshards):
    """`batch` tables (batch, 2^n, 4) split over `nshards` logical ranks on ONE GPU -- a thread and a context per rank, as
    one process per GPU would have, the all-reduce an in-process device sum -- through gkr_sumcheck_mle_sharded_dev.
    Returns every rank's (C, L, R, exchanges)."""
    import threading
    log_p = _log2(nshards)
    tables = np.ascontiguousarray(tables_limbs, dtype=np.uint64).reshape(-1, 1 << n, 4)
    batch = tables.shape[0]
    limbs = exchange_limbs_mle(n, log_p, batch)
    coll = ThreadedDeviceSum(nshards, device, 1, limbs)
    out, errs = [None] * nshards, []

    def run(rank):
        try:
            mine = np.stack([mle_shard(tables[b], n, log_p, rank) for b in range(batch)]) if log_p else tables
            with Context(device) as ctx:
                d = ctx.alloc(mine.nbytes)
                try:
                    ctx.upload(d, mine)
                    out[rank] = sumcheck_mle_sharded_raw(ctx, d, n, log_p, rank, coll.for_rank(rank), batch)
                finally:
                    ctx.free(d)
        except Exception as e:
            errs.append(e)
            coll.abort()
    threads = [threading.Thread(target=run, args=(p,)) for p in range(nshards)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        real = [e for e in errs if not isinstance(e, threading.BrokenBarrierError)]
        raise (real or errs)[0]
    return out


def prove_sumcheck_split_model(shard_values, n, log_p, rank, coll, hasher=multi_hash):
    """The split of gkr_sumcheck_mle_sharded_dev restated on Python integers over a collective (CPU: gloo in the tests): the
    partition, the linearity of the sums, "depends on x_n" as the OR of the shards' neighbour compares, the gather order of
    the tail.  One all-reduce per round here (the library sends the sums of up to five rounds at once -- the same
    transcript).  shard_values: this rank's 2^(n - log_p) integers, mle_shard layout.  -> (proof, r), the same on every rank."""
    P_ = 1 << log_p
    t = [v % MODULUS for v in shard_values]
    dep_last = coll.all_reduce_or(any(t[2 * h] != t[2 * h + 1] for h in range(len(t) // 2)))
    proof, rs = [], []
    keep = min(n - log_p, 6)
    for j in range(n - log_p - keep):
        h = len(t) // 2
        c0, s1 = coll.all_reduce_fr([sum(t[:h]) % MODULUS, sum(t[h:]) % MODULUS])
        c1 = (s1 - c0) % MODULUS
        g = [c1, c0] if c1 else [c0]
        r = hasher(g, 0)
        proof.append(g)
        rs.append(r)
        t = [(t[i] + r * (t[i + h] - t[i])) % MODULUS for i in range(h)]
    # gather: local entry (h, x) is tail entry h * 2P + 2 rank + x
    tail = [0] * (len(t) * P_)
    for i, v in enumerate(t):
        tail[(i >> 1) * 2 * P_ + 2 * rank + (i & 1)] = v
    tail = coll.all_reduce_fr(tail)
    done = len(proof)
    for j in range(done, n):
        h = len(tail) // 2
        c0, c1 = sum(tail[:h]) % MODULUS, (sum(tail[h:]) - sum(tail[:h])) % MODULUS
        g = ([c1, c0] if dep_last else [c0]) if j == n - 1 else ([c1, c0] if c1 else [c0])
        r = hasher(g, 0)
        proof.append(g)
        rs.append(r)
        tail = [(tail[i] + r * (tail[i + h] - tail[i])) % MODULUS for i in range(h)]
    return proof, rs


def tables_differ(ctx: Context, d_a, d_b, count):
    out = ctypes.c_uint32(0)
    ctx._check(N.lib().gkr_device_tables_differ(ctx._h, d_a, d_b, ctypes.c_size_t(count), ctypes.byref(out)))
    return bool(out.value)


def shard_units(total, rank, world):
    """Contiguous split of `total` independent units (proofs, sumchecks) over `world` ranks: the
    no-collective sharding bench.py and proof batches use."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))
