"""Ray-sharded data parallelism: one process per GPU, RCCL over xGMI (torch.distributed 'nccl').

The reference is single-GPU; this is the part of the north-star that has no
reference code.  Rays are independent units, so ranks only ever exchange the gradient of the shared
tables / MLPs: ONE flat bucket all-reduced once per step (161 MB fp32 for the reference composition
at F=8) — sized for xGMI's point-to-point links (a single large ring transfer, not per-tensor calls).
Replica state that must stay identical (occupancy grid, context-window RNG) is either seeded
identically or broadcast from rank 0.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def forced() -> bool:
    """CNC_DIST_FORCE=1 (test hook): treat a ONE-rank world as data-parallel — the process group is created, every
    collective of the N > 1 path is issued (on RCCL they run through the same communicator code as at N = 8), the
    Trainer builds its buckets.  This is how the one GPU of a test box exercises the `nccl` backend."""
    return os.environ.get("CNC_DIST_FORCE") == "1"


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from the torchrun environment (no-op for world 1 unless `forced()`)."""
    rank, local_rank, world = env_world()
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("CNC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_device_index())
        if backend == "nccl":
            dist.init_process_group(backend, device_id=torch.device("cuda", local_device_index()))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def local_device_index() -> int:
    """cuda index of this rank: LOCAL_RANK, or 0 for every rank under CNC_DIST_ONE_DEVICE=1 (test hook:
    lets the N > 1 control flow run on a single-GPU box, with gloo as the backend)."""
    if os.environ.get("CNC_DIST_ONE_DEVICE") == "1":
        return 0
    return env_world()[1]


def _active() -> bool:
    """True when there is a process group with more than one rank (or any group under `forced()`).  WORLD_SIZE > 1 in
    the environment WITHOUT a process group is an error, not a silent single-rank run: replicas would drift apart."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size() > 1 or forced()
    if env_world()[2] > 1:
        raise RuntimeError("WORLD_SIZE > 1 but torch.distributed is not initialised: call cnc_amd.dist.init() first")
    return False


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) share of n units; the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class GradBucket:
    """All gradients of a parameter list as views into ONE contiguous fp32 buffer, so a step's
    exchange is a single all-reduce.  `bind()` points each param.grad at its slice (autograd then
    accumulates in place), `allreduce()` sums across ranks and optionally averages."""

    def __init__(self, params: Iterable[torch.nn.Parameter], tail: int = 0):
        """`tail`: extra fp32 slots behind the gradients that ride in the same all-reduce (`self.tail`): per-step
        scalars every rank needs the sum of — the Trainer's sample count — without a collective of their own."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel + tail, dtype=dt, device=dev)
        self.grads = self.flat[:self.numel]
        self.tail = self.flat[self.numel:]
        self.views = []
        o = 0
        for p in self.params:
            assert p.device == dev and p.dtype == dt
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def bind(self, force: bool = False) -> None:
        """Point every param.grad at its slice.  A gradient that autograd put elsewhere is copied in first,
        unless `force` (the caller is switching buckets and the old gradient must not leak into this one)."""
        for p, v in zip(self.params, self.views):
            if not force and p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v

    def zero(self) -> None:
        self.flat.zero_()

    def allreduce(self, average: bool = True, async_op: bool = False):
        if not _active():
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if average and not async_op:
            self.grads.div_(dist.get_world_size())       # the tail carries sums (sample counts): never averaged
        return work

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()


def broadcast_module_buffers(module: torch.nn.Module, names: Iterable[str], src: int = 0) -> None:
    """Keep replica state (e.g. OccGridEstimator.occs / .binaries) identical to rank `src`."""
    if not _active():
        return
    for n in names:
        t = getattr(module, n)
        if t.dtype == torch.bool:
            u = t.to(torch.uint8)
            dist.broadcast(u, src)
            setattr(module, n, u.to(torch.bool))
        else:
            dist.broadcast(t, src)


def broadcast_parameters(params: Iterable[torch.nn.Parameter], src: int = 0) -> None:
    """Overwrite every rank's parameters with rank `src`'s (in place: optimiser state and views stay valid)."""
    if not _active():
        return
    with torch.no_grad():
        for p in params:
            dist.broadcast(p.data, src)
            p.add_(0)            # in-place no-op that bumps p._version: caches keyed on it (sign plane) refresh


_CHECKSUM_CHUNK = 1 << 22


def _checksums(params) -> torch.Tensor:
    """Two wrapping int64 checksums of every tensor's BITS (sum and sum of squares of its int32 view): [P, 2]."""
    out = []
    for p in params:
        if p.element_size() != 4:
            raise TypeError(f"resync checksums are defined on 4-byte parameters, got {p.dtype}")
        v = p.detach().reshape(-1).view(torch.int32)
        s1 = s2 = None
        for lo in range(0, max(v.numel(), 1), _CHECKSUM_CHUNK):      # transient memory: 8 bytes per chunk element
            c = v[lo:lo + _CHECKSUM_CHUNK]
            a = torch.sum(c, dtype=torch.int64)
            c64 = c.to(torch.int64)
            b = torch.sum(c64 * c64)                                   # wraps mod 2^64 like the one-shot form did
            s1, s2 = (a, b) if s1 is None else (s1 + a, s2 + b)
        out.append(torch.stack([s1, s2]))
    return torch.stack(out)


def resync_parameters(params, src: int = 0):
    """Re-align replicas only where they differ: bit checksums of every parameter tensor are compared across the ranks
    in ONE small collective (max of [c, -c]: max and min at once); the tensors whose checksums disagree on any rank
    are overwritten with rank `src`'s.  Returns (tensors broadcast, bytes broadcast).  The decision is taken from the
    all-reduced vector, so every rank broadcasts the same set."""
    params = list(params)
    if not _active() or not params:
        return 0, 0
    with torch.no_grad():
        c = _checksums(params)
        both = torch.cat([c, -c], dim=1)
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        differ = (both[:, :2] != -both[:, 2:]).any(dim=1).tolist()
        n = nbytes = 0
        for p, d in zip(params, differ):
            if d:
                dist.broadcast(p.data, src)
                p.add_(0)        # bumps p._version: caches keyed on it (sign plane) refresh
                n += 1
                nbytes += p.numel() * p.element_size()
    return n, nbytes


def max_over_ranks(x: float, device) -> float:
    if not _active():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, device):
    """Sum of a float (or of every entry of a list of floats: ONE collective) over the ranks, as Python float(s)."""
    many = isinstance(x, (list, tuple))
    if not _active():
        return list(x) if many else x
    t = torch.tensor(list(x) if many else [x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist() if many else float(t.item())
