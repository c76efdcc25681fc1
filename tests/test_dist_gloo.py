"""The N>1 path on CPU: two processes, gloo backend (no GPU): flat gradient bucket all-reduce,
ray sharding, replica-state broadcast, max/sum-over-ranks timing helpers used by bench.py."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cnc_amd import dist as cd
    from cnc_amd.nerfacc import OccGridEstimator
    r, lr, w = cd.init("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                      # identical replicas
    table = torch.nn.Parameter(torch.zeros(1000, 8))
    mlp = torch.nn.Linear(16, 4)
    bucket = cd.GradBucket([table] + list(mlp.parameters()))
    bucket.bind()
    # each rank's "rays" produce a different gradient
    lo, hi = cd.shard_range(1001, rank, world)
    x = torch.arange(lo, hi, dtype=torch.float32)
    loss = (table[: hi - lo, 0] * x).sum() + mlp(torch.ones(1, 16) * (rank + 1)).sum()
    loss.backward()
    assert table.grad.data_ptr() == bucket.views[0].data_ptr()      # accumulated in the bucket
    bucket.allreduce(average=True)
    est = OccGridEstimator([-1.0] * 3 + [1.0] * 3, resolution=8)
    if rank == 0:
        est.occs.uniform_(0, 1)
        est.binaries = est.occs.view(est.binaries.shape) > 0.5
    cd.broadcast_module_buffers(est, ["occs", "binaries"])
    q.put((rank, (lo, hi), table.grad[:, 0].clone(), mlp.weight.grad.clone(), est.binaries.sum().item(),
           est.occs.sum().item(), cd.max_over_ranks(float(rank + 1), "cpu"), cd.sum_over_ranks(float(hi - lo), "cpu")))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_bucket_allreduce_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, g0, w0, b0, o0, m0, n0), (r1, s1, g1, w1, b1, o1, m1, n1) = res
    assert s0 == (0, 501) and s1 == (501, 1001)                 # contiguous, exhaustive, balanced
    assert torch.equal(g0, g1) and torch.equal(w0, w1)          # replicas hold the same reduced grads
    expect = torch.zeros(1000)
    expect[:501] += torch.arange(0, 501, dtype=torch.float32)
    expect[:500] += torch.arange(501, 1001, dtype=torch.float32)
    assert torch.allclose(g0, expect / 2)                       # mean over ranks
    assert torch.allclose(w0, torch.full((4, 16), 1.5))         # (1 + 2) / 2
    assert b0 == b1 and o0 == o1 and b0 > 0                     # occupancy replicated from rank 0
    assert m0 == m1 == 2.0 and n0 == n1 == 1001.0


def test_shard_range_properties():
    from cnc_amd.dist import shard_range
    for n in (0, 1, 7, 640000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_dist_helpers_refuse_an_uninitialised_world():
    code = ("import os,sys;sys.path.insert(0,%r);os.environ.update(WORLD_SIZE='2',RANK='0',LOCAL_RANK='0');"
            "import torch;from cnc_amd import dist as d;p=torch.nn.Parameter(torch.zeros(4));b=d.GradBucket([p]);"
            "b.allreduce()") % ROOT
    r = __import__('subprocess').run([sys.executable, "-c", code], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')})
    assert r.returncode != 0 and "not initialised" in r.stderr


def _worker_tail_resync(rank, world, port, q):
    """What the data-parallel Trainer adds in round 4, at `world` ranks: the sample count riding in the gradient
    bucket's tail slot, several scalars in one `sum_over_ranks`, and the checksum-triggered re-alignment."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cnc_amd import dist as cd
    cd.init("gloo")
    torch.manual_seed(0)                      # identical replicas
    table = torch.nn.Parameter(torch.randn(257, 8))
    mlp = torch.nn.Linear(16, 4)
    head = torch.nn.Linear(4, 4)
    params = [table] + list(mlp.parameters()) + list(head.parameters())
    bucket = cd.GradBucket(params, tail=1)
    assert bucket.flat.numel() == bucket.numel + 1 and bucket.grads.numel() == bucket.numel
    assert bucket.tail.data_ptr() == bucket.flat[bucket.numel:].data_ptr()
    bucket.zero()
    bucket.bind(force=True)
    (table[rank] * float(rank + 1)).sum().backward()
    bucket.tail.fill_(float(1000 + rank))                 # this rank's sample count
    bucket.allreduce(average=False)
    count = float(bucket.tail[0])
    bucket.grads.div_(world)
    rows = table.grad[:world, 0].clone()
    two = cd.sum_over_ranks([float(rank), 1.0], "cpu")
    # replicas identical: nothing to re-align
    first = cd.resync_parameters(params)
    # an ulp of drift in ONE tensor on ONE rank (what a float atomic in another order does): exactly that tensor
    # is broadcast from rank 0, on every rank
    if rank == world - 1:
        with torch.no_grad():
            v = mlp.weight.view(-1).view(torch.int32)
            v[3] += 1
    second = cd.resync_parameters(params)
    third = cd.resync_parameters(params)
    # +1 ulp and -1 ulp in two entries cancel in a plain bit sum; the second checksum (sum of squares) sees them
    if rank == 1:
        with torch.no_grad():
            v = table.view(-1).view(torch.int32)
            v[5] += 1
            v[9] -= 1
    fourth = cd.resync_parameters(params)
    sums = [float(p.detach().double().sum()) for p in params]
    q.put((rank, count, rows, two, first, second, third, fourth, sums))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bucket_tail_and_checksum_resync(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tail_resync, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mlp_w_bytes, table_bytes = 4 * 16 * 4, 257 * 8 * 4
    for rank, count, rows, two, first, second, third, fourth, sums in res:
        assert count == sum(1000 + r for r in range(world))               # the tail slot came back summed
        assert torch.allclose(rows, torch.arange(1, world + 1, dtype=torch.float32) / world)   # ... the gradients averaged
        assert two == [sum(range(world)), float(world)]
        assert first == (0, 0) and third == (0, 0)
        assert second == (1, mlp_w_bytes) and fourth == (1, table_bytes)
        assert sums == res[0][8]                                          # replicas identical again


def _forced_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      CNC_DIST_FORCE="1")
    from cnc_amd import dist as cd
    assert cd.init("gloo") == (0, 0, 1) and torch.distributed.is_initialized() and cd._active()
    p = [torch.nn.Parameter(torch.arange(5000, dtype=torch.float32)), torch.nn.Parameter(torch.ones(3, 3))]
    b = cd.GradBucket(p, tail=1)
    b.bind()
    p[0].grad.add_(2.0)
    b.tail.fill_(9.0)
    b.allreduce(average=True)
    ok = float(b.tail[0]) == 9.0 and float(p[0].grad[0]) == 2.0        # the tail is never averaged
    w = b.allreduce(average=False, async_op=True)
    w.wait()
    old = cd._CHECKSUM_CHUNK
    one = cd._checksums(p)
    cd._CHECKSUM_CHUNK = 1000                                          # several chunks == one
    many = cd._checksums(p)
    cd._CHECKSUM_CHUNK = old
    q.put((ok, cd.resync_parameters(p), bool(torch.equal(one, many)), cd.sum_over_ranks([1.0, 2.0], "cpu")))
    torch.distributed.destroy_process_group()


def test_forced_one_rank_group_runs_every_collective():
    """CNC_DIST_FORCE=1: a one-rank process group counts as data-parallel (the hook tests/test_gpu_rccl.py uses to put
    the N > 1 path on RCCL with one GPU)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    ok, resync, same, sums = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert ok and resync == (0, 0) and same and sums == [1.0, 2.0]
