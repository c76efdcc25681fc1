"""RCCL itself, on the one GPU of the test box.

Every other multi-rank test swaps gloo in (several ranks on one device cannot form an RCCL communicator).  Here a ONE-rank
process group is created on `backend="nccl"` (= RCCL on ROCm) with `CNC_DIST_FORCE=1` (cnc_amd.dist.forced), so every
collective of the N > 1 path is issued through the communicator: the flat bucket's asynchronous all-reduce and its stream
ordering against kernels of the compute stream, the int64 MAX all-reduce of the replica checksums, the bool -> uint8
broadcast of the occupancy grid, float64 SUM / MAX reductions, bench.py's frame exchange, a data-parallel Trainer step, and
`destroy_process_group`.  (no reference code — the reference is single-GPU.)"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_mapped():
    with open("/proc/self/maps") as fh:
        return any("librccl" in l for l in fh)


@pytest.mark.timeout(600)
def test_collectives_of_the_dp_path_on_rccl(cuda, monkeypatch):
    """In THIS process (so that librccl.so is among the libraries the test run maps)."""
    import torch.distributed as td
    from cnc_amd import dist as cdist
    assert not td.is_initialized()
    for k, v in dict(CNC_DIST_FORCE="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                     MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0").items():
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("CNC_DIST_BACKEND", raising=False)
    monkeypatch.delenv("CNC_DIST_ONE_DEVICE", raising=False)
    dev = torch.device("cuda", 0)
    try:
        assert cdist.init() == (0, 0, 1)
        assert td.is_initialized() and td.get_backend() == "nccl" and cdist._active()
        # ---- the gradient bucket: async all-reduce next to compute, then the join ----
        params = [torch.nn.Parameter(torch.zeros(1 << 22, device=dev)), torch.nn.Parameter(torch.zeros(37, 5, device=dev))]
        bucket = cdist.GradBucket(params, tail=1)
        bucket.bind()
        g = torch.Generator(device=dev).manual_seed(3)
        a = torch.randn(1 << 22, device=dev, generator=g)
        side = torch.cuda.Stream(device=dev)
        for rep in range(3):
            bucket.zero()
            # producer kernels on the compute stream; the collective is enqueued right behind them
            params[0].grad.add_(a * (rep + 1))
            params[1].grad.add_(float(rep))
            bucket.tail.fill_(1234.0 + rep)
            work = bucket.allreduce(average=False, async_op=True)
            with torch.cuda.stream(side):                 # independent work runs next to the collective
                busy = (a * a).sum()
            work.wait()                                   # the compute stream now waits for the communicator's stream
            total = params[0].grad.double().sum() + params[1].grad.double().sum()
            want = (a.double() * (rep + 1)).sum() + rep * 37 * 5
            torch.cuda.current_stream(dev).wait_stream(side)
            assert abs(float(total - want)) <= 1e-6 * abs(float(want)) + 1e-3, rep
            assert float(bucket.tail[0]) == 1234.0 + rep          # a one-rank sum leaves the tail as it was
            assert torch.isfinite(busy)
        # blocking form with the mean: the tail (a sum of counts) is NOT averaged
        bucket.tail.fill_(77.0)
        bucket.allreduce(average=True, async_op=False)
        assert float(bucket.tail[0]) == 77.0
        # ---- replica checksums: one int64 MAX all-reduce; nothing differs in a one-rank world ----
        with torch.no_grad():
            params[0].copy_(a)
        assert cdist.resync_parameters(params) == (0, 0)
        c = cdist._checksums(params)
        v = params[0].detach().view(torch.int32).to(torch.int64)
        assert int(c[0, 0]) == int(v.sum()) and int(c[0, 1]) == int((v * v).sum())      # chunked == one-shot
        with pytest.raises(TypeError):
            cdist._checksums([torch.zeros(4, dtype=torch.float16, device=dev)])
        # ---- parameter and buffer broadcasts (bool travels as uint8) ----
        cdist.broadcast_parameters(params)
        assert torch.equal(params[0].detach(), a)

        class Holder(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.register_buffer("binaries", (torch.arange(4096, device=dev) % 3 == 0).view(1, 16, 16, 16))
                self.register_buffer("occs", torch.linspace(0, 1, 4096, device=dev))
        h = Holder()
        keep = h.binaries.clone()
        cdist.broadcast_module_buffers(h, ["occs", "binaries"])
        assert h.binaries.dtype == torch.bool and torch.equal(h.binaries, keep)
        # ---- float64 reductions of the results line ----
        assert cdist.max_over_ranks(2.5, dev) == 2.5
        assert cdist.sum_over_ranks([1.0, 2.0, 4.0], dev) == [1.0, 2.0, 4.0]
        td.barrier()
        assert _rccl_mapped(), "the nccl backend ran without librccl mapped?"
    finally:
        if td.is_initialized():
            td.destroy_process_group()
    assert not td.is_initialized()


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CNC_DIST_BACKEND",
                        "CNC_DIST_ONE_DEVICE", "CNC_BENCH_BACKEND", "CNC_BENCH_ONE_DEVICE")}
    env.update(CNC_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), **extra)
    return env


@pytest.mark.timeout(1200)
def test_bench_frame_exchange_on_rccl(cuda):
    """bench.py's own exchange path — the table gradient's async all-reduce issued at the end of a frame, joined after
    the next frame's march, the mean, the SUM / MAX reductions of the line — through RCCL."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-train-step", "--no-field"], cwd=ROOT, capture_output=True, text=True,
                       timeout=1100, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["ranks"][0]["backend"] == "nccl" and out["n_gpus"] == 1
    ar = [k for k in out["kernels"] if k.startswith("allreduce(grad_table)")]
    assert len(ar) == 1 and out["kernels"][ar[0]]["launches"] == 2          # one exchange per timed frame
    assert out["value"] > 1e8


_TRAINER = r"""
import os, sys, json, torch
sys.path.insert(0, {root!r})
from cnc_amd.trainer import TrainConfig, Trainer
cfg = TrainConfig(lmbda=2e-3, Pg_level=5, Pg_level_2D=3, log2_hashmap_size=12, log2_hashmap_size_2D=9,
                  sample_num=3000, max_context_layer_num=3, n_features=2, n_neurons=32,
                  resolutions_list=(10, 14, 18, 26, 34), resolutions_list_2D=(18, 34, 66),
                  skip_levels_3D=(0, 1, 2), skip_levels_2D=(0,), max_steps=20, init_batch_size=512,
                  target_sample_batch_size=1 << 14, grid_resolution=16, render_step_size=2e-2,
                  milestones=(100, 130), warmup_iters=20, test_views=2, image_size=48, out_dir={out!r},
                  step_update=4)
tr = Trainer(cfg, device="cuda")
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and tr.dp and tr.bucket is not None
stats = [tr.train_step(s) for s in range(12)]
mapped = any("librccl" in l for l in open("/proc/self/maps"))
print("RESULT " + json.dumps(dict(mse=[s["mse"] for s in stats if s], rays=[s["num_rays"] for s in stats if s],
                                  resync=tr.resync, rccl=mapped)), flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
"""


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("planes_graph", ["0", "1"], ids=["joint_entropy_pass", "planes_graph"])
def test_data_parallel_trainer_step_on_rccl(cuda, tmp_path, planes_graph):
    """The DP training step (bucketed ray-loss gradient all-reduced asynchronously while the context pass runs on its own
    stream and host thread, sample count in the bucket's tail, occupancy broadcast, checksum resync) with RCCL as the
    communicator; against the same run without a process group: same loss trajectory up to atomic order."""
    script = tmp_path / "w.py"
    script.write_text(_TRAINER.format(root=ROOT, out=str(tmp_path / "bits")))
    # CNC_PLANES_GRAPH_DP=1: the planes' half of the entropy pass on its own thread / as a captured graph in the DP step too
    r = subprocess.run([sys.executable, str(script)], env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                                                                CNC_PLANES_GRAPH_DP=planes_graph),
                       capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert out["rccl"] and len(out["mse"]) >= 10 and all(m == m and m < 1.0 for m in out["mse"])
    assert out["resync"]["checks"] == 3 and out["resync"]["fired"] == 0
    assert out["mse"][-1] < out["mse"][0]
