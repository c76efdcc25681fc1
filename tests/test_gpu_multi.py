"""The N > 1 control flow on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL (the
CNC_BENCH_ONE_DEVICE / CNC_BENCH_BACKEND and CNC_DIST_ONE_DEVICE / CNC_DIST_BACKEND hooks).
 * `python bench.py --gpus 2` spawns its own ranks, all-reduces the table gradient inside the timed
   region and reports n_gpus = 2;
 * a 2-rank `Trainer` joins the process group by itself and its replicas hold identical parameters at the
   re-alignment points (ray-loss gradients all-reduced while the context backward runs; entropy-loss gradients
   are replica-identical up to atomic order; parameters broadcast every `step_update` steps)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_spawns_two_ranks_and_allreduces(cuda):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-train-step"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=_clean_env(CNC_BENCH_ONE_DEVICE="1", CNC_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]              # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert [x["rank"] for x in out["ranks"]] == [0, 1] and all(x["world_size"] == 2 for x in out["ranks"])
    # one exchange per timed frame: issued asynchronously at the end of the frame, joined after the next march (or at the
    # end of the timed region)
    ar = [k for k in out["kernels"] if k.startswith("allreduce(grad_table)")]
    assert len(ar) == 1 and out["kernels"][ar[0]]["launches"] == 1
    # whole-job value: both ranks' samples over the slowest rank's time
    assert out["value"] > 0 and out["config"]["samples_per_step_rank0"] > 6e7
    assert out["value"] * out["ms_per_step"] * 1e-3 > 1.5 * out["config"]["samples_per_step_rank0"]
    assert "[bench rank 1]" in r.stderr and "[bench rank 0]" in r.stderr


def test_bench_reports_the_data_parallel_training_step(cuda):
    """At N > 1 the line carries a `train_step` block measured on every rank (the whole model, ray-loss gradient
    all-reduced while the context backward runs): aggregate samples / s, the all-reduce alone and what of it the
    step still waits for."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=1500,
                       env=_clean_env(CNC_BENCH_ONE_DEVICE="1", CNC_BENCH_BACKEND="gloo", CNC_BENCH_TRAIN_WARM="20",
                                      CNC_BENCH_TRAIN_STEPS="6"))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    ts = out["train_step"]
    assert ts["world_size"] == 2 and ts["steps"] == 6
    assert ts["allreduce_bytes"] > 150e6                      # the flat bucket: tables + MLPs + context models at F=8
    assert ts["allreduce_alone_ms"] > 0 and ts["allreduce_exposed_ms_per_step"] >= 0
    assert 0.0 <= ts["allreduce_hidden_frac"] <= 1.0
    assert ts["rendered_samples_per_s"] > 0 and ts["samples_per_step"] > 1e5     # both ranks' samples
    # the data-parallel step is the single-GPU step + the exchange: same streams, the planes' graph replayed
    sch = ts["schedule"]
    assert sch["streams"] == 3 and sch["planes_graph"] is not None and sch["planes_graph"]["replays"] > 0, sch


def test_bench_under_the_drivers_launcher(cuda):
    """The driver's own command line for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` (here both ranks on cuda:0 over gloo)."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-train-step"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=_clean_env(CNC_BENCH_ONE_DEVICE="1", CNC_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and [x["world_size"] for x in out["ranks"]] == [2, 2]


def test_bench_refuses_a_world_that_contradicts_gpus(cuda):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300,
                       env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


_TRAINER_WORKER = r"""
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
assert torch.distributed.is_initialized()
assert torch.distributed.get_world_size() == {world}
stats = [tr.train_step(s) for s in range({steps})]        # replicas are compared / re-aligned after steps 3, 7, 11, ...
sums = [float(p.detach().double().sum()) for p in list(tr.field.parameters()) + list(tr.context.parameters())]
absum = [float(p.detach().double().abs().sum()) for p in list(tr.field.parameters()) + list(tr.context.parameters())]
rays = [s["num_rays"] for s in stats if s]
print("RESULT " + json.dumps(dict(rank=tr.rank, device=str(tr.device), sums=sums, absum=absum, rays=rays,
                                  mse=[s["mse"] for s in stats if s], samples=[s["n_rendering_samples"] for s in stats if s],
                                  binaries=int(tr.estimator.binaries.sum()), resync=tr.resync,
                                  planes_graph=[tr.planes_graph.captures, tr.planes_graph.replays] if tr.planes_graph else None)),
      flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
"""


def _run_trainer_ranks(tmp_path, world, steps):
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    script = tmp_path / "worker.py"
    script.write_text(_TRAINER_WORKER.format(root=ROOT, out=str(tmp_path / "bits"), world=world, steps=steps))
    procs = []
    for rank in range(world):
        env = _clean_env(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                         MASTER_PORT=str(port), CNC_DIST_ONE_DEVICE="1", CNC_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=1500)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]))
    return sorted(outs, key=lambda d: d["rank"])


def _check_replicas(outs, steps, step_update=4):
    a = outs[0]
    assert a["device"] == "cuda:0"                             # the one-device hook
    for b in outs[1:]:
        assert b["device"] == "cuda:0"
        assert a["sums"] == b["sums"] and a["absum"] == b["absum"]      # bit-identical replicas at the last comparison
        assert a["binaries"] == b["binaries"]
        # the ray budget follows the ALL-REDUCED sample count of the step before (it rides in the gradient bucket's
        # tail): the same number on every rank, from step 2 on a different one than the initial batch
        assert a["rays"] == b["rays"]
        assert a["mse"] != b["mse"]                             # ... trained on different rays
        assert a["resync"] == b["resync"]                       # every rank took the same decisions
    assert sum(a["absum"]) > 0
    assert a["rays"][0] == a["rays"][1] == 512 and len(set(a["rays"])) > 2
    # what rank 0's budget at step k + 1 was computed from: the mean over the ranks of the counts of step k - 1
    world = len(outs)
    for k in range(1, steps - 1):
        mean = sum(o["samples"][k - 1] for o in outs) / world
        if mean >= 1.0:
            assert a["rays"][k + 1] == int(a["rays"][k - 1] * ((1 << 14) / mean)), k
    r = a["resync"]
    assert r["checks"] == steps // step_update and 0 <= r["fired"] <= r["checks"]
    assert (r["tensors"] == 0) == (r["fired"] == 0) and (r["bytes"] == 0) == (r["fired"] == 0)
    return r


def test_two_rank_trainer_replicas_stay_identical(cuda, tmp_path):
    """20 data-parallel steps = five occupancy refreshes at step_update = 4, with the schedule that ships: the planes' half of
    the entropy pass replayed from its captured graph on every rank (CNC_PLANES_GRAPH_DP defaults to on)."""
    outs = _run_trainer_ranks(tmp_path, 2, 20)
    _check_replicas(outs, 20)
    assert all(o["planes_graph"][0] >= 2 and o["planes_graph"][1] >= 8 for o in outs), [o["planes_graph"] for o in outs]


def test_eight_rank_trainer_and_bench(cuda, tmp_path):
    """The rank-count-dependent paths at the world size the driver uses — `shard_range`, the bucket's division, per-rank
    seeds, `spawn_ranks`, the count in the bucket's tail, the checksum resync — eight ranks on ONE device over gloo
    (CNC_DIST_ONE_DEVICE / CNC_BENCH_ONE_DEVICE): 20 data-parallel training steps, then `bench.py --gpus 8`."""
    outs = _run_trainer_ranks(tmp_path, 8, 20)
    assert [o["rank"] for o in outs] == list(range(8))
    r = _check_replicas(outs, 20)
    print("resync over 20 steps at 8 ranks:", r)
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-train-step", "--no-field"], cwd=ROOT, capture_output=True, text=True,
                       timeout=1800, env=_clean_env(CNC_BENCH_ONE_DEVICE="1", CNC_BENCH_BACKEND="gloo"))
    assert b.returncode == 0, b.stderr[-3000:]
    lines = [l for l in b.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and [x["rank"] for x in out["ranks"]] == list(range(8))
    assert all(x["world_size"] == 8 for x in out["ranks"])
    # eight cameras, eight frames: the whole job's samples over the slowest rank's time
    assert out["value"] * out["ms_per_step"] * 1e-3 > 7.0 * out["config"]["samples_per_step_rank0"] * 0.8


def test_two_rank_cli_on_a_real_scene_layout(cuda, tmp_path):
    """`torchrun`-style launch of `python -m cnc_amd.train` on an on-disk nerf_synthetic scene: the loaders live on each
    rank's own device (here both on cuda:0), every rank trains on its own pixel stream, and only rank 0 evaluates,
    encodes / decodes and writes the results line."""
    from test_gpu_cli import _fabricate
    _fabricate(tmp_path / "data", "lego")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    argv = ["--max_steps", "24", "--n_features", "2", "--sample_num", "20000", "--test_views", "1", "--log2_hashmap_size", "14",
            "--log2_hashmap_size_2D", "12", "--results", str(tmp_path / "out.txt"), "--out_dir", str(tmp_path / "bits"),
            "--data_root", str(tmp_path / "data"), "--scene", "lego"]
    procs = []
    for rank in range(2):
        env = _clean_env(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                         CNC_DIST_ONE_DEVICE="1", CNC_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-m", "cnc_amd.train"] + argv, env=env, cwd=str(tmp_path),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    lines = open(tmp_path / "out.txt").read().strip().splitlines()
    assert len(lines) == 1 and lines[0].split("\t")[0] == "lego"          # ONE results line, rank 0's
    assert sum("results line appended" in o for o in outs) == 1
    assert sum("evaluation:" in o for o in outs) == 1
    assert len([f for f in os.listdir(tmp_path / "bits") if f.endswith(".b")]) == 33
