"""`python -m cnc_amd.train` — the CNC protocol end to end with the reference driver's flags.

Flag names and defaults are those of examples/train_CNC_nerf_synthetic.py:71-133 (and
train_CNC_tank_temples.py for `--dataset tanks`); everything the reference hard-codes (:135-186: resolutions,
20000 steps, 2^18 target samples, lr 6e-3, schedules, aabb +-1.5 / bbox*1.2, step 5e-3 / 4e-3|1e-3) is the
default of `TrainConfig`.  What runs: train -> evaluate -> encode (.b files) -> wipe + decode -> evaluate ->
13-bit MLP quantisation -> evaluate, then ONE tab-separated results line with the reference's columns
(:562-613) appended to ./results/<dataset>/output.txt.

No dataset ships with this repository (no network): `--dataset procedural` (default when --data_root does not
exist) trains on the built-in analytic scene so the whole protocol can be exercised anywhere.
Extensions: --max_steps, --test_views (subset of the test images), --results, --dataset.
"""
from __future__ import annotations

import argparse
import math
import os
import pathlib
import time

import numpy as np
import torch

from .render import NERF_SYNTHETIC_SCENES, TANKS_SCENES, render_image_with_occgrid_test
from .trainer import LoaderDataset, TrainConfig, Trainer, quantize_params


def build_parser():
    ap = argparse.ArgumentParser(prog="python -m cnc_amd.train")
    ap.add_argument("--data_root", type=str, default=str(pathlib.Path.cwd() / "data/nerf_synthetic"),
                    help="the root dir of the dataset")
    ap.add_argument("--train_split", type=str, default="train", choices=["train", "trainval"],
                    help="which train split to use")
    ap.add_argument("--scene", type=str, default="chair", choices=NERF_SYNTHETIC_SCENES + TANKS_SCENES + ["ball"],
                    help="which scene to use")
    ap.add_argument("--lmbda", type=float, default=2e-3)
    ap.add_argument("--Pg_level", type=int, default=12)
    ap.add_argument("--Pg_level_2D", type=int, default=4)
    ap.add_argument("--log2_hashmap_size", type=int, default=19)
    ap.add_argument("--log2_hashmap_size_2D", type=int, default=17)
    ap.add_argument("--sample_num", type=int, default=200000)
    ap.add_argument("--max_context_layer_num", type=int, default=3)
    ap.add_argument("--n_features", type=int, default=4)
    # extensions
    ap.add_argument("--dataset", choices=["nerf_synthetic", "tanks", "procedural"], default=None)
    ap.add_argument("--max_steps", type=int, default=20000)
    ap.add_argument("--test_views", type=int, default=None, help="evaluate this many test images (default: all)")
    ap.add_argument("--image_size", type=int, default=200, help="procedural scene only")
    ap.add_argument("--seed", type=int, default=42, help="set_random_seed value (the reference drivers fix 42, train:135)")
    ap.add_argument("--results", type=str, default=None, help="results file (default ./results/<dataset>/output.txt)")
    ap.add_argument("--out_dir", type=str, default=None, help="bitstream directory (default ./bitstreams/<scene>)")
    return ap


def make_config_and_data(args, device, rank=0, world=1):
    kind = args.dataset
    if kind is None:
        root_has_scene = os.path.isdir(os.path.join(args.data_root, args.scene))
        kind = ("tanks" if args.scene in TANKS_SCENES else "nerf_synthetic") if root_has_scene else "procedural"
    scene = args.scene if kind != "procedural" else "ball"
    kw = dict(scene=scene, lmbda=args.lmbda, Pg_level=args.Pg_level, Pg_level_2D=args.Pg_level_2D,
              log2_hashmap_size=args.log2_hashmap_size, log2_hashmap_size_2D=args.log2_hashmap_size_2D,
              sample_num=args.sample_num, max_context_layer_num=args.max_context_layer_num,
              n_features=args.n_features, max_steps=args.max_steps, image_size=args.image_size, seed=args.seed,
              out_dir=args.out_dir or f"./bitstreams/{scene}",
              weight_decay=2e-5 if scene == "drums" else 2e-6)           # train:170-172
    if args.max_steps != 20000:      # the milestones of a shortened run keep their relative positions
        f = args.max_steps / 20000.0
        kw.update(milestones=tuple(int(m * f) for m in (9000, 12000, 15000, 17000, 19000)),
                  warmup_iters=max(1, int(1000 * f)))
    dataset = None
    if kind == "nerf_synthetic":
        from .datasets import SubjectLoader
        train = SubjectLoader(subject_id=scene, root_fp=args.data_root, split=args.train_split,
                              num_rays=1024, device=device)
        test = SubjectLoader(subject_id=scene, root_fp=args.data_root, split="test", num_rays=None, device=device)
        dataset = LoaderDataset(train, test)
    elif kind == "tanks":
        from .datasets import SubjectLoader_Tanks
        train = SubjectLoader_Tanks(subject_id=scene, root_fp=args.data_root, split="train", num_rays=1024, device=device)
        test = SubjectLoader_Tanks(subject_id=scene, root_fp=args.data_root, split="test", num_rays=None, device=device)
        # near_plane stays 0.0: the reference driver never hands the loader's NEAR to the sampler
        # (train_CNC_tank_temples.py:176)
        kw.update(aabb=tuple(float(v) for v in train.aabb.tolist()), render_step_size=train.render_step_size)
        dataset = LoaderDataset(train, test)
    if dataset is not None and world > 1:
        # data parallelism: every rank draws its OWN images / pixels (the all-reduce then averages world different
        # batches); replica-identical draws (occupancy cells, context windows) stay on the global generators
        dataset.seed_sampling(42 + 1000 * rank)
    n_test = len(dataset) if dataset is not None else 4
    kw["test_views"] = n_test if args.test_views is None else min(args.test_views, n_test)
    return kind, TrainConfig(**kw), dataset


@torch.no_grad()
def evaluate(tr: Trainer, n_views: int):
    """(psnr, lpips, -ssim) averaged over the test views, as train:384-431 (LPIPS unavailable: NaN).
    world > 1: the views are sharded over the ranks (`cdist.shard_range`) and the two sums are
    all-reduced — every rank calls this at the same point of the protocol and gets the same three numbers."""
    from . import dist as cdist
    from .metrics import psnr, ssim
    c = tr.cfg
    tr.field.eval(); tr.estimator.eval()
    lo, hi = cdist.shard_range(n_views, tr.rank, tr.world)
    ps = ss = 0.0
    for i in range(lo, hi):
        d = tr.dataset.view(i)
        rgb, _, _, _ = render_image_with_occgrid_test(1024, tr.field, tr.estimator, d["rays"], near_plane=c.near_plane,
                                                      render_step_size=c.render_step_size, render_bkgd=d["color_bkgd"],
                                                      cone_angle=c.cone_angle, alpha_thre=c.alpha_thre)
        ps += psnr(rgb, d["pixels"])
        ss += -ssim(rgb.permute(2, 0, 1).unsqueeze(0), d["pixels"].permute(2, 0, 1).unsqueeze(0))
    ps, ss = cdist.sum_over_ranks([ps, ss], tr.device)
    return ps / max(n_views, 1), float("nan"), ss / max(n_views, 1)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("cnc_amd.train needs an MI355X (the HIP extension has no CPU fallback)")
    # under `torchrun --nproc-per-node N`: join the group first, so that the loaders are built on THIS rank's GPU
    from . import dist as cdist
    rank, _, world = cdist.init()
    device = torch.device("cuda", cdist.local_device_index() if world > 1 else 0)
    kind, cfg, dataset = make_config_and_data(args, device, rank, world)
    tr = Trainer(cfg, device=device, dataset=dataset)
    r4 = lambda v: str(np.round(v, decimals=4))

    tic = time.time()
    tr.train(steps=cfg.max_steps)
    elapsed = time.time() - tic
    # world > 1: the replicas are identical after training.  Every rank runs the whole tail — its share of each
    # evaluation (views sharded, sums all-reduced), and the codec round trip on files of its own (the coder is
    # deterministic: same bytes on every rank; decode is sequential over levels, so it cannot be sharded) — so no
    # rank waits in a barrier under the communicator's watchdog while another one works.  Rank 0's files are the
    # ones under the canonical prefix, and rank 0 prints and writes the results line.
    say = print if rank == 0 else (lambda *a, **k: None)
    psnr_avg, lpips_avg, ssim_avg = evaluate(tr, cfg.test_views)
    say(f"evaluation: psnr_avg={psnr_avg}, lpips_avg={lpips_avg}, ssim_avg={ssim_avg}")

    tic = time.time()
    Pgs, embed_bits_MB, embed_bits_MB_codec, prefix = tr.encode(
        None if rank == 0 else os.path.join(cfg.out_dir, f".rank{rank}", "b"))
    encoding_time = time.time() - tic
    say(f"encoded: estimated {embed_bits_MB} MB, coded {embed_bits_MB_codec} MB in {encoding_time:.1f} s")
    tic = time.time()
    tr.decode_into_field(Pgs, prefix)
    decoding_time = time.time() - tic
    psnr_c, lpips_c, ssim_c = evaluate(tr, cfg.test_views)
    say(f"evaluation_decoded: psnr_avg_codec={psnr_c}, lpips_avg_codec={lpips_c}, ssim_avg_codec={ssim_c}, "
          f"size_codec={embed_bits_MB_codec}")

    sizes = tr.sizes_MB(embed_bits_MB_codec)
    mlp = {n: p for n, p in tr.field.named_parameters() if "encoding" not in n}
    cols = [cfg.scene, r4(psnr_avg), r4(lpips_avg), r4(ssim_avg), r4(psnr_c), r4(lpips_c), r4(ssim_c),
            r4(embed_bits_MB), r4(embed_bits_MB_codec)]
    per_digit = []
    MBs_orig = 0.0
    for digit in [13]:
        MBs, MBs_orig, q_state = quantize_params(mlp, digits=digit)
        tr.field.load_state_dict(q_state, strict=False)
        p_q, l_q, s_q = evaluate(tr, cfg.test_views)
        total = embed_bits_MB_codec + sizes["context_models"] + sizes["occupancy_grid"] + MBs
        per_digit += [str(digit), r4(MBs), r4(p_q), r4(l_q), r4(s_q), r4(total)]
        say(f"{digit}-bit MLP: psnr={p_q}, total size {total * 1024:.1f} KB")
    cols += [r4(MBs_orig), r4(sizes["context_models"]), r4(sizes["occupancy_grid"])] + per_digit
    cols += [r4(elapsed), r4(encoding_time), r4(decoding_time)]

    if rank != 0:
        import shutil
        shutil.rmtree(os.path.join(cfg.out_dir, f".rank{rank}"), ignore_errors=True)
        return cols
    folder = {"nerf_synthetic": "Synthetic-NeRF", "tanks": "TanksAndTemple", "procedural": "procedural"}[kind]
    out = args.results or os.path.join("./results", folder, "output.txt")
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "a") as fw:
        fw.write("\t".join(cols) + "\n")
    print(f"results line appended to {out}")
    return cols


if __name__ == "__main__":
    main()
