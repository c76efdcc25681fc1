"""`python -m cnc_amd.train`: the reference driver's flags (train_CNC_nerf_synthetic.py:71-133), the
SubjectLoader wiring and the results line (:562-613) — on the procedural scene and on a fabricated
nerf_synthetic scene (no real dataset can travel to the GPU box)."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_COLS = 9 + 3 + 6 + 3      # scene..embed sizes | mlp/ctx/occ sizes | one quantisation block | three times


def _fabricate(root, scene, n_train=6, n_test=2, size=24):
    """A white-ish blob in front of transparent background seen from a ring of cameras (RGBA PNGs +
    transforms_{train,test}.json, the nerf_synthetic layout)."""
    from PIL import Image
    d = root / scene
    rng = np.random.default_rng(0)
    for split, n in (("train", n_train), ("test", n_test)):
        (d / split).mkdir(parents=True, exist_ok=True)
        frames = []
        for i in range(n):
            yy, xx = np.mgrid[:size, :size]
            blob = ((xx - size / 2) ** 2 + (yy - size / 2) ** 2) < (size / 4) ** 2
            rgba = np.zeros((size, size, 4), np.uint8)
            rgba[blob] = [200, 120 + 10 * i, 60, 255]
            Image.fromarray(rgba, "RGBA").save(d / split / f"r_{i}.png")
            a = 2 * math.pi * i / n
            eye = np.array([4 * math.cos(a), 4 * math.sin(a), 0.5])
            fwd = -eye / np.linalg.norm(eye)
            right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
            up = np.cross(right, fwd)
            c2w = np.eye(4)
            c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, eye
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": c2w.tolist()})
        json.dump({"camera_angle_x": 0.6911, "frames": frames}, open(d / f"transforms_{split}.json", "w"))


def test_flag_names_and_defaults_are_the_reference_drivers():
    from cnc_amd.train import build_parser
    a = build_parser().parse_args([])
    assert (a.train_split, a.scene, a.lmbda, a.Pg_level, a.Pg_level_2D, a.log2_hashmap_size, a.log2_hashmap_size_2D,
            a.sample_num, a.max_context_layer_num, a.n_features) == ("train", "chair", 2e-3, 12, 4, 19, 17, 200000, 3, 4)
    assert a.data_root.endswith("data/nerf_synthetic")


@pytest.mark.parametrize("kind", ["procedural", "nerf_synthetic"])
def test_cli_end_to_end_writes_the_results_line(cuda, tmp_path, kind, monkeypatch):
    from cnc_amd import train
    monkeypatch.chdir(tmp_path)
    argv = ["--max_steps", "40", "--n_features", "2", "--sample_num", "20000", "--test_views", "1",
            "--log2_hashmap_size", "14", "--log2_hashmap_size_2D", "12", "--results", str(tmp_path / "out.txt"),
            "--out_dir", str(tmp_path / "bits")]
    if kind == "nerf_synthetic":
        _fabricate(tmp_path / "data", "lego")
        argv += ["--data_root", str(tmp_path / "data"), "--scene", "lego"]
    else:
        argv += ["--dataset", "procedural", "--image_size", "48"]
    cols = train.main(argv)
    line = open(tmp_path / "out.txt").read().strip().split("\t")
    assert line == cols and len(line) == N_COLS
    assert line[0] == ("lego" if kind == "nerf_synthetic" else "ball")
    psnr, psnr_codec, est_MB, coded_MB = float(line[1]), float(line[4]), float(line[7]), float(line[8])
    assert math.isfinite(psnr) and abs(psnr - psnr_codec) < 0.5          # decode reproduces the render
    assert 0 < coded_MB < 1.05 * est_MB + 1e-3
    assert line[12] == "13" and float(line[17]) > coded_MB               # total size includes MLP, context, grid
    assert len([f for f in os.listdir(tmp_path / "bits") if f.endswith(".b")]) == 33
