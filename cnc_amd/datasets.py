"""Thin dataset loaders for the CNC drivers.

`SubjectLoader` reads a NeRF-synthetic scene (transforms_{split}.json + RGBA PNGs) with PIL and
generates rays on the device exactly like examples/datasets/nerf_synthetic.py:53-239 (same
constructor arguments, `__len__`, `__getitem__` -> {"pixels", "rays", "color_bkgd"},
`update_num_rays`); `SubjectLoader_Tanks` adds the Tanks&Temples specifics of
examples/datasets/tanks.py:62-259 (intrinsics / per-image poses from text files, `bbox.txt` -> aabb
and step size).  No dataset ships with this repository (no network): the trainer falls back to the
procedural scene of cnc_amd.trainer.SyntheticBallDataset when no `data_root` is given.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .render import Rays


def _read_rgba(path):
    from PIL import Image
    img = Image.open(path)
    if img.mode != "RGBA":
        img = img.convert("RGBA")
    return np.asarray(img, dtype=np.uint8)


def _load_renderings(root_fp: str, subject_id: str, split: str):
    data_dir = os.path.join(root_fp, subject_id)
    with open(os.path.join(data_dir, f"transforms_{split}.json"), "r") as fp:
        meta = json.load(fp)
    images, camtoworlds = [], []
    for frame in meta["frames"]:
        images.append(_read_rgba(os.path.join(data_dir, frame["file_path"] + ".png")))
        camtoworlds.append(frame["transform_matrix"])
    images = np.stack(images, axis=0)
    camtoworlds = np.stack(camtoworlds, axis=0)
    w = images.shape[2]
    focal = 0.5 * w / np.tan(0.5 * float(meta["camera_angle_x"]))
    return images, camtoworlds, focal


class SubjectLoader(torch.utils.data.Dataset):
    """NeRF-synthetic scene: random training rays over all images, or one whole image per index."""

    SPLITS = ["train", "val", "trainval", "test"]
    SUBJECT_IDS = ["chair", "drums", "ficus", "hotdog", "lego", "materials", "mic", "ship"]
    NEAR, FAR = 2.0, 6.0
    OPENGL_CAMERA = True

    def __init__(self, subject_id: str, root_fp: str, split: str, color_bkgd_aug: str = "white",
                 num_rays: int = None, near: float = None, far: float = None,
                 batch_over_images: bool = True, device: torch.device = torch.device("cpu")):
        super().__init__()
        assert split in self.SPLITS, "%s" % split
        assert color_bkgd_aug in ["white", "black", "random"]
        self.split = split
        self.num_rays = num_rays
        self.near = self.NEAR if near is None else near
        self.far = self.FAR if far is None else far
        self.training = (num_rays is not None) and (split in ["train", "trainval"])
        self.color_bkgd_aug = color_bkgd_aug
        self.batch_over_images = batch_over_images
        if split == "trainval":
            a = _load_renderings(root_fp, subject_id, "train")
            b = _load_renderings(root_fp, subject_id, "val")
            images, c2w, focal = np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]]), a[2]
        else:
            images, c2w, focal = _load_renderings(root_fp, subject_id, split)
        self.HEIGHT, self.WIDTH = images.shape[1:3]
        self.focal = focal
        self.images = torch.from_numpy(images).to(torch.uint8).to(device)
        self.camtoworlds = torch.from_numpy(c2w).to(torch.float32).to(device)
        self.K = torch.tensor([[focal, 0, self.WIDTH / 2.0], [0, focal, self.HEIGHT / 2.0], [0, 0, 1]],
                              dtype=torch.float32, device=device)

    def __len__(self):
        return len(self.images)

    def update_num_rays(self, num_rays):
        self.num_rays = num_rays

    @torch.no_grad()
    def __getitem__(self, index):
        dev = self.images.device
        if self.training:
            n = self.num_rays
            image_id = (torch.randint(0, len(self.images), size=(n,), device=dev)
                        if self.batch_over_images else torch.full((n,), index, device=dev))
            x = torch.randint(0, self.WIDTH, size=(n,), device=dev)
            y = torch.randint(0, self.HEIGHT, size=(n,), device=dev)
        else:
            image_id = torch.tensor([index], device=dev)
            x, y = torch.meshgrid(torch.arange(self.WIDTH, device=dev), torch.arange(self.HEIGHT, device=dev),
                                  indexing="xy")
            x, y = x.flatten(), y.flatten()
        rgba = self.images[image_id, y, x] / 255.0
        c2w = self.camtoworlds[image_id]
        sign = -1.0 if self.OPENGL_CAMERA else 1.0
        camera_dirs = torch.stack([(x - self.K[0, 2] + 0.5) / self.K[0, 0],
                                   (y - self.K[1, 2] + 0.5) / self.K[1, 1] * sign,
                                   torch.full_like(x, sign, dtype=torch.float32)], dim=-1)
        directions = (camera_dirs[:, None, :] * c2w[:, :3, :3]).sum(dim=-1)
        origins = torch.broadcast_to(c2w[:, :3, -1], directions.shape)
        viewdirs = directions / torch.linalg.norm(directions, dim=-1, keepdims=True)
        if self.training:
            shape = (self.num_rays,)
        else:
            shape = (self.HEIGHT, self.WIDTH)
        origins = origins.reshape(*shape, 3)
        viewdirs = viewdirs.reshape(*shape, 3)
        rgba = rgba.reshape(*shape, 4)
        pixels, alpha = torch.split(rgba, [3, 1], dim=-1)
        if self.training and self.color_bkgd_aug == "random":
            color_bkgd = torch.rand(3, device=dev)
        elif self.training and self.color_bkgd_aug == "black":
            color_bkgd = torch.zeros(3, device=dev)
        else:
            color_bkgd = torch.ones(3, device=dev)
        return {"pixels": pixels * alpha + color_bkgd * (1.0 - alpha),
                "rays": Rays(origins=origins, viewdirs=viewdirs), "color_bkgd": color_bkgd}


class SubjectLoader_Tanks(SubjectLoader):
    """Tanks&Temples (NSVF layout: intrinsics.txt, bbox.txt, pose/*.txt, rgb/*.png; split by the
    file-name prefix 0_ train / 1_ test).  OpenCV camera convention; `aabb` = bbox * 1.2 and
    `render_step_size` = 4e-3 if the bbox voxel size >= 0.15 else 1e-3 (tanks.py:135-137)."""

    OPENGL_CAMERA = False
    NEAR, FAR = 0.01, 6.0

    def __init__(self, subject_id: str, root_fp: str, split: str, color_bkgd_aug: str = "white",
                 num_rays: int = None, near: float = None, far: float = None,
                 batch_over_images: bool = True, device: torch.device = torch.device("cpu")):
        torch.utils.data.Dataset.__init__(self)
        assert split in ("train", "test")
        self.split = split
        self.num_rays = num_rays
        self.near = self.NEAR if near is None else near
        self.far = self.FAR if far is None else far
        self.training = (num_rays is not None) and split == "train"
        self.color_bkgd_aug = color_bkgd_aug
        self.batch_over_images = batch_over_images
        data_dir = os.path.join(root_fp, subject_id)
        K = np.loadtxt(os.path.join(data_dir, "intrinsics.txt"), dtype=np.float32)[:3, :3]
        bbox = np.loadtxt(os.path.join(data_dir, "bbox.txt"), dtype=np.float32)
        self.aabb = torch.tensor(bbox[:6] * 1.2, dtype=torch.float32, device=device)
        self.render_step_size = 4e-3 if bbox[-1] >= 0.15 else 1e-3
        prefix = "0_" if split == "train" else "1_"
        names = sorted(f for f in os.listdir(os.path.join(data_dir, "rgb")) if f.startswith(prefix))
        images = np.stack([_read_rgba(os.path.join(data_dir, "rgb", f)) for f in names], 0)
        c2w = np.stack([np.loadtxt(os.path.join(data_dir, "pose", os.path.splitext(f)[0] + ".txt"),
                                   dtype=np.float32) for f in names], 0)
        self.HEIGHT, self.WIDTH = images.shape[1:3]
        self.focal = float(K[0, 0])
        self.images = torch.from_numpy(images).to(device)
        self.camtoworlds = torch.from_numpy(c2w).to(torch.float32).to(device)
        self.K = torch.from_numpy(K).to(device)
