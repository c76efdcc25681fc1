"""Scene loaders for the CNC drivers: `SubjectLoader` (NeRF-synthetic: `transforms_{split}.json`
+ RGBA PNGs) and `SubjectLoader_Tanks` (Tanks&Temples in the NSVF layout: `intrinsics.txt`, `bbox.txt`,
`pose/*.txt`, `rgb/*.png`, split by file-name prefix).

Interface of the reference's loaders (examples/datasets/nerf_synthetic.py:53-239, tanks.py:62-259): same
constructor arguments, `len()`, `loader[i]` -> {"pixels", "rays", "color_bkgd"}, `update_num_rays`, the
attributes the drivers read (`images`, `camtoworlds`, `K`, `HEIGHT`, `WIDTH`, `training`, and for T&T the scene
box and step size).  Images are read with PIL (imageio / cv2 are not available offline) and kept on the device as
uint8; rays are generated on the device.  No dataset ships with this repository: without a `data_root` the
trainer uses the procedural scene of cnc_amd.trainer.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .render import Rays

_BKGD = {"white": 1.0, "black": 0.0}


def _rgba(path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as img:
        return np.asarray(img.convert("RGBA"), dtype=np.uint8)


def _read_blender_split(scene_dir: str, split: str):
    """(images uint8 [n,H,W,4], camera-to-world [n,4,4], focal length in pixels) of one split."""
    with open(os.path.join(scene_dir, f"transforms_{split}.json")) as fp:
        meta = json.load(fp)
    frames = meta["frames"]
    images = np.stack([_rgba(os.path.join(scene_dir, fr["file_path"] + ".png")) for fr in frames])
    poses = np.asarray([fr["transform_matrix"] for fr in frames], dtype=np.float32)
    focal = 0.5 * images.shape[2] / np.tan(0.5 * float(meta["camera_angle_x"]))
    return images, poses, focal


class _PosedImages(torch.utils.data.Dataset):
    """Posed RGBA images on the device + pinhole ray generation.  Subclasses fill `images` (uint8 [n,H,W,4]),
    `camtoworlds` (float [n,3|4,4]) and `K` (3x3) and say whether the camera looks down -z (OpenGL) or +z."""

    OPENGL_CAMERA = True
    TRAIN_SPLITS = ("train", "trainval")

    def _configure(self, split, color_bkgd_aug, num_rays, near, far, batch_over_images):
        if color_bkgd_aug not in ("white", "black", "random"):
            raise AssertionError(color_bkgd_aug)
        self.split, self.num_rays, self.color_bkgd_aug = split, num_rays, color_bkgd_aug
        self.near = self.NEAR if near is None else near
        self.far = self.FAR if far is None else far
        self.batch_over_images = batch_over_images
        self.training = num_rays is not None and split in self.TRAIN_SPLITS

    def __len__(self):
        return self.images.shape[0]

    def update_num_rays(self, num_rays):
        self.num_rays = num_rays

    def seed_sampling(self, seed: int):
        """(extension) draw training images / pixels / random backgrounds from a generator of this loader's own
        instead of the global one — one seed per data-parallel rank."""
        self._gen = torch.Generator(device=self.images.device).manual_seed(int(seed))

    def _pixels_to_sample(self, index, dev):
        """(image ids, x, y, output shape): `num_rays` random pixels when training, else every pixel of image
        `index` in row-major order."""
        if self.training:
            n = self.num_rays
            g = getattr(self, "_gen", None)
            ids = (torch.randint(0, len(self), (n,), device=dev, generator=g) if self.batch_over_images
                   else torch.full((n,), index, device=dev))
            return (ids, torch.randint(0, self.WIDTH, (n,), device=dev, generator=g),
                    torch.randint(0, self.HEIGHT, (n,), device=dev, generator=g), (n,))
        cols, rows = torch.meshgrid(torch.arange(self.WIDTH, device=dev), torch.arange(self.HEIGHT, device=dev),
                                    indexing="xy")
        return torch.tensor([index], device=dev), cols.reshape(-1), rows.reshape(-1), (self.HEIGHT, self.WIDTH)

    def _background(self, dev):
        if self.training and self.color_bkgd_aug == "random":
            return torch.rand(3, device=dev, generator=getattr(self, "_gen", None))
        level = _BKGD.get(self.color_bkgd_aug, 1.0) if self.training else 1.0     # evaluation is always on white
        return torch.full((3,), level, device=dev)

    @torch.no_grad()
    def __getitem__(self, index):
        dev = self.images.device
        ids, x, y, shape = self._pixels_to_sample(index, dev)
        rgba = self.images[ids, y, x].to(torch.float32) / 255.0
        pose = self.camtoworlds[ids]
        # pixel centre -> camera-space direction; OpenGL cameras have y up and look down -z
        flip = -1.0 if self.OPENGL_CAMERA else 1.0
        cam = torch.stack([(x - self.K[0, 2] + 0.5) / self.K[0, 0],
                           (y - self.K[1, 2] + 0.5) / self.K[1, 1] * flip,
                           torch.full(x.shape, flip, dtype=torch.float32, device=dev)], dim=-1)
        world = torch.einsum("nk,njk->nj", cam, pose[:, :3, :3].expand(cam.shape[0], 3, 3))
        dirs = world / world.norm(dim=-1, keepdim=True)
        origins = pose[:, :3, 3].expand_as(dirs)
        colour, alpha = rgba[:, :3], rgba[:, 3:]
        bkgd = self._background(dev)
        return {"pixels": (colour * alpha + bkgd * (1.0 - alpha)).reshape(*shape, 3),
                "rays": Rays(origins=origins.reshape(*shape, 3), viewdirs=dirs.reshape(*shape, 3)),
                "color_bkgd": bkgd}


class SubjectLoader(_PosedImages):
    """One NeRF-synthetic scene."""

    SPLITS = ["train", "val", "trainval", "test"]
    SUBJECT_IDS = ["chair", "drums", "ficus", "hotdog", "lego", "materials", "mic", "ship"]
    NEAR, FAR = 2.0, 6.0
    OPENGL_CAMERA = True

    def __init__(self, subject_id: str, root_fp: str, split: str, color_bkgd_aug: str = "white",
                 num_rays: int = None, near: float = None, far: float = None,
                 batch_over_images: bool = True, device: torch.device = torch.device("cpu")):
        super().__init__()
        if split not in self.SPLITS:
            raise AssertionError(split)
        self._configure(split, color_bkgd_aug, num_rays, near, far, batch_over_images)
        scene_dir = os.path.join(root_fp, subject_id)
        parts = [_read_blender_split(scene_dir, s) for s in (("train", "val") if split == "trainval" else (split,))]
        images = np.concatenate([p[0] for p in parts])
        poses = np.concatenate([p[1] for p in parts])
        self.focal = parts[0][2]
        self.HEIGHT, self.WIDTH = images.shape[1:3]
        self.images = torch.from_numpy(images).to(device)
        self.camtoworlds = torch.from_numpy(poses).to(device)
        self.K = torch.tensor([[self.focal, 0.0, self.WIDTH / 2.0], [0.0, self.focal, self.HEIGHT / 2.0],
                               [0.0, 0.0, 1.0]], dtype=torch.float32, device=device)


class SubjectLoader_Tanks(_PosedImages):
    """One Tanks&Temples scene (NSVF layout; file names starting with 0_ are training views, 1_ test views).
    OpenCV camera convention.  `aabb` = the bounding box of bbox.txt scaled by 1.2 and `render_step_size` =
    4e-3 if the box's voxel size is >= 0.15 else 1e-3 (tanks.py:135-137)."""

    SPLITS = ["train", "test"]
    SUBJECT_IDS = ["Barn", "Caterpillar", "Family", "Ignatius", "Truck"]
    NEAR, FAR = 0.01, 6.0
    OPENGL_CAMERA = False
    TRAIN_SPLITS = ("train",)

    def __init__(self, subject_id: str, root_fp: str, split: str, color_bkgd_aug: str = "white",
                 num_rays: int = None, near: float = None, far: float = None,
                 batch_over_images: bool = True, device: torch.device = torch.device("cpu")):
        super().__init__()
        if split not in self.SPLITS:
            raise AssertionError(split)
        self._configure(split, color_bkgd_aug, num_rays, near, far, batch_over_images)
        scene_dir = os.path.join(root_fp, subject_id)
        tag = "0_" if split == "train" else "1_"
        names = sorted(f for f in os.listdir(os.path.join(scene_dir, "rgb")) if f.startswith(tag))
        images = np.stack([_rgba(os.path.join(scene_dir, "rgb", f)) for f in names])
        poses = np.stack([np.loadtxt(os.path.join(scene_dir, "pose", os.path.splitext(f)[0] + ".txt"), dtype=np.float32)
                          for f in names])
        intrinsics = np.loadtxt(os.path.join(scene_dir, "intrinsics.txt"), dtype=np.float32)[:3, :3]
        box = np.loadtxt(os.path.join(scene_dir, "bbox.txt"), dtype=np.float32)
        self.aabb = torch.tensor(box[:6] * 1.2, dtype=torch.float32, device=device)
        self.scene_bbox = self.aabb.view(2, 3)
        self.render_step_size = 4e-3 if float(box[-1]) >= 0.15 else 1e-3
        self.HEIGHT, self.WIDTH = images.shape[1:3]
        self.focal = float(intrinsics[0, 0])
        self.images = torch.from_numpy(images).to(device)
        self.camtoworlds = torch.from_numpy(poses).to(device)
        self.K = torch.from_numpy(intrinsics).to(device)
