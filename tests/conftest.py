import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a Trainer whose planes' graph cannot be recorded RAISES under test instead of warning and running op by op: the
    # goldens are to hold the schedule that ships (cnc_amd/trainer.py `_ensure_planes_graph`)
    os.environ.setdefault("CNC_PLANES_GRAPH_STRICT", "1")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import cnc_amd._lib as L
    L.lib()  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def make_grid(res_list, log2_T, D, F, seed=0, binary=False):
    """Offsets as GridEncoder builds them (examples/radiance_fields/ngp.py:197-210)."""
    rng = np.random.default_rng(seed)
    offs = [0]
    for R in res_list:
        n = min(2 ** log2_T, int(R) ** D)
        offs.append(offs[-1] + int(np.ceil(n / 8) * 8))
    offs = np.asarray(offs, np.int32)
    res = np.asarray(res_list, np.int32)
    emb = rng.uniform(-1.5, 1.5, size=(offs[-1], F)).astype(np.float32)
    if binary:
        emb = np.where(emb >= 0, 1.0, -1.0).astype(np.float32)
    return offs, res, emb


def ball_occupancy(Rb, D=3, radius=0.35, seed=0):
    ax = (np.arange(Rb) + 0.5) / Rb - 0.5
    g = np.meshgrid(*([ax] * D), indexing="ij")
    r2 = sum(a * a for a in g)
    occ = r2 < radius * radius
    rng = np.random.default_rng(seed)
    occ ^= rng.uniform(size=occ.shape) < 0.02   # speckle so boxes straddle set/unset cells
    return occ
