"""HIP aligner kernels (through the `pack_and_align` mirror) vs the CPU oracle: integer outputs,
so everything is bit-exact."""
import numpy as np
import pytest
import torch

from conftest import ball_occupancy

pytestmark = pytest.mark.gpu


def _query_gpu(dev, pts, vxl, resolution=None, res_list=None):
    from cnc_amd.backends import pack_and_align as pa
    N = pts.shape[0]
    mask = torch.zeros(N, dtype=torch.int16, device=dev)
    ov = torch.zeros(N, dtype=torch.int32, device=dev)
    p = torch.as_tensor(pts, device=dev)
    v = torch.as_tensor(vxl, device=dev)
    if res_list is None:
        pa.query_mask_3D(p, v, mask, ov, resolution, N)
    else:
        pa.query_mask_3D_qlist(p, v, mask, ov, torch.as_tensor(res_list, device=dev), N)
    torch.cuda.synchronize()
    return mask.cpu().numpy(), ov.cpu().numpy()


@pytest.mark.parametrize("D,Rb", [(3, 16), (3, 128), (2, 32), (2, 128), (3, 12)])
@pytest.mark.parametrize("R", [18, 59, 514])
def test_query_mask_scalar_resolution(cuda, oracle, D, Rb, R):
    vxl = ball_occupancy(Rb, D, seed=R)
    rng = np.random.default_rng(R + Rb)
    pts = rng.integers(0, R, size=(20011, D)).astype(np.int16)   # incl. ring vertices 0 and R-1
    want_m, want_o = oracle.query_mask(pts, vxl, resolution=R)
    got_m, got_o = _query_gpu(cuda, pts, vxl, resolution=R)
    assert np.array_equal(got_m, want_m)
    assert np.array_equal(got_o, want_o)
    assert got_m.max() == 1 and got_m.min() == 0


@pytest.mark.parametrize("D", [2, 3])
def test_query_mask_per_point_resolution(cuda, oracle, D):
    Rb = 128 if D == 3 else 64
    vxl = ball_occupancy(Rb, D, seed=5)
    rng = np.random.default_rng(9)
    res_choices = np.array([18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514], np.int64)
    rl = res_choices[rng.integers(0, len(res_choices), size=30000)]
    pts = (rng.uniform(size=(30000, D)) * rl[:, None]).astype(np.int16)
    want_m, want_o = oracle.query_mask(pts, vxl, resolution_list=rl)
    got_m, got_o = _query_gpu(cuda, pts, vxl, res_list=rl)
    assert np.array_equal(got_m, want_m)
    assert np.array_equal(got_o, want_o)


def test_query_mask_empty_and_full_grids(cuda, oracle):
    pts = np.random.default_rng(0).integers(0, 44, size=(1000, 3)).astype(np.int16)
    for fill in (False, True):
        vxl = np.full((16, 16, 16), fill)
        want = oracle.query_mask(pts, vxl, resolution=44)
        got = _query_gpu(cuda, pts, vxl, resolution=44)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert got[0].all() == fill
    m, o = _query_gpu(cuda, np.zeros((0, 3), np.int16), np.ones((8, 8, 8), bool), resolution=18)
    assert m.shape == (0,) and o.shape == (0,)


@pytest.mark.parametrize("F", [1, 8])
@pytest.mark.parametrize("dim", [2, 3])
def test_align_and_pack_roundtrip(cuda, oracle, F, dim):
    from cnc_amd.backends import pack_and_align as pa
    dev = cuda
    rng = np.random.default_rng(F + dim)
    cnt = rng.integers(0, 40, size=777).astype(np.int64)   # ragged, with empty slots
    cnt[5] = 288                                            # the reference's max collision count
    T = int(cnt.sum())
    feat = rng.normal(size=(T, F)).astype(np.float32)
    cumsum = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    N, M = cnt.shape[0], int(cnt.max())
    want = oracle.align_and_pack_forward(feat, cnt, V=0.0)
    got = pa.align_and_pack_forward(torch.as_tensor(feat, device=dev), torch.as_tensor(cnt, device=dev),
                                    torch.as_tensor(cumsum, device=dev), N, M, F, 0.0, dim)
    assert got.shape == (N, M, F)
    assert np.array_equal(got.cpu().numpy(), want)
    # padded positions carry V
    got_v = pa.align_and_pack_forward(torch.as_tensor(feat, device=dev), torch.as_tensor(cnt, device=dev),
                                      torch.as_tensor(cumsum, device=dev), N, M, F, -7.0, dim)
    assert np.array_equal(got_v.cpu().numpy(), oracle.align_and_pack_forward(feat, cnt, V=-7.0))
    # backward scatters back: pack -> unpack is the identity on feat
    dpk = rng.normal(size=(N, M, F)).astype(np.float32)
    want_b = oracle.align_and_pack_backward(dpk, cnt, T)
    got_b = pa.align_and_pack_backward(torch.as_tensor(dpk, device=dev), torch.as_tensor(feat, device=dev),
                                       torch.as_tensor(cnt, device=dev), torch.as_tensor(cumsum, device=dev),
                                       N, M, F, T, dim)
    assert np.array_equal(got_b.cpu().numpy(), want_b)
    back = pa.align_and_pack_backward(got, torch.as_tensor(feat, device=dev), torch.as_tensor(cnt, device=dev),
                                      torch.as_tensor(cumsum, device=dev), N, M, F, T, dim)
    assert np.array_equal(back.cpu().numpy(), feat)


def test_pack_errors(cuda):
    from cnc_amd.backends import pack_and_align as pa
    f = torch.zeros((4, 2))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        pa.align_and_pack_forward(f, torch.ones(2, dtype=torch.int64), torch.zeros(3, dtype=torch.int64), 2, 1, 2, 0.0, 3)
