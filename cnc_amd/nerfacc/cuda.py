"""`nerfacc.cuda` of the reference (nerfacc/cuda/__init__.py:8-41) resolved to the HIP backend.
The reference resolves these names lazily through a JIT/AOT loader; here they are plain
re-exports of cnc_amd.backends.nerfacc_cuda (which fails loudly if libcnc_hip.so is missing)."""
from ..backends.nerfacc_cuda import (  # noqa: F401
    RaySegmentsSpec,
    exclusive_prod_backward,
    exclusive_prod_forward,
    exclusive_sum,
    importance_sampling,
    inclusive_prod_backward,
    inclusive_prod_forward,
    inclusive_sum,
    opencv_lens_undistortion,
    opencv_lens_undistortion_fisheye,
    ray_aabb_intersect,
    searchsorted,
    traverse_grids,
)
