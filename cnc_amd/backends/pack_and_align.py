"""Drop-in for the reference's `pack_and_align` extension (my_cuda_backen/aligner.cpp:4-79).

align_and_pack_* allocate and return (zero-initialised, like torch::zeros at
aligner_kernel.cu:447,529); query_mask_3D* write the caller's mask / overlap buffers.
"""
from __future__ import annotations

import torch

from .. import _lib
from .._lib import check, check_input, ptr, stream


def align_and_pack_forward(voxel_features, unique_count, unique_count_cumsum, N, M, F, V, dim):
    check_input(voxel_features, "voxel_features")
    check_input(unique_count, "unique_count")
    N, M, F = int(N), int(M), int(F)
    packed = torch.zeros((N, M, F), dtype=voxel_features.dtype, device=voxel_features.device)
    if voxel_features.dtype != torch.float32:
        raise RuntimeError("align_and_pack_forward: libcnc_hip builds the fp32 path only")
    rc = _lib.lib().cnc_align_and_pack_forward(ptr(voxel_features), ptr(unique_count),
                                               ptr(unique_count_cumsum.contiguous()), ptr(packed),
                                               N, M, F, float(V), stream(voxel_features.device))
    check(rc, "align_and_pack_forward")
    return packed


def align_and_pack_backward(dL_packed_features, voxel_features, unique_count, unique_count_cumsum,
                            N, M, F, T, dim):
    check_input(dL_packed_features, "dL_packed_features")
    check_input(voxel_features, "voxel_features")
    check_input(unique_count, "unique_count")
    check_input(unique_count_cumsum, "unique_count_cumsum")
    N, M, F, T = int(N), int(M), int(F), int(T)
    d_feat = torch.zeros((T, F), dtype=dL_packed_features.dtype, device=dL_packed_features.device)
    if dL_packed_features.dtype != torch.float32:
        raise RuntimeError("align_and_pack_backward: libcnc_hip builds the fp32 path only")
    rc = _lib.lib().cnc_align_and_pack_backward(ptr(dL_packed_features), ptr(unique_count),
                                                ptr(unique_count_cumsum), ptr(d_feat), N, M, F,
                                                stream(voxel_features.device))
    check(rc, "align_and_pack_backward")
    return d_feat


def _query_checks(points, binary_vxl, mask, overlap):
    check_input(points, "points_n_orig")
    check_input(binary_vxl, "binary_vxl")
    check_input(mask, "mask")
    check_input(overlap, "overlap_area_pool")
    if points.dtype != torch.int16 or mask.dtype != torch.int16:
        raise RuntimeError("expected scalar type Short")     # packed_accessor<short,...> mismatch
    if overlap.dtype != torch.int32:
        raise RuntimeError("expected scalar type Int")
    if binary_vxl.dtype != torch.bool:
        raise RuntimeError("expected scalar type Bool")
    D = points.shape[1]
    if binary_vxl.dim() != D:
        raise RuntimeError(f"expected {D}-D binary_vxl")
    return D


def query_mask_3D(points_n_orig, binary_vxl, mask, overlap_area_pool, resolution, N):
    D = _query_checks(points_n_orig, binary_vxl, mask, overlap_area_pool)
    rc = _lib.lib().cnc_query_mask_3D(ptr(points_n_orig), D, ptr(binary_vxl),
                                      int(binary_vxl.shape[0]), ptr(mask), ptr(overlap_area_pool),
                                      int(resolution), int(mask.shape[0]), stream(mask.device))
    check(rc, "query_mask_3D")


def query_mask_3D_qlist(points_n_orig_list, binary_vxl, mask, overlap_area_pool, resolution_list, N):
    D = _query_checks(points_n_orig_list, binary_vxl, mask, overlap_area_pool)
    check_input(resolution_list, "resolution_list")
    if resolution_list.dtype != torch.int64:
        raise RuntimeError("expected scalar type Long")
    rc = _lib.lib().cnc_query_mask_3D_qlist(ptr(points_n_orig_list), D, ptr(binary_vxl),
                                            int(binary_vxl.shape[0]), ptr(mask),
                                            ptr(overlap_area_pool), ptr(resolution_list),
                                            int(mask.shape[0]), stream(mask.device))
    check(rc, "query_mask_3D_qlist")


def segment_weighted_sum(values, weights, cumsum, mode=0, order=None):
    """(extension) per-slot (weighted) sum / weighted mean / mean of ragged rows, see
    cnc_segment_weighted_sum{,_gathered} in include/cnc_hip.h.  values [T,F] f32, weights [T] f32 or None,
    cumsum int64 [N+1] -> [N,F]; `order` (int64 [T]): ragged row r is values[order[r]]."""
    check_input(values, "values")
    check_input(cumsum, "cumsum")
    if weights is not None:
        check_input(weights, "weights")
    if values.dtype != torch.float32 or cumsum.dtype != torch.int64:
        raise RuntimeError("segment_weighted_sum: values must be float32 and cumsum int64")
    N, F = cumsum.shape[0] - 1, values.shape[1]
    out = torch.empty((N, F), dtype=torch.float32, device=values.device)
    if order is not None:
        check_input(order, "order")
        if order.dtype != torch.int64 or order.shape[0] != values.shape[0]:
            raise RuntimeError("segment_weighted_sum: order must be int64 [T]")
    rc = _lib.lib().cnc_segment_weighted_sum_gathered(ptr(values), ptr(order), ptr(weights), ptr(cumsum), ptr(out), N, F,
                                                      int(mode), stream(values.device))
    check(rc, "segment_weighted_sum")
    return out
