"""ctypes binding of libcnc_hip.so — the only way the package reaches a kernel.

There is no CPU fallback: if the library is missing or a tensor is not on the GPU the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcnc_hip.so")
_lib = None

_vp, _u32, _i32, _i64, _f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_int64, C.c_float


class RaySegments(C.Structure):
    """cnc_ray_segments_t (include/cnc_hip.h)."""
    _fields_ = [("vals", _vp), ("chunk_starts", _vp), ("chunk_cnts", _vp), ("ray_indices", _vp),
                ("is_left", _vp), ("is_right", _vp), ("is_valid", _vp)]


class CtxWindow(C.Structure):
    """cnc_ctx_window_t (include/cnc_hip.h)."""
    _fields_ = [("pos", _vp * 16), ("cnt", _vp * 16), ("val", _vp * 16), ("p_at", _i64 * 17), ("v_at", _i64 * 17),
                ("row0", _i64 * 16), ("level", _i32 * 16), ("res", _i32 * 16), ("n_win", _i32)]


class AdamTable(C.Structure):
    """cnc_adam_table_t (include/cnc_hip.h)."""
    _fields_ = [("p", _vp), ("m", _vp), ("v", _vp), ("step", _vp), ("g", _vp * 4), ("g_lo", C.c_uint64 * 4),
                ("g_hi", C.c_uint64 * 4), ("n", C.c_uint64), ("sign_bits", _vp), ("clip_count", _vp)]


class AdamTables(C.Structure):
    """cnc_adam_tables_t (include/cnc_hip.h)."""
    _fields_ = [("table", AdamTable * 4), ("n_tables", _u32), ("first_block", _u32 * 4)]


class FieldSave(C.Structure):
    """cnc_field_save_t (include/cnc_hip.h)."""
    _fields_ = [("feat", _vp), ("ld_feat", _u32), ("h1", _vp), ("h3", _vp), ("h4", _vp), ("head_in", _vp), ("ld_head", _u32),
                ("raw", _vp), ("selector", _vp), ("xyz", _vp), ("xy", _vp), ("xz", _vp), ("yz", _vp), ("n_live", _u32)]


class FusedField(C.Structure):
    """cnc_fused_field_t (include/cnc_hip.h)."""
    _fields_ = [("aabb", _vp), ("bits", _vp * 4), ("offsets", _vp * 4), ("resolutions", _vp * 4), ("freqs", _vp),
                ("packed_weights", _vp * 5), ("packed_biases", _vp * 5), ("w2_row0", _vp), ("packed_weights16", _vp * 5),
                ("units", _vp), ("n_levels", _u32 * 4),
                ("n_features", _u32), ("n_freqs", _u32), ("n_neurons", _u32), ("geo_feat_dim", _u32), ("flags", _u32),
                ("packed_weights16q", _vp * 5), ("guard", _vp), ("call_id", _u32), ("pack_id", _u32),
                ("debug_features", _vp), ("debug_ld", _u32), ("save", FieldSave), ("n_rows_dev", _vp)]


class FieldBwd(C.Structure):
    """cnc_field_bwd_t (include/cnc_hip.h)."""
    _fields_ = [("N", _u32), ("n_neurons", _u32), ("n_features", _u32), ("n_enc_columns", _u32), ("geo_feat_dim", _u32),
                ("ld_base", _u32), ("ld_g2", _u32), ("ld_x", _u32), ("grad_rgb", _vp), ("grad_density", _vp), ("rgb", _vp),
                ("base_out", _vp), ("selector", _vp), ("h1", _vp), ("h3", _vp), ("h4", _vp), ("packed_weights_t", _vp * 5),
                ("G5", _vp), ("G4", _vp), ("G3", _vp), ("G2", _vp), ("G1", _vp), ("dX", _vp), ("bias_grads", _vp),
                ("g_max", _vp)]


class FieldWGrad(C.Structure):
    """cnc_field_wgrad_t (include/cnc_hip.h)."""
    _fields_ = [("N", _u32), ("G", _vp * 5), ("ldG", _u32 * 5), ("n_out", _u32 * 5), ("A", _vp * 5), ("ldA", _u32 * 5),
                ("n_in", _u32 * 5), ("head_gap_col", _u32), ("dW", _vp * 5), ("ld_dW", _u32 * 5), ("g_max", _vp),
                ("workspace", _vp), ("workspace_bytes", C.c_uint64), ("n_workgroups", _u32)]


class FieldPackLayer(C.Structure):
    """cnc_field_pack_layer_t (include/cnc_hip.h)."""
    _fields_ = [("W", _vp), ("b", _vp), ("H", _u32), ("K", _u32), ("ldw", _u32), ("n_tiles", _u32), ("n_ksteps", _u32),
                ("n_ksteps16", _u32), ("n_colblocks", _u32), ("n_ksteps32", _u32), ("Wp", _vp), ("Bp", _vp),
                ("Wp16", _vp), ("Wq16", _vp), ("k_gap", _u32), ("flags", _u32), ("src_off", _u32)]


class FieldPack(C.Structure):
    """cnc_field_pack_t (include/cnc_hip.h)."""
    _fields_ = [("layer", FieldPackLayer * 5), ("row0", _vp), ("row0_len", _u32), ("guard", _vp), ("pack_id", _u32)]


# name -> argtypes, in the order of include/cnc_hip.h
SIGNATURES = {
    "cnc_grid_encode_forward": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _u32, _u32, _vp],
    "cnc_grid_encode_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp],
    "cnc_grid_vertex_bits_words": [_u32, _u32],
    "cnc_grid_vertex_bits": [_vp, _u32, _u32, _u32, _vp, _vp],
    "cnc_grid_encode_backward_binned_workspace": [_u32, _u32, _u32],
    "cnc_grid_encode_backward_binned": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _u32,
                                        _u32, _u32, _vp, C.c_uint64, _vp],
    "cnc_backward_plan_create": [C.POINTER(_vp)],
    "cnc_backward_plan_destroy": [_vp],
    "cnc_grid_encode_backward_overlapped_workspace": [_u32, _u32, _u32],
    "cnc_grid_encode_backward_overlapped": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _u32,
                                            _u32, _u32, _vp, C.c_uint64, _vp],
    "cnc_pack_sign_bits": [_vp, _vp, C.c_uint64, _u32, _vp, _vp],
    "cnc_grid_encode_forward_bits": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp],
    "cnc_mlp_forward32": [_vp, _u32, _u32, _u32, _vp, _vp, _u32, _vp, _vp, _u32, _vp, _vp, _u32, _vp, _u32, _u32, _vp],
    "cnc_mlp_forward": [_vp, _u32, _u32, _u32, _vp, _vp, _u32, _vp, _vp, _u32, _vp, _vp, _u32, _vp, _u32, _u32, _vp],
    "cnc_cnt_np_embed": [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp],
    "cnc_cnt_np_embed_backward": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp],
    "cnc_cnt_np_plan": [_vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "cnc_cnt_np_plan3": [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "cnc_cnt_np_embed_planned": [_vp, _vp, _vp, _vp, _u32, _u32, _vp],
    "cnc_cnt_np_embed_planned_backward": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp],
    "cnc_cnt_np_embed_planned_backward3": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp],
    "cnc_cnt_np_embed_planned_backward3_xyz": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp],
    "cnc_vote_plan_count": [_vp, _u32, _u32, _vp, _vp, _vp],
    "cnc_vote_plan_fill": [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "cnc_cnt_vote_masks": [_vp, _u32, _u32, _vp, _vp],
    "cnc_vote_fraction_table": [_vp, _u32, _u32, _vp, _vp, _vp],
    "cnc_vote_fraction_table_backward": [_vp, _vp, _u32, _u32, _vp, _vp],
    "cnc_cnt_np_embed_planned_masked": [_vp, _vp, _vp, _vp, _u32, _u32, _vp],
    "cnc_query_mask_3D": [_vp, _u32, _vp, _u32, _vp, _vp, _i32, _u32, _vp],
    "cnc_query_mask_3D_qlist": [_vp, _u32, _vp, _u32, _vp, _vp, _vp, _u32, _vp],
    "cnc_align_and_pack_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp],
    "cnc_align_and_pack_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp],
    "cnc_segment_weighted_sum": [_vp, _vp, _vp, _vp, _u32, _u32, _i32, _vp],
    "cnc_segment_weighted_sum_gathered": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _i32, _vp],
    "cnc_ray_aabb_intersect": [_vp, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp],
    "cnc_traverse_grids": [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                           _f32, _f32, _i32, _i32, C.POINTER(RaySegments), C.POINTER(RaySegments), _vp, _vp],
    "cnc_march_samples": [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                          _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cnc_occupancy_coarse_words": [_i32, _i32, _i32, _i32],
    "cnc_occupancy_coarse_bits": [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "cnc_march_samples_coarse": [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cnc_sample_positions": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "cnc_inclusive_sum": [_vp, _vp, _vp, _vp, _u32, _i64, _i32, _i32, _vp],
    "cnc_exclusive_sum": [_vp, _vp, _vp, _vp, _u32, _i64, _i32, _i32, _vp],
    "cnc_inclusive_prod_forward": [_vp, _vp, _vp, _vp, _u32, _i64, _vp],
    "cnc_exclusive_prod_forward": [_vp, _vp, _vp, _vp, _u32, _i64, _vp],
    "cnc_inclusive_prod_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _i64, _vp],
    "cnc_exclusive_prod_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _i64, _vp],
    "cnc_volrend_forward": [_vp] * 15 + [_u32, _u32, _vp],
    "cnc_volrend_backward": [_vp] * 19 + [_u32, _u32, _vp],
    "cnc_render_visibility": [_vp] * 5 + [_i32, _f32, _f32, _vp, _vp, _vp, _u32, _vp],
    "cnc_compact_samples": [_vp] * 9 + [_u32, _vp],
    "cnc_ray_window_samples": [_vp] * 10 + [_u32, _vp],
    "cnc_ray_transmittance": [_vp] * 6 + [_u32, _vp],
    "cnc_ray_window_next": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _i32, _u32, _vp],
    "cnc_interval_edges_to_samples": [_vp] * 9 + [_u32, _vp],
    "cnc_pack_bounds": [_vp, _i64, _vp, _vp, _i64, _vp],
    "cnc_level_stats_forward": [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp],
    "cnc_level_stats_backward": [_vp, _vp, _u32, _u32, _vp, _vp, C.c_uint64, _vp, _vp],
    "cnc_field_prepare": [_vp, _vp, _u32, _vp, _vp, _vp],
    "cnc_field_pack_layer": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _vp],
    "cnc_field_pack_layer16": [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp],
    "cnc_field_pack_all": [C.POINTER(FieldPack), _vp],
    "cnc_field_backward_chain": [C.POINTER(FieldBwd), _vp],
    "cnc_field_weight_grads_workspace": [C.POINTER(FieldWGrad), C.POINTER(C.c_uint64)],
    "cnc_field_weight_grads": [C.POINTER(FieldWGrad), _vp],
    "cnc_field_fused_forward": [C.POINTER(FusedField), _vp, _vp, _u32, _vp, _vp, _vp],
    "cnc_ste_binary_forward": [_vp, _vp, C.c_uint64, _vp],
    "cnc_ste_binary_backward": [_vp, _vp, _vp, C.c_uint64, _vp],
    "cnc_relu_backward_bias_partials": [_u32],
    "cnc_relu_backward_bias": [_vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "cnc_field_sinusoid": [_vp, _vp, _u32, _u32, _vp, _u32, _u32, _vp],
    "cnc_field_post": [_vp, _u32, _u32, _vp, _vp, _u32, _vp, _vp, _u32, _u32, _vp],
    "cnc_field_post_backward": [_vp, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _vp, _vp],
    "cnc_ctx_mlp_forward": [_vp, _u32, _u32, _vp, _u32, _u32, _vp, _vp, _u32, _u32, _u32] + [_vp] * 6 + [_vp, _vp],
    "cnc_ctx_mlp_backward": [_vp, _u32, _u32, _vp, _u32, _u32, _vp, _vp, _u32, _u32, _u32] + [_vp] * 6 + [_vp] * 10 + [_u32, _u32, _u32, _u32, _vp],
    "cnc_ctx_window_gather": [_vp] * 8,
    "cnc_rows_scatter": [_vp, _vp, _vp, C.c_uint64, _u32, _vp],
    "cnc_table_adam": [_vp] + [C.c_double] * 6 + [_vp],
    "cnc_ray_window_positions": [_vp] * 10 + [_u32, _vp],
    "cnc_scatter_counted": [_vp, _vp, _vp, _vp, C.c_uint64, _vp],
    "cnc_ctx_compact": [_vp, _vp, _vp, _vp, C.c_uint64, _i32, _vp, _vp, _vp, _vp, _vp],
    "cnc_plane_ring_vertices": [_vp, C.c_uint64, _u32, _u32, C.c_uint64, _vp, _vp, _vp],
    "cnc_bernoulli_bits_partials": [C.c_uint64, _u32],
    "cnc_bernoulli_bits_forward": [_vp, _vp, _vp, C.c_uint64, _u32, _vp, _vp],
    "cnc_bernoulli_bits_backward": [_vp, _vp, _vp, _vp, C.c_uint64, _u32, _vp, _vp, _vp],
    "cnc_segment_weighted_sum_backward": [_vp, _vp, _vp, _vp, _u32, C.c_uint64, _u32, _i32, _vp, _vp],
    "cnc_segment_weighted_sum_gathered_backward": [_vp, _vp, _vp, _vp, _vp, _u32, C.c_uint64, _u32, _i32, _vp, _vp],
}

# entry points that return something other than a status code
RESTYPES = {"cnc_grid_encode_backward_binned_workspace": C.c_uint64,
            "cnc_grid_encode_backward_overlapped_workspace": C.c_uint64, "cnc_bernoulli_bits_partials": C.c_uint32,
            "cnc_relu_backward_bias_partials": C.c_uint32, "cnc_occupancy_coarse_words": C.c_uint32}

CNC_FLAG_STE_BINARY = 1
CNC_FLAG_LEVELS_FINEST_FIRST = 2
CNC_FLAG_BIN_LANE_STORES = 4
CNC_FLAG_CELL_MERGE = 8
CNC_FLAG_CELL_CARRY = 16
CNC_FIELD_SH_FP16 = 1
CNC_FIELD_MFMA_F16X3 = 2
CNC_FIELD_TWO_WAVES = 4
CNC_FIELD_WAVES4 = 8
CNC_PACK_TRANSPOSE = 1
CNC_PACK_ZERO_FIRST = 2
CNC_VOLREND_ACCUMULATE = 1
CNC_VOLREND_FINALIZE = 2
ABI_VERSION = 31          # cnc_abi_version() of the library this table was written for


def lib() -> C.CDLL:
    """Load libcnc_hip.so (built by `python -m cnc_amd.build`); raise if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -m cnc_amd.build`.")
        L = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, C.c_int)
        L.cnc_error_string.argtypes = [C.c_int]
        L.cnc_error_string.restype = C.c_char_p
        L.cnc_abi_version.restype = C.c_int
        if L.cnc_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI v{L.cnc_abi_version()}, this package expects v{ABI_VERSION}: "
                               "rebuild it with `python -m cnc_amd.build`")
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().cnc_error_string(rc).decode()}")


def ptr(t):
    """Device pointer of an optional tensor."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream(device=None) -> int:
    """Raw hipStream_t of torch's current stream on `device` (a tensor's device; default: the
    current device).  Through torch's raw-stream getter when it is there: ~0.3 us instead of the ~5 us of building
    a `torch.cuda.Stream` object — this is called once per kernel launch, ~100 times per training step."""
    if _raw_stream is not None:
        idx = None
        if isinstance(device, torch.device):
            idx = device.index
        elif isinstance(device, int):
            idx = device
        if idx is None:
            idx = torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


# TORCH_CHECK-style argument checks of the reference's host wrappers
def check_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")


def check_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def check_input(t, name):
    check_cuda(t, name)
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
