"""GridEncoder — host-side mirror of the reference's hash-grid encoder module.

Reference: examples/radiance_fields/ngp.py — `STE_binary` :22-39, `STE_multistep` :41-47,
`_grid_encode` :49-165, `GridEncoder` :171-315 (`gridencoder/__init__.py:1` imports the class from
an empty file; this module is what makes `from gridencoder import GridEncoder` real).

Same constructor arguments, buffers (`offsets_list`, `resolutions_list`), parameter (`params`,
U(-1e-4, 1e-4)) and the three entry points `forward`, `forward_diff_levels`,
`forward_given_params`, with the same argument meaning and output layout `[..., L_calc * F]`.

MI355X-first difference (opt-in per instance, `fused_ste=True`, the default): when
`ste_binary=True` the reference materialises `STE_binary(params)` — 4-7 elementwise passes over the
whole 128 MB table before every encoder call, and as many again in backward — and then gathers from
the copy.  Here the sign is taken inside the gather kernel and the STE mask `|p| <= 1` inside the
scatter kernel (CNC_FLAG_STE_BINARY), so the table is never copied.  The results are identical
to the unfused path (tests/test_gridencoder_glue.py); `fused_ste=False` reproduces the reference's
op-by-op dataflow.
"""
from __future__ import annotations

import numpy as np
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _caches, _gradsink
from .backends import gridencoder_backend as _backend


class STE_binary(Function):
    """+1 where clamp(x,-1,1) >= 0 else -1; gradient passes where |x| <= 1 (ngp.py:22-39)."""

    @staticmethod
    def _kernel_ok(*ts):
        return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in ts)

    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        if STE_binary._kernel_ok(input):
            from . import _lib
            out = torch.empty_like(input)
            _lib.check(_lib.lib().cnc_ste_binary_forward(input.data_ptr(), out.data_ptr(), input.numel(),
                                                         _lib.stream(input.device)), "ste_binary_forward")
            return out
        c = torch.clamp(input, min=-1, max=1)
        return ((c >= 0) * 1.0 + (c < 0) * -1.0).to(input.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        (input,) = ctx.saved_tensors
        if STE_binary._kernel_ok(input, grad_output):
            from . import _lib
            out = torch.empty_like(input)
            _lib.check(_lib.lib().cnc_ste_binary_backward(input.data_ptr(), grad_output.data_ptr(), out.data_ptr(),
                                                          input.numel(), _lib.stream(input.device)), "ste_binary_backward")
            return out
        return grad_output * ((input >= -1) & (input <= 1)).to(grad_output.dtype)


class STE_multistep(Function):
    """round(x*Q)/Q with identity gradient (ngp.py:41-47)."""

    @staticmethod
    def forward(ctx, input, Q):
        return torch.round(input * Q) / Q

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None


class _grid_encode(Function):
    """Autograd wrapper of the encoder kernels (ngp.py:49-165).

    `min_level_id` is either an int (scalar level window: the offset / resolution tables are
    sliced, ngp.py:86-97) or an int32 tensor [N] (per-point window, ngp.py:98-109).
    `ste` (extension) = take sign(embeddings) on the fly / mask the gradient with |e| <= 1.
    """

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets_list, resolutions_list, calc_grad_inputs=False,
                min_level_id=None, n_levels_calc=1, binary_vxl=None, PV=0, ste=False, bits=None,
                clip_count=None, occ_sat=None, binned=None, vertex_bits=None):
        inputs = inputs.contiguous()
        if calc_grad_inputs:
            # dead in the reference too (ngp.py:58-60)
            raise AssertionError("calc_grad_inputs not applicable!")
        Rb = 128
        if binary_vxl is not None:
            binary_vxl = binary_vxl.contiguous()
            Rb = binary_vxl.shape[-1]
            assert binary_vxl.dim() == inputs.shape[-1]
        N, num_dim = inputs.shape
        n_features = embeddings.shape[1]
        embeddings = embeddings.contiguous()
        # The reference's kernel writes level-major [L, N, F] and the wrapper permutes to [N, L*F]
        # (ngp.py:111); the kernels here take a row stride (out_ld), so the point-major result is
        # written directly: same values, no permute copy (and none of the gradient in backward).
        ld = n_levels_calc * n_features
        outputs = torch.empty(N, ld, device=inputs.device, dtype=embeddings.dtype)

        scalar_window = isinstance(min_level_id, int)
        if scalar_window:
            max_level_id = min_level_id + n_levels_calc
            offs = offsets_list[min_level_id:max_level_id + 1]
            ress = resolutions_list[min_level_id:max_level_id]
            mli = None
        else:
            offs, ress, mli = offsets_list, resolutions_list, min_level_id
        vb_words, vb_offs = vertex_bits if (vertex_bits is not None and binary_vxl is not None) else (None, None)
        if vb_offs is not None and scalar_window:
            vb_offs = vb_offs[min_level_id:max_level_id]
        vb = None if vb_words is None else (vb_words, vb_offs)
        if bits is not None and ste:
            # binarised table gathered from its bit plane (same values, 32x less table traffic)
            _backend.grid_encode_forward_bits(inputs, bits, offs, ress, outputs, N, num_dim,
                                              n_features, n_levels_calc, Rb, binary_vxl, mli, occ_sat,
                                              out_ld=ld, out_col=0, vertex_bits=vb)
        else:
            _backend.grid_encode_forward(inputs, embeddings, offs, ress, outputs, N, num_dim,
                                         n_features, n_levels_calc, 0, Rb, PV, None, binary_vxl, mli,
                                         ste_binary=ste, occ_sat=occ_sat, out_ld=ld, out_col=0, vertex_bits=vb)
        ctx.save_for_backward(inputs, embeddings, offs, ress, binary_vxl, mli, clip_count, occ_sat, vb_words, vb_offs)
        ctx.dims = (N, num_dim, n_features, n_levels_calc, Rb, ste)
        ctx.binned = binned if (binary_vxl is None and mli is None) else None
        # the caller thread's gradient sink (cnc_amd._gradsink), taken HERE: the backward below runs on autograd's
        # device thread, where the caller's thread-local is not visible
        ctx.sink = _gradsink.current()
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offs, ress, binary_vxl, mli, clip_count, occ_sat, vb_words, vb_offs = ctx.saved_tensors
        N, num_dim, n_features, n_levels_calc, Rb, ste = ctx.dims
        grad = grad.contiguous()                       # [N, L*F], read in place (grad_ld)
        # a gradient sink of the training step (cnc_amd._gradsink): the scatter adds straight into the step's buffer
        # for this table and autograd is handed nothing to add
        sink = ctx.sink
        sunk = None if sink is None else sink.table(embeddings)
        grad_embeddings = torch.zeros_like(embeddings) if sunk is None else sunk
        # Masked calls and per-point level windows are the context pass's: lattice vertices in hash-slot order, which
        # share cells inside a 1024-point block but never consecutively -> the cell-merging scatter
        # (csrc/grid_encode_cells.hip).  Its x-neighbour carry pays where the atomic requests are the call's bound — the
        # 3-D windows (1-3 vertices per cell) and the one-level vote tables — and costs where a cell holds dozens of
        # points (the planes' context levels): tools/replay_bwd_calls.py, one training step's calls one by one.
        cells = _CELL_MERGE and (binary_vxl is not None or mli is not None)
        _backend.grid_encode_backward(grad, inputs, embeddings, offs, ress, grad_embeddings, N,
                                      num_dim, n_features, n_levels_calc, 0, Rb, None, None,
                                      binary_vxl, mli, ste_binary=ste, ste_clip_count=clip_count,
                                      occ_sat=occ_sat, binned=ctx.binned,
                                      grad_ld=n_levels_calc * n_features, grad_col=0,
                                      vertex_bits=None if vb_words is None else (vb_words, vb_offs),
                                      cell_merge=cells, cell_carry=cells and (mli is not None or n_levels_calc == 1))
        return (None, grad_embeddings if sunk is None else None) + (None,) * 13


_CELL_MERGE = os.environ.get("CNC_CELL_MERGE", "1") != "0"      # measurement switch (tools/ab_train.py)

grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    def __init__(self, num_dim=3, n_features=2,
                 resolutions_list=(16, 23, 32, 46, 64, 92, 128, 184, 256, 368, 512, 736),
                 log2_hashmap_size=19, ste_binary=False, ste_multistep=False, add_noise=False, Q=1,
                 fused_ste=True, bitplane=True):
        super().__init__()
        resolutions_list = torch.as_tensor(np.asarray(resolutions_list)).to(torch.int)
        n_levels = resolutions_list.numel()
        self.num_dim = num_dim
        self.n_levels = n_levels
        self.n_features = n_features
        self.log2_hashmap_size = log2_hashmap_size
        self.output_dim = n_levels * n_features
        self.ste_binary = ste_binary
        self.ste_multistep = ste_multistep
        self.add_noise = add_noise
        self.Q = Q
        self.fused_ste = fused_ste
        # bit-plane gather for binarised tables (needs the fused STE path); the packed plane is
        # cached until the table is modified in place (optimizer step bumps Tensor._version)
        self.bitplane = bitplane and fused_ste
        # masked calls (binary_vxl): per-level vertex bit planes of the occupancy test (extension, same results)
        self.vertex_bits = os.environ.get("CNC_VERTEX_BITS", "1") == "1"
        self._bits = None
        self._bits_key = None
        self._bits_src = None
        self._clip_count = None
        self._sat = None
        self._sat_key = None
        self._vbits = None
        self._vbits_key = None

        # rows per level = min(2^log2T, R^D) rounded up to a multiple of 8 (ngp.py:197-210)
        self.max_params = 2 ** log2_hashmap_size
        offsets = [0]
        for R in resolutions_list.tolist():
            rows = min(self.max_params, R ** num_dim)
            offsets.append(offsets[-1] + int(np.ceil(rows / 8) * 8))
        # host copies: the binned-backward plan is made without reading device tables
        self._off_host = list(offsets)
        self._res_host = [int(r) for r in resolutions_list.tolist()]
        self.register_buffer("offsets_list", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.register_buffer("resolutions_list", resolutions_list)
        self.n_params = self.offsets_list[-1] * n_features
        self.params = nn.Parameter(torch.empty(offsets[-1], n_features))
        self.reset_parameters()
        self.n_output_dims = n_levels * n_features
        _caches.register(self)     # fused optimizers do not bump params._version

    def reset_parameters(self):
        # no_grad in-place write (not `.data`): bumps params._version, which keys the caches below
        with torch.no_grad():
            self.params.uniform_(-1e-4, 1e-4)
        self.invalidate_caches()

    def invalidate_caches(self):
        """The packed sign plane / clip counter are stale (call after writing `params` through `.data`; optimizer
        steps do it through cnc_amd._caches).  The buffers themselves are kept: `_bit_plane` repacks into them, so their
        addresses are constants of the run."""
        self._bits_key = self._bits_src = None

    def __repr__(self):
        return (f"GridEncoder: num_dim={self.num_dim} n_levels={self.n_levels} "
                f"n_features={self.n_features} resolutions={self.resolutions_list.tolist()} "
                f"log2_hashmap_size={self.log2_hashmap_size} params={tuple(self.params.shape)} "
                f"ste_binary={self.ste_binary}")

    def _bit_plane(self, params):
        """(uint8 sign plane, clip counter) of `params`, repacked only when the tensor changed.
        The counter (#entries with |p| > 1) lets backward skip the STE-mask gather when it is 0."""
        key = (params.data_ptr(), params._version, tuple(params.shape))
        if self._bits is None or self._bits_key != key:
            with torch.no_grad():
                # repacked IN PLACE when the table's shape is the one before: the plane's address is then a constant of
                # the run, which a captured graph of kernels that read it needs (cnc_amd._planes_graph).  Every reader
                # of the old contents has been joined to the caller's stream by then: the table itself has just been
                # rewritten by the optimizer on that stream.
                n_bytes = (params.shape[0] * params.shape[1] + 7) // 8
                same = (self._bits is not None and self._bits.device == params.device and self._bits.numel() == n_bytes
                        and self._clip_count is not None and self._clip_count.device == params.device)
                cc = self._clip_count if same else torch.empty(1, dtype=torch.int32, device=params.device)
                self._bits = _backend.pack_sign_bits(params.detach().contiguous(), self._bits if same else None, cc)
                self._clip_count = cc
            self._bits_key = key
            self._bits_src = (params,)   # keep the storage alive so (data_ptr, version) stays unique (a tuple:
                                         # a bare Parameter attribute would register as a module parameter)
        return self._bits, self._clip_count

    def _binned_plan(self, n_points, lo=0, hi=None, binary_vxl=None):
        """(n_binned, level_rows) for the levels [lo, hi) of this call, or None (see
        `gridencoder_backend.plan_binned_levels`)."""
        if binary_vxl is not None:
            return None
        hi = self.n_levels if hi is None else hi
        return _backend.plan_binned_levels(self._res_host[lo:hi], self._off_host[lo:hi + 1], self.num_dim,
                                           self.n_features, n_points)

    def _occ_sat(self, binary_vxl):
        """Summed-volume table of the occupancy grid handed to a masked call, rebuilt only when the
        grid tensor changes (the estimator replaces it every `step_update` steps)."""
        if binary_vxl is None:
            return None
        key = (binary_vxl.data_ptr(), binary_vxl._version, tuple(binary_vxl.shape))
        if self._sat is None or self._sat_key != key:
            with torch.no_grad():
                self._sat = _backend.occupancy_sat(binary_vxl)
            self._sat_key = key
            self._sat_src = (binary_vxl,)   # keep the storage alive so data_ptr stays unique
        return self._sat

    def _occ_vertex_bits(self, binary_vxl):
        """Per-level vertex bit planes of the occupancy mask for THIS encoder's resolutions (levels of at most
        2^26 vertices), rebuilt with the summed-volume table when the grid tensor changes: the masked kernels read
        one bit per corner instead of 2^D table entries (`gridencoder_backend.occupancy_vertex_bits`)."""
        if binary_vxl is None or not self.vertex_bits:
            return None
        key = (binary_vxl.data_ptr(), binary_vxl._version, tuple(binary_vxl.shape))
        if self._vbits is None or self._vbits_key != key:
            with torch.no_grad():
                self._vbits = _backend.occupancy_vertex_bits(binary_vxl.contiguous(), self._occ_sat(binary_vxl),
                                                             self._res_host)
            self._vbits_key = key
            self._vbits_src = (binary_vxl,)   # keep the storage alive so data_ptr stays unique (as `_occ_sat` does)
        return self._vbits

    # -- embeddings as the kernels should see them --------------------------------------------
    def _embeddings(self, params, test_phase):
        """Returns (table, ste_flag)."""
        if self.ste_binary:
            if self.fused_ste:
                return params, True
            return STE_binary.apply(params), False
        if self.add_noise and not test_phase:
            return params + (torch.rand_like(params) - 0.5) * (1 / self.Q), False
        if self.ste_multistep or (self.add_noise and test_phase):
            return STE_multistep.apply(params, self.Q), False
        return params, False

    def forward(self, inputs, min_level_id=None, max_level_id=None, test_phase=False,
                outspace_params=None, binary_vxl=None, PV=0):
        """inputs [..., num_dim] in [0,1] -> [..., L_calc * F] for levels [min_level_id, max_level_id)."""
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.num_dim)
        params = self.params if outspace_params is None else outspace_params
        embeddings, ste = self._embeddings(params, test_phase)
        bits, clip = self._bit_plane(params) if (ste and self.bitplane) else (None, None)
        min_level_id = 0 if min_level_id is None else max(min_level_id, 0)
        max_level_id = self.n_levels if max_level_id is None else min(max_level_id, self.n_levels)
        n_levels_calc = max_level_id - min_level_id
        outputs = grid_encode(inputs, embeddings, self.offsets_list, self.resolutions_list, False,
                              min_level_id, n_levels_calc, binary_vxl, PV, ste, bits, clip,
                              self._occ_sat(binary_vxl),
                              self._binned_plan(inputs.shape[0], min_level_id, max_level_id, binary_vxl),
                              self._occ_vertex_bits(binary_vxl))
        return outputs.view(prefix_shape + [n_levels_calc * self.n_features])

    def forward_diff_levels(self, inputs, min_level_id_list=None, n_levels_calc=1, test_phase=False,
                            outspace_params=None, binary_vxl=None, PV=0):
        """Per-point level window [min_level_id_list[i], +n_levels_calc) (ngp.py:265-297)."""
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.num_dim)
        params = self.params if outspace_params is None else outspace_params
        embeddings, ste = self._embeddings(params, test_phase)
        bits, clip = self._bit_plane(params) if (ste and self.bitplane) else (None, None)
        outputs = grid_encode(inputs, embeddings, self.offsets_list, self.resolutions_list, False,
                              min_level_id_list.contiguous(), n_levels_calc, binary_vxl, PV, ste, bits, clip,
                              self._occ_sat(binary_vxl), None, self._occ_vertex_bits(binary_vxl))
        return outputs.view(prefix_shape + [n_levels_calc * self.n_features])

    def forward_given_params(self, inputs, offsets_list, resolutions_list, outspace_params=None,
                             binary_vxl=None, PV=0):
        """One dense 2-D level described by the caller's tables; no STE (ngp.py:299-315)."""
        assert inputs.shape[-1] == 2
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, 2)
        outputs = grid_encode(inputs, outspace_params, offsets_list, resolutions_list, False, 0, 1,
                              binary_vxl, PV, False, None, None, self._occ_sat(binary_vxl))
        return outputs.view(prefix_shape + [self.n_features])
