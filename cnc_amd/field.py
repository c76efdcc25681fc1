"""Radiance field on 3-D + three 2-D binarised hash grids — host mirror of
examples/radiance_fields/ngp.py:318-645 (`trunc_exp`, `contract_to_unisphere`,
`NGPRadianceField_mygrid_2D3D`, `Embedder`/`get_embedder`, `compose_3D_2D_embed`).

Same constructor arguments, sub-module names (`mlp_base.encoding_xyz/xy/xz/yz`, `mlp_base.network`,
`mlp_head`, `direction_encoding`) and methods (`query_density`, `_query_rgb`, `forward`,
`update_embedding_params`), so state dicts and the reference drivers line up.

The one third-party piece, tiny-cuda-nn's `SphericalHarmonics degree 4` direction encoding
(ngp.py:412-425), is restated in closed form (`SHEncoding`): tcnn is not part of the reference tree
(unpinned git install, README.md:56) and not installable here, so its numerics — including the fp16
output rounding tcnn applies on NVIDIA — are PARITY UNPINNED; `fp16_round=True` emulates the rounding.
The dense layers are plain `nn.Linear` (hipBLASLt GEMMs); a fused MFMA field kernel is SURVEY §8f-3.
"""
from __future__ import annotations

import os
from typing import Callable, List, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from .gridencoder import GridEncoder
from .mlp import Linear, linear_fn, run_layers


class _TruncExp(Function):
    """exp with the gradient clamped at x<=15 (ngp.py:318-334)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(torch.clamp(x, max=15))


trunc_exp = _TruncExp.apply


def contract_to_unisphere(x, aabb, ord: Union[str, int] = 2, eps: float = 1e-6, derivative: bool = False):
    """mip-NeRF-360 contraction of unbounded space into [0,1]^3 (ngp.py:337-361)."""
    aabb_min, aabb_max = torch.split(aabb, 3, dim=-1)
    x = (x - aabb_min) / (aabb_max - aabb_min)
    x = x * 2 - 1
    mag = torch.linalg.norm(x, ord=ord, dim=-1, keepdim=True)
    mask = mag.squeeze(-1) > 1
    if derivative:
        dev = (2 * mag - 1) / mag ** 2 + 2 * x ** 2 * (1 / mag ** 3 - (2 * mag - 1) / mag ** 4)
        dev[~mask] = 1.0
        return torch.clamp(dev, min=eps)
    x[mask] = (2 - 1 / mag[mask]) * (x[mask] / mag[mask])
    return x / 4 + 0.5


class SHEncoding(nn.Module):
    """Real spherical harmonics up to degree 4 (16 values) of a direction given in [0,1]^3
    (the reference feeds (dir+1)/2, ngp.py:540-541; tcnn maps back to [-1,1] internally)."""

    n_output_dims = 16

    def __init__(self, fp16_round: bool = False):
        super().__init__()
        self.fp16_round = fp16_round

    def forward(self, d01):
        d = d01 * 2.0 - 1.0
        x, y, z = d[..., 0], d[..., 1], d[..., 2]
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out = torch.stack([
            torch.full_like(x, 0.28209479177387814),
            -0.48860251190291987 * y,
            0.48860251190291987 * z,
            -0.48860251190291987 * x,
            1.0925484305920792 * xy,
            -1.0925484305920792 * yz,
            0.94617469575755997 * zz - 0.31539156525251999,
            -1.0925484305920792 * xz,
            0.54627421529603959 * xx - 0.54627421529603959 * yy,
            0.59004358992664352 * y * (-3.0 * xx + yy),
            2.8906114426405538 * xy * z,
            0.45704579946446572 * y * (1.0 - 5.0 * zz),
            0.3731763325901154 * z * (5.0 * zz - 3.0),
            0.45704579946446572 * x * (1.0 - 5.0 * zz),
            1.4453057213202769 * z * (xx - yy),
            0.59004358992664352 * x * (-xx + 3.0 * yy),
        ], dim=-1)
        if self.fp16_round:
            out = out.to(torch.float16).to(d01.dtype)
        return out


class Embedder:
    """Sinusoidal positional embedding: x, then sin/cos(2^k x) for k < num_freqs (ngp.py:569-599)."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        self.include_input = kwargs["include_input"]
        n = kwargs["num_freqs"]
        if kwargs["log_sampling"]:
            self.freq_bands = 2.0 ** torch.linspace(0.0, kwargs["max_freq_log2"], steps=n)
        else:
            self.freq_bands = torch.linspace(2.0 ** 0.0, 2.0 ** kwargs["max_freq_log2"], steps=n)
        self.periodic_fns = kwargs["periodic_fns"]
        self.out_dim = d * (int(self.include_input) + n * len(self.periodic_fns))

    def embed(self, inputs):
        parts = [inputs] if self.include_input else []
        for freq in self.freq_bands:
            for fn in self.periodic_fns:
                parts.append(fn(inputs * freq))
        return torch.cat(parts, -1)


def get_embedder(multires, i=0, with_freqs=False):
    if i == -1:
        return nn.Identity(), 3
    e = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                 log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    fn = (lambda x, eo=e: eo.embed(x))
    return (fn, e.out_dim, e.freq_bands) if with_freqs else (fn, e.out_dim)


class _FusedFeatures(Function):
    """The four grid encoders writing point-major into ONE [N, ld] feature matrix (the base MLP's
    input), and their backward reading the MLP's input gradient in place.

    Same values as `torch.cat([enc_xyz(x), enc_xy(xy), enc_xz(xz), enc_yz(yz), embed(x)], -1)`
    (ngp.py:631-642) without the four [L,N,F]->[N,L*F] permutes, the cat, and the four
    permute+contiguous copies of the gradient.  Only for the CNC configuration: every encoder
    binarised with the fused STE and a bit plane."""

    @staticmethod
    def forward(ctx, x, owner, rows, *params):
        """`rows` >= N: the matrix gets that many rows, those behind the N encoded ones are zero (a bucketed row
        count for the GEMMs that follow: NGPRadianceField_mygrid_2D3D._bucket_rows); the backward reads the
        gradient of the first N rows only."""
        from .backends import gridencoder_backend as be
        encs = owner._encoders()
        N = x.shape[0]
        ld, cols = owner._layout()
        feat = torch.empty(max(N, rows or 0), ld, device=x.device, dtype=torch.float32)
        if feat.shape[0] > N:
            feat[N:].zero_()
        # plane coordinates by slicing (an index list would cost a host->device copy and a sync each)
        xs = (x.contiguous(), x[:, :2].contiguous(), x[:, ::2].contiguous(), x[:, 1:].contiguous())
        saved = []
        for enc, p, xi, col in zip(encs, params, xs, cols):
            bits, clip = enc._bit_plane(p)
            be.grid_encode_forward_bits(xi, bits, enc.offsets_list, enc.resolutions_list, feat, N,
                                        enc.num_dim, enc.n_features, enc.n_levels, 128, None, None,
                                        None, out_ld=ld, out_col=col)
            saved.append(clip)
        c0 = cols[4]
        if owner.embed_fn is not None:
            # x | sin(2^k x) | cos(2^k x), k = 0..9 (ngp.py:583-599) and the zero padding behind it: one kernel
            from . import _lib
            if owner._freqs.device != x.device:
                owner._freqs = owner._freqs.to(x.device)
            fr = owner._freqs.to(torch.float32).contiguous()
            _lib.check(_lib.lib().cnc_field_sinusoid(xs[0].data_ptr(), fr.data_ptr(), fr.numel(), N, feat.data_ptr(), ld,
                                                     c0, _lib.stream(x.device)), "field_sinusoid")
        elif c0 < ld:
            feat[:N, c0:].zero_()
        ctx.save_for_backward(*xs, *params, *saved)
        ctx.owner = owner
        from . import _gradsink
        ctx.sink = _gradsink.current()     # the caller thread's sink; the backward runs on autograd's own thread
        return feat

    @staticmethod
    def backward(ctx, grad):
        t = ctx.saved_tensors
        xs, params, clips = t[0:4], t[4:8], t[8:12]
        # rows of padding behind the N samples (`rows`) carry none
        grads = _FusedFeatures.scatter(ctx.owner, grad.contiguous(), xs, params, clips, xs[0].shape[0], ctx.sink)
        return (None, None, None, *grads)

    @staticmethod
    def scatter(owner, grad, xs, params, clips, N, sink):
        """The four encoders' backward on the gradient of the feature matrix `grad` [>= N, ld] (read in place: the
        encoders' columns of the first N rows): per table the gradient tensor, or None when it went into `sink` (the
        training step's per-table gradient buffers, cnc_amd._gradsink)."""
        from .backends import gridencoder_backend as be
        ld, cols = owner._layout()
        if grad.shape[1] != ld:
            raise RuntimeError("feature gradient: wrong row length")
        grads = []
        for enc, p, xi, clip, col in zip(owner._encoders(), params, xs, clips, cols):
            sunk = None if sink is None else sink.table(p)
            g = torch.zeros_like(p) if sunk is None else sunk
            be.grid_encode_backward(grad, xi, p, enc.offsets_list, enc.resolutions_list, g, N,
                                    enc.num_dim, enc.n_features, enc.n_levels, 0, 128, None, None,
                                    None, None, ste_binary=True, ste_clip_count=clip,
                                    grad_ld=ld, grad_col=col, binned=enc._binned_plan(N),
                                    # inside a training step (a sink is active) the scatter stays on the caller's stream: the
                                    # step runs three streams of its own, and the library's two side streams next to them
                                    # made streams share hardware queues — which ones depended on the order the process had
                                    # created its streams in: 7.3 ms per step, or 9.7 after a bench frame had run first
                                    overlap_streams=sink is None)
            grads.append(g if sunk is None else None)
        return grads


class compose_3D_2D_embed(nn.Module):
    """[3-D grid | xy | xz | yz plane grids | 63 sinusoid features] -> base MLP (ngp.py:620-645).

    `fused=True` (extension): when all four encoders are binarised (the CNC configuration) they
    write straight into the MLP's input matrix — see `_FusedFeatures`."""

    def __init__(self, encoding_xyz, encoding_xy, encoding_xz, encoding_yz, embed_fn, network, sin_encode=False,
                 fused=True, freq_bands=None):
        super().__init__()
        self.encoding_xyz = encoding_xyz
        self.encoding_xy = encoding_xy
        self.encoding_xz = encoding_xz
        self.encoding_yz = encoding_yz
        self.embed_fn = embed_fn
        self.network = network
        self.fused = fused
        self._freqs = freq_bands

    def _encoders(self):
        return (self.encoding_xyz, self.encoding_xy, self.encoding_xz, self.encoding_yz)

    def _layout(self):
        """(row length padded to 4 floats, first column of [xyz, xy, xz, yz, sinusoid])."""
        cols, c = [], 0
        for e in self._encoders():
            cols.append(c)
            c += e.n_output_dims
        cols.append(c)
        if self.embed_fn is not None:
            c += 3 + 6 * self._freqs.numel()
        return (c + 3) // 4 * 4, cols

    def _can_fuse(self, x):
        return (self.fused and x.is_cuda and x.dtype == torch.float32
                and (self.embed_fn is None or self._freqs is not None)
                and isinstance(self.network[0], nn.Linear)
                and all(e.ste_binary and e.fused_ste and e.bitplane for e in self._encoders()))

    def features_fused(self, x, rows=None):
        """[N, ld] feature matrix; columns [width, ld) are zero.  `rows` > N appends zero rows (_FusedFeatures)."""
        return _FusedFeatures.apply(x, self, rows, *(e.params for e in self._encoders()))

    def features(self, x):
        xs, ys, zs = torch.chunk(x, 3, dim=-1)
        parts = [self.encoding_xyz(x),
                 self.encoding_xy(torch.cat([xs, ys], dim=-1)),
                 self.encoding_xz(torch.cat([xs, zs], dim=-1)),
                 self.encoding_yz(torch.cat([ys, zs], dim=-1))]
        if self.embed_fn is not None:
            parts.append(self.embed_fn(x))
        return torch.cat(parts, dim=-1)

    def forward(self, x, last_rows=None, rows=None):
        """`last_rows` = k: only the first k outputs of the network are computed.  `rows` > N (fused path only):
        the network runs on that many rows, the ones behind the N samples being zero features."""
        if self._can_fuse(x):
            feat = self.features_fused(x, rows)
            first = self.network[0]
            pad = feat.shape[1] - first.in_features
            w = F.pad(first.weight, (0, pad)) if pad else first.weight
            return run_layers(self.network, feat, first_weight=w, last_rows=last_rows)
        if rows is not None and rows > x.shape[0]:
            raise ValueError("rows: only on the fused feature path")
        return run_layers(self.network, self.features(x), last_rows=last_rows)


def _default_density_activation(x):
    return trunc_exp(x - 1)


class _FieldPost(Function):
    """base MLP output [N, 1 + geo] -> (density [N, 1], head input [N, ld] = [SH4(dirs) | geo | 0]) in one
    kernel, and one kernel back (cnc_amd/csrc/field_glue.hip): replaces split / trunc_exp / selector
    multiply / SH encoding (~45 elementwise launches) / cat of ngp.py:527-547."""

    @staticmethod
    def forward(ctx, base_out, selector, dirs, geo, sh_fp16=False):
        from . import _lib
        ctx.set_materialize_grads(False)
        base_out = base_out.contiguous()
        N, ldb = base_out.shape
        dev = base_out.device
        density = torch.empty((N, 1), dtype=torch.float32, device=dev)
        ld = (16 + geo + 3) // 4 * 4
        head_in = torch.empty((N, ld), dtype=torch.float32, device=dev) if dirs is not None else None
        if dirs is not None:
            dirs = dirs.contiguous()
        _lib.check(_lib.lib().cnc_field_post(base_out.data_ptr(), ldb, geo, _lib.ptr(selector), _lib.ptr(dirs), N,
                                             density.data_ptr(), _lib.ptr(head_in), ld,
                                             _lib.CNC_FIELD_SH_FP16 if sh_fp16 else 0, _lib.stream(dev)), "field_post")
        ctx.save_for_backward(base_out, selector)
        ctx.dims = (N, ldb, geo, ld)
        if head_in is None:
            return density, None
        return density, head_in

    @staticmethod
    def backward(ctx, g_density, g_head):
        from . import _lib
        base_out, selector = ctx.saved_tensors
        N, ldb, geo, ld = ctx.dims
        g_base = torch.empty((N, ldb), dtype=torch.float32, device=base_out.device)
        gd = None if g_density is None else g_density.contiguous()
        gh = None if g_head is None else g_head.contiguous()
        _lib.check(_lib.lib().cnc_field_post_backward(base_out.data_ptr(), ldb, geo, _lib.ptr(selector), _lib.ptr(gd),
                                                      _lib.ptr(gh), ld, N, g_base.data_ptr(),
                                                      _lib.stream(base_out.device)), "field_post_backward")
        return g_base, None, None, None, None


class FusedFieldForward:
    """Gradient-free `positions -> density [-> rgb]` through ONE kernel (cnc_field_fused_forward,
    cnc_amd/csrc/field_fused{,2}.hip): the four encoders' features are computed into LDS and consumed there by MFMA,
    so the [N, 255] feature matrix, the activations and the head input never touch HBM (ngp.py:506-547).

    Keeps the five layers' weights packed in MFMA fragment order (cnc_field_pack_all: one launch), repacked when a
    parameter changed (`_version`, plus the global optimizer post-step hook: fused Adam does not bump the counter).
    The default three-product fp16 form carries a range guard: a call in which a hidden activation or a weight left
    fp16's range is recomputed by the exact-fp32 kernel, decided on the device (include/cnc_hip.h)."""

    def __init__(self, field: "NGPRadianceField_mygrid_2D3D"):
        from . import _caches
        self.field = field
        self._key = None
        self._buffers = None
        self._units = None
        self._pack_id = 0          # cnc_field_pack_t.pack_id / cnc_fused_field_t.call_id: the range guard's stamps
        self._call_id = 0
        _caches.register(self)

    def invalidate_caches(self):
        self._key = None

    @staticmethod
    def supported(field) -> bool:
        mb = field.mlp_base
        encs = mb._encoders()
        F_ = encs[0].n_features
        H = mb.network[0].out_features
        geo = field.geo_feat_dim
        nt2 = 3 if H == 160 else 2
        return (field.use_viewdirs and geo > 0 and not field.unbounded and field.num_dim == 3
                and field.density_activation is _default_density_activation
                and mb.embed_fn is not None and mb._freqs is not None
                and all(e.ste_binary and e.fused_ste and e.bitplane and e.n_features == F_ for e in encs)
                and encs[0].num_dim == 3 and all(e.num_dim == 2 for e in encs[1:])
                and len({e.n_levels for e in encs[1:]}) == 1
                and F_ in (2, 4, 8) and H in (64, 160) and 1 + geo <= 32 * nt2 and (17 + geo + 31) // 32 * 32 <= H
                and len(mb.network) == 3 and len(field.mlp_head) == 5
                and all(l.out_features == H for l in (field.mlp_head[0], field.mlp_head[2]))
                # the kernel derives the first layer's K from the unit table and the frequency count
                and mb.network[0].in_features == sum(e.n_levels for e in encs) * F_ + 3 + 6 * mb._freqs.numel()
                and mb.network[2].out_features == 1 + geo and field.mlp_head[0].in_features == 16 + geo
                and field.mlp_head[4].out_features == 3
                # every level dense (R^D rows fit) or hashed into a power-of-two table: what the kernels' index
                # arithmetic assumes (no modulo, every index in range; GridEncoder makes nothing else)
                and all(e._res_host[l] ** e.num_dim <= e._off_host[l + 1] - e._off_host[l]
                        or ((e._off_host[l + 1] - e._off_host[l]) & (e._off_host[l + 1] - e._off_host[l] - 1)) == 0
                        for e in encs for l in range(e.n_levels)))

    def _layers(self):
        f = self.field
        return [f.mlp_base.network[0], f.mlp_base.network[2], f.mlp_head[0], f.mlp_head[2], f.mlp_head[4]]

    def _pack(self, dev):
        """The five layers in every fragment order the kernels read — fp32 (exact form and the range guard's fallback),
        fp16 hi / lo for the one-wave and for the two-wave kernels — in ONE launch (cnc_field_pack_all)."""
        from . import _lib
        layers = self._layers()
        key = tuple((l.weight.data_ptr(), l.weight._version, l.bias.data_ptr(), l.bias._version) for l in layers) + (str(dev),)
        if key == self._key:
            return self._buffers
        f = self.field
        H, geo = layers[0].out_features, f.geo_feat_dim
        T, T2 = H // 32, (3 if H == 160 else 2)
        K0 = layers[0].in_features
        r32 = lambda k: (k + 31) // 32
        # per layer: (32-column tiles, K-steps of 8 [fp32 MFMA], K-steps of 16 [32x32x16 halves],
        #             16-column blocks, K-steps of 32 [16x16x32 halves: the two-wave kernels])
        shapes = [(T, r32(K0) * 4, r32(K0) * 2, H // 16, r32(K0)),
                  (T2, H // 8, H // 16, 5 if H == 160 else 4, H // 32),
                  (T, (16 + geo + 7) // 8, (16 + geo + 15) // 16, H // 16, r32(17 + geo)),     # Wq16: a zero column at k = 16
                  (T, H // 8, H // 16, H // 16, H // 32),
                  (1, H // 8, H // 16, 1, H // 32)]
        if self._buffers is None or self._buffers["dev"] != str(dev):
            self._buffers = {"dev": str(dev),
                             "w": [torch.empty(nk * nt * 256, dtype=torch.float32, device=dev) for nt, nk, _, _, _ in shapes],
                             "w16": [torch.empty(nk16 * nt * 1024, dtype=torch.float16, device=dev) for nt, _, nk16, _, _ in shapes],
                             "wq16": [torch.empty(nk32 * ncb * 1024, dtype=torch.float16, device=dev) for _, _, _, ncb, nk32 in shapes],
                             "b": [torch.empty(nt * 32, dtype=torch.float32, device=dev) for nt, _, _, _, _ in shapes],
                             "row0": torch.empty(H, dtype=torch.float32, device=dev),
                             # the fp16 range guard's words (cnc_hip.h): zero once, stamped with ids afterwards
                             # (64 words: a -DCNC_W2_PROF build adds its phase clocks behind the guard's eight)
                             "guard": torch.zeros(64, dtype=torch.int32, device=dev)}
        buf = self._buffers
        self._pack_id += 1
        d = _lib.FieldPack()
        keep = []
        for k, (l, (nt, nk, nk16, ncb, nk32)) in enumerate(zip(layers, shapes)):
            w, b = l.weight.detach(), l.bias.detach()
            if not w.is_contiguous():
                w = w.contiguous()
            keep += [w, b]
            L = d.layer[k]
            L.W, L.b, L.H, L.K, L.ldw = w.data_ptr(), b.data_ptr(), w.shape[0], w.shape[1], w.stride(0)
            L.n_tiles, L.n_ksteps, L.n_ksteps16, L.n_colblocks, L.n_ksteps32 = nt, nk, nk16, ncb, nk32
            L.Wp, L.Bp = buf["w"][k].data_ptr(), buf["b"][k].data_ptr()
            L.Wp16, L.Wq16 = buf["w16"][k].data_ptr(), buf["wq16"][k].data_ptr()
            L.k_gap = 16 if k == 2 else 0          # head.0: [SH4 | raw density (zero weights) | geo], field_fused2.hip
        d.row0, d.row0_len = buf["row0"].data_ptr(), H
        d.guard, d.pack_id = buf["guard"].data_ptr(), self._pack_id
        import ctypes
        _lib.check(_lib.lib().cnc_field_pack_all(ctypes.byref(d), _lib.stream(dev)), "field_pack_all")
        self._key = key
        self._src = [(l.weight, l.bias) for l in layers]        # keep (data_ptr, version) unique while cached
        return buf

    def _descriptor(self, dev, want_rgb: bool, params=None):
        """(cnc_fused_field_t filled in for a call on `dev`, the tensors it points to, the encoders' clip counts).
        `params`: the four tables to read (default: the encoders' own)."""
        from . import _lib
        f = self.field
        mb = f.mlp_base
        buf = self._pack(dev)
        st = _lib.FusedField()
        aabb = f.aabb if f.aabb.is_contiguous() else f.aabb.contiguous()
        st.aabb = aabb.data_ptr()
        keep, clips = [aabb], []
        for k, e in enumerate(mb._encoders()):
            bits, clip = e._bit_plane(e.params if params is None else params[k])
            keep.append(bits)
            clips.append(clip)
            st.bits[k], st.offsets[k], st.resolutions[k] = bits.data_ptr(), e.offsets_list.data_ptr(), e.resolutions_list.data_ptr()
            st.n_levels[k] = e.n_levels
        if mb._freqs.device != dev or mb._freqs.dtype != torch.float32:
            mb._freqs = mb._freqs.to(device=dev, dtype=torch.float32).contiguous()
        st.freqs, st.n_freqs = mb._freqs.data_ptr(), mb._freqs.numel()
        for k in range(5):
            st.packed_weights[k], st.packed_biases[k] = buf["w"][k].data_ptr(), buf["b"][k].data_ptr()
            st.packed_weights16[k] = buf["w16"][k].data_ptr()
            st.packed_weights16q[k] = buf["wq16"][k].data_ptr()
        st.w2_row0 = buf["row0"].data_ptr()
        if self._units is None or self._units.device != dev:
            # the level tables as one record per (encoder, level) unit, in feature-row order (static per model)
            recs = []
            for k, e in enumerate(mb._encoders()):
                for l in range(e.n_levels):
                    recs.append([e._off_host[l], e._off_host[l + 1] - e._off_host[l], e._res_host[l], k])
            self._units = torch.tensor(recs, dtype=torch.int32).to(dev).contiguous()
        st.units = self._units.data_ptr()
        st.n_features, st.n_neurons, st.geo_feat_dim = mb.encoding_xyz.n_features, mb.network[0].out_features, f.geo_feat_dim
        flags = _lib.CNC_FIELD_SH_FP16 if f.sh_fp16_round else 0
        if f.fused_field_precision == "f16x3":
            flags |= _lib.CNC_FIELD_MFMA_F16X3
            if f.fused_field_kernel == "w2":
                # two cooperating waves per tile; the density-only kernel fits four waves per SIMD, the colour kernel three
                flags |= _lib.CNC_FIELD_TWO_WAVES
                waves = f.fused_field_waves or (3 if want_rgb else 4)
                if waves >= 4:
                    flags |= _lib.CNC_FIELD_WAVES4
        st.flags = flags
        self._call_id = self._call_id % 0xFFFFFFF0 + 1
        st.guard, st.call_id, st.pack_id = buf["guard"].data_ptr(), self._call_id, self._pack_id
        return st, keep, clips

    @torch.no_grad()
    def __call__(self, positions: torch.Tensor, directions=None, debug_features=None, n_rows_dev=None):
        """density [N, 1] (and rgb [N, 3] when `directions` is given) of world positions [N, 3].  `debug_features` (test
        hook, density-only two-wave calls): a float32 [N, >= roundup32(K0)] tensor that receives the first layer's input
        rows as the kernel computed them.  `n_rows_dev` (an int64 tensor of one element on the device): only the first
        min(N, n_rows_dev) rows are evaluated — N is then the capacity of `positions`, the rows behind the count are
        neither read nor written (cnc_fused_field_t.n_rows_dev: the sampler's depth windows)."""
        from . import _lib
        x = positions.reshape(-1, 3)
        if x.dtype != torch.float32 or not x.is_cuda:
            raise RuntimeError("FusedFieldForward: positions must be a float32 CUDA tensor")
        x = x.contiguous()
        N, dev = x.shape[0], x.device
        d = None
        if directions is not None:
            d = directions.reshape(-1, 3).to(torch.float32).contiguous()
        st, keep, _ = self._descriptor(dev, d is not None)
        if debug_features is not None:
            assert debug_features.dtype == torch.float32 and debug_features.is_contiguous() and debug_features.shape[0] == N
            st.debug_features, st.debug_ld = debug_features.data_ptr(), debug_features.shape[1]
        if n_rows_dev is not None:
            if n_rows_dev.dtype != torch.int64 or n_rows_dev.device != dev or n_rows_dev.numel() != 1:
                raise RuntimeError("FusedFieldForward: n_rows_dev must be one int64 on the positions' device")
            st.n_rows_dev = n_rows_dev.data_ptr()
        density = torch.empty((N, 1), dtype=torch.float32, device=dev)
        rgb = torch.empty((N, 3), dtype=torch.float32, device=dev) if d is not None else None
        import ctypes
        _lib.check(_lib.lib().cnc_field_fused_forward(ctypes.byref(st), x.data_ptr(), _lib.ptr(d), N, density.data_ptr(),
                                                      _lib.ptr(rgb), _lib.stream(dev)), "field_fused_forward")
        return (density, rgb) if d is not None else density

    @torch.no_grad()
    def save_forward(self, positions: torch.Tensor, directions: torch.Tensor, rows: int, params=None):
        """The gradient pass's forward (cnc_field_save_t): rgb [rows, 3], density [rows, 1] of the N <= `rows` world
        positions — rows behind them are evaluated as points outside the box — and, in a dict, everything the backward
        reads, written by the same kernel: feat, h1, h3, h4, head_in, raw, selector, xyz / xy / xz / yz, + the encoders'
        clip counts.  Needs the two-wave three-product kernel (the default)."""
        from . import _lib
        f = self.field
        if f.fused_field_precision != "f16x3" or f.fused_field_kernel != "w2":
            raise RuntimeError("save_forward: the two-wave fp16 kernel only")
        x = positions.reshape(-1, 3).contiguous()
        d = directions.reshape(-1, 3).contiguous()
        if x.dtype != torch.float32 or d.dtype != torch.float32 or not x.is_cuda or x.shape != d.shape:
            raise RuntimeError("save_forward: float32 CUDA positions and directions of one shape")
        N, dev = x.shape[0], x.device
        Np = max(int(rows), N)
        st, keep, clips = self._descriptor(dev, True, params)
        mb = f.mlp_base
        H, geo = mb.network[0].out_features, f.geo_feat_dim
        ld = (mb.network[0].in_features + 31) // 32 * 32
        ldh = (17 + geo + 31) // 32 * 32
        new = lambda *shape, dtype=torch.float32: torch.empty(shape, dtype=dtype, device=dev)
        out = {"feat": new(Np, ld), "h1": new(Np, H), "h3": new(Np, H), "h4": new(Np, H), "head_in": new(Np, ldh),
               "raw": new(Np, 1), "selector": new(Np, dtype=torch.uint8), "xyz": new(Np, 3), "xy": new(Np, 2),
               "xz": new(Np, 2), "yz": new(Np, 2)}
        sv = st.save
        sv.feat, sv.ld_feat, sv.h1, sv.h3, sv.h4 = out["feat"].data_ptr(), ld, out["h1"].data_ptr(), out["h3"].data_ptr(), out["h4"].data_ptr()
        sv.head_in, sv.ld_head, sv.raw, sv.selector = out["head_in"].data_ptr(), ldh, out["raw"].data_ptr(), out["selector"].data_ptr()
        sv.xyz, sv.xy, sv.xz, sv.yz, sv.n_live = out["xyz"].data_ptr(), out["xy"].data_ptr(), out["xz"].data_ptr(), out["yz"].data_ptr(), N
        density, rgb = new(Np, 1), new(Np, 3)
        import ctypes
        _lib.check(_lib.lib().cnc_field_fused_forward(ctypes.byref(st), x.data_ptr(), d.data_ptr(), Np, density.data_ptr(),
                                                      rgb.data_ptr(), _lib.stream(dev)), "field_fused_forward(save)")
        self._train_calls = True
        out["clips"] = clips
        return rgb, density, out

    def guard_words(self):
        """The range guard's words on the host (a synchronisation): [last call that saturated / overflowed, layer flags]."""
        return None if self._buffers is None else self._buffers["guard"][:6].tolist()

    def range_guard_fired(self) -> bool:
        """True when the LAST call left fp16's range and was recomputed by the exact kernel (reads one word back: a
        host synchronisation — for tests and diagnostics, never on the hot path)."""
        return self._buffers is not None and int(self._buffers["guard"][0].item()) == self._call_id


class _FieldChain(Function):
    """base MLP -> density activation / head input -> head MLP -> sigmoid on a [Np, ld] feature matrix, with the WHOLE
    input-gradient chain of its backward as one kernel (cnc_field_backward_chain, cnc_amd/csrc/field_bwd.hip).

    Forward: the library GEMMs with bias + ReLU epilogues and the fused post kernel, exactly the ops of the layer-by-layer
    path (`run_layers`, `_FieldPost`) — same values.  Backward: autograd through ngp.py:506-547 from (d rgb, d density)
    to the gradient of the feature matrix's encoder columns in one launch — sigmoid', Linear^T, ReLU', Linear^T, ReLU',
    Linear^T, geo split + trunc_exp's clamped derivative, Linear^T, ReLU', Linear^T — which also leaves the gradients
    with respect to every Linear's output in HBM for the five weight gradients (split-K batched GEMMs, as before) and
    bias gradients.  Replaces five `g @ W` GEMMs, three ReLU-backward passes, the post kernel's backward, the sigmoid's
    backward and the slices between them."""

    @staticmethod
    def forward(ctx, feat, selector, dirs, field, W1, b1, W2, b2, W3, b3, W4, b4, W5, b5):
        from . import _lib
        geo, Np, ld = field.geo_feat_dim, feat.shape[0], feat.shape[1]
        K0 = W1.shape[1]
        W1p = F.pad(W1, (0, ld - K0)) if ld != K0 else W1
        h1 = torch._addmm_activation(b1, feat, W1p.t())
        base_out = F.linear(h1, W2, b2)
        density = torch.empty((Np, 1), dtype=torch.float32, device=feat.device)
        ld_h = (16 + geo + 3) // 4 * 4
        head_in = torch.empty((Np, ld_h), dtype=torch.float32, device=feat.device)
        dirs = dirs.contiguous()
        _lib.check(_lib.lib().cnc_field_post(base_out.data_ptr(), base_out.shape[1], geo, _lib.ptr(selector), dirs.data_ptr(), Np,
                                             density.data_ptr(), head_in.data_ptr(), ld_h,
                                             _lib.CNC_FIELD_SH_FP16 if field.sh_fp16_round else 0, _lib.stream(feat.device)),
                   "field_post")
        W3p = F.pad(W3, (0, ld_h - W3.shape[1])) if ld_h != W3.shape[1] else W3
        h3 = torch._addmm_activation(b3, head_in, W3p.t())
        h4 = torch._addmm_activation(b4, h3, W4.t())
        rgb = torch.sigmoid(F.linear(h4, W5, b5))
        ctx.save_for_backward(feat, selector, h1, base_out, head_in, h3, h4, rgb, W1, W2, W3, W4, W5)
        ctx.field = field
        return rgb, density

    @staticmethod
    def backward(ctx, g_rgb, g_density):
        feat, selector, h1, base_out, head_in, h3, h4, rgb, W1, W2, W3, W4, W5 = ctx.saved_tensors
        need = ctx.needs_input_grad
        dX, layer_grads = _FieldChain.chain(ctx.field, g_rgb, g_density, feat, selector, h1, base_out, head_in, h3, h4, rgb,
                                            (W1, W2, W3, W4, W5), need[4:14], ctx.field.mlp_base._layout()[0], gap=False)
        return (dX if need[0] else None, None, None, None, *layer_grads)

    @staticmethod
    def chain(field, g_rgb, g_density, feat, selector, h1, base_out, head_in, h3, h4, rgb, Ws, need, ld_x, gap):
        """(dX [Np, ld_x]: the gradient of the feature matrix's encoder columns; [dW1, db1, ..., dW5, db5], None where
        `need` says so) from the gradients of rgb / density and what the forward kept.  `base_out`: [Np, >= 1], column 0
        = the raw density.  `gap`: `head_in` is in the fused kernel's layout [SH4 | 0 | geo] (cnc_field_save_t)."""
        from . import _lib
        from .mlp import splitk_weight_grad
        W1, W2, W3, W4, W5 = Ws
        dev, Np = feat.device, feat.shape[0]
        H, geo, K0 = W1.shape[0], field.geo_feat_dim, W1.shape[1]
        n_enc = sum(e.n_output_dims for e in field.mlp_base._encoders())
        wt = field._chain_weights_t(W1, W2, W3, W4, W5, n_enc)
        ld2 = (1 + geo + 3) // 4 * 4
        G5 = torch.empty((Np, 4), dtype=torch.float32, device=dev)
        G4, G3, G1 = (torch.empty((Np, H), dtype=torch.float32, device=dev) for _ in range(3))
        G2 = torch.empty((Np, ld2), dtype=torch.float32, device=dev)
        dX = torch.empty((Np, ld_x), dtype=torch.float32, device=dev)     # only the encoder columns are written — and read
        st = _lib.FieldBwd()
        st.N, st.n_neurons, st.n_features, st.n_enc_columns, st.geo_feat_dim = Np, H, field.mlp_base.encoding_xyz.n_features, n_enc, geo
        st.ld_base, st.ld_g2, st.ld_x = base_out.shape[1], ld2, ld_x
        gr = None if g_rgb is None else g_rgb.contiguous()
        gd = None if g_density is None else g_density.contiguous()
        st.grad_rgb, st.grad_density, st.rgb, st.base_out = _lib.ptr(gr), _lib.ptr(gd), rgb.data_ptr(), base_out.data_ptr()
        st.selector, st.h1, st.h3, st.h4 = selector.data_ptr(), h1.data_ptr(), h3.data_ptr(), h4.data_ptr()
        for k in range(5):
            st.packed_weights_t[k] = wt[k].data_ptr()
        st.G5, st.G4, st.G3, st.G2, st.G1, st.dX = (t.data_ptr() for t in (G5, G4, G3, G2, G1, dX))
        # column sums of G4 | G3 | G1 | G2 | G5 (the bias gradients), + 8 words: the largest |G_l| (the weight gradients' scales)
        bsum = torch.zeros(3 * H + 84 + 8, dtype=torch.float32, device=dev)
        st.bias_grads = bsum.data_ptr()
        st.g_max = bsum.data_ptr() + (3 * H + 84) * 4
        gb_of = (bsum[2 * H:3 * H], bsum[3 * H:3 * H + 1 + geo], bsum[H:2 * H], bsum[:H], bsum[3 * H + 80:3 * H + 83])
        import ctypes
        _lib.check(_lib.lib().cnc_field_backward_chain(ctypes.byref(st), _lib.stream(dev)), "field_backward_chain")
        pairs = ((G1, feat, K0, H), (G2, h1, H, 1 + geo), (G3, head_in, 16 + geo, H), (G4, h3, H, H), (G5, h4, H, 3))
        gws = [None] * 5
        if any(need[0::2]):
            if field.fused_wgrad:
                # the five weight gradients: ONE kernel over the G_l and the layers' inputs + one reduction
                d = _lib.FieldWGrad()
                d.N, d.head_gap_col, d.g_max = Np, (16 if gap else 0xFFFFFFFF), st.g_max
                gws = [torch.empty((n_out, n_in), dtype=torch.float32, device=dev) for _, _, n_in, n_out in pairs]
                for i, (G, A, n_in, n_out) in enumerate(pairs):
                    d.G[i], d.ldG[i], d.n_out[i] = G.data_ptr(), G.shape[1], n_out
                    d.A[i], d.ldA[i], d.n_in[i] = A.data_ptr(), A.shape[1], n_in
                    d.dW[i], d.ld_dW[i] = gws[i].data_ptr(), n_in
                ws = field._wgrad_ws
                if ws is None or ws[0] != (dev, tuple(A.shape[1] for _, A, _, _ in pairs)):
                    nbytes = ctypes.c_uint64(0)
                    _lib.check(_lib.lib().cnc_field_weight_grads_workspace(ctypes.byref(d), ctypes.byref(nbytes)),
                               "field_weight_grads_workspace")
                    ws = field._wgrad_ws = ((dev, tuple(A.shape[1] for _, A, _, _ in pairs)),
                                            torch.empty(nbytes.value // 4, dtype=torch.float32, device=dev))
                d.workspace, d.workspace_bytes = ws[1].data_ptr(), ws[1].numel() * 4
                _lib.check(_lib.lib().cnc_field_weight_grads(ctypes.byref(d), _lib.stream(dev)), "field_weight_grads")
            else:
                for i, (G, A, n_in, n_out) in enumerate(pairs):
                    if need[2 * i]:
                        gw = splitk_weight_grad(G[:, :n_out] if G.shape[1] != n_out else G, A)
                        if i == 2 and gap:        # [SH4 | the raw density's slot | geo]: the slot is not an input of the layer
                            gw = torch.cat([gw[:, :16], gw[:, 17:17 + geo]], dim=1)
                        elif gw.shape[1] != n_in:
                            gw = gw[:, :n_in]
                        gws[i] = gw
        out = []
        for i in range(5):
            out += [gws[i] if need[2 * i] else None, gb_of[i] if need[2 * i + 1] else None]
        return dX, out


class _FieldTrain(Function):
    """The radiance field of the gradient pass, world positions + view directions -> (rgb, density), as TWO kernels and
    the weight gradients: forward = the fused evaluator in its saving form (cnc_field_save_t: the four encoders, the
    sinusoids, both MLPs, the activations — one launch that also writes what the backward reads), backward =
    cnc_field_backward_chain, the five weight gradients, and the four encoders' scatter (`_FusedFeatures.scatter`).
    Replaces `_prepare` + `_FusedFeatures` + `_FieldChain`: 4 encoder launches, the sinusoid kernel, 5 library GEMMs, the
    post kernel, the sigmoid, 3 slice copies and their fills.

    Values: the layers in the three-product fp16 form (~5e-7 per term) where the op chain runs fp32
    library GEMMs; the encoders' features are bit-identical.  Rows [N, Np) are rows of padding (`_bucket_rows`)."""

    @staticmethod
    def forward(ctx, positions, dirs, field, Np, *params_and_layers):
        params, layers = params_and_layers[:4], params_and_layers[4:]
        rgb, density, kept = field._field_fused.save_forward(positions, dirs, Np, params)
        ctx.save_for_backward(kept["feat"], kept["selector"], kept["h1"], kept["raw"], kept["head_in"], kept["h3"], kept["h4"],
                              rgb, kept["xyz"], kept["xy"], kept["xz"], kept["yz"], *params, *kept["clips"], *layers[0::2])
        ctx.field, ctx.N = field, positions.shape[0]
        from . import _gradsink
        ctx.sink = _gradsink.current()     # the caller thread's sink; the backward runs on autograd's own thread
        ctx.mark_non_differentiable(kept["selector"])
        return rgb, density

    @staticmethod
    def backward(ctx, g_rgb, g_density):
        t = ctx.saved_tensors
        feat, selector, h1, raw, head_in, h3, h4, rgb = t[:8]
        xs, params, clips, Ws = t[8:12], t[12:16], t[16:20], t[20:25]
        field, need = ctx.field, ctx.needs_input_grad
        owner = field.mlp_base
        dX, layer_grads = _FieldChain.chain(field, g_rgb, g_density, feat, selector, h1, raw, head_in, h3, h4, rgb, Ws,
                                            need[8:18], owner._layout()[0], gap=True)
        grads = [None] * 4
        if any(need[4:8]):
            grads = _FusedFeatures.scatter(owner, dX, xs, params, clips, ctx.N, ctx.sink)
        return (None, None, None, None, *grads, *layer_grads)


class NGPRadianceField_mygrid_2D3D(nn.Module):
    def __init__(self, aabb: Union[torch.Tensor, List[float]], num_dim: int = 3, use_viewdirs: bool = True,
                 density_activation: Callable = _default_density_activation, unbounded: bool = False,
                 geo_feat_dim: int = 15,
                 resolutions_list=(16, 22, 31, 42, 57, 78, 106, 146, 199, 273, 374, 512),
                 log2_hashmap_size: int = 19, resolutions_list_2D=(64, 128, 256, 512, 1024),
                 log2_hashmap_size_2D=17, n_features_per_level=2, n_neurons=64, ste_binary=True,
                 ste_multistep=False, add_noise=False, Q=10, sh_fp16_round=True, fused_ste=True,
                 fused_features=True) -> None:
        """`sh_fp16_round` (default True): the direction encoding's 16 values are rounded through half precision, as the
        reference's CUDA path sees them — tiny-cuda-nn writes its encoding to a half tensor unless told otherwise
        (ngp.py:412-425 passes no dtype) and `torch.cat` promotes it back (ngp.py:540-547).  False = full float32
        harmonics (the closed-form stand-in the round-3 goldens were made with)."""
        super().__init__()
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        self.register_buffer("aabb", aabb)
        self.num_dim = num_dim
        self.use_viewdirs = use_viewdirs
        self.density_activation = density_activation
        self.unbounded = unbounded
        self.geo_feat_dim = min(127, max(15, n_features_per_level * 10 - 1))     # ngp.py:398-401
        self.resolutions_list, self.log2_hashmap_size = resolutions_list, log2_hashmap_size
        self.fused_head = os.environ.get("CNC_FUSED_HEAD", "1") == "1"
        # normalise / selector, density activation, SH encoding and the head-input concat as single kernels
        self.fused_glue = fused_features and os.environ.get("CNC_FUSED_GLUE", "1") == "1"
        self.sh_fp16_round = bool(sh_fp16_round)
        self._head_fused = None
        # gradient-free calls as ONE kernel, positions -> density (-> rgb): FusedFieldForward (CNC_FUSED_FIELD=0: the
        # chain of encoder launches, library GEMMs and glue kernels that the gradient path uses)
        self.fused_field = fused_features and os.environ.get("CNC_FUSED_FIELD", "1") == "1"
        # "f16x3" (default): the layers on the fp16 matrix pipe, three products per term (~5e-7 per term, against
        # fp32's 6e-8); "f32": v_mfma_f32_32x32x2_f32, an exact fmaf chain per output — 2x the time
        self.fused_field_precision = os.environ.get("CNC_FUSED_FIELD_MFMA", "f16x3")
        if self.fused_field_precision not in ("f16x3", "f32"):
            raise ValueError("CNC_FUSED_FIELD_MFMA must be f16x3 or f32")
        # "w2" (default): two cooperating waves per 32-sample tile (csrc/field_fused2.hip); "w1": one wave per tile
        self.fused_field_kernel = os.environ.get("CNC_FUSED_FIELD_KERNEL", "w2")
        if self.fused_field_kernel not in ("w1", "w2"):
            raise ValueError("CNC_FUSED_FIELD_KERNEL must be w1 or w2")
        self.fused_field_waves = int(os.environ.get("CNC_FUSED_FIELD_WAVES", "0"))     # 0: per kernel (4 density, 3 colour)
        self._field_fused = None
        # the gradient pass's input-gradient chain as one kernel (`_FieldChain`; CNC_FUSED_CHAIN=0: layer by layer)
        self.fused_chain = fused_features and os.environ.get("CNC_FUSED_CHAIN", "1") == "1"
        # ... and its forward as the fused evaluator in its saving form (`_FieldTrain`; CNC_FUSED_TRAIN=0: library GEMMs)
        self.fused_train = self.fused_chain and os.environ.get("CNC_FUSED_TRAIN", "1") == "1"
        self._guard_seen = 1
        self._guard_host = self._guard_evt = self._guard_ids = None       # `snapshot_range_guard`
        # the five weight gradients as one kernel (cnc_field_weight_grads; CNC_FUSED_WGRAD=0: split-K library GEMMs)
        self.fused_wgrad = os.environ.get("CNC_FUSED_WGRAD", "1") == "1"
        self._wgrad_ws = None
        self._chain_supported = None
        self._chain_key = self._chain_wt = self._chain_src = None
        from . import _caches
        _caches.register(self)          # the transposed fragments: fused Adam updates weights without bumping `_version`
        # sample counts from here on run at a bucketed row count (`_bucket_rows`); CNC_ROW_BUCKET_MIN=0 pads every call
        self.row_bucket_min = int(os.environ.get("CNC_ROW_BUCKET_MIN", "4096"))
        self._head_shape_ok = n_neurons == 160       # the 32-row kernel is instantiated for 160-wide layers
        self.resolutions_list_2D, self.log2_hashmap_size_2D = resolutions_list_2D, log2_hashmap_size_2D

        if self.use_viewdirs:
            self.direction_encoding = SHEncoding(fp16_round=sh_fp16_round)

        def grid(D, res, T):
            return GridEncoder(num_dim=D, n_features=n_features_per_level, resolutions_list=res,
                               log2_hashmap_size=T, ste_binary=ste_binary, ste_multistep=ste_multistep,
                               add_noise=add_noise, Q=Q, fused_ste=fused_ste)

        encoding_xyz = grid(3, resolutions_list, log2_hashmap_size)
        encoding_xy = grid(2, resolutions_list_2D, log2_hashmap_size_2D)
        encoding_xz = grid(2, resolutions_list_2D, log2_hashmap_size_2D)
        encoding_yz = grid(2, resolutions_list_2D, log2_hashmap_size_2D)
        embed_fn, input_ch, freq_bands = get_embedder(10, 0, with_freqs=True)
        in_chs = (encoding_xyz.n_output_dims + encoding_xy.n_output_dims + encoding_xz.n_output_dims
                  + encoding_yz.n_output_dims + input_ch)
        network = nn.Sequential(Linear(in_chs, n_neurons), nn.ReLU(inplace=True),
                                Linear(n_neurons, 1 + self.geo_feat_dim))
        self.mlp_base = compose_3D_2D_embed(encoding_xyz, encoding_xy, encoding_xz, encoding_yz, embed_fn, network,
                                            fused=fused_features, freq_bands=freq_bands)
        if self.geo_feat_dim > 0:
            head_in = (self.direction_encoding.n_output_dims if self.use_viewdirs else 0) + self.geo_feat_dim
            self.mlp_head = nn.Sequential(Linear(head_in, n_neurons), nn.ReLU(inplace=True),
                                          Linear(n_neurons, n_neurons), nn.ReLU(inplace=True),
                                          Linear(n_neurons, 3))

    def update_embedding_params(self, params_q_xyz_rec, params_q_xy_rec, params_q_xz_rec, params_q_yz_rec):
        self.mlp_base.encoding_xyz.params = nn.Parameter(params_q_xyz_rec)
        self.mlp_base.encoding_xy.params = nn.Parameter(params_q_xy_rec)
        self.mlp_base.encoding_xz.params = nn.Parameter(params_q_xz_rec)
        self.mlp_base.encoding_yz.params = nn.Parameter(params_q_yz_rec)
        print("embedding_params updated!")

    def _glue_ok(self, x):
        return (self.fused_glue and x.is_cuda and x.dtype == torch.float32 and not self.unbounded and self.num_dim == 3
                and self.density_activation is _default_density_activation and 1 + self.geo_feat_dim <= 128)

    def invalidate_caches(self):
        """After an optimizer step (cnc_amd._caches): the packed transposed weights of the gradient chain are stale."""
        self._chain_key = None

    def _chain_ok(self, x_unit) -> bool:
        """The fused gradient chain (`_FieldChain`) applies: gradients wanted, the shapes the kernels are built for."""
        if not (self.fused_chain and torch.is_grad_enabled() and x_unit.is_cuda and self.mlp_base._can_fuse(x_unit)):
            return False
        if self._chain_supported is None:
            self._chain_supported = bool(FusedFieldForward.supported(self)
                                         and sum(e.n_output_dims for e in self.mlp_base._encoders()) % 4 == 0
                                         and sum(e.n_output_dims for e in self.mlp_base._encoders()) <= 192
                                         and (1 + self.geo_feat_dim + 31) // 32 * 32 <= self.mlp_base.network[0].out_features
                                         # cnc_field_backward_chain's own limits (field_bwd.hip): the bias-sum layout keeps
                                         # 80 (H = 160) / 64 (H = 64) slots for the base network's last layer
                                         and 1 + self.geo_feat_dim <= (80 if self.mlp_base.network[0].out_features == 160 else 64))
        if not self._chain_supported:
            return False
        # ... and its 32-bit byte offsets: rows x the widest row it addresses (the padded feature matrix or a hidden layer)
        H = self.mlp_base.network[0].out_features
        ld = max(H, (sum(e.n_output_dims for e in self.mlp_base._encoders()) + 3 + 60 + 31) // 32 * 32 + 32)
        if x_unit.shape[0] * ld * 4 >= (1 << 32):
            return False
        return self._chain_supported

    def _train_ok(self, positions, directions) -> bool:
        """The gradient pass as the saving fused kernel + the gradient chain (`_FieldTrain`) applies."""
        if not (self.fused_train and self.fused_field and torch.is_grad_enabled() and positions.is_cuda
                and positions.dtype == torch.float32 and not positions.requires_grad and not directions.requires_grad
                and self.fused_field_precision == "f16x3" and self.fused_field_kernel == "w2" and self._glue_ok(positions)):
            return False
        if not self._chain_ok(positions.reshape(-1, 3)):
            return False
        if self._field_fused is None:
            self._field_fused = FusedFieldForward(self) if FusedFieldForward.supported(self) else False
        return bool(self._field_fused)

    def check_range_guard(self) -> bool:
        """For the training loop, at a point where it synchronises anyway: True when a saving forward since the last check
        met a value beyond fp16's range (it saturated: cnc_field_save_t).  The gradient pass then leaves the fused kernel
        for the library-GEMM forward (`_FieldChain`), which has no such range."""
        ff = self._field_fused
        if not ff or not getattr(ff, "_train_calls", False) or not self.fused_train:
            return False
        words = ff.guard_words()
        ff._train_calls = False
        if words and (words[0] != 0 and words[0] >= self._guard_seen or any(w == ff._pack_id for w in words[1:6])):
            import warnings
            warnings.warn("cnc_amd: a value left fp16's range in the fused training forward; the gradient pass continues "
                          "on the fp32 library path")
            self.fused_train = False
            return True
        self._guard_seen = ff._call_id + 1
        return False

    def snapshot_range_guard(self) -> None:
        """Behind a training forward: the guard's words on their way to pinned host memory (one 24-byte copy, no
        synchronisation); `poll_range_guard` looks at them once they have arrived — in a training loop, the step after.
        Round 6: the check used to happen only where the loop synchronises anyway (every `step_update` steps)."""
        ff = self._field_fused
        if not ff or not getattr(ff, "_train_calls", False) or not self.fused_train or ff._buffers is None:
            return
        if self._guard_evt is not None and not self._guard_evt.query():
            return                                      # the previous snapshot is still in flight: it will be looked at first
        g = ff._buffers["guard"]
        if self._guard_host is None:
            self._guard_host = torch.zeros(6, dtype=g.dtype).pin_memory()
        self._guard_host.copy_(g[:6], non_blocking=True)
        self._guard_evt = torch.cuda.Event()
        self._guard_evt.record()
        self._guard_ids = (ff._pack_id, ff._call_id)

    def poll_range_guard(self) -> bool:
        """True (once) when a snapshot has arrived that shows a value beyond fp16's range in a saving forward: the gradient
        pass then leaves the fused kernel for the fp32 library path, as `check_range_guard` does.  Never waits."""
        if self._guard_evt is None or not self._guard_evt.query() or not self.fused_train:
            return False
        words, (pack_id, call_id) = self._guard_host.tolist(), self._guard_ids
        self._guard_evt = None
        if words[0] != 0 and words[0] >= self._guard_seen or any(w == pack_id for w in words[1:6]):
            import warnings
            warnings.warn("cnc_amd: a value left fp16's range in the fused training forward; the gradient pass continues "
                          "on the fp32 library path")
            self.fused_train = False
            return True
        self._guard_seen = max(self._guard_seen, call_id + 1)
        return False

    def _chain_weights_t(self, W1, W2, W3, W4, W5, n_enc):
        """The five layers TRANSPOSED in the 16x16x32 fragment order (cnc_field_pack_all, one launch), cached on the
        weights' versions: what cnc_field_backward_chain multiplies the gradients with."""
        from . import _lib
        ws = (W5, W4, W3, W2, W1)
        key = tuple((w.data_ptr(), w._version) for w in ws)
        if self._chain_key == key:
            return self._chain_wt
        H, geo = W1.shape[0], self.geo_feat_dim
        r32 = lambda k: (k + 31) // 32
        nb2 = 5 if H == 160 else 4
        # (W, packed outputs, K, column blocks, K-steps of 32, flags, src_off)
        T, ZF = _lib.CNC_PACK_TRANSPOSE, _lib.CNC_PACK_ZERO_FIRST
        layers = [(W5, H, 3, H // 16, 1, T, 0), (W4, H, H, H // 16, H // 32, T, 0),
                  (W3, 1 + geo, H, nb2, H // 32, T | ZF, 16), (W2, H, 1 + geo, H // 16, r32(1 + geo), T, 0),
                  (W1, n_enc, H, (n_enc + 15) // 16, H // 32, T, 0)]
        if self._chain_wt is None or self._chain_wt[0].device != W1.device:
            self._chain_wt = [torch.empty(nk * ncb * 1024, dtype=torch.float16, device=W1.device) for _, _, _, ncb, nk, _, _ in layers]
        d = _lib.FieldPack()
        keep = []
        for k, (w, Hp, K, ncb, nk, fl, off) in enumerate(layers):
            wc = w.detach()
            if not wc.is_contiguous():
                wc = wc.contiguous()
            keep.append(wc)
            L = d.layer[k]
            L.W, L.H, L.K, L.ldw = wc.data_ptr(), Hp, K, wc.stride(0)
            L.n_colblocks, L.n_ksteps32, L.flags, L.src_off = ncb, nk, fl, off
            L.Wq16 = self._chain_wt[k].data_ptr()
        import ctypes
        _lib.check(_lib.lib().cnc_field_pack_all(ctypes.byref(d), _lib.stream(W1.device)), "field_pack_all(transposed)")
        self._chain_key, self._chain_src = key, ws
        return self._chain_wt

    def _bucket_rows(self, n: int) -> int:
        """Row count the field's kernels and GEMMs run at for `n` samples: `n` rounded up to a multiple of ~3 % of
        itself (2^(floor(log2 n) - 5)).  The sample count of a training step is new every step, and a hipBLASLt
        GEMM whose row count the library has not seen yet costs 77-127 us of HOST time against 19-30 us for a known
        one (tools/gemm_launch_probe.py) — ~20 GEMMs per step, in the part of the step that is launch-bound.  With
        the rows bucketed a run meets a handful of distinct sizes.  The extra rows are zero feature rows with
        selector 0 (zero density); their outputs are sliced away, so they receive zero gradient, and the encoders
        neither compute nor back-propagate them (`_FusedFeatures`)."""
        if n < self.row_bucket_min:
            return n
        g = 1 << max(6, n.bit_length() - 6)
        return (n + g - 1) // g * g

    def _prepare(self, x):
        """(unit-cube positions [N,3], selector u8 [Np], Np) of N world positions: one kernel.  Np =
        `_bucket_rows(N)` (when the base network can take extra rows); the selector of the rows of padding is 0."""
        from . import _lib
        p = x.reshape(-1, 3).contiguous()
        N = p.shape[0]
        Np = self._bucket_rows(N) if self.mlp_base._can_fuse(p) else N
        x_unit = torch.empty_like(p)
        selector = torch.empty(Np, dtype=torch.uint8, device=p.device)
        _lib.check(_lib.lib().cnc_field_prepare(p.data_ptr(), self.aabb.contiguous().data_ptr(), N, x_unit.data_ptr(),
                                                selector.data_ptr(), _lib.stream(p.device)), "field_prepare")
        if Np != N:
            selector[N:].zero_()
        return x_unit, selector, Np

    def _fused_forward(self, x):
        """The one-kernel evaluator for a gradient-free call on `x`, or None (gradients wanted, switched off, a
        configuration outside the kernel's shapes, host tensors)."""
        if torch.is_grad_enabled() or not self.fused_field or not self._glue_ok(x):
            return None
        if self._field_fused is None:
            self._field_fused = FusedFieldForward(self) if FusedFieldForward.supported(self) else False
        return self._field_fused or None

    def query_density(self, x, return_feat: bool = False):
        if not return_feat:
            fused = self._fused_forward(x)
            if fused is not None:
                return fused(x).view(list(x.shape[:-1]) + [1])
        if self._glue_ok(x):
            lead = list(x.shape[:-1])
            x_unit, selector, Np = self._prepare(x)
            N = x_unit.shape[0]
            if not return_feat and not torch.is_grad_enabled():
                # the sampler's visibility pass (6-8x the samples of the gradient pass once a surface has formed: 1.5-2 M
                # against 2^18) reads the density only: unit 0 of the last layer instead of all 1 + geo_feat_dim
                h = self.mlp_base(x_unit, last_rows=1, rows=Np)
                density, _ = _FieldPost.apply(h, selector, None, 0)
                return (density if Np == N else density[:N]).view(lead + [1])
            h = self.mlp_base(x_unit, rows=Np)
            density, _ = _FieldPost.apply(h, selector, None, self.geo_feat_dim)
            if Np != N:
                density, h = density[:N], h[:N]
            density = density.view(lead + [1])
            return (density, h[:, 1:].view(lead + [self.geo_feat_dim])) if return_feat else density
        if self.unbounded:
            x = contract_to_unisphere(x, self.aabb)
        else:
            aabb_min, aabb_max = torch.split(self.aabb, self.num_dim, dim=-1)
            x = (x - aabb_min) / (aabb_max - aabb_min)
        selector = ((x > 0.0) & (x < 1.0)).all(dim=-1)
        h = self.mlp_base(x.view(-1, self.num_dim)).view(list(x.shape[:-1]) + [1 + self.geo_feat_dim]).to(x)
        density_before_activation, base_mlp_out = torch.split(h, [1, self.geo_feat_dim], dim=-1)
        density = self.density_activation(density_before_activation) * selector[..., None]
        return (density, base_mlp_out) if return_feat else density

    def _query_rgb(self, dir, embedding, apply_act: bool = True):
        if self.use_viewdirs:
            d = self.direction_encoding(((dir + 1.0) / 2.0).reshape(-1, dir.shape[-1]))
            h = torch.cat([d, embedding.reshape(-1, self.geo_feat_dim)], dim=-1)
        else:
            h = embedding.reshape(-1, self.geo_feat_dim)
        if (self.fused_head and h.is_cuda and not torch.is_grad_enabled() and self._head_shape_ok
                and h.shape[0] >= (1 << 17)):      # below that the 64-row tiles do not fill the 1024 SIMDs
            # gradient-free evaluation: the three layers in one MFMA kernel (32 rows per wave), hidden
            # activations stay in LDS; the input gets a 16-byte aligned row stride
            if self._head_fused is None:
                from .mlp import FusedMLPForward
                self._head_fused = FusedMLPForward(self.mlp_head, rows_per_wave=32)
            pad = (-h.shape[1]) % 4
            hp = F.pad(h, (0, pad)) if pad else h
            rgb = self._head_fused(hp[:, :h.shape[1]])
        else:
            rgb = run_layers(self.mlp_head, h)
        rgb = rgb.reshape(list(embedding.shape[:-1]) + [3]).to(embedding)
        return torch.sigmoid(rgb) if apply_act else rgb

    def _head(self, h):
        """mlp_head on a [N, ld] input whose columns beyond the head's width are zero padding."""
        n_in = self.mlp_head[0].in_features
        if (self.fused_head and not torch.is_grad_enabled() and self._head_shape_ok and h.shape[0] >= (1 << 17)):
            if self._head_fused is None:
                from .mlp import FusedMLPForward
                self._head_fused = FusedMLPForward(self.mlp_head, rows_per_wave=32)
            return self._head_fused(h[:, :n_in])
        first = self.mlp_head[0]
        pad = h.shape[1] - n_in
        return run_layers(self.mlp_head, h, first_weight=F.pad(first.weight, (0, pad)) if pad else first.weight)

    def forward(self, positions: torch.Tensor, directions: torch.Tensor = None):
        if self.use_viewdirs and (directions is not None):
            assert positions.shape == directions.shape, f"{positions.shape} v.s. {directions.shape}"
        if directions is not None and self.use_viewdirs and directions.is_cuda:
            fused = self._fused_forward(positions)
            if fused is not None:
                density, rgb = fused(positions, directions)
                lead = list(positions.shape[:-1])
                return rgb.view(lead + [3]), density.view(lead + [1])
        if directions is not None and self.use_viewdirs and self.geo_feat_dim > 0 and self._glue_ok(positions) \
                and directions.is_cuda and directions.dtype == torch.float32:
            lead = list(positions.shape[:-1])
            if self._train_ok(positions, directions):
                # the gradient pass: ONE kernel forward (the fused evaluator, saving what the backward reads), the
                # gradient chain kernel + weight gradients + the encoders' scatter backward (`_FieldTrain`)
                p = positions.reshape(-1, 3)
                N = p.shape[0]
                Np = self._bucket_rows(N)
                mb, mh = self.mlp_base.network, self.mlp_head
                rgb, density = _FieldTrain.apply(p, directions.reshape(-1, 3), self, Np,
                                                 *(e.params for e in self.mlp_base._encoders()),
                                                 mb[0].weight, mb[0].bias, mb[2].weight, mb[2].bias, mh[0].weight, mh[0].bias,
                                                 mh[2].weight, mh[2].bias, mh[4].weight, mh[4].bias)
                if Np != N:
                    rgb, density = rgb[:N], density[:N]
                return rgb.view(lead + [3]), density.view(lead + [1])
            x_unit, selector, Np = self._prepare(positions)
            N = x_unit.shape[0]
            dirs = directions.reshape(-1, 3)
            if Np != N:          # rows of padding (`_bucket_rows`): any direction will do
                dirs = torch.cat([dirs, dirs.new_zeros((Np - N, 3))])
            if self._chain_ok(x_unit):
                # the gradient pass: library GEMMs forward, ONE kernel for the whole input-gradient chain backward
                feat = self.mlp_base.features_fused(x_unit, Np)
                mb, mh = self.mlp_base.network, self.mlp_head
                rgb, density = _FieldChain.apply(feat, selector, dirs, self, mb[0].weight, mb[0].bias, mb[2].weight, mb[2].bias,
                                                 mh[0].weight, mh[0].bias, mh[2].weight, mh[2].bias, mh[4].weight, mh[4].bias)
            else:
                h = self.mlp_base(x_unit, rows=Np)
                density, head_in = _FieldPost.apply(h, selector, dirs, self.geo_feat_dim, self.sh_fp16_round)
                rgb = torch.sigmoid(self._head(head_in))
            if Np != N:
                rgb, density = rgb[:N], density[:N]
            return rgb.view(lead + [3]), density.view(lead + [1])
        density, embedding = self.query_density(positions, return_feat=True)
        rgb = self._query_rgb(directions, embedding=embedding)
        return rgb, density
