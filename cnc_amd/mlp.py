"""Dense layers of the radiance field / context models.

The layers are ordinary `nn.Linear` modules (same parameters, same state-dict keys as the
reference's `nn.Sequential(nn.Linear, ...)`, ngp.py:475-504, utils_bpp_acc.py:378-393) whose GEMMs run
on hipBLASLt's fp32 MFMA kernels.  One thing is done differently: the weight gradient.

For these shapes — a reduction over N = 10^5..10^6 samples into a tiny [out, in] matrix — hipBLASLt's
default kernel selection on MI355X runs at 0.4–39 TFLOP/s (tools/gemm_probe.py: 0.45 ms for a 160x3
layer at N=2^18), while forward / input-gradient GEMMs of the same layers reach 60–100 TFLOP/s.
`dW = dY^T X` is therefore computed as a batched GEMM over S row-slabs followed by a sum over slabs
(split-K by hand): 4–17x faster on the same library, fp32 throughout.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _caches


def _split(n_rows: int) -> int:
    """Number of row slabs: ~1024+ rows per slab, power of two, at most 256."""
    s = 1
    while s < 256 and n_rows // (s * 2) >= 1024:
        s *= 2
    return s


def splitk_weight_grad(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """g^T x ([out, in]) as a batched GEMM over row slabs + a sum over the slabs (see the module docstring)."""
    n = x.shape[0]
    s = _split(n)
    if s == 1:
        return g.t() @ x
    m = n // s
    head = m * s
    gw = torch.bmm(g[:head].reshape(s, m, -1).transpose(1, 2), x[:head].reshape(s, m, -1)).sum(0)
    if head < n:
        gw = gw + g[head:].t() @ x[head:]
    return gw


class _LinearSplitK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        g = grad_out.reshape(-1, grad_out.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        grad_x = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_x = (g @ weight).reshape(x.shape)
        if ctx.needs_input_grad[1]:
            n = x2.shape[0]
            s = _split(n)
            if s == 1:
                grad_w = g.t() @ x2
            else:
                m = n // s
                head = m * s
                gs = g[:head].view(s, m, -1)
                xs = x2[:head].view(s, m, -1)
                grad_w = torch.bmm(gs.transpose(1, 2), xs).sum(0)
                if head < n:
                    grad_w = grad_w + g[head:].t() @ x2[head:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_b = g.sum(0)
        return grad_x, grad_w, grad_b


class _LinearReLUSplitK(torch.autograd.Function):
    """relu(x W^T + b) with the bias + ReLU in the GEMM epilogue; backward masks the incoming gradient
    with (y > 0) and continues as `_LinearSplitK`."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = torch._addmm_activation(bias, x, weight.t())
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, y = ctx.saved_tensors
        grad_x = grad_w = grad_b = None
        grad_out = grad_out.contiguous()
        C = y.shape[1]
        if (ctx.needs_input_grad[2] and y.dtype == torch.float32 and grad_out.dtype == torch.float32 and C % 4 == 0
                and C <= 256 and y.is_contiguous()):
            # ReLU's backward and the column sums for the bias gradient in one pass over the gradient
            from . import _lib
            L, n = _lib.lib(), y.shape[0]
            g = torch.empty_like(y)
            partial = torch.empty((int(L.cnc_relu_backward_bias_partials(n)), C), dtype=torch.float32, device=y.device)
            _lib.check(L.cnc_relu_backward_bias(grad_out.data_ptr(), y.data_ptr(), n, C, g.data_ptr(), partial.data_ptr(),
                                                _lib.stream(y.device)), "relu_backward_bias")
            grad_b = partial.sum(0)
        else:
            g = torch.ops.aten.threshold_backward(grad_out, y, 0.0)      # nn.ReLU's own backward kernel
        if ctx.needs_input_grad[0]:
            grad_x = g @ weight
        if ctx.needs_input_grad[1]:
            n = x.shape[0]
            s = _split(n)
            if s == 1:
                grad_w = g.t() @ x
            else:
                m = n // s
                head = m * s
                grad_w = torch.bmm(g[:head].view(s, m, -1).transpose(1, 2), x[:head].view(s, m, -1)).sum(0)
                if head < n:
                    grad_w = grad_w + g[head:].t() @ x[head:]
        if ctx.needs_input_grad[2] and grad_b is None:
            grad_b = g.sum(0)
        return grad_x, grad_w, grad_b


class Linear(nn.Linear):
    """nn.Linear with the split-K weight gradient (CUDA tensors only; CPU uses the stock path)."""

    def forward(self, input):
        if input.is_cuda and torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad):
            return _LinearSplitK.apply(input, self.weight, self.bias)
        return F.linear(input, self.weight, self.bias)


def linear_fn(x, weight, bias):
    """Functional form of `Linear.forward` (for callers that pad or slice the weight)."""
    if x.is_cuda and torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return _LinearSplitK.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def run_layers(layers, h, first_weight=None, last_rows=None):
    """`nn.Sequential(*layers)(h)`, with every Linear -> ReLU pair as ONE hipBLASLt GEMM with bias +
    ReLU in the epilogue (`torch._addmm_activation`) instead of a GEMM and a separate pass over the
    activations; under autograd through `_LinearReLUSplitK`.
    `first_weight` replaces the first Linear's weight (zero-padded input columns); `last_rows` = k keeps only the
    first k output units of the LAST Linear (a caller that reads nothing else: the density query)."""
    layers = list(layers)
    fuse = h.is_cuda and h.dim() == 2
    i = 0
    while i < len(layers):
        m = layers[i]
        if isinstance(m, nn.Linear):
            w = first_weight if (i == 0 and first_weight is not None) else m.weight
            if last_rows is not None and i == len(layers) - 1:
                h = linear_fn(h, w[:last_rows], None if m.bias is None else m.bias[:last_rows])
                i += 1
                continue
            if fuse and m.bias is not None and i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU):
                if torch.is_grad_enabled() and (h.requires_grad or w.requires_grad):
                    h = _LinearReLUSplitK.apply(h, w, m.bias)
                else:
                    h = torch._addmm_activation(m.bias, h, w.t())
                i += 2
                continue
            h = linear_fn(h, w, m.bias)
        else:
            h = m(h)
        i += 1
    return h


def _round16(n: int) -> int:
    return (n + 15) // 16 * 16


class FusedMLPForward:
    """Gradient-free forward of `nn.Sequential(Linear, ReLU, Linear[, ReLU, Linear])` through the
    fused MFMA kernel (cnc_amd/csrc/mlp.hip).  Keeps zero-padded copies of the weights, refreshed
    when a parameter is modified in place (optimizer step) or replaced."""

    def __init__(self, seq: nn.Sequential, rows_per_wave: int = 16):
        # 16: v_mfma_f32_16x16x4 kernel, any widths <= 160; 32: v_mfma_f32_32x32x2 kernel, only the
        # radiance field's two shapes (hidden 160, second width <= 96 or 160, output <= 32)
        self.rows_per_wave = rows_per_wave
        self.linears = [m for m in seq if isinstance(m, nn.Linear)]
        acts = [m for m in seq if not isinstance(m, nn.Linear)]
        if len(self.linears) not in (2, 3) or len(acts) != len(self.linears) - 1 \
                or not all(isinstance(a, nn.ReLU) for a in acts):
            raise ValueError("FusedMLPForward supports Linear-ReLU-Linear[-ReLU-Linear]")
        if any(l.out_features > 160 for l in self.linears) or any(l.bias is None for l in self.linears):
            raise ValueError("FusedMLPForward: widths <= 160 and biases required")
        self._key = None
        self._packed = None
        _caches.register(self)     # fused optimizers do not bump Tensor._version

    def _pack(self):
        key = tuple((l.weight.data_ptr(), l.weight._version, l.bias.data_ptr(), l.bias._version)
                    for l in self.linears)
        if key != self._key:
            packed = []
            wide = self.rows_per_wave == 32
            kp = (self.linears[0].in_features + 7) // 8 * 8 if wide else _round16(self.linears[0].in_features)
            for l in self.linears:
                hp = (l.out_features + 31) // 32 * 32 if wide else _round16(l.out_features)
                w = torch.zeros((hp, kp), dtype=torch.float32, device=l.weight.device)
                w[: l.out_features, : l.in_features] = l.weight.detach()
                b = torch.zeros(hp, dtype=torch.float32, device=l.weight.device)
                b[: l.out_features] = l.bias.detach()
                packed.append((w, b, hp))
                kp = hp
            self._packed, self._key = packed, key
            # pin the sources: a freed weight whose address and version are reused must not alias the key
            self._src = [(l.weight, l.bias) for l in self.linears]
        return self._packed

    def invalidate(self):
        """Drop the packed copies (call after writing a weight through `.data`, which skips `_version`)."""
        self._key = self._packed = None

    invalidate_caches = invalidate

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        from . import _lib
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDA tensor")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).to(torch.float32)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        n, k0 = x2.shape
        packed = self._pack()
        n_out = self.linears[-1].out_features
        y = torch.empty((n, n_out), dtype=torch.float32, device=x.device)
        (w1, b1, h1), (w2, b2, h2) = packed[0], packed[1]
        w3, b3, h3 = packed[2] if len(packed) == 3 else (None, None, 0)
        fn = _lib.lib().cnc_mlp_forward32 if self.rows_per_wave == 32 else _lib.lib().cnc_mlp_forward
        rc = fn(x2.data_ptr(), n, x2.stride(0), k0, w1.data_ptr(), b1.data_ptr(), h1,
                                        w2.data_ptr(), b2.data_ptr(), h2, _lib.ptr(w3), _lib.ptr(b3), h3,
                                        y.data_ptr(), n_out, n_out, _lib.stream(x.device))
        _lib.check(rc, "mlp_forward")
        return y.reshape(*lead, n_out)
