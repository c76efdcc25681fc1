"""Derived copies of parameters (the encoders' packed sign planes, the fused MLP kernel's padded weights) are
keyed on `Tensor._version`, which every in-place op bumps — except the fused optimizers
(`torch.optim.Adam(fused=True)` and friends update the parameters without touching the counter).  So every
holder of such a copy registers here, and ONE global optimizer post-step hook drops the copies after any
optimizer step: a stale sign plane would silently freeze training.
"""
import weakref

_holders = weakref.WeakSet()
_hooked = False


def invalidate_all(*_):
    for h in list(_holders):
        h.invalidate_caches()


def register(holder):
    """`holder.invalidate_caches()` is called after every optimizer step from now on."""
    global _hooked
    _holders.add(holder)
    if not _hooked:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(invalidate_all)
        _hooked = True
