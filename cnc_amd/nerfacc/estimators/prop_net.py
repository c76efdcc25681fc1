"""`nerfacc.estimators.prop_net` exists so that the reference's examples/utils.py:19 imports unchanged; proposal-network
sampling is not on the CNC path (out of scope) and is not built."""
from .base import AbstractEstimator


class PropNetEstimator(AbstractEstimator):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("PropNetEstimator is outside the CNC path (the CNC drivers use OccGridEstimator)")


def get_proposal_requires_grad_fn(*args, **kwargs):
    raise NotImplementedError("proposal networks are outside the CNC path")
