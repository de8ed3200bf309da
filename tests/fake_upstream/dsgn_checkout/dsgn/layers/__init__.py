"""Python wrappers over the checkout's compiled extension ``dsgn._C`` - the stand-in for upstream DSGN's layer package: every file here
does ``from dsgn import _C`` at import time, as the upstream wrappers do, so this package is importable only where an extension of that
name exists (a CUDA build upstream; tests/fake_upstream/reference_C.py or eval_driving_safety_amd.upstream_shims.ext_C here)."""
from .build_cost_volume import BuildCostVolume, build_cost_volume
from .nms import nms
from .sigmoid_focal_loss import SigmoidFocalLoss

__all__ = ["BuildCostVolume", "build_cost_volume", "nms", "SigmoidFocalLoss"]
