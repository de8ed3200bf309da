"""Modules that stand where an upstream checkout's COMPILED extensions stand, so that the upstream Python model code runs on an
MI355X without hipifying anything: ``install()`` registers them in ``sys.modules`` under the names the reference's files import.

  model.roi_layers        -> upstream_shims.roi_layers   (ROIAlign, roi_align, nms on csrc/roi.hip)

``cli/upstream.py`` calls ``install()`` before it imports the Stereo R-CNN checkout's ``model.*``; a user's own script does
``import eval_driving_safety_amd.upstream_shims as s; s.install()`` first."""
import sys

from . import roi_layers

_NAMES = {"model.roi_layers": roi_layers,
          "model.roi_layers.roi_align": roi_layers,       # upstream sub-modules a checkout may import by their own names
          "model.roi_layers.nms": roi_layers}


def install(force=True):
    """register the shims; ``force=False`` keeps a ``model.roi_layers`` that is already imported.  Returns the names installed."""
    done = []
    for name, mod in _NAMES.items():
        if force or name not in sys.modules:
            sys.modules[name] = mod
            done.append(name)
    if "model.roi_layers" in done:
        parent = sys.modules.get("model")
        if parent is None:
            try:                                     # the checkout's own ``model`` package (importable once its lib/ is on sys.path)
                import importlib
                parent = importlib.import_module("model")
            except ImportError:
                parent = None                        # ``from model.roi_layers import ...`` still resolves through sys.modules later
        if parent is not None:
            setattr(parent, "roi_layers", roi_layers)
    return done


def installed():
    return sys.modules.get("model.roi_layers") is roi_layers
