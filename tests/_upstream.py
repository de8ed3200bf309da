"""Helpers for the tests that run the stand-in upstream checkouts (tests/fake_upstream/): which extension module stands at ``dsgn._C``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_upstream")
UPSTREAM_NAMES = ("dsgn", "env_utils", "model", "roi_data_layer", "_init_paths")


def forget_upstream():
    for m in [m for m in sys.modules if m.split(".")[0] in UPSTREAM_NAMES]:
        del sys.modules[m]


def reference_C():
    """tests/fake_upstream/reference_C.py as a module object (plain torch: what a CUDA build of the checkout's extension would compute)"""
    name = "_fake_upstream_reference_C"
    if name not in sys.modules:
        spec = importlib.util.spec_from_file_location(name, os.path.join(FAKE, "reference_C.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        sys.modules[name] = mod
    return sys.modules[name]


def bind_dsgn_extension(ext):
    """``ext``: "reference" - the plain-torch restatement stands at ``dsgn._C``; "shim" - libadvengine's (upstream_shims.install("dsgn"));
    None - nothing (the checkout's ``from dsgn import _C`` then fails, as on a box without any build)"""
    sys.modules.pop("dsgn._C", None)
    if ext == "reference":
        sys.modules["dsgn._C"] = reference_C()
    elif ext == "shim":
        from eval_driving_safety_amd import upstream_shims
        upstream_shims.install("dsgn")
    elif ext is not None:
        raise ValueError(ext)
