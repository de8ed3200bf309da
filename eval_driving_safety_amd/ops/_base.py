"""Host-side operators over torch ROCm tensors -> libadvengine.so.

Each function is the counterpart of one inline block of the reference's attack scripts (cited
per function, paths relative to the reference root) and keeps its argument meaning.  Tensors
stay where they are: the kernels are enqueued on the caller's current torch stream, nothing
is copied, allocated (except a result tensor when ``out`` is not given) or synchronised.

torch is plumbing here (device memory + streams); all arithmetic is in the HIP library.
"""
import ctypes

import torch
import torch.nn.functional as F

from .. import _lib, routes
from .._lib import AdvSpace

_lib.load()  # fail at import time if the HIP library is not built


class Space:
    """Per-channel pixel space (adv_space_t)."""

    def __init__(self, c_space, name):
        self.c = c_space
        self.name = name

    @staticmethod
    def dsgn(reference_on_gpu=False):
        """ImageNet-normalised RGB in [0,1]: attack/DSGN/pgd_attack.py:153-154,196-207,349-350.
        ``reference_on_gpu``: re-normalise with the float32 reciprocal, as torch's GPU kernels evaluate ``tensor / std[c]`` -
        bit-identical to a GPU run of the reference script; the default is bit-identical to a CPU run."""
        s = AdvSpace()
        if reference_on_gpu:
            _lib.load().adv_space_dsgn_gpu_reference(ctypes.byref(s))
            return Space(s, "dsgn_norm01_gpu_reference")
        _lib.load().adv_space_dsgn(ctypes.byref(s))
        return Space(s, "dsgn_norm01")

    @staticmethod
    def srcnn():
        """BGR minus PIXEL_MEANS on 0..255: attack/Stereo-RCNN/pgd_attack.py:189-207."""
        s = AdvSpace()
        _lib.load().adv_space_srcnn(ctypes.byref(s))
        return Space(s, "srcnn_meansub255")

    @property
    def affine(self):
        return self.c.kind in (_lib.ADV_SPACE_AFFINE, _lib.ADV_SPACE_AFFINE_RCP)

    @property
    def lo(self):
        return tuple(self.c.lo)

    @property
    def hi(self):
        return tuple(self.c.hi)

    def ref(self):
        return ctypes.byref(self.c)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _img(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError("%s must be a CUDA/ROCm tensor (there is no CPU path)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.shape[1] != 3:
        raise ValueError("%s must be [N,3,H,W] or [3,H,W], got %s" % (name, tuple(t.shape)))
    return t


def _same(a, b, na, nb):
    if a.shape != b.shape or a.device != b.device:
        raise ValueError("%s %s/%s and %s %s/%s differ" % (na, tuple(a.shape), a.device, nb, tuple(b.shape), b.device))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _f3(v):
    return None if v is None else (ctypes.c_float * 3)(*[float(x) for x in v])


class _on:
    """make the tensor's device current for the launch (no-op in the one-process-per-GPU case)"""

    def __init__(self, t):
        self.dev = t.device
        self.ctx = None

    def __enter__(self):
        if torch.cuda.current_device() != self.dev.index:
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _feat(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous float32 CUDA tensor" % name)
    return t


__all__ = [n for n in dir() if not n.startswith("__")]
