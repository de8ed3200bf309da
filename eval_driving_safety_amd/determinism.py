"""Byte-reproducible attacks: what the PRODUCT does about the layers it leaves to torch (README "Reproducibility").

This package's own kernels accumulate in a fixed order and the committed route table (routes_gfx950.json) fixes which kernel a layer
takes, so their results depend on the inputs only.  Layers still computed by torch - MIOpen convolutions (strided / 7x7 layers, the
table's "" routes) and rocBLAS products - may take split-K solvers that accumulate with float atomics, and MIOpen answers a shape's
FIRST call in a process with a fallback solver while it looks for a better one.  Two measures, applied by the attack drivers and the
adapters themselves (not by the caller, not by the tests):

  ``solvers()`` / ``@deterministic``   every detector step runs under ``torch.backends.cudnn.flags(deterministic=True, benchmark=False)``:
                                       MIOpen is asked for deterministic solvers only and never times alternatives at run time;
  warm-up (inside ``@deterministic``)  the first time an adapter sees an input shape it runs ONE throw-away step before the real one, so that
                                       iteration 1 of image 1 is computed by the same solvers as every later iteration.  An adapter of
                                       your own: decorate its ``loss_and_grad`` with ``determinism.deterministic`` (or call
                                       ``determinism.warm_adapter`` yourself).

``cli/_common.setup_device`` additionally sets the two flags process-wide for the command-line scripts.  north_star asks for bit-exact
box indices: a sign flip of one gradient element moves one pixel by 2 alpha, so reproducible gradients are what that rests on."""
import contextlib
import functools

import torch

_warmed = set()


def _flags():
    """deterministic solvers, no run-time search; ``enabled`` stays what the user set (torch's context manager would default it to False)"""
    return torch.backends.cudnn.flags(enabled=torch.backends.cudnn.enabled, deterministic=True, benchmark=False)


@contextlib.contextmanager
def solvers():
    with _flags():
        yield


def _throw_away(step):
    """run a warm-up step without consuming the caller's random numbers: an upstream detector's forward may sample (proposal targets),
    and the real step must draw what it would have drawn without the warm-up"""
    import random
    py, cpu = random.getstate(), torch.random.get_rng_state()
    cuda = torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None
    try:
        import numpy as np
        npst = np.random.get_state()
    except Exception:                             # noqa: BLE001
        np = npst = None
    try:
        step()
    finally:
        random.setstate(py)
        torch.random.set_rng_state(cpu)
        if cuda is not None:
            torch.cuda.set_rng_state_all(cuda)
        if np is not None:
            np.random.set_state(npst)


def deterministic(fn):
    """decorator for a detector step (``loss_and_grad(self, x, extra)`` / ``detect(self, x, extra)``): the call - and the autograd backward
    it runs; the flags are process-wide, not thread-local - under ``solvers()``, preceded, the FIRST time this object sees an input of
    this shape on a ROCm device, by one throw-away call of the same function (the warm-up: MIOpen's first answer for a shape may come from
    another solver than its later ones).  The caller sees one call and one result either way; ``x`` is not modified by a step."""
    @functools.wraps(fn)
    def inner(self, x, *args, **kwargs):
        with _flags():
            if isinstance(x, torch.Tensor) and x.is_cuda and not torch.cuda.is_current_stream_capturing():
                seen = self.__dict__.setdefault("_adv_warm_shapes", set())
                key = (tuple(x.shape), str(x.device))                                # one key format with warm_adapter(): either warms for both
                if key not in seen:
                    seen.add(key)
                    _throw_away(lambda: fn(self, x, *args, **kwargs))
            return fn(self, x, *args, **kwargs)
    inner.__wrapped_deterministic__ = True
    return inner


def under_solvers(fn):
    """decorator for a driver entry point (PgdAttack.run_batch, PatchTrainer.train_batch ...): the call under ``solvers()``, no warm-up"""
    @functools.wraps(fn)
    def inner(*args, **kwargs):
        with _flags():
            return fn(*args, **kwargs)
    return inner


def set_process_defaults():
    """the same two flags as process-wide defaults (the CLIs); returns the previous (deterministic, benchmark)"""
    old = (torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    return old


def warm_once(key, fn):
    """run ``fn()`` once per ``key`` per process (a throw-away detector step on a new (detector, shape)); True when it ran"""
    if key in _warmed:
        return False
    _warmed.add(key)
    fn()
    return True


def warm_adapter(adapter, x, extra):
    """ONE throw-away ``adapter.loss_and_grad(x, extra)`` per (adapter object, input shape, device) on a ROCm tensor: MIOpen's first
    answer for a shape may come from another solver than its later ones.  ``x`` is not modified.  True when the step ran."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        return False
    seen = getattr(adapter, "_adv_warm_shapes", None)
    if seen is None:
        seen = set()
        try:
            adapter._adv_warm_shapes = seen
        except AttributeError:              # an adapter without a __dict__: warm it every time rather than never
            pass
    key = (tuple(x.shape), str(x.device))
    if key in seen:
        return False
    seen.add(key)
    _throw_away(lambda: adapter.loss_and_grad(x, extra))
    return True


def forget_warm_ups():
    _warmed.clear()
