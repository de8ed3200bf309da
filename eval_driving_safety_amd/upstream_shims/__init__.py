"""Modules that stand where an upstream checkout's COMPILED extensions stand, so that the upstream Python model code runs on an
MI355X without hipifying anything: ``install()`` registers them in ``sys.modules`` under the names the checkouts import.

  group "srcnn" (Stereo R-CNN; names from the reference's own files, attack/Stereo-RCNN/stereo_rcnn.py:18, pgd_attack.py:25-27)
    model.roi_layers, model.roi_layers.roi_align, model.roi_layers.nms   -> roi_layers   (ROIAlign, roi_align, nms on csrc/roi.hip)
  group "dsgn"  (DSGN; the compiled pieces SURVEY 2.2 lists behind attack/DSGN/pgd_attack.py:220,308,324)        [UPSTREAM-UNVERIFIED names]
    dsgn._C                                                             -> ext_C        (build_cost_volume_*, sigmoid_focalloss_*, nms,
                                                                                         roi_align_* on csrc/psv.hip, volume.hip, roi.hip)
  offered, not installed by default (a checkout whose wrapper package itself cannot be imported):
    model._C                                                            -> ext_C
    dsgn.layers, dsgn.layers.build_cost_volume, dsgn.layers.sigmoid_focal_loss, dsgn.layers.nms, dsgn.layers.roi_align -> dsgn_layers

The DSGN sources are not in the reference tree, so WHERE its extension lives is a fact of the user's checkout: ``TABLE`` is data, and
``install(names, table={"their.module.name": "ext_C" | "dsgn_layers" | "roi_layers" | a module object})`` (CLI: ``--shim name=key``)
puts a shim under any other name.  ``cli/upstream.py`` calls ``install("srcnn")`` / ``install("dsgn")`` before it imports the checkout's
model code; a user's own script does ``import eval_driving_safety_amd.upstream_shims as s; s.install()`` first."""
import importlib
import sys
import types

from . import dsgn_layers, ext_C, roi_layers

SHIMS = {"roi_layers": roi_layers, "ext_C": ext_C, "dsgn_layers": dsgn_layers}

# module name in the checkout -> shim key
TABLE = {
    "model.roi_layers": "roi_layers",
    "model.roi_layers.roi_align": "roi_layers",           # upstream sub-modules a checkout may import by their own names
    "model.roi_layers.nms": "roi_layers",
    "model._C": "ext_C",
    "dsgn._C": "ext_C",
    "dsgn.layers": "dsgn_layers",
    "dsgn.layers.build_cost_volume": "dsgn_layers",
    "dsgn.layers.sigmoid_focal_loss": "dsgn_layers",
    "dsgn.layers.nms": "dsgn_layers",
    "dsgn.layers.roi_align": "dsgn_layers",
}
GROUPS = {
    "srcnn": ("model.roi_layers", "model.roi_layers.roi_align", "model.roi_layers.nms"),
    "dsgn": ("dsgn._C",),
}
_installed = {}


def _resolve(names, table):
    if names is None:
        names = ("srcnn", "dsgn")
    if isinstance(names, str):
        names = (names,)
    out = []
    for n in names:
        for m in GROUPS.get(n, (n,)):
            if m not in table:
                raise KeyError("no shim is listed for %r: pass table={%r: 'ext_C' | 'dsgn_layers' | 'roi_layers'} (known names: %s)"
                               % (m, m, ", ".join(sorted(table))))
            if m not in out:
                out.append(m)
    return out


def _module(key):
    if isinstance(key, types.ModuleType):
        return key
    if key not in SHIMS:
        raise KeyError("unknown shim %r (have: %s)" % (key, ", ".join(sorted(SHIMS))))
    return SHIMS[key]


def install(names=None, table=None, force=True):
    """Register shims in ``sys.modules``.  ``names``: groups ("srcnn", "dsgn") and / or module names, default both groups; ``table``:
    extra / overriding {module name: shim key or module}, e.g. {"dsgn.ops._ext": "ext_C"}; its names are installed too.  ``force=False``
    keeps a module of that name that is already imported.  Returns the names installed."""
    merged = dict(TABLE)
    if table:
        merged.update(table)
    wanted = _resolve(names, merged)
    if table:
        wanted += [n for n in table if n not in wanted]
    done = []
    for name in wanted:
        mod = _module(merged[name])
        if not force and name in sys.modules and sys.modules[name] is not mod:
            continue
        sys.modules[name] = mod
        _installed[name] = mod
        done.append(name)
        parent_name, _, leaf = name.rpartition(".")
        if not parent_name:
            continue
        parent = sys.modules.get(parent_name)
        if parent is not None and any(parent is m for m in SHIMS.values()):
            continue                                 # a sub-module name of a shimmed package: the shim already holds that attribute (a function)
        if parent is None:
            try:                                     # the checkout's own package (importable once the checkout is on sys.path)
                parent = importlib.import_module(parent_name)
            except Exception:
                parent = None                        # ``from pkg.ext import ...`` still resolves through sys.modules later
        if parent is not None:
            setattr(parent, leaf, mod)
    return done


def parse_shim_flags(pairs):
    """``--shim their.module=ext_C`` (repeatable) -> the ``table`` argument of ``install``"""
    table = {}
    for p in pairs or ():
        name, sep, key = str(p).partition("=")
        if not sep or not name or key not in SHIMS:
            raise ValueError("--shim takes module.name=%s, got %r" % ("|".join(sorted(SHIMS)), p))
        table[name] = key
    return table


def installed(name="model.roi_layers"):
    """is ``name`` currently bound to its shim (default: the Stereo R-CNN RoI package)"""
    return name in _installed and sys.modules.get(name) is _installed[name]
