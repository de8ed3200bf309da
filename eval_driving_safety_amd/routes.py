"""Which kernel computes a convolution - a COMMITTED TABLE, not a stopwatch.

Several layers have more than one implementation in libadvengine (direct implicit GEMM, Winograd F(2x2,3x3)) next to torch's
own operator (MIOpen); they sum in different float orders, so the choice is part of the RESULT: two runs of one attack, or two
ranks of an image-sharded job, must take the same route for the same layer or their gradients differ in the last bits, the sign
maps flip somewhere and the adversarial PNGs differ.  Round 3 chose by timing at first use; this module replaces that:

  ADV_ROUTES=table   (default)  ``routes_gfx950.json`` next to this file: "direction|kernel|shape..." -> route, generated on an
                                MI355X by tools/make_routes.py from per-layer measurements and committed.  A shape the table
                                does not hold gets the FIXED RULE below - never a timer.
  ADV_ROUTES=fixed              the fixed rule only (what a table miss gets).
  ADV_ROUTES=measure            round 3's behaviour (time every candidate the first time a shape is seen, keep the fastest for
                                the process): for tools/make_routes.py and the per-layer benches, which is how the table is made.
  ADV_ROUTES=/path/to.json      another table (e.g. one a user generated for their own layer shapes with tools/make_routes.py).

The fixed rule: a layer with a Winograd kernel (3x3 / 3x3x3, stride 1) takes it, everything else this package's direct kernel;
torch's operator only where the table says so (it won 3 % of the measured layer shapes, all small maps).  ``table_hash()``
identifies the table in bench lines and output manifests.
"""
import hashlib
import json
import os

import torch

_DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "routes_gfx950.json")
ROUTES = ("hip", "wino", "wino4", "", "direct")         # "" = torch's operator (MIOpen / rocBLAS); "direct"/"wino"/"wino4" for the 3D stride-1 layers
                                                         # "wino" = Winograd F(2x2,3x3) (csrc/wino2d.hip), "wino4" = F(4x4,3x3) (csrc/wino4.hip)

_state = {"mode": None, "table": None, "path": None, "measured": {}, "misses": set(), "used": {}, "memo": {}}


def key_str(key):
    """a route key (nested tuples of ints / bools / strings) -> the flat string the JSON table is indexed by"""
    out = []
    for v in key:
        if isinstance(v, (tuple, list, torch.Size)):
            out.append("x".join(str(int(q)) for q in v))
        elif isinstance(v, bool):
            out.append("1" if v else "0")
        else:
            out.append(str(v))
    return "|".join(out)


def _load(path):
    with open(path) as f:
        doc = json.load(f)
    routes = doc.get("routes", {})
    bad = {k: v for k, v in routes.items() if v not in ROUTES}
    if bad:
        raise ValueError("%s: unknown routes %r" % (path, bad))
    return routes


def configure(mode=None):
    """(re)read ADV_ROUTES (or take ``mode``): "table" | "fixed" | "measure" | a path"""
    mode = mode if mode is not None else os.environ.get("ADV_ROUTES", "table")
    _state["measured"], _state["misses"], _state["used"], _state["memo"] = {}, set(), {}, {}
    if mode in ("fixed", "measure"):
        _state.update(mode=mode, table={}, path=None)
    else:
        path = _DEFAULT if mode == "table" else mode
        if not os.path.exists(path):
            if mode != "table":
                raise FileNotFoundError("ADV_ROUTES=%s: no such route table" % mode)
            _state.update(mode="table", table={}, path=None)           # no table shipped: everything by the fixed rule
        else:
            _state.update(mode="table", table=_load(path), path=path)
    return _state["mode"]


def mode():
    if _state["mode"] is None:
        configure()
    return _state["mode"]


def table_hash():
    """12 hex digits identifying the decisions in force (the table's content, or the mode when there is none)"""
    m = mode()
    if m != "table":
        return m
    blob = json.dumps(sorted(_state["table"].items()), separators=(",", ":")).encode()
    return hashlib.sha256(blob).hexdigest()[:12]


def _wino4_pays(key):
    """a 3x3 / 3x3x3 stride-1 layer the table does not know: does F(4x4,3x3) pay?  Its workgroup computes 16 x 32 outputs of 64 channels
    in one plane (two images side by side on maps of at most 15 columns).  It wins (a) where the launch is at least ~0.6 rounds of the
    chip and the tiles are at least 45 % real outputs (maps of 5 rows are two thirds padding), and (b) - 2D layers, since the K-split
    launch - on SMALL maps with a long contraction (at most 128 tiles x channel blocks and at least 256 input channels: the contraction is
    dealt to several workgroups per tile); F(2x2,3x3) otherwise.  Fitted to profiles/r06_routes_measured.jsonl: the sign of the measured
    difference on 182 of the 188 layers measured both ways, 0.4 % over the per-layer optimum summed over all of them (the round-5 rule on
    the same table: 149 of 188, 5.7 %).  A function of the key alone: no clock."""
    try:
        if key[0] in ("f", "b"):
            if key[1] != 3:
                return False
            out_ch, in_ch = (key[3], key[2]) if key[0] == "f" else (key[2], key[3])
            shape = key[5]
            planes, h, w, two_d = shape[0], shape[2], shape[3], True
        elif key[0] in ("f3", "b3"):
            out_ch, in_ch = (key[2], key[1]) if key[0] == "f3" else (key[1], key[2])
            shape = key[3]
            planes, h, w, two_d = shape[0] * shape[2], shape[3], shape[4], False
        else:
            return False
        pair = two_d and w <= 15 and planes >= 2
        tiles = ((h + 15) // 16) * (1 if pair else (w + 31) // 32)
        wgs = tiles * ((planes + 1) // 2 if pair else planes) * ((out_ch + 63) // 64)
        filled = h * w * (2 if pair else 1) >= 0.45 * tiles * 512
        if out_ch < 16 or not filled:
            return False
        return wgs >= 150 or (two_d and wgs <= 128 and in_ch >= 256)
    except (TypeError, IndexError, ValueError):
        return False


def fixed_rule(names, key=None):
    """the route of a shape the table does not know: Winograd where the layer has it - F(4x4,3x3) on large maps (_wino4_pays), F(2x2,3x3)
    otherwise - else this package's direct kernel"""
    if "wino4" in names and key is not None and _wino4_pays(key):
        return "wino4"
    if "wino" in names:
        return "wino"
    return "hip" if "hip" in names else next(iter(names))


def _time(fn):
    fn()                                     # solver search / first touch outside the timed calls
    best = float("inf")
    for _ in range(3):                       # the best of three groups of three: one noisy group must not decide a layer
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def choose(key, fns):
    """the name of the route for ``key`` among ``fns`` (name -> callable, in order of preference for ties).  Table / fixed rule:
    no launch, no clock, the same answer in every process.  Measure mode: times each candidate once per key (outside stream
    captures), remembers the winner - and the timings, for tools/make_routes.py."""
    m = mode()
    if m != "measure":              # the answer is a function of (key, candidates): remembered - the string key costs more than the lookup
        try:
            hit = _state["memo"].get((key, tuple(fns)))
        except TypeError:           # an unhashable key (a list inside): not memoised
            hit = None
        if hit is not None:
            _state["used"][hit[1]] = hit[0]          # (used() stays the record of every decision asked for)
            return hit[0]
    ks = key_str(key)
    if m == "measure":
        got = _state["measured"].get(ks)
        if got is None:
            if torch.cuda.is_current_stream_capturing():
                return fixed_rule(fns, key)
            t = {name: _time(fn) for name, fn in fns.items()}
            got = _state["measured"][ks] = (min(t, key=t.get), t)
        r = got[0]
    else:
        r = _state["table"].get(ks) if m == "table" else None
        if r is None or r not in fns:
            if m == "table":
                _state["misses"].add(ks)
            r = fixed_rule(fns, key)
    _state["used"][ks] = r
    if m != "measure":
        try:
            _state["memo"][(key, tuple(fns))] = (r, ks)
        except TypeError:
            pass
    return r


def measured():
    """measure mode: {key string: (winner, {route: ms of three calls})}"""
    return dict(_state["measured"])


def used():
    """{key string: route} of every decision this process has asked for"""
    return dict(_state["used"])


def misses():
    """table mode: the key strings that were looked up and not found (they took the fixed rule)"""
    return sorted(_state["misses"])


def summary():
    """how many of the decisions of this process so far came from where (for bench lines)"""
    return {"mode": mode(), "table": os.path.basename(_state["path"]) if _state["path"] else None, "hash": table_hash(),
            "table_entries": len(_state["table"] or {}), "fixed_rule_lookups": len(_state["misses"]),
            "measured_shapes": len(_state["measured"])}
