#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING the reference.

Runs only in the development container (needs /root/reference and CPU torch);
nothing here travels to the GPU box except the .npz/.json files it writes.

The reference scripts cannot be imported (module-level imports of the un-vendored
DSGN / Stereo R-CNN repos, argparse at import time), and their hot loops are inline
in ``main()`` / ``__main__``.  So this script parses each file with ``ast`` and

  * executes top-level helper ``def``s and constant assignments as they stand
    (``denormalize``, ``normalize``, ``tensor2im``, ``save_img``, ``init_patch``,
    ``generate_round_mask``, ``kitti_output`` ...), and
  * lifts the hot-loop STATEMENTS out of ``main()`` by source line range and executes
    those very AST nodes in a namespace holding seeded inputs

so every expected output below was computed by the reference's own statements under
torch-CPU, not by a restatement.  What is synthetic is only the input: images, the
gradient that autograd would have left in ``img.grad``, patch start values.

No reference source text is written to the fixtures - only arrays, scalars, digests.

usage:  python tests/golden/make_golden.py           (rewrites tests/golden/*.npz)
"""
import ast
import hashlib
import io
import json
import os
import random
import sys
import tempfile
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = os.environ.get("ADV_REFERENCE_ROOT", "/root/reference")
warnings.filterwarnings("ignore")
torch.set_num_threads(1)


# --------------------------------------------------------------------------- AST helpers
def _parse(rel):
    path = os.path.join(REF, rel)
    with open(path) as f:
        src = f.read()
    return ast.parse(src, filename=path), path


def exec_toplevel(rel, names, ns):
    """exec top-level FunctionDef / Assign nodes whose (target) name is in `names`."""
    tree, path = _parse(rel)
    picked = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            picked.append(node)
        elif isinstance(node, ast.Assign) and any(
                isinstance(t, ast.Name) and t.id in names for t in node.targets):
            picked.append(node)
    found = {n.name if isinstance(n, ast.FunctionDef) else n.targets[0].id for n in picked}
    missing = set(names) - found
    assert not missing, (rel, missing)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


def exec_lines(rel, lo, hi, ns):
    """exec the outermost statements of `rel` lying wholly inside source lines [lo, hi]."""
    tree, path = _parse(rel)
    picked = []

    def visit(body):
        for node in body:
            if node.lineno >= lo and node.end_lineno <= hi:
                picked.append(node)
                continue
            for field in ("body", "orelse", "finalbody"):
                sub = getattr(node, field, None)
                if isinstance(sub, list) and sub and isinstance(sub[0], ast.stmt):
                    visit(sub)

    visit(tree.body)
    assert picked, (rel, lo, hi)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def leaf(a):
    t = torch.from_numpy(np.array(a, copy=True))
    t.requires_grad = True
    return t


# --------------------------------------------------------------------------- DSGN PGD
DSGN_PGD = "attack/DSGN/pgd_attack.py"


def dsgn_pgd_case(seed, h, w, crop_h, crop_w, alpha, eps, n_iter, grad_scale=1.0,
                  specials=False, keep_arrays=True, padded=False):
    from PIL import Image
    ns = {"torch": torch, "np": np, "Image": Image}
    exec_toplevel(DSGN_PGD, ["mean", "std", "tensor2im", "save_img", "denormalize", "normalize"], ns)
    if padded:      # the image is crop_h x crop_w, zero-padded in normalised space to the network size h x w
        x0L = synth.dsgn_padded(seed, crop_h, crop_w, h, w)
        x0R = synth.dsgn_padded(seed + 1, crop_h, crop_w, h, w)
    else:
        x0L = synth.dsgn_normalised(seed, h, w)
        x0R = synth.dsgn_normalised(seed + 1, h, w)
    ns.update(imgL=torch.from_numpy(x0L.copy()), imgR=torch.from_numpy(x0R.copy()),
              alpha=alpha, eps=eps)
    exec_lines(DSGN_PGD, 254, 255, ns)          # ori_img*_data
    exec_lines(DSGN_PGD, 297, 298, ns)          # clean_img*_data
    cleanL = ns["clean_imgL_data"].numpy().copy()
    cleanR = ns["clean_imgR_data"].numpy().copy()
    assert np.array_equal(cleanL, ns["ori_imgL_data"].numpy())

    def export(t):
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "x.png")
            ns["save_img"](t.clone().detach_().cpu()[0], p, crop_w, crop_h)
            return np.array(Image.open(p).convert("RGB"))

    out = {"x0L": x0L, "x0R": x0R, "cleanL": cleanL, "cleanR": cleanR,
           "u8L_0": export(ns["imgL"]), "u8R_0": export(ns["imgR"])}
    digests = {}
    for k in range(n_iter):
        gL = synth.gradient(1000 * seed + 2 * k, x0L.shape, grad_scale, specials=specials)
        gR = synth.gradient(1000 * seed + 2 * k + 1, x0R.shape, grad_scale, specials=specials)
        ns["imgL"].requires_grad = True           # pgd_attack.py:305-306
        ns["imgR"].requires_grad = True
        ns["imgL"].grad = torch.from_numpy(gL.copy())
        ns["imgR"].grad = torch.from_numpy(gR.copy())
        exec_lines(DSGN_PGD, 339, 354, ns)        # THE reference PGD step
        xl, xr = ns["imgL"].numpy().copy(), ns["imgR"].numpy().copy()
        u8l, u8r = export(ns["imgL"]), export(ns["imgR"])
        digests["xL_%d" % (k + 1)] = sha(xl)
        digests["xR_%d" % (k + 1)] = sha(xr)
        digests["u8L_%d" % (k + 1)] = sha(u8l)
        digests["u8R_%d" % (k + 1)] = sha(u8r)
        if keep_arrays:
            out["gL_%d" % k], out["gR_%d" % k] = gL, gR
            out["xL_%d" % (k + 1)], out["xR_%d" % (k + 1)] = xl, xr
            out["u8L_%d" % (k + 1)], out["u8R_%d" % (k + 1)] = u8l, u8r
    meta = dict(seed=seed, h=h, w=w, crop_h=crop_h, crop_w=crop_w, alpha=alpha, eps=eps,
                n_iter=n_iter, grad_scale=grad_scale, specials=specials, digests=digests)
    if padded:
        meta["padded"] = True
    if not keep_arrays:
        out = {k: v for k, v in out.items() if k.startswith("u8") and False}
    return out, meta


# --------------------------------------------------------------------------- S-RCNN PGD
SR_PGD = "attack/Stereo-RCNN/pgd_attack.py"


def srcnn_pgd_case(seed, h, w, alpha, eps_arg, n_iter, grad_scale=1.0, specials=False,
                   keep_arrays=True, zero_ties=False):
    cfg = types.SimpleNamespace(PIXEL_MEANS=np.array([[list(synth.SRCNN_PIXEL_MEANS)]]))
    ns = {"torch": torch, "np": np, "cfg": cfg}
    x0L = synth.srcnn_meansub(seed, h, w)
    x0R = synth.srcnn_meansub(seed + 1, h, w)
    if zero_ties:   # +-0 and tiny values, so that clamp bounds of -0.0 / +0.0 (eps = 0) meet zeros of either sign
        for a, s0 in ((x0L, seed), (x0R, seed + 1)):
            rs = np.random.RandomState(s0)
            pick = rs.randint(0, 6, size=a.shape)
            a[pick == 0] = np.float32(0.0)
            a[pick == 1] = np.float32(-0.0)
            a[pick == 2] = np.float32(1e-42)
            a[pick == 3] = np.float32(-1e-42)
    eps = 255 * eps_arg                           # pgd_attack.py:57  (args.eps * 255)
    ns.update(im_left_data=torch.from_numpy(x0L.copy()), im_right_data=torch.from_numpy(x0R.copy()),
              alpha=alpha, eps=eps)
    exec_lines(SR_PGD, 122, 123, ns)              # clean_im_*_data = im_*_data.data
    out = {"x0L": x0L, "x0R": x0R}
    digests = {}

    def hwc_plus_means(name):
        # pgd_attack.py:233-236 (left) / 239-242 (right) minus the cv2.imwrite line
        sub = dict(ns)
        if name == "L":
            exec_lines(SR_PGD, 233, 236, sub)
            return sub["img_left"].copy()
        exec_lines(SR_PGD, 239, 242, sub)
        return sub["img_right"].copy()

    for k in range(n_iter):
        gL = synth.gradient(2000 * seed + 2 * k, x0L.shape, grad_scale, specials=specials)
        gR = synth.gradient(2000 * seed + 2 * k + 1, x0R.shape, grad_scale, specials=specials)
        ns["im_left_data"].requires_grad = True   # pgd_attack.py:153-154
        ns["im_right_data"].requires_grad = True
        ns["im_left_data"].grad = torch.from_numpy(gL.copy())
        ns["im_right_data"].grad = torch.from_numpy(gR.copy())
        exec_lines(SR_PGD, 177, 217, ns)          # THE reference PGD step
        xl, xr = ns["im_left_data"].numpy().copy(), ns["im_right_data"].numpy().copy()
        fl, fr = hwc_plus_means("L"), hwc_plus_means("R")
        digests["xL_%d" % (k + 1)] = sha(xl)
        digests["xR_%d" % (k + 1)] = sha(xr)
        digests["hwcL_%d" % (k + 1)] = sha(fl)
        if keep_arrays:
            out["gL_%d" % k], out["gR_%d" % k] = gL, gR
            out["xL_%d" % (k + 1)], out["xR_%d" % (k + 1)] = xl, xr
            out["hwcL_%d" % (k + 1)], out["hwcR_%d" % (k + 1)] = fl, fr
    meta = dict(seed=seed, h=h, w=w, alpha=alpha, eps_arg=eps_arg, eps=eps, n_iter=n_iter,
                grad_scale=grad_scale, specials=specials, digests=digests)
    return out, meta


# --------------------------------------------------------------------------- patch attacks
DSGN_PATCH = "attack/DSGN/patch_attack.py"
SR_PATCH = "attack/Stereo-RCNN/patch_attack.py"


def patch_case(model, seed, ratio, eps, iters, grad_scale, zero_patch=False):
    """Full-size (shape is hard-coded in the reference) paste + update trace."""
    if model == "dsgn":
        rel, H, W = DSGN_PATCH, 384, 1248
        L, R = "imgL", "imgR"
        pad_lines, paste_lines, upd_lines = (326, 333), (369, 376), (416, 430)
        mk = synth.dsgn_normalised
        plo, phi = -2.0, 2.5
    else:
        rel, H, W = SR_PATCH, 600, 1987
        L, R = "im_left_data", "im_right_data"
        pad_lines, paste_lines, upd_lines = (178, 185), (221, 230), (257, 281)
        mk = synth.srcnn_meansub
        plo, phi = -140.0, 170.0                  # beyond the per-channel clamp on both sides
    ns = {"torch": torch, "np": np, "nn": nn, "random": random, "os": os}
    exec_toplevel(rel, ["init_patch", "generate_round_mask"], ns)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            patch_dim, radius, patch0 = ns["init_patch"](ratio, "p")
        finally:
            os.chdir(cwd)
    assert patch0.shape == (1, 3, patch_dim, patch_dim) and not patch0.any()
    if not zero_patch:
        patch0 = synth.patch_init(seed + 5, patch_dim, plo, phi)
    random.seed(seed)
    center_l, center_r, mask_l, mask_r = ns["generate_round_mask"](radius)
    ns.update(center_l=center_l, center_r=center_r, radius=radius,
              mask_l=torch.from_numpy(mask_l), mask_r=torch.from_numpy(mask_r),
              patch=torch.from_numpy(patch0.copy()), alpha=1e3, eps=eps)
    exec_lines(rel, *pad_lines, ns)
    xL, xR = mk(seed + 10, H, W), mk(seed + 11, H, W)
    ns[L], ns[R] = leaf(xL), leaf(xR)
    cy, cxl, cxr, r = center_l[0], center_l[1], center_r[1], radius
    win = lambda a, cx: np.ascontiguousarray(a[:, :, cy - r:cy + r + 1, cx - r:cx + r + 1])
    out = {"patch_0": patch0, "mask_win": np.ascontiguousarray(mask_l[0, 0, cy - r:cy + r + 1, cxl - r:cxl + r + 1])}
    digests = {"mask_l": sha(mask_l[0, 0]), "mask_r": sha(mask_r[0, 0])}
    for k in range(iters):
        exec_lines(rel, *paste_lines, ns)         # THE reference paste
        pl, pr = ns[L].data.numpy(), ns[R].data.numpy()
        digests["pastedL_%d" % k], digests["pastedR_%d" % k] = sha(pl), sha(pr)
        out["pastedL_win_%d" % k], out["pastedR_win_%d" % k] = win(pl, cxl), win(pr, cxr)
        # autograd ACCUMULATES into the same leaf across inner iterations (the reference never
        # zeroes img.grad): emulate by adding this iteration's gradient to .grad
        gL = synth.gradient(3000 * seed + 2 * k, xL.shape, grad_scale)
        gR = synth.gradient(3000 * seed + 2 * k + 1, xR.shape, grad_scale)
        for name, g in ((L, gL), (R, gR)):
            t = ns[name]
            t.grad = torch.from_numpy(g.copy()) if t.grad is None else t.grad + torch.from_numpy(g)
        out["gradL_win_%d" % k] = win(ns[L].grad.numpy(), cxl)
        out["gradR_win_%d" % k] = win(ns[R].grad.numpy(), cxr)
        digests["gradL_%d" % k], digests["gradR_%d" % k] = sha(ns[L].grad.numpy()), sha(ns[R].grad.numpy())
        exec_lines(rel, *upd_lines, ns)           # THE reference patch update
        out["patch_%d" % (k + 1)] = ns["patch"].numpy().copy()
    meta = dict(model=model, seed=seed, ratio=ratio, eps=eps, iters=iters, grad_scale=grad_scale,
                zero_patch=zero_patch, H=H, W=W, patch_dim=int(patch_dim), radius=int(radius),
                center_l=[int(v) for v in center_l], center_r=[int(v) for v in center_r],
                mask_area=int(mask_l[0, 0].sum()), digests=digests)
    return out, meta


def mask_and_dims_cases():
    """init_patch dims for a sweep of ratios; generate_round_mask centres per seed, for the two
    trainers and for the two detect-under-attack scripts' atk_mode column bands."""
    res = {"init_patch": [], "centers": []}
    for model, rel, short in (("dsgn", DSGN_PATCH, 384), ("srcnn", SR_PATCH, 600)):
        ns = {"np": np, "os": os, "random": random}
        exec_toplevel(rel, ["init_patch", "generate_round_mask"], ns)
        for ratio in (0.05, 0.1, 0.13, 0.2, 0.25, 0.26, 0.2605, 0.3, 0.333, 0.5):
            cwd = os.getcwd()
            with tempfile.TemporaryDirectory() as td:
                os.chdir(td)
                try:
                    d, r, p = ns["init_patch"](ratio, "p")
                    saved = np.load(os.path.join(td, "p", "epoch0", "patch.npy"))
                finally:
                    os.chdir(cwd)
            assert saved.dtype == np.float32 and saved.shape == (1, 3, d, d)
            res["init_patch"].append(dict(model=model, ratio=ratio, patch_dim=int(d), radius=int(r)))
        r = {"dsgn": 38, "srcnn": 30}[model]
        for seed in range(6):
            random.seed(seed)
            cl, cr, ml, mr = ns["generate_round_mask"](r)
            res["centers"].append(dict(model=model, script="patch_attack", atk_mode="random", seed=seed,
                                       radius=r, center_l=[int(v) for v in cl], center_r=[int(v) for v in cr],
                                       area=int(ml[0, 0].sum()), mask_l=sha(ml[0, 0]), mask_r=sha(mr[0, 0]),
                                       mask_shape=list(ml.shape), mask_dtype=str(ml.dtype)))
    for model, rel in (("dsgn", "attack/DSGN/predict_and_save_patch.py"),
                       ("srcnn", "attack/Stereo-RCNN/predict_and_save_patch.py")):
        r = {"dsgn": 38, "srcnn": 30}[model]
        for mode in ("random", "sp_left", "sp_straight", "sp_right"):
            ns = {"np": np, "os": os, "random": random, "args": types.SimpleNamespace(atk_mode=mode)}
            exec_toplevel(rel, ["generate_round_mask"], ns)
            for seed in range(4):
                random.seed(seed)
                cl, cr, ml, mr = ns["generate_round_mask"](r)
                res["centers"].append(dict(model=model, script="predict_and_save_patch", atk_mode=mode,
                                           seed=seed, radius=r, center_l=[int(v) for v in cl],
                                           center_r=[int(v) for v in cr], area=int(ml[0, 0].sum()),
                                           mask_l=sha(ml[0, 0]), mask_r=sha(mr[0, 0]),
                                           mask_shape=list(ml.shape), mask_dtype=str(ml.dtype)))
    return res


# --------------------------------------------------------------------------- label writer
def label_case():
    """kitti_output's text formatting (attack/DSGN/predict_and_save_pgd.py:250-284).
    `get_dimensions` is upstream DSGN code that is not in the reference; it is stubbed with a
    function that reads (h, w, l, ry) off the test vector, so only the reference's own centre /
    alpha / y-shift arithmetic and its format string are pinned here."""
    rel = "attack/DSGN/predict_and_save_pgd.py"
    rs = np.random.RandomState(7)
    n = 5
    labels = torch.tensor([2, 2, 1, 3, 2])
    bbox = torch.from_numpy((rs.rand(n, 4) * 300).astype(np.float32))
    scores = torch.from_numpy(rs.rand(n).astype(np.float32))
    corners = torch.from_numpy((rs.randn(n, 24) * 3 + 10).astype(np.float32))
    dims = (rs.rand(n, 4) * 3).astype(np.float32)
    state = {"i": 0}

    def get_dimensions(c):
        d = dims[state["i"]]
        state["i"] += 1
        return float(d[0]), float(d[1]), float(d[2]), float(d[3] - 1.5)

    class Pred:
        def __init__(self):
            self.bbox = bbox
            self.f = {"labels": labels, "scores": scores, "box_corner3d": corners}

        def get_field(self, k):
            return self.f[k]

        def has_field(self, k):
            return k in self.f

    cfg = types.SimpleNamespace(learn_viewpoint=False)
    ns = {"np": np, "os": os, "torch": torch, "cfg": cfg, "get_dimensions": get_dimensions,
          "print": lambda *a, **k: None}
    exec_toplevel(rel, ["kitti_output"], ns)
    with tempfile.TemporaryDirectory() as td:
        ns["kitti_output"]([Pred()], [42], td)
        with open(os.path.join(td, "000042.txt")) as f:
            text = f.read()
    return dict(labels=labels.tolist(), bbox=bbox.numpy().tolist(), scores=scores.numpy().tolist(),
                corners=corners.numpy().tolist(),
                dims=[[float(d[0]), float(d[1]), float(d[2]), float(d[3] - 1.5)] for d in dims],
                image_index=42, text=text)


# --------------------------------------------------------------------------- depth statistics
def depth_stats_case():
    """error_estimating / depth_error_estimating / project_disp_to_depth_map / project_disp_to_depth of
    attack/DSGN/predict_and_save_pgd.py:202-247,304-329 executed as they stand (cfg.max_depth and the upstream
    calibration object replaced by plain stand-ins: P, f_u, and an identity project_image_to_velo)."""
    rel = "attack/DSGN/predict_and_save_pgd.py"
    cfg = types.SimpleNamespace(max_depth=40.4)
    ns = {"torch": torch, "np": np, "cfg": cfg}
    exec_toplevel(rel, ["error_estimating", "depth_error_estimating", "project_disp_to_depth_map", "project_disp_to_depth"], ns)
    pred, gt = synth.depth_stats_inputs(77)
    P = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
    PR = P.copy()
    PR[0, 3] = -339.5242
    calib = types.SimpleNamespace(P=P, f_u=721.5377, project_image_to_velo=lambda pts: pts)
    calib_R = types.SimpleNamespace(P=PR)
    tp, tg = torch.from_numpy(pred), torch.from_numpy(gt)
    out = {"seed": 77, "max_depth": cfg.max_depth, "P": P.tolist(), "P_R": PR.tolist(), "f_u": 721.5377}
    out["error_estimating"] = list(ns["error_estimating"](tp, tg))
    out["error_estimating_maxdisp50"] = list(ns["error_estimating"](tp, tg, maxdisp=50))
    out["error_estimating_valid_images"] = list(ns["error_estimating"](tp[[0, 2]], tg[[0, 2]]))
    out["depth_error_depth"] = list(ns["depth_error_estimating"](tp, tg, depth_disp=True, calib_batch=[calib] * 3, calib_R_batch=[calib_R] * 3))
    disp = torch.from_numpy(np.abs(pred) + 1)
    out["depth_error_disp"] = list(ns["depth_error_estimating"](disp, tg, depth_disp=False, calib_batch=[calib] * 3, calib_R_batch=[calib_R] * 3))
    out["depth_map_depth"] = sha(ns["project_disp_to_depth_map"](calib, pred[0].copy(), max_high=1., baseline=0.54, depth_disp=True))
    out["depth_map_disp"] = sha(ns["project_disp_to_depth_map"](calib, pred[2].copy(), max_high=1., baseline=0.532, depth_disp=False))
    cloud = ns["project_disp_to_depth"](calib, pred[0].copy(), max_high=30., baseline=0.54, depth_disp=True)
    out["cloud_rows"], out["cloud"] = int(cloud.shape[0]), sha(cloud)
    return out


# --------------------------------------------------------------------------- consumer side
def scenario_case(label_text):
    """What the reference's own consumer makes of a label file: ``load_label`` (evaluation/convert_scenarios.py:52-95,
    executed as it stands) and the obstacle loop of ``convert_scenario`` (:116-133, the very statements) with the
    CommonRoad classes replaced by recorders (CommonRoad is not installed here; only constructor arguments matter)."""
    rel = "evaluation/convert_scenarios.py"
    ns = {"np": np, "os": os}
    exec_toplevel(rel, ["load_label"], ns)
    extra = ("Van -1 -1 0.1 1 2 3 4 2.0 1.9 5.1 -3.5 1.6 14.25 4.5 0.7\n"        # ry > pi: wrapped by the consumer
             "Truck -1 -1 0.1 1 2 3 4 3.0 2.5 9.0 6.0 1.7 30.0 -3.9 0.6\n"       # ry < -pi
             "Tram -1 -1 0.1 1 2 3 4 3.0 2.5 9.0 6.0 1.7 30.0 0.0 0.6\n")        # ignored type
    text = label_text + extra
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "000042.txt")
        with open(path, "w") as f:
            f.write(text)
        label = ns["load_label"](path)
    added = []

    class Rec:
        def __init__(self, *a, **k):
            self.a, self.k = a, k

    class Scenario:
        n = 100

        def generate_object_id(self):
            Scenario.n += 1
            return Scenario.n

        def add_objects(self, o):
            shape, state = o.a[2], o.a[3]
            added.append(dict(id=o.a[0], width=float(shape.k["width"]), length=float(shape.k["length"]),
                              position=[float(v) for v in state.k["position"]], orientation=float(state.k["orientation"]),
                              time_step=int(state.k["time_step"])))

    ns.update(label=label, scenario=Scenario(), Rectangle=Rec, State=Rec, StaticObstacle=Rec,
              ObstacleType=types.SimpleNamespace(PARKED_VEHICLE="parked"))
    exec_lines(rel, 116, 133, ns)
    return dict(text=text, label=label, obstacles=added)


# --------------------------------------------------------------------------- main
def cli_flags_case():
    """the command-line surface of the eight attack / detect scripts: per script, every
    ``add_argument`` call's option strings, default VALUE (the default expression evaluated), type and action - data read off
    the reference's own parser set-up, for tests/test_host_logic.py to hold this package's CLIs against"""
    out = {}
    for model in ("DSGN", "Stereo-RCNN"):
        for script in ("pgd_attack", "patch_attack", "predict_and_save_pgd", "predict_and_save_patch"):
            rel = "attack/%s/%s.py" % (model, script)
            tree, _ = _parse(rel)
            flags = []
            for n in ast.walk(tree):
                if not (isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr == "add_argument"):
                    continue
                entry = {"options": [a.value for a in n.args if isinstance(a, ast.Constant)], "lineno": n.lineno}
                for k in n.keywords:
                    if k.arg == "default":
                        entry["default"] = eval(compile(ast.Expression(k.value), rel, "eval"), {})
                    elif k.arg == "type":
                        entry["type"] = ast.unparse(k.value)
                    elif k.arg in ("action", "dest"):
                        entry[k.arg] = ast.literal_eval(k.value)
                flags.append(entry)
            out[rel] = sorted(flags, key=lambda e: e["lineno"])
    return out


def objective_cases():
    """a4 / a14: the OBJECTIVE statements of the two PGD scripts - attack/DSGN/pgd_attack.py:269-270 (mask), :301-336 (forward call,
    depth term, RPN3D term, zero_grad, backward) and attack/Stereo-RCNN/pgd_attack.py:153-174 (forward call, the six
    uncertainty-weighted terms, backward) - executed as they stand around the stand-in detectors of stub_models.py.  Stored: the
    loss and what ``backward()`` left in the image gradients.  Inputs regenerate from the seeds (synth + torch.Generator)."""
    import torch.nn.functional as F
    import stub_models
    out = {}
    arrays = {}
    h, w = 10, 14
    for name, loss_disp, rpn in (("dsgn_both", True, True), ("dsgn_depth_only", True, False), ("dsgn_rpn_only", False, True)):
        cfg = types.SimpleNamespace(PlaneSweepVolume=True, loss_disp=loss_disp, RPN3D_ENABLE=rpn, min_depth=2.0, max_depth=40.4, stub_gain=0.7)
        gen = torch.Generator().manual_seed(41)
        disp_true = torch.rand((1, h, w), generator=gen) * 50.0
        targets = (torch.randn((h, w), generator=gen),)
        ns = {"torch": torch, "F": F, "cfg": cfg, "RPN3DLoss": stub_models.StubRpn3dLoss, "model": stub_models.StubDsgn(5).eval(),
              "imgL": torch.from_numpy(synth.dsgn_normalised(51, h, w).copy()), "imgR": torch.from_numpy(synth.dsgn_normalised(52, h, w).copy()),
              "disp_true": disp_true, "targets": targets, "calib": None, "calib_R": None, "ious": 0.3, "labels_map": None,
              "calibs_fu": torch.tensor([721.5377]), "calibs_baseline": torch.tensor([0.54]),
              "calibs_Proj": torch.arange(12, dtype=torch.float64).view(1, 3, 4) / 10, "calibs_Proj_R": torch.ones(1, 3, 4, dtype=torch.float64)}
        exec_lines(DSGN_PGD, 269, 270, ns)            # mask = (disp_true > cfg.min_depth) & (disp_true <= cfg.max_depth); detach_
        exec_lines(DSGN_PGD, 301, 302, ns)            # loss = 0.; losses = dict()
        exec_lines(DSGN_PGD, 305, 306, ns)            # requires_grad
        exec_lines(DSGN_PGD, 308, 336, ns)            # forward, both terms, zero_grad, retain_grad, backward
        arrays[name + "_gradL"], arrays[name + "_gradR"] = ns["imgL"].grad.numpy().copy(), ns["imgR"].grad.numpy().copy()
        arrays[name + "_loss"] = ns["loss"].detach().numpy().copy()
        out[name] = {"h": h, "w": w, "loss_disp": loss_disp, "RPN3D_ENABLE": rpn, "loss": float(ns["loss"]), "mask_count": int(ns["mask"].sum()),
                     "loss_keys": sorted(ns["losses"].keys())}
    gen = torch.Generator().manual_seed(43)
    uncert = torch.randn(6, generator=gen) * 0.4
    ns = {"torch": torch, "stereoRCNN": stub_models.StubStereoRcnn(6).eval(), "uncert": uncert,
          "im_left_data": torch.from_numpy(synth.srcnn_meansub(61, h, w).copy()), "im_right_data": torch.from_numpy(synth.srcnn_meansub(62, h, w).copy()),
          "im_info": torch.tensor([[float(h), float(w), 1.6]]), "gt_boxes_left": torch.full((1, 30, 5), 0.25), "gt_boxes_right": torch.zeros(1, 30, 5),
          "gt_boxes_merge": torch.zeros(1, 30, 5), "gt_dim_orien": torch.zeros(1, 30, 5), "gt_kpts": torch.zeros(1, 30, 6), "num_boxes": torch.tensor([1])}
    exec_lines("attack/Stereo-RCNN/pgd_attack.py", 153, 174, ns)
    arrays["srcnn_gradL"], arrays["srcnn_gradR"] = ns["im_left_data"].grad.numpy().copy(), ns["im_right_data"].grad.numpy().copy()
    arrays["srcnn_loss"] = ns["loss"].detach().numpy().copy()
    arrays["srcnn_uncert"] = uncert.numpy().copy()
    out["srcnn"] = {"h": h, "w": w, "loss": float(ns["loss"])}
    out["bytes"] = save_npz("objectives.npz", arrays)
    return out


def upsample_add_case():
    """the FPN top-down step: the reference's own ``_upsample_add`` (attack/Stereo-RCNN/stereo_rcnn.py:91-108), lifted as a function and
    executed on seeded maps - the pyramid's ragged ~2x steps and an exact 2x - with what backward() leaves in the small map.  Pins
    oracle_np.bilinear_up / bilinear_up_bwd and csrc/resize.hip (float32 rounding apart: torch's CPU kernel associates the four products
    its own way)."""
    import torch.nn.functional as F
    ns = exec_lines("attack/Stereo-RCNN/stereo_rcnn.py", 91, 108, {"F": F, "torch": torch})
    fn = ns["_upsample_add"]
    arrays, out = {}, {"cases": []}
    gen = torch.Generator().manual_seed(77)
    for i, (hw, size) in enumerate((((5, 8), (10, 15)), ((4, 6), (8, 12)), ((19, 13), (38, 25)), ((7, 9), (7, 9)))):
        x = torch.randn((2, 3) + hw, generator=gen)
        y = torch.randn((2, 3) + size, generator=gen)
        g = torch.randn((2, 3) + size, generator=gen)
        xl = x.clone().requires_grad_(True)
        res = fn(None, xl, y)
        res.backward(g)
        for k, v in (("x", x), ("y", y), ("g", g), ("out", res.detach()), ("grad_x", xl.grad)):
            arrays["%s%d" % (k, i)] = v.numpy().copy()
        out["cases"].append({"in": list(hw), "out": list(size)})
    out["bytes"] = save_npz("upsample_add.npz", arrays)
    return out


def pyramid_roi_case():
    """RoI pooling over the feature pyramid: the reference's own ``PyramidRoI_Feat`` (attack/Stereo-RCNN/stereo_rcnn.py:110-141) lifted as
    a function and executed - level per roi, per-level pooling at ``feat.size(2) / im_info[0][0]``, concatenation, sort back into roi
    order - with the per-level operator held fixed (``RCNN_roi_align`` = the oracle's RoIAlign, 7x7; ``RCNN_roi_kpts_align`` = 14x14).
    Pins surrogates.pyramid_roi_feat and ops.PyramidRoIAlign (level assignment, scales, row order)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import oracle_np as O
    ns = exec_lines("attack/Stereo-RCNN/stereo_rcnn.py", 110, 141, {"torch": torch})
    fn = ns["PyramidRoI_Feat"]

    def pool(p):
        return lambda feat, rois, scale: torch.from_numpy(O.roi_align(feat.numpy(), rois.numpy(), p, float(scale)))
    me = types.SimpleNamespace(RCNN_roi_align=pool(7), RCNN_roi_kpts_align=pool(14))
    gen = torch.Generator().manual_seed(91)
    im_info = torch.tensor([[600.0, 1000.0, 1.6]])
    feats = [torch.randn((1, 2, int(np.ceil(600 / s)), int(np.ceil(1000 / s))), generator=gen) for s in (4, 8, 16, 32)]
    sizes = [18, 30, 47, 60, 90, 130, 140, 200, 230, 330, 380, 520, 45, 300]            # square-ish boxes across the four levels, ragged order
    rois = []
    for i, sz in enumerate(sizes):
        x1, y1 = float(20 + 37 * i), float(10 + 23 * (i % 7))
        rois.append([0.0, x1, y1, x1 + sz * (1.0 + 0.1 * (i % 3)), y1 + sz * (1.0 - 0.05 * (i % 4))])
    rois = torch.tensor(rois)
    arrays = {"rois": rois.numpy().copy(), "im_info": im_info.numpy().copy()}
    for l, f in enumerate(feats):
        arrays["feat%d" % l] = f.numpy().copy()
    arrays["pooled7"] = fn(me, feats, rois, im_info).numpy().copy()
    arrays["pooled14"] = fn(me, feats, rois, im_info, kpts=True).numpy().copy()
    h = rois[:, 4] - rois[:, 2] + 1
    w = rois[:, 3] - rois[:, 1] + 1
    levels = torch.round(torch.log(torch.sqrt(h * w) / 224.0) + 4).clamp(2, 5)
    return {"bytes": save_npz("pyramid_roi.npz", arrays), "rois": len(sizes), "levels_used": sorted(set(int(v) for v in levels))}


def upstream_call_sites_case():
    """How the reference's own files USE the upstream operator package ``model.roi_layers`` (a compiled CUDA extension upstream;
    eval_driving_safety_amd/upstream_shims/roi_layers.py here): names imported, constructor and call arities, keyword names - read
    off the ASTs, so the shim's signatures are checked against the call sites themselves."""
    out = {"imports": {}, "calls": {}}
    files = ["attack/Stereo-RCNN/stereo_rcnn.py", "attack/Stereo-RCNN/pgd_attack.py", "attack/Stereo-RCNN/patch_attack.py",
             "attack/Stereo-RCNN/predict_and_save_pgd.py", "attack/Stereo-RCNN/predict_and_save_patch.py", "attack/Stereo-RCNN/stereo_rpn.py"]
    for rel in files:
        tree, _ = _parse(rel)
        names = []
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module == "model.roi_layers":
                names += [a.name for a in node.names]
        if names:
            out["imports"][rel] = sorted(names)
        # attributes assigned from ROIAlign(...): their later calls are ROIAlign.forward call sites
        holders = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and getattr(node.value.func, "id", None) == "ROIAlign":
                for t in node.targets:
                    if isinstance(t, ast.Attribute):
                        holders.add(t.attr)
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            f = node.func
            what = None
            if isinstance(f, ast.Name) and f.id in ("ROIAlign", "nms"):
                what = f.id
            elif isinstance(f, ast.Attribute) and f.attr in holders:
                what = "ROIAlign.forward"
            if what:
                out["calls"].setdefault(what, []).append({"file": rel, "line": node.lineno, "positional": len(node.args),
                                                          "keywords": sorted(k.arg for k in node.keywords)})
    for v in out["calls"].values():
        v.sort(key=lambda c: (c["file"], c["line"]))
    return out


def save_npz(name, arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    return os.path.getsize(path)


def main():
    index = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}

    pgd = [
        # name,               seed  h   w  ch  cw  alpha     eps      N  gscale specials
        ("dsgn_pgd_default",    1, 24, 40, 21, 37, 1 / 255,  0.3,     4, 1.0, False),   # script defaults (:53-55)
        ("dsgn_pgd_fgsm",       2, 24, 40, 24, 40, 8 / 255,  8 / 255, 1, 1.0, False),   # BASELINE config 1
        ("dsgn_pgd_cfg2",       3, 16, 28, 15, 27, 1 / 255,  0.03,   20, 1e-3, False),  # BASELINE config 2
        ("dsgn_pgd_specials",   4, 12, 20, 12, 20, 2 / 255,  0.05,    3, 1.0, True),    # nan/inf/-0 gradients
        ("dsgn_pgd_ragged",     5,  7, 13,  5, 11, 1 / 255,  0.01,    3, 1.0, False),   # W not a multiple of 4
    ]
    for name, seed, h, w, ch, cw, a, e, n, gs, sp in pgd:
        arrays, meta = dsgn_pgd_case(seed, h, w, ch, cw, a, e, n, gs, sp)
        meta["bytes"] = save_npz(name + ".npz", arrays)
        index["cases"][name] = meta
    # one full-size KITTI-shaped pair: digests only, inputs regenerate from the seed
    _, meta = dsgn_pgd_case(6, 384, 1248, 375, 1242, 1 / 255, 0.03, 2, 1e-3, False, keep_arrays=False)
    index["cases"]["dsgn_pgd_fullsize"] = meta
    # the same with the loader's zero padding (normalised space) around a crop_h x crop_w image: small with arrays,
    # and one KITTI-shaped 375x1242 pair inside 384x1248 as digests
    arrays, meta = dsgn_pgd_case(8, 16, 28, 13, 22, 1 / 255, 0.03, 4, 1.0, False, padded=True)
    meta["bytes"] = save_npz("dsgn_pgd_padded.npz", arrays)
    index["cases"]["dsgn_pgd_padded"] = meta
    _, meta = dsgn_pgd_case(9, 384, 1248, 375, 1242, 1 / 255, 0.03, 2, 1e-3, False, keep_arrays=False, padded=True)
    index["cases"]["dsgn_pgd_fullsize_padded"] = meta

    sr = [
        ("srcnn_pgd_default",  11, 20, 33, 1.0, 0.3,  4, 1.0, False),   # script defaults (:41-44)
        ("srcnn_pgd_cfg3",     12, 14, 27, 1.0, 0.03, 20, 1.0, False),  # BASELINE config 3
        ("srcnn_pgd_specials", 13, 10, 18, 2.5, 0.05, 3, 1.0, True),
    ]
    for name, seed, h, w, a, e, n, gs, sp in sr:
        arrays, meta = srcnn_pgd_case(seed, h, w, a, e, n, gs, sp)
        meta["bytes"] = save_npz(name + ".npz", arrays)
        index["cases"][name] = meta
    # eps = 0 and alpha = 0 / tiny with signed zeros in the image: pins how torch.clamp resolves +-0 ties
    for name, seed, a, e in (("srcnn_pgd_zero_eps", 15, 0.0, 0.0), ("srcnn_pgd_zero_eps_tiny_alpha", 16, 1e-42, 0.0)):
        arrays, meta = srcnn_pgd_case(seed, 9, 16, a, e, 3, 1.0, True, zero_ties=True)
        meta["bytes"] = save_npz(name + ".npz", arrays)
        index["cases"][name] = meta
    _, meta = srcnn_pgd_case(14, 600, 1987, 1.0, 0.03, 2, 1.0, False, keep_arrays=False)
    index["cases"]["srcnn_pgd_fullsize"] = meta

    pt = [
        ("dsgn_patch_default", "dsgn", 21, 0.2, 8 / 255, 2, 4e-5, False),   # script defaults (:53-56)
        ("dsgn_patch_zero",    "dsgn", 22, 0.2, 8 / 255, 2, 4e-5, True),    # fresh zeros patch (init_patch)
        ("dsgn_patch_100px",   "dsgn", 23, 0.2605, 8 / 255, 2, 4e-5, False),  # BASELINE config 4 (D=101)
        ("srcnn_patch_default", "srcnn", 31, 0.1, 0.1, 2, 1.5e-4, False),   # script defaults (:43-47)
    ]
    for name, model, seed, ratio, eps, iters, gs, zp in pt:
        arrays, meta = patch_case(model, seed, ratio, eps, iters, gs, zp)
        meta["bytes"] = save_npz(name + ".npz", arrays)
        index["cases"][name] = meta

    index["masks"] = mask_and_dims_cases()
    index["label"] = label_case()
    index["scenario"] = scenario_case(index["label"]["text"])
    index["depth_stats"] = depth_stats_case()
    index["cli_flags"] = cli_flags_case()
    index["objectives"] = objective_cases()
    index["upstream_call_sites"] = upstream_call_sites_case()
    index["upsample_add"] = upsample_add_case()
    index["pyramid_roi"] = pyramid_roi_case()
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE)
              if f.endswith((".npz", ".json")))
    print("wrote %d cases, %.1f KiB" % (len(index["cases"]), tot / 1024))


if __name__ == "__main__":
    main()
