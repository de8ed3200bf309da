#!/usr/bin/env python3
"""Timeline of the F(4x4,3x3) kernel's stage loop from in-kernel s_memtime stamps (a -DADV_WINO4_STAMPS build of csrc/wino4.hip:
tools/build_variant.sh stamps wino4.hip -DADV_WINO4_STAMPS; run with ADVENGINE_LIB=tools/_build/libadv_stamps.so).  Per wave: cycles from the
stage's start to each half-step (sampled: stage s stamps half-step s % n), to the last matrix instruction, to the LDS drain, to the barrier's release."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eval_driving_safety_amd import ops, _lib  # noqa: E402


def run(name, fn, half_steps, last_stage=60, life_only=False):
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros((8, 8, 64, 5), dtype=np.uint64)
    rc = lib.adv_debug_wino4_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    assert rc == 0, rc
    s = buf.astype(np.int64)
    life = s[:, :, 63, :]                               # kernel entry, first stage's start, loop end, kernel end (after the stores have left)
    okl = life[..., 0] > 0
    if okl.any():
        print(json.dumps({"case": name, "workgroup_life_cycles_median": {"prologue": int(np.median((life[..., 1] - life[..., 0])[okl])),
                          "stage_loop": int(np.median((life[..., 2] - life[..., 1])[okl])), "epilogue_incl_store_drain": int(np.median((life[..., 3] - life[..., 2])[okl]))}}), flush=True)
    p0 = s[:, :, 60, :]                                # the prologue: set-up done, requests issued, tiles landed, first barrier passed
    okp = p0[..., 0] > 0
    if okp.any():
        medp = lambda a: int(np.median(a[okp]))
        print(json.dumps({"case": name, "prologue_cycles_median": {
            "entry_to_setup_done": medp(p0[..., 0] - life[..., 0]), "requests_issued": medp(p0[..., 1] - p0[..., 0]), "wait_for_tiles": medp(p0[..., 2] - p0[..., 1]),
            "edge_fix_and_barrier": medp(p0[..., 3] - p0[..., 2]), "first_transform_and_barrier": medp(life[..., 1] - p0[..., 3])}}), flush=True)
    e1, e2 = s[:, :, 61, :], s[:, :, 62, :]          # the epilogue: round 0 in four steps, then every round's end
    oke = e1[..., 0] > 0
    if oke.any():
        med = lambda a: int(np.median(a[oke]))
        print(json.dumps({"case": name, "epilogue_cycles_median": {
            "loop_end_to_epilogue_start": med(e1[..., 0] - life[..., 2]), "round0_lds_writes_issued": med(e1[..., 1] - e1[..., 0]), "round0_barrier": med(e1[..., 2] - e1[..., 1]),
            "round0_reads_and_transform": med(e1[..., 3] - e1[..., 2]), "round0_stores_issued": med(e1[..., 4] - e1[..., 3]), "round0_end_barrier": med(e2[..., 0] - e1[..., 4]),
            "round1": med(e2[..., 1] - e2[..., 0]), "round2": med(e2[..., 2] - e2[..., 1]), "round3": med(e2[..., 3] - e2[..., 2]),
            "store_drain": med(life[..., 3] - np.maximum.reduce([e2[..., 0], e2[..., 1], e2[..., 2], e2[..., 3]]))}}), flush=True)
    for wave in range(0 if not life_only else 8, 8):
        ns = half_steps[wave]
        st = s[:, wave, 4:last_stage, :]                        # steady-state stages of the eight stamped workgroups
        ok = st[..., 0] > 0
        if not ok.any():
            continue
        total = (st[..., 4] - st[..., 0])[ok]
        last = (st[..., 2] - st[..., 0])[ok]
        drain = (st[..., 3] - st[..., 2])[ok]
        bar = (st[..., 4] - st[..., 3])[ok]
        prof = {}
        stages = np.arange(4, last_stage)
        for h in range(ns):
            sel = (stages % ns) == h
            d = (st[:, sel, 1] - st[:, sel, 0])[ok[:, sel]]
            if d.size:
                prof[h] = int(np.median(d))
        print(json.dumps({"case": name, "wave": wave, "stage_cycles_median": int(np.median(total)), "to_last_matrix_issue": int(np.median(last)),
                          "lds_drain": int(np.median(drain)), "barrier_wait": int(np.median(bar)), "barrier_wait_p90": int(np.percentile(bar, 90)),
                          "half_step_start": prof}), flush=True)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if "--small" in sys.argv:      # a map with fewer tiles than compute units: the workgroup's life, whole and as parts of a K-split launch
        xs = torch.randn((2, 256, 38, 125), device=dev)
        ps = ops.ConvWino4Prep(torch.randn((256, 256, 3, 3), device=dev) * 0.02)
        for tile in (0, 1):
            for splits in (1,):
                run("256->256 [2,256,38,125] tile %d, %d part(s)" % (tile, splits), lambda: ops.conv_wino4(xs, ps, tile=tile, splits=splits), [36] * 8, life_only=True)
        x3 = torch.randn((1, 32, 48, 96, 312), device=dev)
        p3 = ops.ConvWino4Prep(torch.randn((32, 32, 3, 3, 3), device=dev) * 0.05)
        b3 = torch.randn((32,), device=dev)
        run("3D 32->32 [48,96,312] tile 3 (whole chip)", lambda: ops.conv_wino4(x3, p3, b3, None, True, tile=3), [18] * 8, last_stage=22, life_only=True)
        return
    x = torch.randn((2, 256, 150, 497), device=dev)
    prep = ops.ConvWino4Prep(torch.randn((256, 256, 3, 3), device=dev) * 0.02)
    for dbg in [int(a) for a in sys.argv[1:]] or [0]:        # phase ablations need the -DADV_TEST_HOOKS build of the stamped library
        if dbg:
            os.environ["ADV_WINO4_DBG"] = str(dbg)
        run("256->256 [2,256,150,497] tile 1 dbg %d" % dbg, lambda: ops.conv_wino4(x, prep, tile=1), [36] * 8, last_stage=31)
        os.environ.pop("ADV_WINO4_DBG", None)
    x3 = torch.randn((1, 32, 48, 96, 312), device=dev)
    p3 = ops.ConvWino4Prep(torch.randn((32, 32, 3, 3, 3), device=dev) * 0.05)
    run("3D 32->32 [48,96,312] tile 3", lambda: ops.conv_wino4(x3, p3, tile=3), [18] * 8, last_stage=22)


if __name__ == "__main__":
    main()
