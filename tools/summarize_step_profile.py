#!/usr/bin/env python3
"""kernel_stats.csv of a profiled end-to-end leg (tools/gpu_profile_cmd.sh over tools/bench_end_to_end.py) -> who computes the step:
shares of libadvengine's kernels, MIOpen / rocBLAS / CK, torch's element-wise and copy kernels.  MIOpen's one-time solver search (its
``naive_conv_*`` reference kernels, run once per new layer shape before the timed steps) is listed apart and excluded from the shares.
With a third argument - the kernel statistics of the WARM-UP alone (tools/gpu_profile_step.sh) - every kernel's warm-up time and calls are
subtracted first: what remains are the steady-state steps.
usage: tools/summarize_step_profile.py gpurun_out/prof_r04r101/kernel_stats.csv [out.json [warmup_kernel_stats.csv]]"""
import csv
import json
import re
import sys

OURS = re.compile(r"conv_wino|bilinear_up_\w+|conv2d_\w+_mfma|conv3d_k3\w*|convt3d\w*|roi_\w+|nms_\w+|psv_\w+|pgd_step\w*|affine_\w+|export_u8\w*|patch_\w+|depth_regress\w*|grid_sample3d\w*|"
                  r"gs_to_channels_last|bias_act_kernel|relu_backward_kernel|bev_fold\w*|focal_\w+|space_to_depth2|disc_mask\w*|clean_index\w*|import_u8\w*|dense_align\w*|\w+_prep_kernel|conv3d_k3_prep")


def _library_kernels():
    """the kernel names of the in-tree libadvengine.so (its device stubs): the exact list instead of a pattern that lags behind new kernels"""
    import os
    import subprocess
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eval_driving_safety_amd", "libadvengine.so")
    try:
        out = subprocess.run(["nm", "-C", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    except Exception:
        return None
    names = set(re.findall(r"__device_stub__([A-Za-z0-9_]+)", out))
    return re.compile(r"(?:^|[\s:])(" + "|".join(sorted(names, key=len, reverse=True)) + r")[<(]") if names else None


LIBS = re.compile(r"miopen|Cijk_|igemm_|Col2Im|Im2d2Col|Im2Col|batched_transpose|ck::|_ZN2ck|SubTensorOp|gridwise|MIOpen|Op\dd|transpose_")


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    if len(sys.argv) > 3:
        warm = {r["Name"]: (float(r["TotalDurationNs"]), int(r["Calls"])) for r in csv.DictReader(open(sys.argv[3]))}
        kept = []
        for r in rows:
            wt, wc = warm.get(r["Name"], (0.0, 0))
            t, c = float(r["TotalDurationNs"]) - wt, int(r["Calls"]) - wc
            if c > 0 and t > 0:
                r = dict(r, TotalDurationNs=t, Calls=c)
                kept.append(r)
        rows = kept
    cats = {"libadvengine": [], "miopen_rocblas_ck": [], "torch_elementwise_and_copies": [], "miopen_solver_search_one_time": []}
    exact = _library_kernels()
    for r in rows:
        n = r["Name"]
        t = float(r["TotalDurationNs"])
        if "naive_conv" in n:
            cats["miopen_solver_search_one_time"].append((n, t, int(r["Calls"])))
        elif (exact.search(n) if exact else None) or OURS.search(n):
            cats["libadvengine"].append((n, t, int(r["Calls"])))
        elif LIBS.search(n):
            cats["miopen_rocblas_ck"].append((n, t, int(r["Calls"])))
        else:
            cats["torch_elementwise_and_copies"].append((n, t, int(r["Calls"])))
    total = sum(t for k, v in cats.items() if k != "miopen_solver_search_one_time" for _, t, _ in v)
    out = {"source": sys.argv[1], "warm_up_subtracted": len(sys.argv) > 3, "total_ms_excluding_solver_search": total / 1e6, "shares": {}, "top": {}}
    for k, v in cats.items():
        s = sum(t for _, t, _ in v)
        out["shares"][k] = {"ms": s / 1e6, "share": (s / total) if k != "miopen_solver_search_one_time" else None, "launches": sum(c for _, _, c in v)}
        out["top"][k] = [{"kernel": re.sub(r"\(.*", "", n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))[:110], "ms": t / 1e6, "calls": c}
                         for n, t, c in sorted(v, key=lambda e: -e[1])[:8]]
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(out, f, indent=1)
    for k, v in out["shares"].items():
        print("%-34s %9.2f ms  %s  %6d launches" % (k, v["ms"], ("%5.1f %%" % (100 * v["share"])) if v["share"] is not None else "  (one-time)", v["launches"]))


if __name__ == "__main__":
    main()
