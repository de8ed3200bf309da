#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/gpu_profile.sh) into the small files kept
under profiles/: the rocprofv3 --stats rows of this library's kernels and the HBM traffic per launch
from the two PMC passes, corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

usage: tools/summarize_prof.py gpurun_out/prof_r01 r01
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

OURS = ("clean_index_build_identity", "clean_index_build_vec4", "clean_lut_kernel", "pgd_step_shifted", "pgd_step_vec4", "pgd_step_scalar", "affine_vec4", "affine_scalar", "export_u8_vec4", "export_u8_scalar",
        "patch_paste_kernel", "patch_delta_kernel", "patch_apply_kernel", "disc_mask_kernel", "psv_")


def short(name):
    for k in OURS:
        if k in name:
            i = name.index(k)
            j = name.find("(", i)
            return name[i:j if j > 0 else None]
    return None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    for leg, suffix in (("trace", ""), ("trace_other", "_other_legs")):      # the headline leg alone; the Stereo R-CNN-shape + delivered-iterates legs
        found = glob.glob(os.path.join(src, leg, "**", "*_kernel_stats.csv"), recursive=True)
        if not found:
            continue
        rows = list(csv.DictReader(open(found[0])))
        with open(os.path.join(dst, "%s_kernel_stats%s.csv" % (tag, suffix)), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                s = short(r["Name"])
                w.writerow([s if s else r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
    line = os.path.join(src, "bench_line_of_the_trace.json")
    if os.path.exists(line) and os.path.getsize(line):      # the line the traced command itself printed: its roofline belongs to this trace's averages
        with open(os.path.join(dst, "%s_kernel_stats_bench_line.json" % tag), "w") as f:
            json.dump(json.loads(open(line).readline()), f, indent=1)
    pmc = defaultdict(lambda: defaultdict(list))
    for leg in ("fetch", "write"):
        for path in glob.glob(os.path.join(src, leg, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                s = short(r["Kernel_Name"])
                if s:
                    pmc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 "
                     "--warmup 0 --no-cpu-baseline --no-end-to-end --no-delivered --no-srcnn`", "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                     "(gfx950: FETCH_SIZE reads half of a wide coalesced stream)", "kernels": {}}
    for k, d in sorted(pmc.items()):
        f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
        wv = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
        out["kernels"][k] = {"launches_fetch_pass": len(d["FETCH_SIZE"]), "launches_write_pass": len(d["WRITE_SIZE"]),
                             "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": wv,
                             "hbm_read_bytes_per_launch": 2 * f * 1024, "hbm_write_bytes_per_launch": wv * 1024,
                             "hbm_bytes_per_launch": (2 * f + wv) * 1024}
    lines = os.path.join(src, "bench_lines.jsonl")
    if os.path.exists(lines):
        out["bench_lines_under_profiler"] = [json.loads(l) for l in open(lines) if l.startswith("{")]
    with open(os.path.join(dst, "%s_pmc_hbm.json" % tag), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["kernels"].items():
        print("%-40s %.4f GB/launch" % (k, v["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
