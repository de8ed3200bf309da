#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ of tools/gpu_profile_kernels.sh into profiles/<tag>_secondary_kernels.json:
per (kernel, grid size) the average duration from the kernel trace and the HBM bytes from the two PMC passes
(2*FETCH_SIZE + WRITE_SIZE KiB, gfx950 correction of MI355X_MICROARCH.md).
usage: tools/summarize_prof_kernels.py gpurun_out/prof_r01k r01"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

KEEP = re.compile(r"(psv_\w+|affine_\w+|export_u8_\w+|pgd_step_\w+|patch_\w+_kernel|disc_mask_kernel)(<[^>]*>)?")


def short(name):
    m = KEEP.search(name)
    return m.group(0) if m else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(src, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            s = short(r["Kernel_Name"])
            if s:
                dur[(s, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    pmc = defaultdict(lambda: defaultdict(list))
    for leg in ("fetch", "write"):
        for path in glob.glob(os.path.join(src, leg, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                s = short(r["Kernel_Name"])
                if s:
                    pmc[(s, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"source": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (three separate runs) of "
                     "tools/bench_kernels.py", "correction": "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024", "rows": []}
    for key in sorted(dur):
        d = dur[key]
        skip = min(3, len(d) - 1)                       # drop the warm-up launches
        avg = sum(d[skip:]) / max(1, len(d[skip:]))
        f = pmc[key]["FETCH_SIZE"]
        w = pmc[key]["WRITE_SIZE"]
        row = {"kernel": key[0], "grid_threads": key[1], "launches": len(d), "avg_ns": avg}
        if f and w:
            fb, wb = 2 * 1024 * sum(f[skip:]) / len(f[skip:]), 1024 * sum(w[skip:]) / len(w[skip:])
            row.update(hbm_read_bytes=fb, hbm_write_bytes=wb, hbm_GBps=(fb + wb) / avg)
        out["rows"].append(row)
    with open(os.path.join(root, "profiles", "%s_secondary_kernels.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    for r in out["rows"]:
        print("%-28s grid %9d  %9.1f us  %s" % (r["kernel"], r["grid_threads"], r["avg_ns"] / 1e3,
                                                  ("%.1f MB read %.1f MB write -> %.0f GB/s" % (r["hbm_read_bytes"] / 1e6, r["hbm_write_bytes"] / 1e6, r["hbm_GBps"]))
                                                  if "hbm_GBps" in r else ""))


if __name__ == "__main__":
    main()
