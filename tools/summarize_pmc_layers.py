#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ of tools/gpu_profile_layers.sh -> one JSON: per case of tools/pmc_layers.py the kernel's average duration
(kernel trace), matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) (as profiles/r0[12]_*pmc.json),
MFMA instruction count, wave-cycle breakdown, LDS bank-conflict share, and HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (the gfx950
correction of MI355X_MICROARCH.md).  Dispatches are attributed to cases by launch order (the manifest), warm-up launches dropped.
usage: tools/summarize_pmc_layers.py gpurun_out/prof_r04layers profiles/r04_conv_pmc.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

KERNELS = re.compile(r"::(conv_wino4|conv_wino|conv2d_1x1_mfma|conv2d_3x3_mfma|conv3d_k3_s2_mfma|convt3d_k3_s2_mfma|conv3d_k3_mfma|roi_align_bwd_lds)<|::(roi_align_bwd_tab)\(")


def dispatches(src, leg, value):
    """[(dispatch id, kernel family, full kernel name, grid, {counter: value})] in launch order"""
    rows = defaultdict(lambda: [None, None, 0, {}])
    for path in glob.glob(os.path.join(src, leg, "**", "*_counter_collection.csv" if value else "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            m = KERNELS.search(r["Kernel_Name"])
            if not m:
                continue
            did = int(r["Dispatch_Id"])
            e = rows[did]
            e[0], e[1] = m.group(1) or m.group(2), r["Kernel_Name"]
            if value:
                e[2] = int(r["Grid_Size"])
                e[3][r["Counter_Name"]] = e[3].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            else:
                e[2] = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                e[3]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return [(d,) + tuple(rows[d]) for d in sorted(rows)]


def per_case(src, leg, value, manifest):
    ds = dispatches(src, leg, value)
    out, i = [], 0
    for c in manifest["cases"]:
        mine = []
        while i < len(ds) and len(mine) < c["launches"]:
            if ds[i][1] == c["kernel"]:
                mine.append(ds[i])
            i += 1
        out.append(mine[c["warm"]:])
    return out


def kernel_name(full):
    """'void (anonymous namespace)::conv_wino4<2, 16, 2, false, 0, false>(float const*, ...)' -> 'conv_wino4<2, 16, 2, false, 0, false>'"""
    name = full.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):           # cut at the argument list: the first '(' outside the template brackets
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def mean(rows, key):
    v = [r[4][key] for r in rows if key in r[4]]
    return sum(v) / len(v) if v else None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    manifest = json.load(open(os.path.join(src, "manifest.json")))
    legs = {"trace": per_case(src, "trace", False, manifest), "pmc1": per_case(src, "pmc1", True, manifest), "pmc2": per_case(src, "pmc2", True, manifest),
            "fetch": per_case(src, "fetch", True, manifest), "write": per_case(src, "write", True, manifest)}
    rows = []
    for k, c in enumerate(manifest["cases"]):
        t, p1, p2, f, w = (legs[n][k] for n in ("trace", "pmc1", "pmc2", "fetch", "write"))
        row = {"case": c["case"], "kernel": kernel_name(t[0][2]) if t else c["kernel"], "launches_averaged": len(t)}
        ns = mean(t, "ns")
        if ns:
            row["avg_us"] = ns / 1e3
            if c["direct_flops_per_launch"]:
                row["direct_equiv_tflops"] = c["direct_flops_per_launch"] / ns / 1e3
                executed = c["direct_flops_per_launch"] / {"conv_wino": 2.25, "conv_wino4": 4.0}.get(c["kernel"], 1.0)
                row["executed_tflops"] = executed / ns / 1e3
                row["executed_over_peak_157.3"] = executed / ns / 1e3 / 157.3
        gui = mean(p1, "GRBM_GUI_ACTIVE")
        if gui:
            row["mfma_pipe_utilisation"] = mean(p1, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui / 8.0)
            row["mfma_instructions"] = mean(p1, "SQ_INSTS_VALU_MFMA_F32")
            wc = mean(p1, "SQ_WAVE_CYCLES")
            if wc:
                row["wave_cycles"] = {"waiting_any": mean(p1, "SQ_WAIT_ANY") / wc, "waiting_inst": mean(p1, "SQ_WAIT_INST_ANY") / wc, "issuing": mean(p1, "SQ_ACTIVE_INST_ANY") / wc}
            row["sq_busy_over_gpu_cycles"] = mean(p1, "SQ_BUSY_CYCLES") / (gui / 8.0) if mean(p1, "SQ_BUSY_CYCLES") else None
        gui2 = mean(p2, "GRBM_GUI_ACTIVE")
        if gui2 and mean(p2, "SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_over_lds_active"] = mean(p2, "SQ_LDS_BANK_CONFLICT") / mean(p2, "SQ_LDS_IDX_ACTIVE")
            row["lds_active_over_gpu_cycles_per_cu"] = mean(p2, "SQ_LDS_IDX_ACTIVE") / (256.0 * gui2 / 8.0)
            row["valu_instructions"], row["lds_instructions"] = mean(p2, "SQ_INSTS_VALU"), mean(p2, "SQ_INSTS_LDS")
            if row.get("mfma_instructions") is not None and row["valu_instructions"]:
                # a float32 matrix instruction and the other wave's vector instructions do not overlap on a SIMD (profiles/r06_mfma_valu_probe.json):
                # the vector instructions' ~4 cycles each come out of the same budget - what the matrix pipe could reach at most in this kernel
                other = max(0.0, row["valu_instructions"] - row["mfma_instructions"])
                row["vector_share_of_simd_cycles"] = 4.0 * other / (1024.0 * gui2 / 8.0)
                row["matrix_plus_vector_share"] = row.get("mfma_pipe_utilisation", 0.0) + row["vector_share_of_simd_cycles"]
        fs, ws = mean(f, "FETCH_SIZE"), mean(w, "WRITE_SIZE")
        if fs is not None and ws is not None:
            row["hbm_read_bytes"], row["hbm_write_bytes"] = 2 * 1024 * fs, 1024 * ws
            if ns:
                row["hbm_GBps"] = (row["hbm_read_bytes"] + row["hbm_write_bytes"]) / ns
        rows.append(row)
    out = {"source": "tools/gpu_profile_layers.sh over tools/pmc_layers.py (rocprofv3 --pmc, four separate passes, + a kernel trace; the program directly after --)",
           "how": "mfma_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); LDS = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                  "vector_share_of_simd_cycles = 4 x (SQ_INSTS_VALU - SQ_INSTS_VALU_MFMA_F32) / the same SIMD cycles (float32 matrix and vector instructions share a SIMD's datapath on gfx950); "
                  "hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; executed = direct FLOPs / 2.25 for the F(2x2,3x3) kernel, / 4 for the F(4x4,3x3) kernel", "rows": rows}
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    for r in rows:
        print("%-52s %8.1f us  direct %6.1f TF  mfma busy %s  lds conflict %s  hbm %s GB/s" % (
            r["case"][:52], r.get("avg_us", 0), r.get("direct_equiv_tflops", 0), "%.2f" % r["mfma_pipe_utilisation"] if "mfma_pipe_utilisation" in r else "-",
            "%.2f" % r["lds_bank_conflict_over_lds_active"] if "lds_bank_conflict_over_lds_active" in r else "-", "%.0f" % r["hbm_GBps"] if "hbm_GBps" in r else "-"))


if __name__ == "__main__":
    main()
