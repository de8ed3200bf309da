"""profiles/ must describe the library that ships: every `*_kernel_stats.csv` that profiles/README.md lists as CURRENT (the block between
`<!-- current-profiles` and `-->`) has to (a) contain the kernel families the README says its benchmark leg launches, (b) contain no
kernel of this library that the current libadvengine.so no longer has (a renamed or removed kernel = a stale profile).  VERDICT r2 item 7."""
import csv
import os
import re
import subprocess

from eval_driving_safety_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THIRD_PARTY = ("at::", "Cijk_", "miopen", "MIOpen", "igemm", "ck::", "_ZN2ck", "rocprim", "__amd", "naive_conv", "Im2d2Col", "Col2Im", "batched_transpose",
               "SubTensorOp", "gemm", "hipcub", "void rocprim", "Op", "transpose", "rocblas", "Tensile", "kernel_grouped", "void at", "elementwise",
               "reduce", "softmax", "index", "cat", "copy", "fill", "scan", "sort", "gridwise", "distribution", "upsample", "pool", "nll", "sigmoid")


def _kernels_in_library():
    out = subprocess.run(["nm", "-C", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    return set(re.findall(r"__device_stub__([A-Za-z0-9_]+)", out))


def _current_profiles():
    text = open(os.path.join(ROOT, "profiles", "README.md")).read()
    m = re.search(r"<!-- current-profiles\n(.*?)-->", text, re.S)
    assert m, "profiles/README.md has no current-profiles block"
    table = {}
    for line in m.group(1).strip().splitlines():
        name, fams = line.split(":", 1)
        table[name.strip()] = fams.split()
    return table


def _base(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.split(r"[<(]", name, 1)[0].strip()


def test_current_profiles_contain_the_kernels_the_library_launches():
    lib = _kernels_in_library()
    assert {"pgd_step_vec4_idx", "conv3d_k3_mfma", "conv2d_3x3_mfma", "conv2d_1x1_mfma", "conv_wino", "roi_align_bwd_lds"} <= lib
    table = _current_profiles()
    assert len(table) >= 4
    for fname, families in table.items():
        path = os.path.join(ROOT, "profiles", fname)
        assert os.path.exists(path), "%s is listed as current but missing" % fname
        rows = list(csv.DictReader(open(path)))
        names = [_base(r.get("Kernel") or r.get("Name")) for r in rows]
        for fam in families:
            assert any(k.startswith(fam) for k in lib), "%s: the README expects kernel family %s, which libadvengine.so does not have" % (fname, fam)
            assert any(n.startswith(fam) for n in names), "%s lacks kernel family %s - re-profile the leg" % (fname, fam)
        for n in set(names):
            if n and not n.startswith(THIRD_PARTY) and re.fullmatch(r"[a-z0-9_]+", n):
                assert n in lib, "%s holds kernel %s, which the current libadvengine.so does not have - a stale profile" % (fname, n)


def test_the_headline_trace_reproduces_the_bench_line_the_traced_command_printed():
    """the rocprofv3 summary committed for the headline leg must give the `roofline.frac` of the bench line - the line the
    SAME traced command printed is kept beside it (tools/gpu_profile.sh traces the headline leg alone; tools/summarize_prof.py copies both):
    algorithmic bytes per launch / the trace's average duration of `pgd_step_vec4_idx` / 8 TB/s within 2 % of that line's frac."""
    import json
    table = _current_profiles()
    stats = next(n for n in table if re.fullmatch(r"r\d+_kernel_stats\.csv", n))
    tag = stats.split("_")[0]
    line_path = os.path.join(ROOT, "profiles", "%s_kernel_stats_bench_line.json" % tag)
    assert os.path.exists(line_path), "%s: the bench line of the traced command is missing (tools/summarize_prof.py writes it)" % line_path
    line = json.load(open(line_path))
    rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", stats))) if _base(r["Kernel"]).startswith("pgd_step_vec4_idx")]
    assert rows, "%s holds no pgd_step_vec4_idx row" % stats
    row = max(rows, key=lambda r: int(r["Calls"]))
    avg_s = float(row["AverageNs"]) * 1e-9
    roof = line["roofline"]
    frac = roof["algorithmic_bytes_per_launch"] / avg_s / 1e9 / roof["peak"]
    assert abs(frac - roof["frac"]) <= 0.02 * roof["frac"], (frac, roof["frac"], row)
    assert float(row["StdDev"]) <= 0.1 * float(row["AverageNs"]), "the trace mixes this kernel with another leg (copy-engine traffic beside it): %r" % row
