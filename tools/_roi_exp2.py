import sys, os, json, torch, numpy as np
sys.path.insert(0, '/root/repo')
from eval_driving_safety_amd import ops, _lib
dev = torch.device("cuda", 0)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rs = np.random.RandomState(0)
fh, fw, stride = 150, 497, 4
with _lib.using(_lib.HOOKS_LIB_PATH):
    for label, n, spread, wpx, hpx in (("uniform 4x13", 512, 1.0, 15, 51), ("clustered 4x13", 512, 0.08, 15, 51)):
        x1 = 900 + (rs.rand(n) - 0.5) * 1900 * spread; y1 = 300 + (rs.rand(n) - 0.5) * 500 * spread
        rois = torch.tensor(np.stack([np.zeros(n), x1, y1, x1 + wpx, y1 + hpx], 1).astype(np.float32), device=dev)
        g = torch.randn((n, 256, 7, 7), device=dev)
        for dbg in ("0", "1", "2", "3"):
            os.environ["ADV_ROI_DBG"] = dbg
            ms = t(lambda: ops.roi_align_bwd(g, rois, (1, 256, fh, fw), 1.0 / stride, 0))
            print(json.dumps({"rois": label, "dbg(1=no fma,2=no classify)": dbg, "bwd_ms": round(ms, 3)}), flush=True)
