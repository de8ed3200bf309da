import sys, json, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from eval_driving_safety_amd import ops
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
for label, n, spread, wpx, hpx in (("uniform 8x8", 512, 1.0, 32, 32), ("uniform 4x13", 512, 1.0, 15, 51), ("clustered 4x13", 512, 0.08, 15, 51), ("clustered 8x8", 512, 0.08, 32, 32),
                                   ("clustered 4x13 n=128", 128, 0.08, 15, 51), ("uniform 30x30", 512, 1.0, 120, 120), ("identical 4x13", 512, 0.0, 15, 51)):
    x1 = 900 + (rs.rand(n) - 0.5) * 1900 * spread; y1 = 300 + (rs.rand(n) - 0.5) * 500 * spread
    rois = torch.tensor(np.stack([np.zeros(n), x1, y1, x1 + wpx, y1 + hpx], 1).astype(np.float32), device=dev)
    for pooled in (7, 14):
        g = torch.randn((n, 256, pooled, pooled), device=dev)
        ms = t(lambda: ops.roi_align_bwd(g, rois, (1, 256, fh, fw), 1.0 / stride, 0))
        msf = t(lambda: ops.roi_align(torch.zeros((1,256,fh,fw), device=dev), rois, pooled, 1.0/stride, 0))
        print(json.dumps({"rois": label, "n": n, "pooled": pooled, "bwd_ms": round(ms, 3), "fwd_ms": round(msf, 3)}), flush=True)
