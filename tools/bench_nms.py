import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from eval_driving_safety_amd import ops
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
for n in (2000, 4000):
    ctr = rs.rand(n, 2) * 600; wh = rs.rand(n, 2) * 80 + 4
    boxes = torch.tensor(np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32), device=dev)
    sc = torch.zeros(n, device=dev)
    for _ in range(3): ops.nms_padded(boxes, sc, 0.7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.nms_padded(boxes, sc, 0.7)
    e1.record(); torch.cuda.synchronize()
    print(n, "nms_padded us per call (zero + mask + scan + 2 fills):", round(e0.elapsed_time(e1) / 20 * 1e3, 1))
