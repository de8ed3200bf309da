import sys, collections, torch
sys.path.insert(0, "/root/repo")
from eval_driving_safety_amd import adapters, data, ops
import eval_driving_safety_amd.ops.elementwise as E
import eval_driving_safety_amd.ops.conv2d as C2
import eval_driving_safety_amd.ops.conv3d as C3
import traceback
seen = collections.Counter()
orig = E.relu_backward
def logged(g, y):
    fr = [f for f in traceback.extract_stack()[:-1] if "ops/" in f.filename][-1]
    seen[(tuple(g.shape), fr.filename.split("/")[-1], fr.lineno)] += 1
    return orig(g, y)
for m in (E, C2, C3, ops):
    if hasattr(m, "relu_backward"): setattr(m, "relu_backward", logged)
import eval_driving_safety_amd.ops.volume as V
if hasattr(V, "relu_backward"): V.relu_backward = logged
dev = torch.device("cuda", 0)
net = adapters.DsgnShapedAdapter(dev, seed=0)
batch = next(iter(data.SyntheticStereo(1, "dsgn", batch=1, seed=0)))
batch.extra = net.synthetic_extra(batch, seed=1)
x = torch.cat([batch.imgL, batch.imgR]).to(dev)
net.loss_and_grad(x, batch.extra); seen.clear()
net.loss_and_grad(x, batch.extra)
for k, v in sorted(seen.items(), key=lambda kv: -torch.Size(kv[0][0]).numel()):
    print(v, k, "%.1f MB" % (torch.Size(k[0]).numel() * 4 / 1e6))
