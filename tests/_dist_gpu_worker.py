"""One rank of the 2-rank universal-patch run of tests/test_gpu_dist.py: HIP ops on cuda:0 (the box has one GPU, so both
ranks share it and the collective runs over gloo), a REAL Comm, gradients recorded for the host replay."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    out_dir, n_pairs, average = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
    import synth
    from eval_driving_safety_amd import adapters, attacks
    from eval_driving_safety_amd.dist import Comm
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    comm = Comm.from_env(backend="gloo")
    H, W = 384, 1248
    pairs = [(synth.dsgn_normalised(70 + 2 * i, H, W), synth.dsgn_normalised(71 + 2 * i, H, W)) for i in range(n_pairs)]

    class Recorder:
        def __init__(self, inner):
            self.inner, self.grads = inner, []

        def loss_and_grad(self, x, extra=None):
            loss, g = self.inner.loss_and_grad(x, extra)
            self.grads.append(g.detach().cpu().numpy().copy())
            return loss, g

    rec = Recorder(adapters.ToyStereoAdapter(dev, seed=4))

    def factory():
        return [attacks.StereoBatch(torch.from_numpy(l.copy()), torch.from_numpy(r.copy()), ["%06d" % i], [(1242, 375)])
                for i, (l, r) in enumerate(pairs)]

    tr = attacks.PatchTrainer("dsgn", 0.2, 8 / 255, 2, 1, out_root=out_dir, seed=9, comm=comm, device=dev, average=average)
    patch = tr.train(factory, rec)
    np.save(os.path.join(out_dir, "patch_rank%d.npy" % comm.rank), patch.cpu().numpy())
    np.save(os.path.join(out_dir, "grads_rank%d.npy" % comm.rank), np.stack(rec.grads) if rec.grads else np.zeros((0,)))
    comm.close()


if __name__ == "__main__":
    main()
