"""Counterpart of attack/DSGN/pgd_attack.py (flags :35-56, scaffolding :58-147, loop :229-374)."""
import argparse

import torch

from . import _common, upstream
from .. import adapters, data
from ..attacks import PgdAttack
from ..dist import Comm


def build_parser():
    parser = argparse.ArgumentParser(description="PGD / FGSM perturbation of KITTI stereo pairs against DSGN (MI355X engine; counterpart of attack/DSGN/pgd_attack.py)")
    _common.add_scaffolding(parser)
    parser.add_argument("--iter", type=int, default=4, help="number of PGD steps (1 = FGSM)")
    parser.add_argument("--alpha", type=float, default=(1.0 / 255), help="step size per iteration (in the model's pixel units: 1/255 of the [0,1] range for DSGN, grey levels for Stereo R-CNN)")
    parser.add_argument("--eps", type=float, default=0.3, help="L-infinity budget (DSGN: fraction of the [0,1] range; Stereo R-CNN: multiplied by 255)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.debugnum is None:
        args.debugnum = 100                                   # :64-65
    dev, args.devices_resolved = _common.setup_device(args.devices)
    comm = Comm.from_env(device=dev)
    torch.manual_seed(args.seed)                              # :86-87
    torch.cuda.manual_seed(args.seed)
    if args.model in ("toy", "shaped", "layerlist"):
        batch = args.btest if args.btest else 1
        workers = args.loader_workers if args.loader_workers is not None else (0 if args.debug else 12)
        loader = data.SyntheticStereo(args.synthetic, "dsgn", batch, seed=args.seed) if args.synthetic \
            else data.KittiFolder(args.data_path, args.split_file, batch, workers=workers)
        if args.model == "toy":
            adapter = adapters.ToyStereoAdapter(dev, seed=args.seed)
        else:   # plane-sweep volume (HIP) -> 3D hourglass on the float32 matrix cores -> depth loss; synthetic sparse depth
            adapter = _common.layerlist_dsgn(dev, args) if args.model == "layerlist" else \
                adapters.PsvStereoAdapter(dev, seed=args.seed, hourglass=True, dsgn_head=True)
            loader = _common.WithExtra(loader, adapter.synthetic_extra)
    else:
        rt = _common.upstream_or_exit(lambda: upstream.DsgnRuntime(args, dev, attack=True))
        adapter = adapters.DsgnAdapter(rt.model, rt.cfg, rt.RPN3DLoss)
        loader = upstream.dsgn_attack_loader(rt)
    if args.graph and not getattr(adapter, "graph_safe", False):
        raise SystemExit("--graph needs a detector step without host read-backs or data-dependent shapes (--model toy / shaped / layerlist); "
                         "an upstream DSGN model is not known to be one")
    atk = PgdAttack("dsgn", args.alpha, args.eps, args.iter, out_root=args.out_root, save_every=args.save_every, device=dev,
                    reference_on_gpu=args.reference_on_gpu, graph=args.graph)
    n = atk.run(loader, adapter, comm, debugnum=args.debugnum if args.debug else None)
    print("rank %d attacked %d stereo pairs" % (comm.rank, n))
    comm.close()


if __name__ == "__main__":
    main()
