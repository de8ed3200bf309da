"""Counterpart of attack/DSGN/patch_attack.py (flags :35-57, scaffolding :59-152, loop :278-443)."""
import argparse

import torch

from . import _common, upstream
from .. import adapters, data
from ..attacks import PatchTrainer
from ..dist import Comm


def build_parser():
    parser = argparse.ArgumentParser(description="Train a universal adversarial patch against DSGN (MI355X engine; counterpart of attack/DSGN/patch_attack.py)")
    _common.add_scaffolding(parser)
    parser.add_argument("--iter", type=int, default=2, help="inner updates of the patch per frame")
    parser.add_argument("--eps", type=float, default=(8.0 / 255), help="L-infinity budget (DSGN: fraction of the [0,1] range; Stereo R-CNN: multiplied by 255)")
    parser.add_argument("--epochs", type=int, default=80, help="passes over the split")
    parser.add_argument("--ratio", type=float, default=0.2, help="patch diameter as a fraction of the image height")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.debugnum is None:
        args.debugnum = 100
    dev, args.devices_resolved = _common.setup_device(args.devices)
    comm = Comm.from_env(device=dev)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    if args.model in ("toy", "shaped", "layerlist"):
        batch = args.btest if args.btest else 1
        workers = args.loader_workers if args.loader_workers is not None else (0 if args.debug else 12)
        base = (lambda: data.SyntheticStereo(args.synthetic, "dsgn", batch, seed=args.seed)) if args.synthetic \
            else (lambda: data.KittiFolder(args.data_path, args.split_file, batch, workers=workers))
        if args.model == "toy":
            adapter, factory = adapters.ToyStereoAdapter(dev, seed=args.seed), base
        else:
            adapter = _common.layerlist_dsgn(dev, args) if args.model == "layerlist" else \
                adapters.PsvStereoAdapter(dev, seed=args.seed, hourglass=True, dsgn_head=True)
            factory = lambda: _common.WithExtra(base(), adapter.synthetic_extra)
    else:
        rt = _common.upstream_or_exit(lambda: upstream.DsgnRuntime(args, dev, attack=True))
        adapter = adapters.DsgnAdapter(rt.model, rt.cfg, rt.RPN3DLoss)
        factory = lambda: upstream.dsgn_attack_loader(rt)
    trainer = PatchTrainer("dsgn", args.ratio, args.eps, args.iter, args.epochs, out_root=args.out_root,
                           seed=args.pos_seed, comm=comm, device=dev)
    trainer.train(factory, adapter, debugnum=args.debugnum if args.debug else None)
    comm.close()


if __name__ == "__main__":
    main()
