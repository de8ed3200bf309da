"""Counterpart of attack/DSGN/patch_attack.py (flags :35-57, loop :278-443)."""
import argparse

import torch

from . import _common
from .. import adapters, data
from ..attacks import PatchTrainer
from ..dist import Comm


def build_parser():
    parser = argparse.ArgumentParser(description="Patch attack")
    _common.add_scaffolding(parser)
    parser.add_argument("--iter", type=int, default=2, help="iteration number of patch attack")
    parser.add_argument("--eps", type=float, default=(8.0 / 255))
    parser.add_argument("--epochs", type=int, default=80)
    parser.add_argument("--ratio", type=float, default=0.2)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.debugnum is None:
        args.debugnum = 100
    dev = _common.setup_device()
    comm = Comm.from_env(device=dev)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    batch = args.btest if args.btest else 1
    if args.model == "toy":
        adapter = adapters.ToyStereoAdapter(dev, seed=args.seed)
    else:
        try:
            from dsgn.models import StereoNet                 # noqa: F401  (upstream)
        except Exception:
            _common.upstream_unavailable("dsgn (upstream DSGN)")
        raise SystemExit("wire your DSGN checkpoint through adapters.DsgnAdapter(model, cfg, RPN3DLoss); see INTEGRATION.md")
    factory = (lambda: data.SyntheticStereo(args.synthetic, "dsgn", batch, seed=args.seed)) if args.synthetic \
        else (lambda: data.KittiFolder(args.data_path, args.split_file, batch))
    trainer = PatchTrainer("dsgn", args.ratio, args.eps, args.iter, args.epochs, out_root=args.out_root,
                           seed=args.pos_seed, comm=comm, device=dev)
    trainer.train(factory, adapter, debugnum=args.debugnum if args.debug else None)
    comm.close()


if __name__ == "__main__":
    main()
