"""Counterpart of attack/Stereo-RCNN/patch_attack.py (flags :38-55, loop :99-293)."""
import argparse

import torch

from . import _common
from .. import adapters, data
from ..attacks import PatchTrainer
from ..dist import Comm


def build_parser():
    parser = argparse.ArgumentParser(description="Attack the Stereo R-CNN network")
    parser.add_argument("--iter", type=int, default=2, help="iteration number of pgd attack")
    parser.add_argument("--eps", dest="eps", type=float, default=0.1)
    parser.add_argument("--epochs", dest="epochs", type=int, default=40)
    parser.add_argument("--ratio", dest="ratio", type=float, default=0.1)
    parser.add_argument("--debug", action="store_true", default=False, help="debug mode")
    parser.add_argument("--debugnum", default=None, type=int, help="debug mode")
    parser.add_argument("--seed", type=int, default=3, help="cfg.RNG_SEED upstream")
    _common.add_engine_flags(parser)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    dev = _common.setup_device()
    comm = Comm.from_env(device=dev)
    if args.model == "toy":
        adapter = adapters.ToyStereoAdapter(dev, seed=args.seed, planes=(0, 8, 16, 32))
    else:
        try:
            from model.stereo_rcnn.resnet import resnet       # noqa: F401  (upstream)
        except Exception:
            _common.upstream_unavailable("model.stereo_rcnn (upstream Stereo R-CNN)")
        raise SystemExit("wire your checkpoint through adapters.StereoRcnnAdapter(model, uncert); see INTEGRATION.md")
    if not args.synthetic:
        raise SystemExit("the Stereo R-CNN roidb loader is upstream code; use --synthetic N or drive PatchTrainer from your loader")
    factory = lambda: data.SyntheticStereo(args.synthetic, "srcnn", 1, seed=args.seed)
    trainer = PatchTrainer("srcnn", args.ratio, args.eps, args.iter, args.epochs, out_root=args.out_root,
                           seed=args.pos_seed, comm=comm, device=dev)
    trainer.train(factory, adapter, debugnum=args.debugnum if args.debug else None)
    comm.close()


if __name__ == "__main__":
    main()
