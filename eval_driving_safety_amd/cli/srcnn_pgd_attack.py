"""Counterpart of attack/Stereo-RCNN/pgd_attack.py (flags :36-51, loop :105-243)."""
import argparse

import torch

from . import _common
from .. import adapters, data
from ..attacks import PgdAttack
from ..dist import Comm


def build_parser():
    parser = argparse.ArgumentParser(description="Attack the Stereo R-CNN network")
    parser.add_argument("--iter", type=int, default=4, help="iteration number of pgd attack")
    parser.add_argument("--alpha", dest="alpha", type=float, default=1.0)
    parser.add_argument("--eps", dest="eps", type=float, default=0.3)      # scaled by 255 (:57)
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
        raise SystemExit("the Stereo R-CNN roidb loader is upstream code; use --synthetic N or drive PgdAttack from your loader")
    loader = data.SyntheticStereo(args.synthetic, "srcnn", 1, seed=args.seed)
    atk = PgdAttack("srcnn", args.alpha, args.eps, args.iter, out_root=args.out_root, save_every=args.save_every, device=dev)
    n = atk.run(loader, adapter, comm, debugnum=args.debugnum if args.debug else None)
    print("rank %d attacked %d stereo pairs" % (comm.rank, n))
    comm.close()


if __name__ == "__main__":
    main()
