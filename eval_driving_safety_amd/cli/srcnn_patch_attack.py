"""Counterpart of attack/Stereo-RCNN/patch_attack.py (flags :38-55, scaffolding :99-150, loop :152-293)."""
import argparse

from . import _common, upstream
from .. import adapters, data
from ..attacks import PatchTrainer
from ..dist import Comm


def build_parser():
    parser = argparse.ArgumentParser(description="Train a universal adversarial patch against Stereo R-CNN (counterpart of attack/Stereo-RCNN/patch_attack.py)")
    parser.add_argument("--iter", type=int, default=2, help="inner updates of the patch per frame")
    parser.add_argument("--eps", dest="eps", type=float, default=0.1, help="L-infinity budget (DSGN: fraction of the [0,1] range; Stereo R-CNN: multiplied by 255)")
    parser.add_argument("--epochs", dest="epochs", type=int, default=40, help="passes over the split")
    parser.add_argument("--ratio", dest="ratio", type=float, default=0.1, help="patch diameter as a fraction of the image height")
    parser.add_argument("--debug", action="store_true", default=False, help="stop after --debugnum frames and read the data in the main process")
    parser.add_argument("--debugnum", default=None, type=int, help="how many frames a --debug run handles (the scripts differ by one, see DESIGN.md Q15)")
    parser.add_argument("--seed", type=int, default=3, help="seed of torch's generators")
    parser.add_argument("--devices", "-d", type=str, default="0", help="GPU index (the reference always uses the current device)")
    _common.add_engine_flags(parser)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    dev, _ = _common.setup_device(args.devices)
    comm = Comm.from_env(device=dev)
    if args.model in ("toy", "shaped", "layerlist"):
        if not args.synthetic:
            raise SystemExit("--model %s needs --synthetic N (the Stereo R-CNN roidb loader is upstream code)" % args.model)
        base = lambda: data.SyntheticStereo(args.synthetic, "srcnn", 1, seed=args.seed)
        if args.model == "toy":
            adapter, factory = adapters.ToyStereoAdapter(dev, seed=args.seed, planes=(0, 8, 16, 32)), base
        else:
            import torch
            from .. import surrogates
            if args.model == "layerlist":   # ResNet-101-FPN layer list on the route table's kernels; the scripts' checkpoint if it is there
                net, uncert = _common.layerlist_srcnn(dev, args)
            else:
                net, uncert = surrogates.StereoRcnnShaped(seed=args.seed).to(dev).eval(), torch.zeros(6, device=dev)
            adapter = adapters.StereoRcnnAdapter(net, uncert)
            factory = lambda: _common.WithExtra(base(), lambda b: surrogates.synthetic_srcnn_extra(b, dev))
    else:
        rt = _common.upstream_or_exit(lambda: upstream.SrcnnRuntime(dev, training=True, workers=8, adopt=args.adopt))     # :128-129
        adapter = adapters.StereoRcnnAdapter(rt.model, rt.uncert)
        factory = lambda: upstream.srcnn_loader(rt)
    trainer = PatchTrainer("srcnn", args.ratio, args.eps, args.iter, args.epochs, out_root=args.out_root,
                           seed=args.pos_seed, comm=comm, device=dev)
    trainer.train(factory, adapter, debugnum=(args.debugnum - 1) if (args.debug and args.debugnum is not None) else None)
    comm.close()


if __name__ == "__main__":
    main()
