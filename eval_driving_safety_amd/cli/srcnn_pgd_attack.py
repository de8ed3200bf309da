"""Counterpart of attack/Stereo-RCNN/pgd_attack.py (flags :36-51, scaffolding :55-99, loop :105-243)."""
import argparse

from . import _common, upstream
from .. import adapters, data
from ..attacks import PgdAttack
from ..dist import Comm


def build_parser():
    parser = argparse.ArgumentParser(description="PGD / FGSM perturbation of KITTI stereo pairs against Stereo R-CNN (counterpart of attack/Stereo-RCNN/pgd_attack.py)")
    parser.add_argument("--iter", type=int, default=4, help="number of PGD steps (1 = FGSM)")
    parser.add_argument("--alpha", default=1.0, type=float, help="step size per iteration (in the model's pixel units: 1/255 of the [0,1] range for DSGN, grey levels for Stereo R-CNN)")
    parser.add_argument("--eps", default=0.3, type=float,                  # scaled by 255 (:57)
                        help="L-infinity budget (DSGN: fraction of the [0,1] range; Stereo R-CNN: multiplied by 255)")
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
    print("Start iteration: ", args.iter)                                   # :59
    if args.model in ("toy", "shaped", "layerlist"):
        if not args.synthetic:
            raise SystemExit("--model %s needs --synthetic N (the Stereo R-CNN roidb loader is upstream code)" % args.model)
        loader = data.SyntheticStereo(args.synthetic, "srcnn", 1, seed=args.seed)
        if args.model == "toy":
            adapter = adapters.ToyStereoAdapter(dev, seed=args.seed, planes=(0, 8, 16, 32))
        else:   # FPN + stereo RPN + RoIAlign heads on this package's kernels, random weights, synthetic ground truth
            import torch
            from .. import surrogates
            if args.model == "layerlist":   # ResNet-101-FPN layer list on the route table's kernels; the scripts' checkpoint if it is there
                net, uncert = _common.layerlist_srcnn(dev, args)
            else:
                net, uncert = surrogates.StereoRcnnShaped(seed=args.seed).to(dev).eval(), torch.zeros(6, device=dev)
            adapter = adapters.StereoRcnnAdapter(net, uncert)
            loader = _common.WithExtra(loader, lambda b: surrogates.synthetic_srcnn_extra(b, dev))
    else:
        rt = _common.upstream_or_exit(lambda: upstream.SrcnnRuntime(dev, training=True, workers=0, adopt=args.adopt))
        adapter = adapters.StereoRcnnAdapter(rt.model, rt.uncert)
        loader = upstream.srcnn_loader(rt)
    if args.graph:
        raise SystemExit("--graph: not offered for the Stereo R-CNN scripts.  The layer-list surrogates' iteration can be captured from Python "
                         "(attacks.PgdAttack(graph=True) with surrogates.StereoRcnnShaped.allow_graph_capture, reused across label sets with the "
                         "same host constants), but with the PNG export running beside the replays it faulted on this torch / ROCm stack; "
                         "upstream models read back and compact")
    atk = PgdAttack("srcnn", args.alpha, args.eps, args.iter, out_root=args.out_root, save_every=args.save_every, device=dev)
    # `if args.debug and i >= args.debugnum: break` (:107-108): debugnum - 1 is the last index attacked
    n = atk.run(loader, adapter, comm, debugnum=(args.debugnum - 1) if (args.debug and args.debugnum is not None) else None)
    print("rank %d attacked %d stereo pairs" % (comm.rank, n))
    comm.close()


if __name__ == "__main__":
    main()
