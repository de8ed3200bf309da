"""Counterpart of attack/DSGN/predict_and_save_pgd.py (flags :34-62, loop :332-455): detect on PGD-attacked image folders
(``dsgn_pgd_iters_k`` swapped in for ``image_2/3``, attack/DSGN/README.md:30,69) and write the KITTI label files."""
import argparse

import torch

from . import _common, _dsgn_detect, upstream


def build_parser():
    parser = argparse.ArgumentParser(description="Run DSGN on a folder of PGD iterates and write KITTI label files (counterpart of attack/DSGN/predict_and_save_pgd.py)")
    _common.add_scaffolding(parser)
    _common.add_detect_flags(parser)
    parser.add_argument("--iter", type=int, help="which iterate folder (<model>_pgd_iters_<k>) to run the detector on")
    parser.add_argument("--alpha", type=float, help="step size the attacked folder was made with (part of the folder name only)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.debugnum is None:
        args.debugnum = 100
    dev, args.devices_resolved = _common.setup_device(args.devices)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    if args.model in ("shaped", "layerlist"):
        return _dsgn_detect.run_shaped(args, "pgd", dev)
    rt = _common.upstream_or_exit(lambda: upstream.DsgnRuntime(args, dev, attack=False))
    if args.alpha and args.iter:                                               # :92-93 (after the debug / _train tags)
        args.tag += "_iter{0}_alpha{1}".format(str(args.iter), str(args.alpha))
    written, label_dir = _dsgn_detect.run(args, rt, "pgd", dev)
    print("wrote %d label files to %s" % (written, label_dir))


if __name__ == "__main__":
    main()
