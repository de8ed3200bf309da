"""Counterpart of attack/DSGN/predict_and_save_patch.py (flags :35-66, loop :394-554): paste the trained patch at a
position drawn from the ``--atk_mode`` column band into every clean pair, detect, write the KITTI label files."""
import argparse

import torch

from . import _common, _dsgn_detect, upstream


def build_parser():
    parser = argparse.ArgumentParser(description="Run DSGN on frames carrying a trained patch and write KITTI label files (counterpart of attack/DSGN/predict_and_save_patch.py)")
    _common.add_scaffolding(parser, loadmodel="./outputs/temp/DSGN_car_pretrained/finetune_53.tar", btest=1, devices=0)
    _common.add_detect_flags(parser)
    parser.add_argument("--ratio", dest="ratio", type=float, default=0.2, help="patch diameter as a fraction of the image height")
    parser.add_argument("--epochs", dest="epochs", type=int, default=80, help="passes over the split")
    parser.add_argument("--patch_dir", dest="patch_dir", type=str, help="folder holding the trained patches (<model>_patch_ratio_<r>/epoch<k>/patch.npy)")
    parser.add_argument("--atk_mode", dest="atk_mode", type=str, default="random",
                        help="where the patch is pasted: random | sp_left | sp_straight | sp_right (column bands of the image)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.debugnum is None:
        args.debugnum = 100
    dev, args.devices_resolved = _common.setup_device(args.devices)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    if args.model in ("shaped", "layerlist"):
        return _dsgn_detect.run_shaped(args, "patch", dev)
    rt = _common.upstream_or_exit(lambda: upstream.DsgnRuntime(args, dev, attack=False))
    if args.ratio or args.epochs:                                              # :96-97
        args.tag += "_ratio{0}_epochs{1}".format(args.ratio, args.epochs)
    written, label_dir = _dsgn_detect.run(args, rt, "patch", dev)
    print("wrote %d label files to %s" % (written, label_dir))


if __name__ == "__main__":
    main()
