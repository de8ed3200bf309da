"""Counterpart of attack/Stereo-RCNN/predict_and_save_pgd.py (flags :37-53, loop :72-427): detect on the PGD-attacked,
network-scale PNG folders (``stereo_rcnn_pgd_iters_k`` swapped in for image_2/3) and write the KITTI result files."""
import argparse

from . import _common, _srcnn_detect, upstream


def build_parser():
    parser = argparse.ArgumentParser(description="Run Stereo R-CNN on a folder of PGD iterates and write its result files (counterpart of attack/Stereo-RCNN/predict_and_save_pgd.py)")
    parser.add_argument("--iter", dest="iter", help="which iterate folder (<model>_pgd_iters_<k>) to run the detector on", type=int, default=1)
    parser.add_argument("--alpha", dest="alpha", help="step size the attacked folder was made with (part of the folder name only)", type=float, default=1)
    parser.add_argument("--save_feat_map", action="store_true", help="also dump the detector's intermediate feature maps")
    parser.add_argument("--save_feat_path", type=str, default="", help="folder for --save_feat_map")
    parser.add_argument("--devices", "-d", type=str, default="0", help="GPU index (the reference always uses the current device)")
    _common.add_engine_flags(parser)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    dev, _ = _common.setup_device(args.devices)
    from .. import ops
    rt = _common.upstream_or_exit(lambda: upstream.SrcnnRuntime(dev, training=False, workers=0, normalize=False, adopt=args.adopt))      # :78-124
    written, result_dir = _srcnn_detect.run(args, rt, "pgd", dev, ops)
    print("wrote %d detections to %s" % (written, result_dir))


if __name__ == "__main__":
    main()
