"""Counterpart of attack/Stereo-RCNN/predict_and_save_patch.py (flags :38-57, loop :133-539): paste the trained patch at a
position drawn from the ``--atk_mode`` column band into every clean pair, detect, write the KITTI result files."""
import argparse

from . import _common, _srcnn_detect, upstream


def build_parser():
    parser = argparse.ArgumentParser(description="Run Stereo R-CNN on frames carrying a trained patch and write its result files (counterpart of attack/Stereo-RCNN/predict_and_save_patch.py)")
    parser.add_argument("--debug", action="store_true", default=False, help="stop after --debugnum frames and read the data in the main process")
    parser.add_argument("--debugnum", default=None, type=int, help="how many frames a --debug run handles (the scripts differ by one, see DESIGN.md Q15)")
    parser.add_argument("--ratio", dest="ratio", type=float, default=0.1, help="patch diameter as a fraction of the image height")
    parser.add_argument("--epochs", dest="epochs", type=int, default=40, help="passes over the split")
    parser.add_argument("--patch_dir", dest="patch_dir", type=str, help="folder holding the trained patches (<model>_patch_ratio_<r>/epoch<k>/patch.npy)")
    parser.add_argument("--atk_mode", dest="atk_mode", type=str, default="random",
                        help="where the patch is pasted: random | sp_left | sp_straight | sp_right (column bands of the image)")
    parser.add_argument("--save_feat_map", action="store_true", help="also dump the detector's intermediate feature maps")
    parser.add_argument("--save_feat_path", type=str, default="", help="folder for --save_feat_map")
    parser.add_argument("--devices", "-d", type=str, default="0", help="GPU index (the reference always uses the current device)")
    _common.add_engine_flags(parser)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    dev, _ = _common.setup_device(args.devices)
    from .. import ops
    rt = _common.upstream_or_exit(lambda: upstream.SrcnnRuntime(dev, training=False, workers=0, normalize=False, adopt=args.adopt))
    written, result_dir = _srcnn_detect.run(args, rt, "patch", dev, ops)
    print("wrote %d detections to %s" % (written, result_dir))


if __name__ == "__main__":
    main()
