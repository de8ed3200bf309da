import argparse
import os
import sys

import torch


def add_scaffolding(parser):
    """flags shared by the DSGN scripts (attack/DSGN/pgd_attack.py:35-51)"""
    parser.add_argument("-cfg", "--cfg", "--config", default=None, help="config path")
    parser.add_argument("--data_path", default="./data/kitti/training", help="select model")
    parser.add_argument("--loadmodel", default=None, help="loading model")
    parser.add_argument("--seed", type=int, default=1, metavar="S", help="random seed (default: 1)")
    parser.add_argument("--split_file", default="./data/kitti/val.txt", help="split file")
    parser.add_argument("--btest", "-btest", type=int, default=None)
    parser.add_argument("--devices", "-d", type=str, default=None)
    parser.add_argument("--tag", "-t", type=str, default="")
    parser.add_argument("--debug", action="store_true", default=False, help="debug mode")
    parser.add_argument("--debugnum", default=None, type=int, help="debug mode")
    add_engine_flags(parser)


def add_engine_flags(parser):
    g = parser.add_argument_group("engine (not in the reference)")
    g.add_argument("--model", default="upstream", choices=["upstream", "toy"],
                   help="'upstream' imports the user's DSGN / Stereo R-CNN checkout; 'toy' runs the plumbing "
                        "with a fixed-seed differentiable stand-in on synthetic KITTI-shaped pairs")
    g.add_argument("--synthetic", type=int, default=0, metavar="N", help="attack N synthetic pairs instead of --data_path")
    g.add_argument("--out_root", default=".", help="where the *_pgd_iters_k / *_patch_ratio_r folders go")
    g.add_argument("--save_every", type=int, default=1, help="write every k-th iterate (reference: every one)")
    g.add_argument("--pos_seed", type=int, default=None, help="seed of the patch-position stream (reference: unseeded)")


def setup_device():
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("no ROCm device: this engine has no CPU path")
    torch.cuda.set_device(local)
    return torch.device("cuda", local)


def upstream_unavailable(what):
    sys.exit("%s is not importable. The detectors are third-party checkouts the reference expects you to clone "
             "(attack/DSGN/README.md:18, attack/Stereo-RCNN/README.md:18); run from inside that checkout, or pass "
             "--model toy --synthetic N to exercise the attack engine without it." % what)
