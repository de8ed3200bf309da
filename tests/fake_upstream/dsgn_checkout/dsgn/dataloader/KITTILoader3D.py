import os


def dataloader(filepath, split, depth_disp=False, cfg=None, is_train=False):
    with open(split) as f:
        ids = [l.strip() for l in f if l.strip()]
    left = [os.path.join(filepath, "image_2", i + ".png") for i in ids]
    right = [os.path.join(filepath, "image_3", i + ".png") for i in ids]
    return left, right, ["depth:" + i for i in ids]
