import os
import types

N_IMAGES = int(os.environ.get("FAKE_SRCNN_IMAGES", "3"))


def combined_roidb(name):
    assert name == "kitti_val"
    imdb = types.SimpleNamespace(classes=("__background__", "Car"), num_classes=2, _classes=("__background__", "Car"))
    roidb = [{"img_left": "data/kitti/object/training/image_2/%06d.png" % (7 + 3 * i)} for i in range(N_IMAGES)]
    return imdb, roidb, [1.0] * N_IMAGES, list(range(N_IMAGES))
