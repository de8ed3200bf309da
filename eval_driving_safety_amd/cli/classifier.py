"""Counterpart of dynamic_vehicles/{train,validate}.py and driving_constraint/{train,validate}.py (the reference's scripts
take no flags: paths and hyper-parameters are constants at the top of each file; they are the defaults here)."""
import argparse
import os

import torch

from .. import classifiers as C


def build_parser():
    p = argparse.ArgumentParser(description="scenario-context classifiers (dynamic vehicles / driving constraint)")
    p.add_argument("task", choices=sorted(C.TASKS))
    p.add_argument("mode", choices=["train", "validate"])
    p.add_argument("--image_dir", default=None, help="default: data/training_image_2/ (dynamic_vehicles) or data/image_2/ (driving_constraint)")
    p.add_argument("--val_image_dir", default=None)
    p.add_argument("--train_csv", default=None)
    p.add_argument("--val_csv", default=None)
    p.add_argument("--save_dir", default="model/")
    p.add_argument("--checkpoint", default=None, help="validate: default model/pretrained_model/cnn_20.pth (validate.py:37)")
    p.add_argument("--imagenet_weights", default=None, help="torchvision state dict of the backbone (pretrained=True of Model.py:19; no network here)")
    p.add_argument("--epochs", type=int, default=None, help="passes over the split")
    p.add_argument("--device", default=None)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    cls, _, _, batch, _, _, flip = C.TASKS[args.task]
    dev = torch.device(args.device if args.device else ("cuda" if torch.cuda.is_available() else "cpu"))      # train.py:11
    dyn = args.task == "dynamic_vehicles"
    img = args.image_dir or ("data/training_image_2/" if dyn else "data/image_2/")
    vimg = args.val_image_dir or ("data/validation_image_2/" if dyn else "data/image_2/")
    tcsv = args.train_csv or ("training_csv.csv" if dyn else "data/training_csv.csv")
    vcsv = args.val_csv or ("validation_csv.csv" if dyn else "data/validation_csv.csv")
    kw = dict(shuffle=True, num_workers=1, pin_memory=dev.type == "cuda")                                     # train.py:26-28
    val = torch.utils.data.DataLoader(C.CsvImageDataset(vimg, vcsv, cls.mean, cls.std), batch_size=32 if args.mode == "validate" else batch, **kw)
    model = cls()
    if args.mode == "validate":
        C.load_checkpoint(model, args.checkpoint or "model/pretrained_model/cnn_20.pth")
        C.check_accuracy(val, model.to(dev), dev)
        return
    if args.imagenet_weights:
        print("loaded %d backbone tensors" % C.load_imagenet_backbone(model, args.imagenet_weights))
    else:
        print("warning: no --imagenet_weights: the backbone starts from random weights (the reference starts from ImageNet)")
    train = torch.utils.data.DataLoader(C.CsvImageDataset(img, tcsv, cls.mean, cls.std, random_flip=flip), batch_size=batch, **kw)
    C.train(args.task, train, val, dev, args.save_dir, epochs=args.epochs, model=model)


if __name__ == "__main__":
    main()
