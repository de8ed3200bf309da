"""The two scenario-context classifiers of the reference on PyTorch-ROCm:

    dynamic_vehicles/     VGG16 + a 2-layer sigmoid head on 224x224 car crops      (Model.py:15-34, train.py:22-125)
    driving_constraint/   ResNet50 + a sigmoid fc on 224x224 scene images           (Model.py:15-30, train.py:12-124)

They are NOT on the perturbation hot path (standard fine-tunes whose only link to the attacks is through files: the
evaluation scripts read their decisions as the presence of a label file, evaluation/convert_scenarios.py:40-42,109-112),
so this is plain torch running on MIOpen - no custom kernels.  What is reproduced: the architectures with torchvision's
parameter names (torchvision is absent here; the reference's checkpoints ``cnn_N.pth`` and torchvision's ImageNet
weights load by key), the heads, the frozen-backbone rule, the transforms, the optimiser / learning-rate rules, the
accuracy count and the checkpoint format.
"""
import csv
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ backbones
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


class _Vgg16(nn.Module):
    """torchvision.models.vgg16 layout: ``features`` (13 conv3x3 + ReLU, 5 max-pools), ``avgpool`` to 7x7, ``classifier``"""

    def __init__(self, classifier):
        super().__init__()
        layers, cin = [], 3
        for v in VGG16_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = classifier

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)     # stride on the 3x3 (ResNet v1.5)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        return F.relu(self.bn3(self.conv3(y)) + idt)


class _ResNet50(nn.Module):
    """torchvision.models.resnet50 layout: conv1/bn1, layer1..4 = (3, 4, 6, 3) bottlenecks, global average pool, ``fc``"""

    def __init__(self, fc):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (width, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            mods = []
            for b in range(blocks):
                s = stride if b == 0 else 1
                down = None
                if s != 1 or cin != width * 4:
                    down = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride=s, bias=False), nn.BatchNorm2d(width * 4))
                mods.append(_Bottleneck(cin, width, s, down))
                cin = width * 4
            setattr(self, "layer%d" % i, nn.Sequential(*mods))
        self.fc = fc

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


class _Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class _ReLU(nn.Module):
    def forward(self, x):
        return F.relu(x)


# ------------------------------------------------------------------------------------------------ the two models
class DynamicVehicleCNN(nn.Module):
    """dynamic_vehicles/Model.py:15-34: VGG16 whose classifier is Flatten, Linear(25088, 4096), Dropout(0.1), ReLU,
    Linear(4096, 1), Dropout(0.1), Sigmoid; output squeezed to [B].  Keys: ``vgg16.features.*``, ``vgg16.classifier.{1,4}.*``."""

    mean, std = (0.3091, 0.3181, 0.3248), (0.2328, 0.2308, 0.2337)          # train.py:16-17

    def __init__(self, num_classes=1):
        super().__init__()
        head = nn.Sequential(_Flatten(), nn.Linear(25088, 4096, bias=True), nn.Dropout(0.1), _ReLU(),
                             nn.Linear(4096, num_classes, bias=True), nn.Dropout(0.1), nn.Sigmoid())
        self.vgg16 = _Vgg16(head)

    backbone = property(lambda self: self.vgg16)
    head_key = "classifier"

    def forward(self, images):
        return self.vgg16(images).squeeze(1)


class DrivingConstraintCNN(nn.Module):
    """driving_constraint/Model.py:15-30: ResNet50 whose fc is Linear(2048, 1), Dropout(0.5), Sigmoid.  Keys ``resnet50.*``."""

    mean, std = (0.3775, 0.3923, 0.3839), (0.3110, 0.3154, 0.3180)          # train.py:34-35

    def __init__(self, num_classes=1):
        super().__init__()
        self.resnet50 = _ResNet50(nn.Sequential(nn.Linear(2048, num_classes, bias=True), nn.Dropout(0.5), nn.Sigmoid()))

    backbone = property(lambda self: self.resnet50)
    head_key = "fc"

    def forward(self, images):
        return self.resnet50(images).squeeze(1)


def load_imagenet_backbone(model, path):
    """``models.vgg16(pretrained=True)`` / ``resnet50(pretrained=True)`` (Model.py:19): there is no network here, so the
    torchvision state dict is read from a file; the replaced head keeps its fresh initialisation."""
    state = torch.load(path, map_location="cpu")
    head = model.head_key + "."
    own = model.backbone.state_dict()
    picked = {k: v for k, v in state.items() if not k.startswith(head) and k in own and own[k].shape == v.shape}
    missing = [k for k in own if not k.startswith(head) and k not in picked and not k.endswith("num_batches_tracked")]
    if missing:
        raise KeyError("ImageNet weights lack %d backbone tensors, e.g. %s" % (len(missing), missing[:3]))
    model.backbone.load_state_dict(picked, strict=False)
    return len(picked)


def freeze_backbone(model, train_backbone=False):
    """train.py:51-55 / :62-66: only the head trains unless ``train_CNN`` / ``pretrained`` is set"""
    for name, p in model.backbone.named_parameters():
        p.requires_grad = True if model.head_key in name else train_backbone


# ------------------------------------------------------------------------------------------------ data
def to_tensor_224(img, mean, std, flip=False):
    """transforms.Resize((224, 224)) [PIL bilinear] -> (RandomHorizontalFlip) -> ToTensor -> Normalize"""
    from PIL import Image
    img = img.convert("RGB").resize((224, 224), Image.BILINEAR)
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1))).float().div_(255.0)
    return (x - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)


class CsvImageDataset(torch.utils.data.Dataset):
    """DynamicVehicleDataset / DrivingConstraintDataset (Dataset.py:8-25): rows ``img_name,label`` after a header line"""

    def __init__(self, root_dir, annotation_file, mean, std, random_flip=False, seed=0):
        with open(annotation_file, newline="") as f:
            rows = list(csv.reader(f))[1:]
        self.items = [(r[0], float(r[1])) for r in rows if r]
        self.root, self.mean, self.std, self.random_flip = root_dir, mean, std, random_flip
        self.rng = np.random.RandomState(seed)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        name, label = self.items[i]
        with Image.open(os.path.join(self.root, name)) as im:
            x = to_tensor_224(im, self.mean, self.std, flip=self.random_flip and self.rng.rand() < 0.5)
        return x, torch.tensor(label)


# ------------------------------------------------------------------------------------------------ loops
def count_correct(scores, labels):
    """the scripts' accuracy rule (train.py:70-73): score >= 0.5 -> 1.0, compared with the float label"""
    pred = (scores >= 0.5).to(labels.dtype)
    return int((pred == labels).sum()), int(pred.numel())


def check_accuracy(loader, model, device, log=print):
    num_correct = num_samples = 0
    model.eval()
    with torch.no_grad():
        for x, y in loader:
            c, n = count_correct(model(x.to(device)), y.to(device))
            num_correct += c
            num_samples += n
    model.train()
    acc = float(num_correct) / float(max(1, num_samples)) * 100
    log("Got {} / {} with accuracy {:.2f}".format(num_correct, num_samples, acc))
    return acc


TASKS = {
    # task: (model, lr, epochs, batch, validate every, save every, flip)        train.py of each folder
    "dynamic_vehicles": (DynamicVehicleCNN, 0.000001, 20, 16, 2, 1, False),      # :22-25,48-49,101,117-125
    "driving_constraint": (DrivingConstraintCNN, 0.001, 20, 8, 2, 5, True),      # :18-21,64-67,105,109-118
}


def train(task, train_loader, val_loader, device, save_dir, epochs=None, model=None, train_backbone=False, log=print):
    """the training loops of dynamic_vehicles/train.py:87-125 and driving_constraint/train.py:91-118.

    dynamic_vehicles halves the learning rate once from epoch 6 and once from epoch 10 (:104-114).  driving_constraint
    builds a MultiStepLR([5, 10, 15], 0.1) but never steps it (:66-67 and no ``scheduler.step()`` anywhere), so its
    learning rate stays 1e-3 - kept."""
    cls, lr, n_epochs, _, val_every, save_every, _ = TASKS[task]
    model = (cls() if model is None else model).to(device)
    freeze_backbone(model, train_backbone)
    criterion = nn.BCELoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    os.makedirs(save_dir, exist_ok=True)
    model.train()
    for epoch in range(1, (epochs or n_epochs) + 1):
        last = None
        for imgs, labels in train_loader:
            loss = criterion(model(imgs.to(device)), labels.to(device))
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            last = float(loss.detach())
        log("Epoch [{}/{}] loss={}".format(epoch, epochs or n_epochs, last))
        if epoch % val_every == 0 and val_loader is not None:
            check_accuracy(val_loader, model, device, log)
        if task == "dynamic_vehicles":
            g = optimizer.param_groups[0]
            if epoch >= 6 and g["lr"] == 0.000001:
                g["lr"] *= 0.5
                log("Updated learning rate: {}".format(g["lr"]))
            if epoch >= 10 and g["lr"] == 0.0000005:
                g["lr"] *= 0.5
                log("Updated learning rate: {}".format(g["lr"]))
        if epoch % save_every == 0:
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict()},
                       os.path.join(save_dir, "cnn_{}.pth".format(epoch)))
    return model


def load_checkpoint(model, path, device="cpu"):
    """validate.py:37-38: ``torch.load(path)['model_state_dict']``"""
    model.load_state_dict(torch.load(path, map_location=device)["model_state_dict"])
    return model
