"""The scenario-context classifiers: torchvision-compatible parameter layout (so the reference's
checkpoints and ImageNet weights load by key), heads, freezing, accuracy rule, learning-rate rules, checkpoint format."""
import os

import numpy as np
import pytest
import torch

from eval_driving_safety_amd import classifiers as C


def test_vgg16_layout_matches_torchvision_and_the_reference_head():
    m = C.DynamicVehicleCNN()
    sd = m.state_dict()
    conv_idx = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]                   # torchvision vgg16.features conv positions
    assert sorted(int(k.split(".")[2]) for k in sd if k.startswith("vgg16.features.") and k.endswith("weight")) == conv_idx
    assert tuple(sd["vgg16.features.28.weight"].shape) == (512, 512, 3, 3) and tuple(sd["vgg16.features.0.weight"].shape) == (64, 3, 3, 3)
    assert sum(v.numel() for k, v in sd.items() if k.startswith("vgg16.features.")) == 14714688        # VGG16 conv parameters
    assert tuple(sd["vgg16.classifier.1.weight"].shape) == (4096, 25088) and tuple(sd["vgg16.classifier.4.weight"].shape) == (1, 4096)
    assert [k for k in sd if k.startswith("vgg16.classifier.")] == ["vgg16.classifier.1.weight", "vgg16.classifier.1.bias",
                                                                    "vgg16.classifier.4.weight", "vgg16.classifier.4.bias"]
    C.freeze_backbone(m)                                                          # train.py:51-55
    assert all(p.requires_grad == ("classifier" in n) for n, p in m.vgg16.named_parameters())


def test_resnet50_layout_matches_torchvision_and_the_reference_head():
    m = C.DrivingConstraintCNN()
    sd = m.state_dict()
    assert tuple(sd["resnet50.conv1.weight"].shape) == (64, 3, 7, 7)
    assert tuple(sd["resnet50.layer1.0.downsample.0.weight"].shape) == (256, 64, 1, 1)
    assert tuple(sd["resnet50.layer2.0.conv2.weight"].shape) == (128, 128, 3, 3)
    assert tuple(sd["resnet50.layer4.2.conv3.weight"].shape) == (2048, 512, 1, 1) and "resnet50.layer3.5.bn3.running_var" in sd
    assert "resnet50.layer3.6.conv1.weight" not in sd and "resnet50.layer2.1.downsample.0.weight" not in sd
    backbone = sum(p.numel() for n, p in m.resnet50.named_parameters() if not n.startswith("fc."))
    assert backbone == 23508032                                                   # torchvision resnet50 without its fc
    assert tuple(sd["resnet50.fc.0.weight"].shape) == (1, 2048)
    C.freeze_backbone(m)
    assert [n for n, p in m.resnet50.named_parameters() if p.requires_grad] == ["fc.0.weight", "fc.0.bias"]


def test_forward_accuracy_rule_and_backbone_loading(tmp_path):
    torch.manual_seed(0)
    m = C.DrivingConstraintCNN().eval()
    x = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        y = m(x)
    assert tuple(y.shape) == (2,) and bool(((y > 0) & (y < 1)).all())
    assert C.count_correct(torch.tensor([0.5, 0.49, 0.9, 0.1]), torch.tensor([1.0, 1.0, 0.0, 0.0])) == (2, 4)    # >= 0.5 -> 1
    # a "torchvision" state dict (backbone keys without the resnet50. prefix, its own 1000-way fc) loads into the backbone only
    tv = {k[len("resnet50."):]: v.clone() + 1 for k, v in m.state_dict().items() if not k.startswith("resnet50.fc.")}
    tv["fc.weight"], tv["fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)
    torch.save(tv, str(tmp_path / "resnet50.pth"))
    before = m.resnet50.fc[0].weight.clone()
    n = C.load_imagenet_backbone(m, str(tmp_path / "resnet50.pth"))
    assert n == len(tv) - 2 and torch.equal(m.resnet50.fc[0].weight, before)
    assert torch.equal(m.resnet50.conv1.weight, tv["conv1.weight"])


def test_training_loop_rules_and_checkpoint_format(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(0)
    os.makedirs(str(tmp_path / "img"))
    rows = ["img_name,label"]
    for i in range(6):
        label = i % 2
        a = rs.randint(0, 80, (50, 70, 3)) + (150 if label else 0)                # bright = 1, dark = 0
        Image.fromarray(a.astype(np.uint8)).save(str(tmp_path / "img" / ("%02d.png" % i)))
        rows.append("%02d.png,%d" % (i, label))
    (tmp_path / "t.csv").write_text("\n".join(rows) + "\n")

    class Tiny(torch.nn.Module):                                                  # the loop is model-agnostic: a tiny stand-in keeps the test fast
        head_key = "fc"

        def __init__(self):
            super().__init__()
            self.net = torch.nn.Sequential()
            self.net.body = torch.nn.Conv2d(3, 2, 3, stride=8)
            self.net.fc = torch.nn.Linear(2, 1)

        backbone = property(lambda self: self.net)

        def forward(self, x):
            return torch.sigmoid(self.net.fc(self.net.body(x).mean(dim=(2, 3)))).squeeze(1)

    ds = C.CsvImageDataset(str(tmp_path / "img"), str(tmp_path / "t.csv"), C.DynamicVehicleCNN.mean, C.DynamicVehicleCNN.std)
    x0, y0 = ds[1]
    assert tuple(x0.shape) == (3, 224, 224) and float(y0) == 1.0 and len(ds) == 6
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False)
    logs = []
    C.train("dynamic_vehicles", loader, loader, torch.device("cpu"), str(tmp_path / "model"), epochs=10, model=Tiny(), log=logs.append)
    assert logs.count("Updated learning rate: 5e-07") == 1 and logs.count("Updated learning rate: 2.5e-07") == 1   # train.py:104-114
    assert sum(l.startswith("Got ") for l in logs) == 5                           # every second epoch (:101)
    ck = torch.load(str(tmp_path / "model" / "cnn_10.pth"))
    assert sorted(ck) == ["epoch", "model_state_dict", "optimizer_state_dict"] and ck["epoch"] == 10
    assert sorted(os.listdir(str(tmp_path / "model"))) == sorted("cnn_%d.pth" % e for e in range(1, 11))           # every epoch (:117-125)
    logs = []
    C.train("driving_constraint", loader, loader, torch.device("cpu"), str(tmp_path / "m2"), epochs=5, model=Tiny(), log=logs.append)
    assert sorted(os.listdir(str(tmp_path / "m2"))) == ["cnn_5.pth"] and not any("learning rate" in l for l in logs)  # every 5th; lr constant


@pytest.mark.gpu
def test_classifiers_run_on_the_gpu():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for cls in (C.DynamicVehicleCNN, C.DrivingConstraintCNN):
        m = cls().to(dev)
        C.freeze_backbone(m)
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        x, y = torch.randn(4, 3, 224, 224, device=dev), torch.tensor([1.0, 0.0, 1.0, 0.0], device=dev)
        losses = []
        for _ in range(3):
            loss = torch.nn.BCELoss()(m(x), y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        assert all(np.isfinite(losses))


def test_classifier_cli_train_then_validate(tmp_path):
    """`cli.classifier driving_constraint train` writes cnn_5.pth (every 5th epoch, train.py:109-118); `validate` loads it by key"""
    import subprocess
    import sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = np.random.RandomState(1)
    os.makedirs(str(tmp_path / "data" / "image_2"))
    rows = ["img_name,label"]
    for i in range(4):
        Image.fromarray(rs.randint(0, 255, (60, 90, 3)).astype(np.uint8)).save(str(tmp_path / "data" / "image_2" / ("%d.png" % i)))
        rows.append("%d.png,%d" % (i, i % 2))
    for name in ("training_csv.csv", "validation_csv.csv"):
        (tmp_path / "data" / name).write_text("\n".join(rows) + "\n")
    env = dict(os.environ, PYTHONPATH=root, OMP_NUM_THREADS="4")
    run = lambda *a: subprocess.run([sys.executable, "-m", "eval_driving_safety_amd.cli.classifier"] + list(a), cwd=str(tmp_path), env=env,
                                    stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    out = run("driving_constraint", "train", "--epochs", "5", "--device", "cpu")
    assert out.returncode == 0, out.stdout[-2000:]
    assert os.listdir(str(tmp_path / "model")) == ["cnn_5.pth"] and "warning: no --imagenet_weights" in out.stdout
    out = run("driving_constraint", "validate", "--checkpoint", "model/cnn_5.pth", "--device", "cpu")
    assert out.returncode == 0 and "Got " in out.stdout and "/ 4 with accuracy" in out.stdout, out.stdout[-2000:]
