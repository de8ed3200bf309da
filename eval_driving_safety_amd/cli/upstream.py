"""Construction of the UPSTREAM detectors, configs and loaders exactly as the reference's scripts do it, so that
``--model upstream`` runs a user's DSGN / Stereo R-CNN checkout through this engine.

Nothing upstream is vendored here: ``dsgn.*`` / ``env_utils`` (DSGN) and ``model.*`` / ``roi_data_layer.*``
(Stereo R-CNN) are imported from the checkout the CLI is run in, which is how the reference is used
(attack/DSGN/README.md:18,32-55; attack/Stereo-RCNN/README.md:18,31-51).  Each builder cites the script lines it
restates.  The tests drive these builders with stand-in packages of the same names (tests/fake_upstream/)."""
import importlib
import os
import re
import types

import numpy as np
import torch

from ..attacks import StereoBatch
from ..determinism import under_solvers


class UpstreamMissing(ImportError):
    pass


def _need(module, what):
    try:
        return importlib.import_module(module)
    except Exception as e:
        raise UpstreamMissing("%s is not importable (%s: %s). The detectors are third-party checkouts the reference expects "
                              "you to clone (attack/DSGN/README.md:18, attack/Stereo-RCNN/README.md:18); run from inside that "
                              "checkout, or pass --model toy --synthetic N to exercise the attack engine without it."
                              % (what, type(e).__name__, e))


# ------------------------------------------------------------------------------------------------ devices
def resolve_devices(devices, mem_info):
    """``--devices`` as the DSGN scripts read it (attack/DSGN/pgd_attack.py:58-65): empty -> the GPU with the least
    memory in use; ``a-b`` -> the inclusive range (a missing end = first / last GPU).  Returns the comma-separated
    string the scripts export as CUDA_VISIBLE_DEVICES (:82).  The detect-under-patch script's default is the INTEGER 0
    (predict_and_save_patch.py:52), which ``if not args.devices`` (:70) also sends to the least-used GPU; "-d 0" is GPU 0."""
    if not devices:
        devices = str(int(np.argmin(mem_info())))
    devices = str(devices)
    if "-" in devices:
        lo, hi = devices.split("-")
        lo = int(lo) if lo.isdigit() else 0
        hi = int(hi) + 1 if hi.isdigit() else len(mem_info())
        devices = ",".join(str(i) for i in range(lo, hi))
    return devices


def _sysfs_vram_used_by_bus_id():
    """{PCI bus id "dddd:bb:dd.f" -> bytes in use} from the amdgpu driver's sysfs counters (no HIP context is created)"""
    import glob
    out = {}
    for p in glob.glob("/sys/class/drm/card*/device/mem_info_vram_used"):
        dev_dir = os.path.realpath(os.path.dirname(p))             # .../0000:c1:00.0
        bus = os.path.basename(dev_dir).lower()
        if re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-9a-f]", bus):
            try:
                out[bus] = int(open(p).read())
            except (OSError, ValueError):
                pass
    return out


def _hip_bus_id(i):
    """PCI bus id of HIP device i in sysfs spelling, or None when this torch does not expose it"""
    props = torch.cuda.get_device_properties(i)
    bus = getattr(props, "pci_bus_id", None)
    if bus is None:
        return None
    if isinstance(bus, int):                                       # torch exposes domain / bus / device as integers
        return "%04x:%02x:%02x.0" % (int(getattr(props, "pci_domain_id", 0)), bus, int(getattr(props, "pci_device_id", 0)))
    return str(bus).lower()


def used_memory_per_gpu():
    """stand-in for upstream ``env_utils.mem_info`` when it is absent: bytes in use per visible HIP device, in HIP's enumeration
    order.  Read from the driver's sysfs counters matched to the HIP devices BY PCI BUS ID (DRM card numbering need not follow HIP's
    order - mixed or partitioned GPUs, visibility masks); any device the match cannot place falls back to ``torch.cuda.mem_get_info``
    for all of them."""
    n = torch.cuda.device_count()                   # counts devices without initialising them
    try:
        by_bus = _sysfs_vram_used_by_bus_id()
        ids = [_hip_bus_id(i) for i in range(n)]
        if n and all(b is not None and b in by_bus for b in ids) and len(set(ids)) == n:
            return [by_bus[b] for b in ids]
    except Exception:
        pass
    out = []
    for i in range(n):
        free, total = torch.cuda.mem_get_info(i)
        out.append(total - free)
    return out


def pick_device(devices, mem_info=None):
    """One process drives ONE GPU (the reference's nn.DataParallel over a single device is only a checkpoint-key shim,
    SURVEY 2.3): under torchrun LOCAL_RANK decides, otherwise the first id of the resolved ``--devices`` list."""
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device: this engine has no CPU path")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        ids = [int(v) for v in resolve_devices(devices, mem_info or used_memory_per_gpu).split(",")] if devices else []
        # an explicit --devices list (a-b or a,b,c) is indexed by LOCAL_RANK; without one rank k drives GPU k
        index = 0 if os.environ.get("ADV_SHARE_GPU") == "1" else (ids[local] if local < len(ids) and len(ids) > 1 else local)   # ADV_SHARE_GPU: 1-GPU test boxes
        resolved = str(index)
    else:
        resolved = resolve_devices(devices, mem_info or used_memory_per_gpu)
        index = int(resolved.split(",")[0])
    print("Using GPU:{}".format(resolved))                                   # pgd_attack.py:81
    torch.cuda.set_device(index)
    return torch.device("cuda", index), resolved


# ------------------------------------------------------------------------------------------------ adoption
def adopt_model(model, mode, dev, first_call=None):
    """--adopt: the checkout's detector onto libadvengine's convolution kernels with its own weights (adopt.adopt); returns the report"""
    if mode == "off" or dev.type != "cuda":
        return None
    from .. import adopt as _adopt
    verify = None
    if mode == "verify":
        args, kwargs = first_call()
        verify = _adopt.Call(args, kwargs)
    rep = _adopt.adopt(model, verify=verify)
    native = sum(1 for _, what in rep["replaced"] if "torch +" not in what)
    print("adopted {} convolution modules ({} on libadvengine kernels, {} BatchNorms folded, {} ReLUs fused{}){}".format(
        len(rep["replaced"]), native, rep["folded_bn"], rep["fused_relu"],
        "; grid_sample / depth regression bound in {}".format(", ".join(sorted({m for m, _ in rep["functional"]}))) if rep.get("functional") else "",
        "; {} outputs verified within 1e-4".format(rep["verified_outputs"]) if "verified_outputs" in rep else ""))
    return rep


# ------------------------------------------------------------------------------------------------ DSGN
class DsgnAttackCollator:
    """BatchCollator of the two DSGN attack scripts (attack/DSGN/pgd_attack.py:103-126): the dict the loop reads."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __call__(self, batch):
        cols = list(zip(*batch))
        ret = {"imgL": torch.cat(cols[0], dim=0), "imgR": torch.cat(cols[1], dim=0), "disp_L": torch.stack(cols[2], dim=0),
               "calib": cols[3], "calib_R": cols[4], "image_indexes": cols[5]}
        if self.cfg.RPN3D_ENABLE:
            ret["targets"], ret["ious"], ret["labels_map"] = cols[6], cols[7], cols[8]
        return ret


class DsgnDetectCollator:
    """BatchCollator of the two DSGN detect scripts (attack/DSGN/predict_and_save_pgd.py:110-123): a 7-list."""

    def __call__(self, batch):
        cols = list(zip(*batch))
        return [torch.cat(cols[0], dim=0), torch.cat(cols[1], dim=0), torch.stack(cols[2], dim=0), cols[3], cols[4], cols[5], cols[6]]


def calib_tensors(calib, calib_R, absolute_baseline):
    """fu / baseline / projection matrices of a batch (attack/DSGN/pgd_attack.py:262-266; the detect scripts do not take
    the absolute value of the baseline, predict_and_save_pgd.py:356-358)"""
    fu = torch.as_tensor([c.f_u for c in calib])
    base = torch.as_tensor([(c.P[0, 3] - c_R.P[0, 3]) / c.P[0, 0] for c, c_R in zip(calib, calib_R)])
    if absolute_baseline:
        base = torch.abs(base)
    return fu, base, torch.as_tensor(np.array([c.P for c in calib])), torch.as_tensor(np.array([c.P for c in calib_R]))


class DsgnRuntime:
    """cfg + model + loader of one DSGN script run"""

    def __init__(self, args, dev, attack):
        """attack=True: the scaffolding of pgd_attack.py / patch_attack.py (:70-147: Experimenter config, is_train=True
        loader with targets, 12 workers unless --debug); attack=False: that of predict_and_save_*.py (:76-175)."""
        env_utils = _need("env_utils", "env_utils (upstream DSGN)")
        if dev.type == "cuda":
            # upstream's dsgn._C (cost-volume build, sigmoid focal loss, NMS: CUDA sources) does not exist on an MI355X: libadvengine's
            # entry points stand in its place, registered BEFORE the checkout's layers run ``from dsgn import _C``
            from .. import upstream_shims
            done = upstream_shims.install("dsgn", table=upstream_shims.parse_shim_flags(getattr(args, "shim", None)))
            print("{} -> eval_driving_safety_amd.upstream_shims (cost volume / focal loss / NMS on csrc/psv.hip, volume.hip, roi.hip)".format(", ".join(done)))
        models = _need("dsgn.models", "dsgn.models (upstream DSGN)")
        ls = _need("dsgn.dataloader.KITTILoader3D", "dsgn.dataloader.KITTILoader3D")
        DA = _need("dsgn.dataloader.KITTILoader_dataset3d", "dsgn.dataloader.KITTILoader_dataset3d")
        self.RPN3DLoss = _need("dsgn.models.loss3d", "dsgn.models.loss3d").RPN3DLoss
        self.make_postprocessor = _need("dsgn.models.inference3d", "dsgn.models.inference3d").make_fcos3d_postprocessor
        self.args, self.dev, self.attack = args, dev, attack
        if args.loadmodel is None:
            raise SystemExit("--loadmodel is required with --model upstream (the config is read from its directory, pgd_attack.py:70)")
        self.cfg = cfg = env_utils.Experimenter(os.path.dirname(args.loadmodel), args.cfg).config       # :70-71
        if args.debug:                                                                                   # :73-79
            args.btest = len(str(args.devices_resolved).split(","))
            workers = 0
            cfg.debug = True
            args.tag += "debug{}".format(args.debugnum)
        else:
            workers = 12
        if not attack and getattr(args, "train", False):                                                 # predict_and_save_pgd.py:88-90
            args.split_file = "./data/kitti/train.txt"
            args.tag += "_train"
        if not args.btest:
            raise SystemExit("--btest is required (pgd_attack.py:81 asserts it)")
        depth_disp = True if attack else cfg.eval_depth
        left, right, disp = ls.dataloader(args.data_path, args.split_file, depth_disp=depth_disp, cfg=cfg, is_train=attack)   # :93-97
        self.dataset = DA.myImageFloder(left, right, disp, attack, split=args.split_file, cfg=cfg)       # :99-100
        collate = DsgnAttackCollator(cfg) if attack else DsgnDetectCollator()
        self.torch_loader = torch.utils.data.DataLoader(self.dataset, batch_size=args.btest, shuffle=False, num_workers=workers,
                                                        collate_fn=collate, **({} if attack else {"drop_last": False}))    # :130-133
        model = models.StereoNet(cfg=cfg)                                                                # :136-147
        model = torch.nn.DataParallel(model, device_ids=[dev.index]) if dev.type == "cuda" else torch.nn.DataParallel(model)
        model.to(dev)
        model.eval()
        if args.loadmodel is not None and args.loadmodel.endswith("tar"):
            state = torch.load(args.loadmodel, map_location=dev)
            model.load_state_dict(state["state_dict"])
            print("Loaded {}".format(args.loadmodel))
        else:
            print("------------------------------ Load Nothing ---------------------------------")
        print("Number of model parameters: {}".format(sum(p.data.nelement() for p in model.parameters())))
        self.model = model
        self.adoption = adopt_model(model, getattr(args, "adopt", "on"), dev, self._first_sample_call if getattr(args, "adopt", "on") == "verify" else None)

    def _first_sample_call(self):
        """the arguments of one forward call on the first sample of the split (for --adopt verify)"""
        collate = self.torch_loader.collate_fn
        batch = collate([self.dataset[0]])
        if self.attack:
            imgL, imgR, calib, calib_R = batch["imgL"], batch["imgR"], batch["calib"], batch["calib_R"]
        else:
            imgL, imgR, calib, calib_R = batch[0], batch[1], batch[3], batch[4]
        fu, base, proj, proj_r = calib_tensors(calib, calib_R, absolute_baseline=self.attack)
        return (imgL.float().to(self.dev), imgR.float().to(self.dev), fu, base, proj), {"calibs_Proj_R": proj_r}

    # -- the attack scripts' view: StereoBatch + everything the objective needs in ``extra`` -------------------------
    def attack_batches(self):
        from PIL import Image
        dev, cfg = self.dev, self.cfg
        for databatch in self.torch_loader:
            idx = databatch["image_indexes"]
            names = ["%06d" % i for i in idx]
            sizes = []
            for n in names:                                          # the original size, for the PNG crop (:272-276)
                with Image.open("{0}/image_2/{1}.png".format(self.args.data_path, n)) as im:
                    sizes.append(im.size)
            targets = databatch.get("targets")
            if targets is not None:                                  # :257-260
                for t in targets:
                    t.bbox = t.bbox.to(dev)
                    t.box3d = t.box3d.to(dev)
            fu, base, proj, proj_r = calib_tensors(databatch["calib"], databatch["calib_R"], absolute_baseline=True)
            extra = types.SimpleNamespace(calibs_fu=fu, calibs_baseline=base, calibs_Proj=proj, calibs_Proj_R=proj_r,
                                          disp_true=torch.as_tensor(databatch["disp_L"], dtype=torch.float32).to(dev),
                                          targets=targets, calib=databatch["calib"], calib_R=databatch["calib_R"],
                                          ious=databatch.get("ious"), labels_map=databatch.get("labels_map"), image_indexes=idx)
            yield StereoBatch(databatch["imgL"].float(), databatch["imgR"].float(), names, sizes, extra)

    # -- the detect scripts' view -------------------------------------------------------------------------------------
    def detect_batches(self):
        for imgL, imgR, gt_disp, calib, calib_R, image_sizes, image_indexes in self.torch_loader:
            fu, base, proj, proj_r = calib_tensors(calib, calib_R, absolute_baseline=False)
            extra = types.SimpleNamespace(calibs_fu=fu, calibs_baseline=base, calibs_Proj=proj, calibs_Proj_R=proj_r, gt_disp=gt_disp,
                                          calib=calib, calib_R=calib_R, image_sizes=image_sizes, image_indexes=image_indexes)
            yield StereoBatch(imgL.float(), imgR.float(), ["%06d" % i for i in image_indexes], None, extra)

    @under_solvers
    def predict(self, x, extra):
        """``test()`` of the scripts (pgd_attack.py:208-226): no-grad forward + FCOS3D post-processing"""
        b = x.shape[0] // 2
        self.model.eval()
        with torch.no_grad():
            out = self.model(x[:b], x[b:], extra.calibs_fu, extra.calibs_baseline, extra.calibs_Proj, calibs_Proj_R=extra.calibs_Proj_R)
        rets = [out["depth_preds"]]
        if self.cfg.RPN3D_ENABLE:
            rets.append(self.make_postprocessor(self.cfg)(out["bbox_cls"], out["bbox_reg"], out["bbox_centerness"],
                                                          image_sizes=extra.image_sizes, calibs_Proj=extra.calibs_Proj))
        return rets


class _Iterable:
    """a re-iterable, len-aware view of a generator method (what PgdAttack.run / PatchTrainer.train iterate)"""

    def __init__(self, gen, n, batch):
        self.gen, self.n, self.batch = gen, n, batch

    def __iter__(self):
        return self.gen()

    def __len__(self):
        return self.n


def dsgn_attack_loader(rt):
    return _Iterable(rt.attack_batches, len(rt.torch_loader), rt.args.btest)


# ------------------------------------------------------------------------------------------------ Stereo R-CNN
MODEL_PTH = "./models_stereo/stereo_rcnn_12_6477.pth"      # hard-coded in all four scripts (pgd_attack.py:94)


class SrcnnRuntime:
    """roidb + loader + network of one Stereo R-CNN script run"""

    def __init__(self, dev, training, workers=0, model_pth=MODEL_PTH, normalize=None, adopt="on"):
        """attack/Stereo-RCNN/pgd_attack.py:61-99 (training=True: ground truth prepared, ``uncert`` loaded) and
        predict_and_save_pgd.py:78-124 (training=False, normalize=False)."""
        _need("_init_paths", "_init_paths (upstream Stereo R-CNN)")
        if dev.type == "cuda":
            # upstream's model.roi_layers is a compiled CUDA extension: on an MI355X the libadvengine package stands in its place, registered
            # BEFORE the checkout's model code runs ``from model.roi_layers import ROIAlign`` / ``nms`` (stereo_rcnn.py:18, proposal layer)
            from .. import upstream_shims
            upstream_shims.install("srcnn")
            print("model.roi_layers -> eval_driving_safety_amd.upstream_shims.roi_layers (ROIAlign / nms on csrc/roi.hip)")
        roidb_mod = _need("roi_data_layer.roidb", "roi_data_layer.roidb")
        loader_mod = _need("roi_data_layer.roibatchLoader", "roi_data_layer.roibatchLoader (the reference's substitute file)")
        self.cfg = cfg = _need("model.utils.config", "model.utils.config").cfg
        resnet = _need("model.stereo_rcnn.resnet", "model.stereo_rcnn.resnet").resnet
        np.random.seed(cfg.RNG_SEED)                                                                     # :61-62
        cfg.TRAIN.USE_FLIPPED = False
        self.imdb, self.roidb, ratio_list, ratio_index = roidb_mod.combined_roidb("kitti_val")          # :66
        kw = {} if normalize is None else {"normalize": normalize}
        self.dataset = loader_mod.roibatchLoader(self.roidb, ratio_list, ratio_index, 1, self.imdb.num_classes, training=training, **kw)
        self.torch_loader = torch.utils.data.DataLoader(self.dataset, batch_size=1, shuffle=False, num_workers=workers,
                                                        pin_memory=(workers == 0))                      # :73-75 / patch :128-129
        net = resnet(self.imdb.classes, 101, pretrained=False)                                           # :91-99
        net.create_architecture()
        checkpoint = torch.load(model_pth, map_location=dev)
        net.load_state_dict(checkpoint["model"])
        self.uncert = checkpoint["uncert"].to(dev) if (training and "uncert" in checkpoint) else None
        net.to(dev)
        net.eval()
        self.model, self.dev = net, dev
        self.adoption = adopt_model(net, adopt, dev, self._first_sample_call if adopt == "verify" else None)

    def _first_sample_call(self):
        data = self.dataset[0]
        t = [torch.as_tensor(v).unsqueeze(0).to(self.dev) for v in data[:8]]
        return (t[0].float(), t[1].float(), t[2], t[3], t[4], t[5], t[6], t[7], torch.as_tensor(data[8]).to(self.dev)), {}

    def image_name(self, i):
        return self.roidb[i]["img_left"].split("/")[-1].strip()                                           # :120

    def batches(self):
        dev = self.dev
        for i, data in enumerate(self.torch_loader):
            extra = types.SimpleNamespace(im_info=data[2].to(dev), gt_boxes_left=data[3].to(dev), gt_boxes_right=data[4].to(dev),
                                          gt_boxes_merge=data[5].to(dev), gt_dim_orien=data[6].to(dev), gt_kpts=data[7].to(dev),
                                          num_boxes=torch.as_tensor(data[8]).to(dev), roidb_index=i)
            yield StereoBatch(data[0].float(), data[1].float(), [self.image_name(i)], None, extra)


def srcnn_loader(rt):
    return _Iterable(rt.batches, len(rt.torch_loader), 1)
