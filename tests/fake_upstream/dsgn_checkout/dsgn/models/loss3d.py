class RPN3DLoss:
    def __init__(self, cfg):
        self.cfg = cfg

    def __call__(self, bbox_cls, bbox_reg, bbox_centerness, targets, calib, calib_R, ious=None, labels_map=None):
        t = targets[0].bbox.sum() * 1e-3 + targets[0].box3d.sum() * 1e-3
        cls, reg, ctr = (bbox_cls - 0.1).abs().mean(), (bbox_reg * bbox_reg).mean(), bbox_centerness.mean() + t
        return cls + reg + ctr, cls, reg, ctr
