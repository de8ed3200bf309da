import torch

from dsgn.layers import nms


class BoxList:
    def __init__(self, bbox, fields):
        self.bbox, self.fields = bbox, fields

    def get_field(self, k):
        return self.fields[k]

    def has_field(self, k):
        return k in self.fields


def make_fcos3d_postprocessor(cfg):
    def post(bbox_cls, bbox_reg, bbox_centerness, image_sizes=None, calibs_Proj=None):
        left = []
        for b in range(bbox_cls.shape[0]):
            s = torch.sigmoid(bbox_cls[b].mean()).reshape(1)
            x0 = 100.0 + 50.0 * torch.tanh(bbox_reg[b].mean())
            box = torch.stack([x0, x0 * 0 + 120.0, x0 + 80.0, x0 * 0 + 200.0]).reshape(1, 4)
            # candidates: the box, three jittered copies of lower score (suppressed) - box NMS through the checkout's compiled operator
            jit = torch.tensor([[0.0, 0, 0, 0], [3, -2, 4, 1], [-5, 4, -3, 2], [6, 6, 5, 7]], device=box.device)
            cand, cand_s = box + jit, s * torch.tensor([1.0, 0.9, 0.8, 0.7], device=box.device)
            keep = nms(cand, cand_s, 0.5)
            assert keep.tolist() == [0], keep
            box = cand[keep]
            c = torch.tensor([[-1.0, 1.0, 20.0]], device=box.device) + bbox_centerness[b].mean()
            d = torch.tensor([[-0.8, -0.8, -2.0], [0.8, -0.8, -2.0], [0.8, 0.8, -2.0], [-0.8, 0.8, -2.0],
                              [-0.8, -0.8, 2.0], [0.8, -0.8, 2.0], [0.8, 0.8, 2.0], [-0.8, 0.8, 2.0]], device=box.device)
            left.append(BoxList(box, {"labels": torch.tensor([2]), "scores": s, "box_corner3d": (c + d).reshape(1, 24)}))
        return [left, None]
    return post
